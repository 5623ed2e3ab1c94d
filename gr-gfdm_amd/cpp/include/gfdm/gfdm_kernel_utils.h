/* Common base of the GPU-backed kernel classes.
 *
 * Mirrors gr-gfdm's include/gfdm/gfdm_kernel_utils.h:37-52 where it is part of the kernel
 * contract: the `gfdm_complex` typedef and calculate_signal_energy().  The reference base also
 * hands out FFTW plans (initialize_fft); plans are an implementation detail of the CPU kernels
 * and have no meaning for a HIP back end, so that helper is intentionally absent.
 */
#ifndef INCLUDED_GFDM_GFDM_KERNEL_UTILS_H
#define INCLUDED_GFDM_GFDM_KERNEL_UTILS_H

#include <gfdm/api.h>
#include <complex>
#include <stdexcept>
#include <string>
#include <vector>

namespace gr {
namespace gfdm {

class GFDM_API gfdm_kernel_utils
{
public:
    typedef std::complex<float> gfdm_complex;

    gfdm_kernel_utils() = default;
    ~gfdm_kernel_utils() = default;

    /* sum |x|^2 over ninput_size samples (host-side helper, not on the hot path) */
    float calculate_signal_energy(const gfdm_complex* p_in, const int ninput_size);

    /* addition: the GPU (HIP device ordinal) on which kernel objects constructed BY THE CALLING THREAD are created from now on;
     * the reference constructors have no such argument, so it is a per-thread setting (default 0).  Returns the previous value.
     * gfdm/sharded_batch.h uses it to put one kernel object on every GPU of a node. */
    static int set_default_device(int device);
    static int default_device();

protected:
    /* translate a gfdm_hip status into the exception the reference would have thrown */
    static void throw_on_error(int status, const char* where);
};

} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_GFDM_KERNEL_UTILS_H */
