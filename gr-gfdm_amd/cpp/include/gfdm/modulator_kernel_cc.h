/* gr::gfdm::modulator_kernel_cc -- same public interface as gr-gfdm's
 * include/gfdm/modulator_kernel_cc.h:41-51, executed by the HIP kernels behind include/gfdm_hip.h.
 *
 * Drop-in for its three callers: lib/simple_modulator_cc_impl.cc:51-52,73,
 * lib/transmitter_kernel.cc:49-50,83 and python/bindings/modulator_python.cc:34-59.
 */
#ifndef INCLUDED_GFDM_MODULATOR_KERNEL_CC_H
#define INCLUDED_GFDM_MODULATOR_KERNEL_CC_H

#include <gfdm/gfdm_kernel_utils.h>

struct gfdm_hip_modulator;

namespace gr {
namespace gfdm {

class GFDM_API modulator_kernel_cc : public gfdm_kernel_utils
{
public:
    /* throws std::invalid_argument when frequency_taps.size() != n_timeslots * overlap,
     * std::runtime_error when no GPU is usable */
    modulator_kernel_cc(int n_timeslots, int n_subcarriers, int overlap, std::vector<gfdm_complex> frequency_taps);
    ~modulator_kernel_cc();
    modulator_kernel_cc(const modulator_kernel_cc&) = delete;
    modulator_kernel_cc& operator=(const modulator_kernel_cc&) = delete;

    /* one block of block_size() symbols, host pointers, synchronous (the reference contract) */
    void generic_work(gfdm_complex* p_out, const gfdm_complex* p_in);
    int block_size() { return d_n_subcarriers * d_n_timeslots; }
    std::vector<gfdm_complex> filter_taps();

    /* --- additions: whole batches per call --- */
    /* nblocks blocks back to back, host pointers */
    void generic_work_batch(gfdm_complex* p_out, const gfdm_complex* p_in, long nblocks);
    /* device pointers, enqueued on hip_stream (hipStream_t as void*), no synchronisation */
    void generic_work_device(void* d_out, const void* d_in, long nblocks, void* hip_stream);
    const char* kernel_name() const;

private:
    int d_n_timeslots;
    int d_n_subcarriers;
    int d_overlap;
    gfdm_hip_modulator* d_handle;
};

} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_MODULATOR_KERNEL_CC_H */
