/* gr::gfdm::add_cyclic_prefix_cc -- public interface of gr-gfdm's include/gfdm/add_cyclic_prefix_cc.h:40-60 (cyclic prefix /
 * suffix with cyclic shift + block-pinching window ramps; prefix removal), executed as HIP copy kernels behind include/gfdm_hip.h.
 * Drop-in for lib/cyclic_prefixer_cc_impl.cc, lib/remove_prefix_cc_impl.cc and python/bindings/cyclic_prefix_python.cc.
 * transmitter_kernel and the receivers' frame interface have the same arithmetic fused into their kernels.
 */
#ifndef INCLUDED_GFDM_ADD_CYCLIC_PREFIX_CC_H
#define INCLUDED_GFDM_ADD_CYCLIC_PREFIX_CC_H

#include <gfdm/api.h>

#include <complex>
#include <vector>

struct gfdm_hip_cyclic_prefixer;

namespace gr {
namespace gfdm {

class GFDM_API add_cyclic_prefix_cc
{
public:
    typedef std::complex<float> gfdm_complex;

    /* throws std::invalid_argument with the reference's message for a wrong number of window taps (lib/add_cyclic_prefix_cc.cc:42-50)
     * and for a cyclic shift the reference would read out of bounds with; std::runtime_error when no GPU is usable */
    add_cyclic_prefix_cc(int block_len, int cp_len, int cs_len, int ramp_len, std::vector<gfdm_complex> window_taps, int cyclic_shift = 0);
    ~add_cyclic_prefix_cc();
    add_cyclic_prefix_cc(const add_cyclic_prefix_cc&) = delete;
    add_cyclic_prefix_cc& operator=(const add_cyclic_prefix_cc&) = delete;

    void generic_work(gfdm_complex* p_out, const gfdm_complex* p_in);
    void add_cyclic_prefix(gfdm_complex* p_out, const gfdm_complex* p_in, const int cyclic_prefix);
    void remove_cyclic_prefix(gfdm_complex* p_out, const gfdm_complex* p_in);
    int block_size() { return d_block_len; }
    int frame_size() { return block_size() + d_cp_len + d_cs_len; }
    int cyclic_shift() const { return d_cyclic_shift; }

    /* --- additions: whole batches per call (blocks / frames back to back), host or device pointers --- */
    void add_cyclic_prefix_batch(gfdm_complex* out, const gfdm_complex* in, int cyclic_shift, long nblocks);
    void remove_cyclic_prefix_batch(gfdm_complex* out, const gfdm_complex* in, long nblocks);
    void add_cyclic_prefix_device(void* d_out, const void* d_in, int cyclic_shift, long nblocks, void* hip_stream);
    void remove_cyclic_prefix_device(void* d_out, const void* d_in, long nblocks, void* hip_stream);

private:
    const int d_block_len;
    const int d_cp_len;
    const int d_cs_len;
    const int d_ramp_len;
    const int d_cyclic_shift;
    gfdm_hip_cyclic_prefixer* d_handle;
};

} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_ADD_CYCLIC_PREFIX_CC_H */
