// gr::gfdm::add_cyclic_prefix_cc over the HIP C-ABI (replaces lib/add_cyclic_prefix_cc.cc of gr-gfdm).
#include <gfdm/add_cyclic_prefix_cc.h>
#include <gfdm/gfdm_kernel_utils.h>
#include <gfdm_hip.h>

#include <stdexcept>
#include <string>

namespace gr {
namespace gfdm {

namespace {
void raise(int status, const char* where)
{
    if (status == GFDM_HIP_OK) return;
    const char* detail = gfdm_hip_last_error();
    std::string msg = (detail && *detail) ? detail : gfdm_hip_strerror(status);
    if (status == GFDM_HIP_EINVAL) throw std::invalid_argument(msg);
    throw std::runtime_error(std::string(where) + ": " + msg);
}
inline float* fp(add_cyclic_prefix_cc::gfdm_complex* p) { return reinterpret_cast<float*>(p); }
inline const float* fp(const add_cyclic_prefix_cc::gfdm_complex* p) { return reinterpret_cast<const float*>(p); }
} // namespace

add_cyclic_prefix_cc::add_cyclic_prefix_cc(int block_len, int cp_len, int cs_len, int ramp_len, std::vector<gfdm_complex> window_taps,
                                           int cyclic_shift)
    : d_block_len(block_len), d_cp_len(cp_len), d_cs_len(cs_len), d_ramp_len(ramp_len), d_cyclic_shift(cyclic_shift), d_handle(nullptr)
{
    raise(gfdm_hip_cyclic_prefixer_create(&d_handle, block_len, cp_len, cs_len, ramp_len, fp(window_taps.data()),
                                          static_cast<int>(window_taps.size()), cyclic_shift, gfdm_kernel_utils::default_device()),
          "add_cyclic_prefix_cc");
}

add_cyclic_prefix_cc::~add_cyclic_prefix_cc() { gfdm_hip_cyclic_prefixer_destroy(d_handle); }

void add_cyclic_prefix_cc::generic_work(gfdm_complex* p_out, const gfdm_complex* p_in) { add_cyclic_prefix(p_out, p_in, d_cyclic_shift); }

void add_cyclic_prefix_cc::add_cyclic_prefix(gfdm_complex* p_out, const gfdm_complex* p_in, const int cyclic_prefix)
{
    add_cyclic_prefix_batch(p_out, p_in, cyclic_prefix, 1);
}

void add_cyclic_prefix_cc::remove_cyclic_prefix(gfdm_complex* p_out, const gfdm_complex* p_in) { remove_cyclic_prefix_batch(p_out, p_in, 1); }

void add_cyclic_prefix_cc::add_cyclic_prefix_batch(gfdm_complex* out, const gfdm_complex* in, int cyclic_shift, long nblocks)
{
    raise(gfdm_hip_cyclic_prefixer_add_host(d_handle, fp(out), fp(in), cyclic_shift, nblocks), "add_cyclic_prefix");
}

void add_cyclic_prefix_cc::remove_cyclic_prefix_batch(gfdm_complex* out, const gfdm_complex* in, long nblocks)
{
    raise(gfdm_hip_cyclic_prefixer_remove_host(d_handle, fp(out), fp(in), nblocks), "remove_cyclic_prefix");
}

void add_cyclic_prefix_cc::add_cyclic_prefix_device(void* d_out, const void* d_in, int cyclic_shift, long nblocks, void* hip_stream)
{
    raise(gfdm_hip_cyclic_prefixer_add_device(d_handle, d_out, d_in, cyclic_shift, nblocks, hip_stream), "add_cyclic_prefix_device");
}

void add_cyclic_prefix_cc::remove_cyclic_prefix_device(void* d_out, const void* d_in, long nblocks, void* hip_stream)
{
    raise(gfdm_hip_cyclic_prefixer_remove_device(d_handle, d_out, d_in, nblocks, hip_stream), "remove_cyclic_prefix_device");
}

} // namespace gfdm
} // namespace gr
