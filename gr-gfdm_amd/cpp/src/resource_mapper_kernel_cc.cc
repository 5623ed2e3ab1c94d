// gr::gfdm::resource_mapper_kernel_cc over the HIP C-ABI (replaces lib/resource_mapper_kernel_cc.cc of gr-gfdm).
#include <gfdm/resource_mapper_kernel_cc.h>
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
inline float* fp(resource_mapper_kernel_cc::gfdm_complex* p) { return reinterpret_cast<float*>(p); }
inline const float* fp(const resource_mapper_kernel_cc::gfdm_complex* p) { return reinterpret_cast<const float*>(p); }
} // namespace

resource_mapper_kernel_cc::resource_mapper_kernel_cc(int timeslots, int subcarriers, int active_subcarriers, std::vector<int> subcarrier_map,
                                                     bool per_timeslot, bool is_mapper)
    : d_block_size(static_cast<size_t>(timeslots > 0 ? timeslots : 0) * static_cast<size_t>(active_subcarriers > 0 ? active_subcarriers : 0)),
      d_frame_size(static_cast<size_t>(timeslots > 0 ? timeslots : 0) * static_cast<size_t>(subcarriers > 0 ? subcarriers : 0)),
      d_is_mapper(is_mapper), d_handle(nullptr)
{
    raise(gfdm_hip_resource_mapper_create(&d_handle, timeslots, subcarriers, active_subcarriers, subcarrier_map.data(),
                                          static_cast<int>(subcarrier_map.size()), per_timeslot ? 1 : 0, gfdm_kernel_utils::default_device()),
          "resource_mapper_kernel_cc");
}

resource_mapper_kernel_cc::~resource_mapper_kernel_cc() { gfdm_hip_resource_mapper_destroy(d_handle); }

void resource_mapper_kernel_cc::map_to_resources(gfdm_complex* p_out, const gfdm_complex* p_in, const size_t ninput_size)
{
    map_to_resources_batch(p_out, p_in, ninput_size, 1);
}

void resource_mapper_kernel_cc::demap_from_resources(gfdm_complex* p_out, const gfdm_complex* p_in, const size_t noutput_size)
{
    demap_from_resources_batch(p_out, p_in, noutput_size, 1);
}

namespace {
// sizes above block_size() are the reference's std::invalid_argument (lib/resource_mapper_kernel_cc.cc:78-82, 95-99); anything that
// does not fit an int is above it
int checked_size(size_t n, size_t block_size, const char* what)
{
    if (n > block_size)
        throw std::invalid_argument(std::string(what) + " vector size(" + std::to_string(n) + ") MUST not exceed active_subcarriers * timeslots(" +
                                    std::to_string(block_size) + ")!");
    return static_cast<int>(n);
}
} // namespace

void resource_mapper_kernel_cc::map_to_resources_batch(gfdm_complex* out, const gfdm_complex* in, size_t ninput_size, long nblocks)
{
    raise(gfdm_hip_resource_mapper_map_host(d_handle, fp(out), fp(in), checked_size(ninput_size, d_block_size, "input"), nblocks), "map_to_resources");
}

void resource_mapper_kernel_cc::demap_from_resources_batch(gfdm_complex* out, const gfdm_complex* in, size_t noutput_size, long nblocks)
{
    raise(gfdm_hip_resource_mapper_demap_host(d_handle, fp(out), fp(in), checked_size(noutput_size, d_block_size, "output"), nblocks),
          "demap_from_resources");
}

void resource_mapper_kernel_cc::map_to_resources_device(void* d_out, const void* d_in, size_t ninput_size, long nblocks, void* hip_stream)
{
    raise(gfdm_hip_resource_mapper_map_device(d_handle, d_out, d_in, checked_size(ninput_size, d_block_size, "input"), nblocks, hip_stream),
          "map_to_resources_device");
}

void resource_mapper_kernel_cc::demap_from_resources_device(void* d_out, const void* d_in, size_t noutput_size, long nblocks, void* hip_stream)
{
    raise(gfdm_hip_resource_mapper_demap_device(d_handle, d_out, d_in, checked_size(noutput_size, d_block_size, "output"), nblocks, hip_stream),
          "demap_from_resources_device");
}

} // namespace gfdm
} // namespace gr
