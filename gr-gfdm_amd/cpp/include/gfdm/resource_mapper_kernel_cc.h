/* gr::gfdm::resource_mapper_kernel_cc -- public interface of gr-gfdm's include/gfdm/resource_mapper_kernel_cc.h:38-60 (data
 * symbols <-> the [subcarriers][timeslots] resource grid), executed as HIP gather kernels behind include/gfdm_hip.h.  Drop-in for
 * lib/resource_mapper_cc_impl.cc, lib/resource_demapper_cc_impl.cc and python/bindings/resource_mapper_python.cc.
 * Where the mapper sits next to a modulator or a receiver, transmitter_kernel / the receivers' frame interface do the same work
 * inside their kernels without the grid ever reaching memory.
 */
#ifndef INCLUDED_GFDM_RESOURCE_MAPPER_KERNEL_CC_H
#define INCLUDED_GFDM_RESOURCE_MAPPER_KERNEL_CC_H

#include <gfdm/api.h>

#include <complex>
#include <cstddef>
#include <vector>

struct gfdm_hip_resource_mapper;

namespace gr {
namespace gfdm {

class GFDM_API resource_mapper_kernel_cc
{
public:
    typedef std::complex<float> gfdm_complex;

    /* throws std::invalid_argument with the reference's messages (lib/resource_mapper_kernel_cc.cc:44-69), std::runtime_error when
     * no GPU is usable */
    resource_mapper_kernel_cc(int timeslots,
                              int subcarriers,
                              int active_subcarriers,
                              std::vector<int> subcarrier_map,
                              bool per_timeslot = true,
                              bool is_mapper = true);
    ~resource_mapper_kernel_cc();
    resource_mapper_kernel_cc(const resource_mapper_kernel_cc&) = delete;
    resource_mapper_kernel_cc& operator=(const resource_mapper_kernel_cc&) = delete;

    size_t frame_size() { return d_frame_size; }
    size_t block_size() { return d_block_size; }
    size_t input_vector_size() { return d_is_mapper ? d_block_size : d_frame_size; }
    size_t output_vector_size() { return d_is_mapper ? d_frame_size : d_block_size; }
    void map_to_resources(gfdm_complex* p_out, const gfdm_complex* p_in, const size_t ninput_size);
    void demap_from_resources(gfdm_complex* p_out, const gfdm_complex* p_in, const size_t noutput_size);

    /* --- additions: whole batches per call (blocks back to back), host or device pointers --- */
    void map_to_resources_batch(gfdm_complex* out, const gfdm_complex* in, size_t ninput_size, long nblocks);
    void demap_from_resources_batch(gfdm_complex* out, const gfdm_complex* in, size_t noutput_size, long nblocks);
    void map_to_resources_device(void* d_out, const void* d_in, size_t ninput_size, long nblocks, void* hip_stream);
    void demap_from_resources_device(void* d_out, const void* d_in, size_t noutput_size, long nblocks, void* hip_stream);

private:
    size_t d_block_size;
    size_t d_frame_size;
    bool d_is_mapper;
    gfdm_hip_resource_mapper* d_handle;
};

} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_RESOURCE_MAPPER_KERNEL_CC_H */
