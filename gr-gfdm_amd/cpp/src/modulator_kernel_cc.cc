// gr::gfdm::modulator_kernel_cc over the HIP C-ABI (replaces lib/modulator_kernel_cc.cc of gr-gfdm).
#include <gfdm/modulator_kernel_cc.h>
#include <gfdm_hip.h>

namespace gr {
namespace gfdm {

modulator_kernel_cc::modulator_kernel_cc(int n_timeslots, int n_subcarriers, int overlap, std::vector<gfdm_complex> frequency_taps)
    : d_n_timeslots(n_timeslots), d_n_subcarriers(n_subcarriers), d_overlap(overlap), d_handle(nullptr)
{
    throw_on_error(gfdm_hip_modulator_create(&d_handle, n_timeslots, n_subcarriers, overlap,
                                             reinterpret_cast<const float*>(frequency_taps.data()),
                                             static_cast<int>(frequency_taps.size()), default_device()),
                   "modulator_kernel_cc");
}

modulator_kernel_cc::~modulator_kernel_cc() { gfdm_hip_modulator_destroy(d_handle); }

std::vector<modulator_kernel_cc::gfdm_complex> modulator_kernel_cc::filter_taps()
{
    std::vector<gfdm_complex> taps(static_cast<size_t>(d_n_timeslots) * d_overlap);
    throw_on_error(gfdm_hip_modulator_filter_taps(d_handle, reinterpret_cast<float*>(taps.data())), "filter_taps");
    return taps;
}

void modulator_kernel_cc::generic_work(gfdm_complex* p_out, const gfdm_complex* p_in) { generic_work_batch(p_out, p_in, 1); }

void modulator_kernel_cc::generic_work_batch(gfdm_complex* p_out, const gfdm_complex* p_in, long nblocks)
{
    throw_on_error(gfdm_hip_modulator_work_host(d_handle, reinterpret_cast<float*>(p_out), reinterpret_cast<const float*>(p_in), nblocks),
                   "modulator generic_work");
}

void modulator_kernel_cc::generic_work_device(void* d_out, const void* d_in, long nblocks, void* hip_stream)
{
    throw_on_error(gfdm_hip_modulator_work_device(d_handle, d_out, d_in, nblocks, hip_stream), "modulator generic_work_device");
}

const char* modulator_kernel_cc::kernel_name() const { return gfdm_hip_modulator_kernel_name(d_handle); }

} // namespace gfdm
} // namespace gr
