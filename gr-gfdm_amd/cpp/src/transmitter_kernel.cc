// gr::gfdm::transmitter_kernel over the HIP C-ABI (replaces lib/transmitter_kernel.cc, and with it the uses of
// resource_mapper_kernel_cc / add_cyclic_prefix_cc inside the transmitter, of gr-gfdm).
#include <gfdm/transmitter_kernel.h>
#include <gfdm_hip.h>

namespace gr {
namespace gfdm {

namespace {
void raise(int status, const char* where)
{
    if (status == GFDM_HIP_OK) return;
    const char* detail = gfdm_hip_last_error();
    std::string msg = (detail && *detail) ? detail : gfdm_hip_strerror(status);
    if (status == GFDM_HIP_EINVAL_TAPS || status == GFDM_HIP_EINVAL_OVERLAP || status == GFDM_HIP_EINVAL) throw std::invalid_argument(msg);
    throw std::runtime_error(std::string(where) + ": " + msg);
}
inline float* fp(transmitter_kernel::gfdm_complex* p) { return reinterpret_cast<float*>(p); }
inline const float* fp(const transmitter_kernel::gfdm_complex* p) { return reinterpret_cast<const float*>(p); }
} // namespace

transmitter_kernel::transmitter_kernel(int timeslots, int subcarriers, int active_subcarriers, int cp_len, int cs_len, int ramp_len,
                                       std::vector<int> subcarrier_map, bool per_timeslot, int overlap,
                                       std::vector<gfdm_complex> frequency_taps, std::vector<gfdm_complex> window_taps,
                                       std::vector<int> cyclic_shifts, std::vector<std::vector<gfdm_complex>> preambles)
    : d_cyclic_shifts(cyclic_shifts), d_handle(nullptr)
{
    if (cyclic_shifts.size() != preambles.size() || preambles.empty())            // lib/transmitter_kernel.cc:57-60
        throw std::invalid_argument("Number of cyclic shifts and number of preambles do not match!");
    const size_t plen = preambles[0].size();
    std::vector<gfdm_complex> flat;
    flat.reserve(plen * preambles.size());
    for (const auto& p : preambles) {
        if (p.size() != plen) throw std::invalid_argument("All preambles must have equal size!");   // :62-66
        flat.insert(flat.end(), p.begin(), p.end());
    }
    raise(gfdm_hip_transmitter_create(&d_handle, timeslots, subcarriers, active_subcarriers, cp_len, cs_len, ramp_len,
                                      subcarrier_map.data(), static_cast<int>(subcarrier_map.size()), per_timeslot ? 1 : 0, overlap,
                                      fp(frequency_taps.data()), static_cast<int>(frequency_taps.size()), fp(window_taps.data()),
                                      static_cast<int>(window_taps.size()), cyclic_shifts.data(), static_cast<int>(cyclic_shifts.size()),
                                      fp(flat.data()), static_cast<int>(plen), gfdm_kernel_utils::default_device()),
          "transmitter_kernel");
}

transmitter_kernel::~transmitter_kernel() { gfdm_hip_transmitter_destroy(d_handle); }

int transmitter_kernel::input_vector_size() { return gfdm_hip_transmitter_input_vector_size(d_handle); }
int transmitter_kernel::output_vector_size() { return gfdm_hip_transmitter_output_vector_size(d_handle); }

void transmitter_kernel::generic_work(gfdm_complex* p_out, const gfdm_complex* p_in, const int ninput_size)
{
    gfdm_complex* outs[1] = { p_out };
    generic_work_batch(outs, 1, p_in, ninput_size, 1);
}

void transmitter_kernel::generic_work_batch(gfdm_complex* const* outs, int n_ports, const gfdm_complex* in, int ninput_size, long nframes)
{
    std::vector<float*> raw(n_ports > 0 ? n_ports : 0);
    for (int i = 0; i < n_ports; ++i) raw[i] = fp(outs[i]);
    raise(gfdm_hip_transmitter_work_host(d_handle, raw.data(), n_ports, fp(in), ninput_size, nframes), "transmitter generic_work");
}

void transmitter_kernel::generic_work_device(void* const* d_outs, int n_ports, const void* d_in, int ninput_size, long nframes, void* hip_stream)
{
    raise(gfdm_hip_transmitter_work_device(d_handle, d_outs, n_ports, d_in, ninput_size, nframes, hip_stream), "transmitter generic_work_device");
}

void transmitter_kernel::modulate(gfdm_complex* out, const gfdm_complex* in, const int ninput_size)
{
    raise(gfdm_hip_transmitter_modulate_host(d_handle, fp(out), fp(in), ninput_size, 1), "transmitter modulate");
}

void transmitter_kernel::add_frame(gfdm_complex* out, const gfdm_complex* in, const int cyclic_shift)
{
    raise(gfdm_hip_transmitter_add_frame_host(d_handle, fp(out), fp(in), cyclic_shift, 1), "transmitter add_frame");
}

const char* transmitter_kernel::kernel_name() const { return gfdm_hip_transmitter_kernel_name(d_handle); }

} // namespace gfdm
} // namespace gr
