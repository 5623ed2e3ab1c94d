// gr::gfdm::advanced_receiver_kernel_cc over the HIP C-ABI (replaces lib/advanced_receiver_kernel_cc.cc of gr-gfdm).
#include <gfdm/advanced_receiver_kernel_cc.h>
#include <gfdm/preamble_channel_estimator_cc.h>
#include <gfdm_hip.h>
#include <cmath>
#include <limits>

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
inline float* fp(gr_complex_t* p) { return reinterpret_cast<float*>(p); }
inline const float* fp(const gr_complex_t* p) { return reinterpret_cast<const float*>(p); }
} // namespace

std::shared_ptr<constellation> constellation::qpsk()
{
    const float s = std::sqrt(0.5f);
    return std::make_shared<constellation>(std::vector<std::complex<float>>{ { -s, -s }, { s, -s }, { -s, s }, { s, s } }, QPSK);
}

std::shared_ptr<constellation> constellation::bpsk()
{
    return std::make_shared<constellation>(std::vector<std::complex<float>>{ { -1.f, 0.f }, { 1.f, 0.f } }, BPSK);
}

unsigned int constellation::decision_maker(const std::complex<float>* sample) const
{
    if (d_rule == QPSK) return 2u * (sample->imag() > 0.f) + (sample->real() > 0.f);
    if (d_rule == BPSK) return sample->real() > 0.f;
    unsigned int best = 0;
    float dist = std::numeric_limits<float>::infinity();
    for (unsigned int i = 0; i < d_points.size(); ++i) {
        const float d = std::norm(*sample - d_points[i]);
        if (d < dist) { dist = d; best = i; }
    }
    return best;
}

advanced_receiver_kernel_cc::advanced_receiver_kernel_cc(int timeslots, int subcarriers, int overlap,
                                                         std::vector<gr_complex_t> frequency_taps, std::vector<int> subcarrier_map,
                                                         int ic_iter, gr::gfdm::constellation_sptr constellation,
                                                         int do_phase_compensation)
    : d_block_len(timeslots * subcarriers), d_handle(nullptr)
{
    if (!constellation) throw std::invalid_argument("advanced_receiver_kernel_cc: constellation is NULL");
    const auto& pts = constellation->points();
    raise(gfdm_hip_advanced_receiver_create(&d_handle, timeslots, subcarriers, overlap, fp(frequency_taps.data()),
                                            static_cast<int>(frequency_taps.size()), subcarrier_map.data(),
                                            static_cast<int>(subcarrier_map.size()), ic_iter,
                                            reinterpret_cast<const float*>(pts.data()), static_cast<int>(pts.size()),
                                            static_cast<int>(constellation->rule()), do_phase_compensation, gfdm_kernel_utils::default_device()),
          "advanced_receiver_kernel_cc");
}

advanced_receiver_kernel_cc::~advanced_receiver_kernel_cc() { gfdm_hip_advanced_receiver_destroy(d_handle); }

void advanced_receiver_kernel_cc::generic_work(gr_complex_t* p_out, const gr_complex_t* p_in) { generic_work_batch(p_out, p_in, nullptr, 1); }

void advanced_receiver_kernel_cc::generic_work_equalize(gr_complex_t* out, const gr_complex_t* in, const gr_complex_t* f_eq_in)
{
    if (!f_eq_in) throw std::invalid_argument("generic_work_equalize: f_eq_in is NULL");
    generic_work_batch(out, in, f_eq_in, 1);
}

void advanced_receiver_kernel_cc::generic_work_batch(gr_complex_t* out, const gr_complex_t* in, const gr_complex_t* f_eq_in, long nblocks)
{
    raise(gfdm_hip_advanced_receiver_work_host(d_handle, fp(out), fp(in), fp(f_eq_in), nblocks), "advanced receiver generic_work");
}

void advanced_receiver_kernel_cc::generic_work_device(void* d_out, const void* d_in, const void* d_f_eq, long nblocks, void* hip_stream)
{
    raise(gfdm_hip_advanced_receiver_work_device(d_handle, d_out, d_in, d_f_eq, nblocks, hip_stream), "advanced receiver generic_work_device");
}

void advanced_receiver_kernel_cc::set_ic(int ic_iter) { raise(gfdm_hip_advanced_receiver_set_ic(d_handle, ic_iter), "set_ic"); }
int advanced_receiver_kernel_cc::get_ic(void) { return gfdm_hip_advanced_receiver_get_ic(d_handle); }
void advanced_receiver_kernel_cc::set_phase_compensation(int do_phase_compensation)
{
    raise(gfdm_hip_advanced_receiver_set_phase_compensation(d_handle, do_phase_compensation), "set_phase_compensation");
}
int advanced_receiver_kernel_cc::get_phase_compensation() { return gfdm_hip_advanced_receiver_get_phase_compensation(d_handle); }
void advanced_receiver_kernel_cc::configure_frames(int frame_len, int cp_len, std::vector<int> subcarrier_map, bool per_timeslot)
{
    raise(gfdm_hip_advanced_receiver_configure_frames(d_handle, frame_len, cp_len, subcarrier_map.data(),
                                                      static_cast<int>(subcarrier_map.size()), per_timeslot ? 1 : 0),
          "configure_frames");
}

void advanced_receiver_kernel_cc::generic_work_frames_batch(gr_complex_t* out, const gr_complex_t* in, const gr_complex_t* f_eq_in,
                                                            int noutput_size, long nframes)
{
    raise(gfdm_hip_advanced_receiver_work_frames_host(d_handle, fp(out), fp(in), fp(f_eq_in), noutput_size, nframes),
          "advanced receiver generic_work_frames");
}

void advanced_receiver_kernel_cc::generic_work_frames_device(void* d_out, const void* d_in, const void* d_f_eq, int noutput_size,
                                                             long nframes, void* hip_stream)
{
    raise(gfdm_hip_advanced_receiver_work_frames_device(d_handle, d_out, d_in, d_f_eq, noutput_size, nframes, hip_stream),
          "advanced receiver generic_work_frames_device");
}

void advanced_receiver_kernel_cc::set_channel_estimator(preamble_channel_estimator_cc* estimator)
{
    raise(gfdm_hip_advanced_receiver_set_channel_estimator(d_handle, estimator ? estimator->handle() : nullptr), "set_channel_estimator");
}

void advanced_receiver_kernel_cc::generic_work_estimated_batch(gr_complex_t* out, const gr_complex_t* in, const gr_complex_t* rx_preambles,
                                                               int preamble_stride, int noutput_size, long nblocks)
{
    raise(gfdm_hip_advanced_receiver_work_estimated_host(d_handle, fp(out), fp(in), fp(rx_preambles), preamble_stride, noutput_size, nblocks),
          "advanced receiver generic_work_estimated");
}

void advanced_receiver_kernel_cc::generic_work_estimated_device(void* d_out, const void* d_in, const void* d_rx_preambles, int preamble_stride,
                                                                int noutput_size, long nblocks, void* hip_stream)
{
    raise(gfdm_hip_advanced_receiver_work_estimated_device(d_handle, d_out, d_in, d_rx_preambles, preamble_stride, noutput_size, nblocks, hip_stream),
          "advanced receiver generic_work_estimated_device");
}

advanced_receiver_kernel_cc::io_layout_t advanced_receiver_kernel_cc::io_layout(bool estimated, int noutput_size) const
{
    io_layout_t ly{ 0, 0, 0 };
    raise(gfdm_hip_advanced_receiver_io_layout(d_handle, estimated ? 1 : 0, noutput_size, &ly.n_in, &ly.n_out, &ly.est_fft_len), "io_layout");
    return ly;
}

const char* advanced_receiver_kernel_cc::kernel_name() const { return gfdm_hip_advanced_receiver_kernel_name(d_handle); }

} // namespace gfdm
} // namespace gr
