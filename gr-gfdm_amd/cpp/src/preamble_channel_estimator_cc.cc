// gr::gfdm::preamble_channel_estimator_cc over the HIP C-ABI (replaces lib/preamble_channel_estimator_cc.cc of gr-gfdm).
#include <gfdm/preamble_channel_estimator_cc.h>
#include <gfdm_hip.h>

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
typedef preamble_channel_estimator_cc::gfdm_complex cplx;
inline float* fp(cplx* p) { return reinterpret_cast<float*>(p); }
inline const float* fp(const cplx* p) { return reinterpret_cast<const float*>(p); }
} // namespace

preamble_channel_estimator_cc::preamble_channel_estimator_cc(int timeslots, int fft_len, int active_subcarriers, bool is_dc_free,
                                                             int which_estimator, std::vector<gfdm_complex> preamble)
    : d_timeslots(timeslots), d_fft_len(fft_len), d_active_subcarriers(active_subcarriers), d_is_dc_free(is_dc_free),
      d_which_estimator(which_estimator), d_handle(nullptr)
{
    raise(gfdm_hip_channel_estimator_create(&d_handle, timeslots, fft_len, active_subcarriers, is_dc_free ? 1 : 0, which_estimator,
                                            fp(preamble.data()), static_cast<int>(preamble.size()), default_device()),
          "preamble_channel_estimator_cc");
}

preamble_channel_estimator_cc::~preamble_channel_estimator_cc() { gfdm_hip_channel_estimator_destroy(d_handle); }

std::vector<float> preamble_channel_estimator_cc::preamble_filter_taps()
{
    std::vector<float> t(9);
    gfdm_hip_channel_estimator_preamble_filter_taps(d_handle, t.data());
    return t;
}

void preamble_channel_estimator_cc::estimate_preamble_channel(gfdm_complex* fd_preamble_channel, const gfdm_complex* rx_preamble)
{
    raise(gfdm_hip_channel_estimator_estimate_preamble_channel_host(d_handle, fp(fd_preamble_channel), fp(rx_preamble), 1), "estimate_preamble_channel");
}

void preamble_channel_estimator_cc::filter_preamble_estimate(gfdm_complex* filtered, const gfdm_complex* estimate)
{
    raise(gfdm_hip_channel_estimator_filter_preamble_estimate_host(d_handle, fp(filtered), fp(estimate), 1), "filter_preamble_estimate");
}

void preamble_channel_estimator_cc::interpolate_frame(gfdm_complex* frame_estimate, const gfdm_complex* estimate)
{
    raise(gfdm_hip_channel_estimator_interpolate_frame_host(d_handle, fp(frame_estimate), fp(estimate), 1), "interpolate_frame");
}

void preamble_channel_estimator_cc::estimate_frame(gfdm_complex* frame_estimate, const gfdm_complex* rx_preamble)
{
    estimate_frame_batch(frame_estimate, rx_preamble, 1);
}

void preamble_channel_estimator_cc::estimate_frame_batch(gfdm_complex* frame_estimates, const gfdm_complex* rx_preambles, long nframes)
{
    raise(gfdm_hip_channel_estimator_estimate_frame_host(d_handle, fp(frame_estimates), fp(rx_preambles), nframes), "estimate_frame");
}

void preamble_channel_estimator_cc::estimate_frame_device(void* d_frame_estimates, const void* d_rx_preambles, long nframes, void* hip_stream)
{
    raise(gfdm_hip_channel_estimator_estimate_frame_device(d_handle, d_frame_estimates, d_rx_preambles, nframes, hip_stream), "estimate_frame");
}

void preamble_channel_estimator_cc::prepare_for_zf(gfdm_complex* transformed_frame, const gfdm_complex* frame_estimate)
{
    raise(gfdm_hip_channel_estimator_prepare_for_zf_host(d_handle, fp(transformed_frame), fp(frame_estimate), 1), "prepare_for_zf");
}

float preamble_channel_estimator_cc::estimate_snr(std::vector<float>& cnrs, const gfdm_complex* rx_preamble)
{
    cnrs.resize(d_active_subcarriers);
    float snr_lin = 0.0f;
    estimate_snr_batch(&snr_lin, cnrs.data(), rx_preamble, 1);
    return snr_lin;
}

void preamble_channel_estimator_cc::estimate_snr_batch(float* snr_lin, float* cnrs, const gfdm_complex* rx_preambles, long nframes)
{
    raise(gfdm_hip_channel_estimator_estimate_snr_host(d_handle, snr_lin, cnrs, fp(rx_preambles), nframes), "estimate_snr");
}

} // namespace gfdm
} // namespace gr
