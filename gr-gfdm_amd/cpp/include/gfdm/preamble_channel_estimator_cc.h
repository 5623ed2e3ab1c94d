/* gr::gfdm::preamble_channel_estimator_cc -- public interface of gr-gfdm's
 * include/gfdm/preamble_channel_estimator_cc.h:45-78 (received preamble -> frequency-domain channel estimate of a frame),
 * executed as one LDS-resident HIP kernel per call behind include/gfdm_hip.h.
 * Drop-in for lib/channel_estimator_cc_impl.cc and the estimator inside lib/receiver_cc_impl.cc.
 */
#ifndef INCLUDED_GFDM_PREAMBLE_CHANNEL_ESTIMATOR_CC_H
#define INCLUDED_GFDM_PREAMBLE_CHANNEL_ESTIMATOR_CC_H

#include <gfdm/gfdm_kernel_utils.h>

struct gfdm_hip_channel_estimator;

namespace gr {
namespace gfdm {

class GFDM_API preamble_channel_estimator_cc : public gfdm_kernel_utils
{
public:
    /* throws std::invalid_argument for unusable arguments, std::runtime_error when no GPU is usable */
    preamble_channel_estimator_cc(int timeslots,
                                  int fft_len,
                                  int active_subcarriers,
                                  bool is_dc_free,
                                  int which_estimator,
                                  std::vector<gfdm_complex> preamble);
    ~preamble_channel_estimator_cc();
    preamble_channel_estimator_cc(const preamble_channel_estimator_cc&) = delete;
    preamble_channel_estimator_cc& operator=(const preamble_channel_estimator_cc&) = delete;

    void estimate_preamble_channel(gfdm_complex* fd_preamble_channel, const gfdm_complex* rx_preamble);
    int fft_len() { return d_fft_len; };
    int timeslots() { return d_timeslots; };
    int frame_len() { return d_timeslots * d_fft_len; };
    int active_subcarriers() { return d_active_subcarriers; };
    bool is_dc_free() { return d_is_dc_free; };
    std::vector<float> preamble_filter_taps();
    void filter_preamble_estimate(gfdm_complex* filtered, const gfdm_complex* estimate);
    void interpolate_frame(gfdm_complex* frame_estimate, const gfdm_complex* estimate);
    void estimate_frame(gfdm_complex* frame_estimate, const gfdm_complex* rx_preamble);
    void prepare_for_zf(gfdm_complex* transformed_frame, const gfdm_complex* frame_estimate);
    float estimate_snr(std::vector<float>& cnrs, const gfdm_complex* rx_preamble);

    /* --- additions: whole batches per call (nframes preambles / estimates back to back) --- */
    void estimate_frame_batch(gfdm_complex* frame_estimates, const gfdm_complex* rx_preambles, long nframes);
    void estimate_frame_device(void* d_frame_estimates, const void* d_rx_preambles, long nframes, void* hip_stream);
    /* snr_lin[nframes], cnrs[nframes * active_subcarriers] */
    void estimate_snr_batch(float* snr_lin, float* cnrs, const gfdm_complex* rx_preambles, long nframes);
    /* C-ABI handle, for receiver_kernel_cc / advanced_receiver_kernel_cc::set_channel_estimator */
    const gfdm_hip_channel_estimator* handle() const { return d_handle; }

private:
    int d_timeslots;
    int d_fft_len;
    int d_active_subcarriers;
    bool d_is_dc_free;
    int d_which_estimator;
    gfdm_hip_channel_estimator* d_handle;
};

} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_PREAMBLE_CHANNEL_ESTIMATOR_CC_H */
