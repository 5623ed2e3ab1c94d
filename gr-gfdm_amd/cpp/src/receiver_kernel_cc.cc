// gr::gfdm::receiver_kernel_cc over the HIP C-ABI (replaces lib/receiver_kernel_cc.cc of gr-gfdm).
#include <gfdm/receiver_kernel_cc.h>
#include <gfdm/preamble_channel_estimator_cc.h>
#include <gfdm_hip.h>
#include <algorithm>

namespace gr {
namespace gfdm {

namespace {
inline float* fp(receiver_kernel_cc::gfdm_complex* p) { return reinterpret_cast<float*>(p); }
inline const float* fp(const receiver_kernel_cc::gfdm_complex* p) { return reinterpret_cast<const float*>(p); }
} // namespace

receiver_kernel_cc::receiver_kernel_cc(int n_timeslots, int n_subcarriers, int overlap, std::vector<gfdm_complex> frequency_taps)
    : d_n_subcarriers(n_subcarriers), d_n_timeslots(n_timeslots), d_block_len(n_timeslots * n_subcarriers), d_overlap(overlap),
      d_handle(nullptr)
{
    throw_on_error(gfdm_hip_receiver_create(&d_handle, n_timeslots, n_subcarriers, overlap, fp(frequency_taps.data()),
                                            static_cast<int>(frequency_taps.size()), default_device()),
                   "receiver_kernel_cc");
}

receiver_kernel_cc::~receiver_kernel_cc() { gfdm_hip_receiver_destroy(d_handle); }

std::vector<receiver_kernel_cc::gfdm_complex> receiver_kernel_cc::filter_taps() const
{
    std::vector<gfdm_complex> taps(static_cast<size_t>(d_n_timeslots) * d_overlap);
    throw_on_error(gfdm_hip_receiver_filter_taps(d_handle, fp(taps.data())), "filter_taps");
    return taps;
}

std::vector<receiver_kernel_cc::gfdm_complex> receiver_kernel_cc::ic_filter_taps() const
{
    std::vector<gfdm_complex> taps(static_cast<size_t>(d_n_timeslots));
    throw_on_error(gfdm_hip_receiver_ic_filter_taps(d_handle, fp(taps.data())), "ic_filter_taps");
    return taps;
}

void receiver_kernel_cc::generic_work(gfdm_complex* out, const gfdm_complex* in) { generic_work_batch(out, in, nullptr, 1); }

void receiver_kernel_cc::generic_work_equalize(gfdm_complex* out, const gfdm_complex* in, const gfdm_complex* f_eq_in)
{
    if (!f_eq_in) throw std::invalid_argument("generic_work_equalize: f_eq_in is NULL");
    generic_work_batch(out, in, f_eq_in, 1);
}

void receiver_kernel_cc::generic_work_batch(gfdm_complex* out, const gfdm_complex* in, const gfdm_complex* f_eq_in, long nblocks)
{
    throw_on_error(gfdm_hip_receiver_demodulate_host(d_handle, fp(out), fp(in), fp(f_eq_in), nblocks), "receiver generic_work");
}

void receiver_kernel_cc::fft_filter_downsample(gfdm_complex* p_out, const gfdm_complex* p_in)
{
    throw_on_error(gfdm_hip_receiver_fft_filter_downsample_host(d_handle, fp(p_out), fp(p_in), nullptr, 1), "fft_filter_downsample");
}

void receiver_kernel_cc::fft_equalize_filter_downsample(gfdm_complex* p_out, const gfdm_complex* p_in, const gfdm_complex* f_eq_in)
{
    if (!f_eq_in) throw std::invalid_argument("fft_equalize_filter_downsample: f_eq_in is NULL");
    throw_on_error(gfdm_hip_receiver_fft_filter_downsample_host(d_handle, fp(p_out), fp(p_in), fp(f_eq_in), 1),
                   "fft_equalize_filter_downsample");
}

void receiver_kernel_cc::transform_subcarriers_to_td(gfdm_complex* p_out, const gfdm_complex* p_in)
{
    throw_on_error(gfdm_hip_receiver_transform_subcarriers_to_td_host(d_handle, fp(p_out), fp(p_in), 1), "transform_subcarriers_to_td");
}

void receiver_kernel_cc::cancel_sc_interference(gfdm_complex* p_out, const gfdm_complex* p_td_in, const gfdm_complex* p_fd_in)
{
    throw_on_error(gfdm_hip_receiver_cancel_sc_interference_host(d_handle, fp(p_out), fp(p_td_in), fp(p_fd_in), 1),
                   "cancel_sc_interference");
}

void receiver_kernel_cc::generic_work_device(void* d_out, const void* d_in, const void* d_f_eq, long nblocks, void* hip_stream)
{
    throw_on_error(gfdm_hip_receiver_demodulate_device(d_handle, d_out, d_in, d_f_eq, nblocks, hip_stream), "receiver generic_work_device");
}

void receiver_kernel_cc::fft_filter_downsample_device(void* d_out, const void* d_in, const void* d_f_eq, long nblocks, void* hip_stream)
{
    throw_on_error(gfdm_hip_receiver_fft_filter_downsample_device(d_handle, d_out, d_in, d_f_eq, nblocks, hip_stream),
                   "fft_filter_downsample_device");
}

void receiver_kernel_cc::transform_subcarriers_to_td_device(void* d_out, const void* d_in, long nblocks, void* hip_stream)
{
    throw_on_error(gfdm_hip_receiver_transform_subcarriers_to_td_device(d_handle, d_out, d_in, nblocks, hip_stream),
                   "transform_subcarriers_to_td_device");
}

void receiver_kernel_cc::cancel_sc_interference_device(void* d_out, const void* d_td_in, const void* d_fd_in, long nblocks, void* hip_stream)
{
    throw_on_error(gfdm_hip_receiver_cancel_sc_interference_device(d_handle, d_out, d_td_in, d_fd_in, nblocks, hip_stream),
                   "cancel_sc_interference_device");
}

void receiver_kernel_cc::configure_frames(int frame_len, int cp_len, std::vector<int> subcarrier_map, bool per_timeslot)
{
    throw_on_error(gfdm_hip_receiver_configure_frames(d_handle, frame_len, cp_len, subcarrier_map.data(),
                                                      static_cast<int>(subcarrier_map.size()), per_timeslot ? 1 : 0),
                   "configure_frames");
}

void receiver_kernel_cc::generic_work_frames_batch(gfdm_complex* out, const gfdm_complex* in, const gfdm_complex* f_eq_in, int noutput_size,
                                                   long nframes)
{
    throw_on_error(gfdm_hip_receiver_demodulate_frames_host(d_handle, fp(out), fp(in), fp(f_eq_in), noutput_size, nframes),
                   "receiver generic_work_frames");
}

void receiver_kernel_cc::generic_work_frames_device(void* d_out, const void* d_in, const void* d_f_eq, int noutput_size, long nframes,
                                                    void* hip_stream)
{
    throw_on_error(gfdm_hip_receiver_demodulate_frames_device(d_handle, d_out, d_in, d_f_eq, noutput_size, nframes, hip_stream),
                   "receiver generic_work_frames_device");
}

void receiver_kernel_cc::set_channel_estimator(preamble_channel_estimator_cc* estimator)
{
    throw_on_error(gfdm_hip_receiver_set_channel_estimator(d_handle, estimator ? estimator->handle() : nullptr), "set_channel_estimator");
}

void receiver_kernel_cc::generic_work_estimated_batch(gfdm_complex* out, const gfdm_complex* in, const gfdm_complex* rx_preambles,
                                                      int preamble_stride, int noutput_size, long nblocks)
{
    throw_on_error(gfdm_hip_receiver_demodulate_estimated_host(d_handle, fp(out), fp(in), fp(rx_preambles), preamble_stride, noutput_size, nblocks),
                   "receiver generic_work_estimated");
}

void receiver_kernel_cc::generic_work_estimated_device(void* d_out, const void* d_in, const void* d_rx_preambles, int preamble_stride,
                                                       int noutput_size, long nblocks, void* hip_stream)
{
    throw_on_error(gfdm_hip_receiver_demodulate_estimated_device(d_handle, d_out, d_in, d_rx_preambles, preamble_stride, noutput_size, nblocks,
                                                                 hip_stream),
                   "receiver generic_work_estimated_device");
}

receiver_kernel_cc::io_layout_t receiver_kernel_cc::io_layout(bool estimated, int noutput_size) const
{
    io_layout_t ly{ 0, 0, 0 };
    throw_on_error(gfdm_hip_receiver_io_layout(d_handle, estimated ? 1 : 0, noutput_size, &ly.n_in, &ly.n_out, &ly.est_fft_len), "io_layout");
    return ly;
}

const char* receiver_kernel_cc::kernel_name() const { return gfdm_hip_receiver_kernel_name(d_handle); }

// ---- legacy 2-D API: [subcarrier][timeslot] vectors <-> the flat subcarrier-major block ----

void receiver_kernel_cc::vectorize_2d(matrix_t& out_vector, const gfdm_complex* p_in)
{
    for (int k = 0; k < d_n_subcarriers; ++k) std::copy_n(p_in + static_cast<size_t>(k) * d_n_timeslots, d_n_timeslots, out_vector[k].begin());
}

void receiver_kernel_cc::serialize_output(gfdm_complex out[], matrix_t& sc_symbols)
{
    for (int k = 0; k < d_n_subcarriers; ++k) std::copy_n(sc_symbols[k].begin(), d_n_timeslots, out + static_cast<size_t>(k) * d_n_timeslots);
}

void receiver_kernel_cc::filter_superposition(matrix_t& out, const gfdm_complex* in)
{
    d_flat_a.resize(d_block_len);
    fft_filter_downsample(d_flat_a.data(), in);
    vectorize_2d(out, d_flat_a.data());
}

void receiver_kernel_cc::demodulate_subcarrier(matrix_t& out, matrix_t& sc_fdomain)
{
    d_flat_a.resize(d_block_len);
    d_flat_b.resize(d_block_len);
    serialize_output(d_flat_a.data(), sc_fdomain);
    transform_subcarriers_to_td(d_flat_b.data(), d_flat_a.data());
    vectorize_2d(out, d_flat_b.data());
}

void receiver_kernel_cc::remove_sc_interference(matrix_t& sc_symbols, matrix_t& sc_fdomain)
{
    std::vector<gfdm_complex> td(d_block_len), fd(d_block_len), res(d_block_len);
    serialize_output(td.data(), sc_symbols);
    serialize_output(fd.data(), sc_fdomain);
    cancel_sc_interference(res.data(), td.data(), fd.data());
    vectorize_2d(sc_symbols, res.data());
}

} // namespace gfdm
} // namespace gr
