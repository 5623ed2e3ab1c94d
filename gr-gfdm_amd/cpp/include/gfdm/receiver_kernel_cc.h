/* gr::gfdm::receiver_kernel_cc -- same public interface as gr-gfdm's
 * include/gfdm/receiver_kernel_cc.h:52-89, executed by the HIP kernels behind include/gfdm_hip.h.
 *
 * Drop-in for lib/simple_receiver_cc_impl.cc:50-51,71, lib/advanced_receiver_kernel_cc.cc:44-47
 * and python/bindings/demodulator_python.cc:35-205.
 */
#ifndef INCLUDED_GFDM_RECEIVER_KERNEL_CC_H
#define INCLUDED_GFDM_RECEIVER_KERNEL_CC_H

#include <gfdm/gfdm_kernel_utils.h>

struct gfdm_hip_receiver;

namespace gr {
namespace gfdm {

class preamble_channel_estimator_cc;

class GFDM_API receiver_kernel_cc : public gfdm_kernel_utils
{
public:
    typedef std::vector<std::vector<gfdm_complex>> matrix_t;

    /* throws std::invalid_argument for a wrong tap count or overlap < 2 */
    receiver_kernel_cc(int n_timeslots, int n_subcarriers, int overlap, std::vector<gfdm_complex> frequency_taps);
    ~receiver_kernel_cc();
    receiver_kernel_cc(const receiver_kernel_cc&) = delete;
    receiver_kernel_cc& operator=(const receiver_kernel_cc&) = delete;

    /* single-block, host-pointer calls with the reference semantics */
    void generic_work(gfdm_complex* out, const gfdm_complex* in);
    void generic_work_equalize(gfdm_complex* out, const gfdm_complex* in, const gfdm_complex* f_eq_in);
    void fft_filter_downsample(gfdm_complex* p_out, const gfdm_complex* p_in);
    void fft_equalize_filter_downsample(gfdm_complex* p_out, const gfdm_complex* p_in, const gfdm_complex* f_eq_in);
    void transform_subcarriers_to_td(gfdm_complex* p_out, const gfdm_complex* p_in);
    void cancel_sc_interference(gfdm_complex* p_out, const gfdm_complex* p_td_in, const gfdm_complex* p_fd_in);

    /* legacy 2-D vector API of the reference (no caller in gr-gfdm); routed through the same kernels */
    void filter_superposition(matrix_t& out, const gfdm_complex* in);
    void demodulate_subcarrier(matrix_t& out, matrix_t& sc_fdomain);
    void serialize_output(gfdm_complex out[], matrix_t& sc_symbols);
    void vectorize_2d(matrix_t& out_vector, const gfdm_complex* p_in);
    void remove_sc_interference(matrix_t& sc_symbols, matrix_t& sc_fdomain);

    int block_size() const { return d_block_len; }
    std::vector<gfdm_complex> filter_taps() const;
    std::vector<gfdm_complex> ic_filter_taps() const;
    int timeslots() const { return d_n_timeslots; }
    int subcarriers() const { return d_n_subcarriers; }
    int overlap() const { return d_overlap; }

    /* --- additions: whole batches per call (f_eq_in may be nullptr = no equaliser; one vector per block otherwise) --- */
    void generic_work_batch(gfdm_complex* out, const gfdm_complex* in, const gfdm_complex* f_eq_in, long nblocks);
    void generic_work_device(void* d_out, const void* d_in, const void* d_f_eq, long nblocks, void* hip_stream);
    void fft_filter_downsample_device(void* d_out, const void* d_in, const void* d_f_eq, long nblocks, void* hip_stream);
    void transform_subcarriers_to_td_device(void* d_out, const void* d_in, long nblocks, void* hip_stream);
    void cancel_sc_interference_device(void* d_out, const void* d_td_in, const void* d_fd_in, long nblocks, void* hip_stream);
    /* --- additions: raw frames in, demapped symbols out (what remove_prefix -> receiver -> resource_demapper compute in
     * the reference flowgraph, as ONE kernel).  configure_frames declares the layout once: frames of frame_len samples whose
     * block starts cp_len samples in; with a non-empty subcarrier_map only the active subcarriers' symbols are written, in
     * resource-mapper order, noutput_size per frame (<= 0: all).  f_eq_in (may be nullptr) stays one block per frame. --- */
    void configure_frames(int frame_len, int cp_len, std::vector<int> subcarrier_map, bool per_timeslot);
    void generic_work_frames_batch(gfdm_complex* out, const gfdm_complex* in, const gfdm_complex* f_eq_in, int noutput_size, long nframes);
    void generic_work_frames_device(void* d_out, const void* d_in, const void* d_f_eq, int noutput_size, long nframes, void* hip_stream);

    /* --- additions: the channel estimator fused in front (channel_estimator_cc -> f_eq input in the reference flowgraph, as ONE
     * kernel).  After set_channel_estimator, generic_work_estimated_* take each block's received core preamble (preamble b at
     * rx_preambles + b * preamble_stride, 0 = packed) instead of an equaliser vector; the block I/O follows configure_frames when
     * that was called, else plain blocks (noutput_size ignored).  The estimator object must outlive its use; nullptr detaches. --- */
    void set_channel_estimator(preamble_channel_estimator_cc* estimator);
    /* sizes of one generic_work_frames_* (estimated = false) or generic_work_estimated_* (true) call with this noutput_size, in
     * complex samples per frame / block -- asked from the library, which alone knows what its kernels write; throws where the call
     * itself would (no configure_frames / estimator, noutput_size without a subcarrier map or above active * timeslots) */
    struct io_layout_t { int n_in; int n_out; int est_fft_len; };
    io_layout_t io_layout(bool estimated, int noutput_size) const;
    void generic_work_estimated_batch(gfdm_complex* out, const gfdm_complex* in, const gfdm_complex* rx_preambles, int preamble_stride, int noutput_size, long nblocks);
    void generic_work_estimated_device(void* d_out, const void* d_in, const void* d_rx_preambles, int preamble_stride, int noutput_size, long nblocks,
                                       void* hip_stream);
    const char* kernel_name() const;

private:
    int d_n_subcarriers;
    int d_n_timeslots;
    int d_block_len;
    int d_overlap;
    gfdm_hip_receiver* d_handle;
    std::vector<gfdm_complex> d_flat_a, d_flat_b;   /* scratch of the legacy 2-D API */
};

} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_RECEIVER_KERNEL_CC_H */
