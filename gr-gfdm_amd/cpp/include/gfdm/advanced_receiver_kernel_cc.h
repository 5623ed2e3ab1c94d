/* gr::gfdm::advanced_receiver_kernel_cc -- interface of gr-gfdm's
 * include/gfdm/advanced_receiver_kernel_cc.h:37-78 (receiver + interference cancellation),
 * executed as ONE fused HIP kernel per batch behind include/gfdm_hip.h.
 *
 * Drop-in for lib/advanced_receiver_sb_cc_impl.cc:69-76,99,109.
 */
#ifndef INCLUDED_GFDM_ADVANCED_RECEIVER_KERNEL_CC_H
#define INCLUDED_GFDM_ADVANCED_RECEIVER_KERNEL_CC_H

#include <gfdm/constellation.h>
#include <gfdm/gfdm_kernel_utils.h>

#if defined(GFDM_WITH_GNURADIO) || (defined(__has_include) && __has_include(<gnuradio/digital/constellation.h>))
#include <gnuradio/digital/constellation.h>
#define GFDM_HAVE_GR_CONSTELLATION 1
#endif

struct gfdm_hip_advanced_receiver;

namespace gr {
namespace gfdm {

class preamble_channel_estimator_cc;

#ifndef GFDM_HAVE_GR_CONSTELLATION
typedef std::complex<float> gr_complex_t;
#else
typedef gr_complex gr_complex_t;
#endif

class GFDM_API advanced_receiver_kernel_cc
{
public:
    /* argument order of the reference constructor; `constellation` carries points + decision rule */
    advanced_receiver_kernel_cc(int timeslots,
                                int subcarriers,
                                int overlap,
                                std::vector<gr_complex_t> frequency_taps,
                                std::vector<int> subcarrier_map,
                                int ic_iter,
                                gr::gfdm::constellation_sptr constellation,
                                int do_phase_compensation);
#ifdef GFDM_HAVE_GR_CONSTELLATION
    /* the reference signature: points() is read once, the decision rule is inferred from them */
    advanced_receiver_kernel_cc(int timeslots, int subcarriers, int overlap, std::vector<gr_complex_t> frequency_taps,
                                std::vector<int> subcarrier_map, int ic_iter, gr::digital::constellation_sptr constellation,
                                int do_phase_compensation)
        : advanced_receiver_kernel_cc(timeslots, subcarriers, overlap, frequency_taps, subcarrier_map, ic_iter,
                                      gr::gfdm::constellation::from_points(constellation->points()), do_phase_compensation)
    {
    }
#endif
    ~advanced_receiver_kernel_cc();
    advanced_receiver_kernel_cc(const advanced_receiver_kernel_cc&) = delete;
    advanced_receiver_kernel_cc& operator=(const advanced_receiver_kernel_cc&) = delete;

    void generic_work(gr_complex_t* p_out, const gr_complex_t* p_in);
    void generic_work_equalize(gr_complex_t* out, const gr_complex_t* in, const gr_complex_t* f_eq_in);
    void set_ic(int ic_iter);
    int get_ic(void);
    int block_size() { return d_block_len; }
    void set_phase_compensation(int do_phase_compensation);
    int get_phase_compensation();

    /* --- additions: whole batches per call --- */
    void generic_work_batch(gr_complex_t* out, const gr_complex_t* in, const gr_complex_t* f_eq_in, long nblocks);
    void generic_work_device(void* d_out, const void* d_in, const void* d_f_eq, long nblocks, void* hip_stream);
    /* --- additions: raw frames in, demapped symbols out (what remove_prefix -> receiver -> resource_demapper compute in
     * the reference flowgraph, as ONE kernel).  configure_frames declares the layout once: frames of frame_len samples whose
     * block starts cp_len samples in; with a non-empty subcarrier_map only the active subcarriers' symbols are written, in
     * resource-mapper order, noutput_size per frame (<= 0: all).  f_eq_in (may be nullptr) stays one block per frame. --- */
    void configure_frames(int frame_len, int cp_len, std::vector<int> subcarrier_map, bool per_timeslot);
    void generic_work_frames_batch(gr_complex_t* out, const gr_complex_t* in, const gr_complex_t* f_eq_in, int noutput_size, long nframes);
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
    void generic_work_estimated_batch(gr_complex_t* out, const gr_complex_t* in, const gr_complex_t* rx_preambles, int preamble_stride, int noutput_size, long nblocks);
    void generic_work_estimated_device(void* d_out, const void* d_in, const void* d_rx_preambles, int preamble_stride, int noutput_size, long nblocks,
                                       void* hip_stream);
    const char* kernel_name() const;

private:
    int d_block_len;
    gfdm_hip_advanced_receiver* d_handle;
};

} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_ADVANCED_RECEIVER_KERNEL_CC_H */
