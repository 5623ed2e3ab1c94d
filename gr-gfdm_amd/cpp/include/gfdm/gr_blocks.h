/* GNU Radio blocks over the GPU-backed kernel classes, work() = one batched launch per scheduler call (batched_work.h).
 *
 * Compiled ONLY where GNU Radio's headers exist (__has_include below): this image and the GPU boxes have no GNU Radio, there this
 * header is empty and nothing stands in for the missing headers.  In a gr-gfdm checkout the same one-line work() bodies replace the
 * per-block loops of lib/{simple_modulator_cc,simple_receiver_cc,advanced_receiver_sb_cc}_impl.cc; the
 * classes here are the stand-alone form for a flowgraph that links libgfdm_kernels.so directly.
 *
 * Item accounting and tag handling follow the reference: lib/simple_modulator_cc_impl.cc:44-80, lib/simple_receiver_cc_impl.cc:42-77,
 * lib/advanced_receiver_sb_cc_impl.cc:55-123.
 */
#ifndef INCLUDED_GFDM_GR_BLOCKS_H
#define INCLUDED_GFDM_GR_BLOCKS_H

#if defined(__has_include)
#if __has_include(<gnuradio/sync_block.h>) && __has_include(<gnuradio/io_signature.h>)
#define GFDM_HAVE_GNURADIO 1
#endif
#endif

#ifdef GFDM_HAVE_GNURADIO
#include <gnuradio/io_signature.h>
#include <gnuradio/sync_block.h>

#include <gfdm/advanced_receiver_kernel_cc.h>
#include <gfdm/batched_work.h>
#include <gfdm/modulator_kernel_cc.h>
#include <gfdm/receiver_kernel_cc.h>

#include <memory>

namespace gr {
namespace gfdm {

class hip_simple_modulator_cc : public gr::sync_block
{
public:
    typedef std::shared_ptr<hip_simple_modulator_cc> sptr;
    static sptr make(int n_timeslots, int n_subcarriers, int overlap, std::vector<gr_complex> frequency_taps)
    {
        return sptr(new hip_simple_modulator_cc(n_timeslots, n_subcarriers, overlap, frequency_taps));
    }
    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        return batched::sync_work(*d_kernel, noutput_items, static_cast<const gr_complex*>(input_items[0]), static_cast<gr_complex*>(output_items[0]));
    }

private:
    hip_simple_modulator_cc(int n_timeslots, int n_subcarriers, int overlap, std::vector<gr_complex> frequency_taps)
        : gr::sync_block("hip_simple_modulator_cc", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(gr_complex))),
          d_kernel(std::make_unique<modulator_kernel_cc>(n_timeslots, n_subcarriers, overlap, frequency_taps))
    {
        set_output_multiple(d_kernel->block_size());
    }
    std::unique_ptr<modulator_kernel_cc> d_kernel;
};

class hip_simple_receiver_cc : public gr::sync_block
{
public:
    typedef std::shared_ptr<hip_simple_receiver_cc> sptr;
    static sptr make(int n_timeslots, int n_subcarriers, int overlap, std::vector<gr_complex> frequency_taps)
    {
        return sptr(new hip_simple_receiver_cc(n_timeslots, n_subcarriers, overlap, frequency_taps));
    }
    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        return batched::sync_work(*d_kernel, noutput_items, static_cast<const gr_complex*>(input_items[0]), static_cast<gr_complex*>(output_items[0]));
    }

private:
    hip_simple_receiver_cc(int n_timeslots, int n_subcarriers, int overlap, std::vector<gr_complex> frequency_taps)
        : gr::sync_block("hip_simple_receiver_cc", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(gr_complex))),
          d_kernel(std::make_unique<receiver_kernel_cc>(n_timeslots, n_subcarriers, overlap, frequency_taps))
    {
        set_output_multiple(d_kernel->block_size());
    }
    std::unique_ptr<receiver_kernel_cc> d_kernel;
};

class hip_advanced_receiver_sb_cc : public gr::sync_block
{
public:
    typedef std::shared_ptr<hip_advanced_receiver_sb_cc> sptr;
    static sptr make(int n_timeslots, int n_subcarriers, int overlap, int ic_iter, std::vector<gr_complex> frequency_taps,
                     gr::gfdm::constellation_sptr constellation, std::vector<int> subcarrier_map, int do_phase_compensation)
    {
        return sptr(new hip_advanced_receiver_sb_cc(n_timeslots, n_subcarriers, overlap, ic_iter, frequency_taps, constellation, subcarrier_map,
                                                    do_phase_compensation));
    }
    void set_ic(int ic_iter) { d_kernel->set_ic(ic_iter); }
    int get_ic() { return d_kernel->get_ic(); }
    void set_phase_compensation(int v) { d_kernel->set_phase_compensation(v); }
    int get_phase_compensation() { return d_kernel->get_phase_compensation(); }

    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        const bool eq = input_items.size() > 1;                       // port 1 connected: per-block equaliser vectors
        const int produced = batched::sync_work_equalize(*d_kernel, noutput_items, static_cast<const gr_complex*>(input_items[0]),
                                                         eq ? static_cast<const gr_complex*>(input_items[1]) : nullptr,
                                                         static_cast<gr_complex*>(output_items[0]));
        std::vector<tag_t> tags;                                      // lib/advanced_receiver_sb_cc_impl.cc:106-120
        get_tags_in_window(tags, eq ? 1 : 0, 0, produced);
        for (auto& t : tags) add_item_tag(0, t);
        return produced;
    }

private:
    hip_advanced_receiver_sb_cc(int n_timeslots, int n_subcarriers, int overlap, int ic_iter, std::vector<gr_complex> frequency_taps,
                                gr::gfdm::constellation_sptr constellation, std::vector<int> subcarrier_map, int do_phase_compensation)
        : gr::sync_block("hip_advanced_receiver_sb_cc", gr::io_signature::make(1, 2, sizeof(gr_complex)),
                         gr::io_signature::make(1, 1, sizeof(gr_complex))),
          d_kernel(std::make_unique<advanced_receiver_kernel_cc>(n_timeslots, n_subcarriers, overlap, frequency_taps, subcarrier_map, ic_iter,
                                                                 constellation, do_phase_compensation))
    {
        set_output_multiple(d_kernel->block_size());
        set_tag_propagation_policy(TPP_DONT);
    }
    std::unique_ptr<advanced_receiver_kernel_cc> d_kernel;
};

} // namespace gfdm
} // namespace gr
#endif /* GFDM_HAVE_GNURADIO */

#endif /* INCLUDED_GFDM_GR_BLOCKS_H */
