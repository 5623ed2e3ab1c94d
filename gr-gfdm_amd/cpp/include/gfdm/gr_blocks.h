/* GNU Radio blocks over the GPU-backed kernel classes, work() = one batched launch per scheduler call (batched_work.h).
 *
 * Compiled ONLY where GNU Radio's headers exist (__has_include below): this image and the GPU boxes have no GNU Radio, there this
 * header is empty and nothing stands in for the missing headers in the build.  (tests/test_boundary.py syntax-checks it against
 * tests/mock_gnuradio, a declarations-only test double of the block API names used here.)  In a gr-gfdm checkout the same one-line work() bodies replace the
 * per-block loops of lib/{simple_modulator_cc,simple_receiver_cc,advanced_receiver_sb_cc}_impl.cc; the
 * classes here are the stand-alone form for a flowgraph that links libgfdm_kernels.so directly.
 *
 * Item accounting and tag handling follow the reference: lib/simple_modulator_cc_impl.cc:44-80, lib/simple_receiver_cc_impl.cc:42-77,
 * lib/advanced_receiver_sb_cc_impl.cc:55-123, lib/transmitter_cc_impl.cc:64-195, lib/channel_estimator_cc_impl.cc:44-120,
 * lib/resource_mapper_cc_impl.cc:43-106, lib/resource_demapper_cc_impl.cc:43-105, lib/cyclic_prefixer_cc_impl.cc:43-110.
 */
#ifndef INCLUDED_GFDM_GR_BLOCKS_H
#define INCLUDED_GFDM_GR_BLOCKS_H

#if defined(__has_include)
#if __has_include(<gnuradio/sync_block.h>) && __has_include(<gnuradio/io_signature.h>) && __has_include(<pmt/pmt.h>)
#define GFDM_HAVE_GNURADIO 1
#endif
#endif

#ifdef GFDM_HAVE_GNURADIO
#include <gnuradio/io_signature.h>
#include <gnuradio/sync_block.h>
#include <pmt/pmt.h>

#include <gfdm/add_cyclic_prefix_cc.h>
#include <gfdm/advanced_receiver_kernel_cc.h>
#include <gfdm/batched_work.h>
#include <gfdm/modulator_kernel_cc.h>
#include <gfdm/preamble_channel_estimator_cc.h>
#include <gfdm/receiver_kernel_cc.h>
#include <gfdm/resource_mapper_kernel_cc.h>
#include <gfdm/transmitter_kernel.h>

#include <memory>
#include <string>

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

/* transmitter_cc (lib/transmitter_cc_impl.cc:64-195): fixed-rate general block, one output port per cyclic shift, the optional
 * tagged-stream length tag rewritten from input_vector_size to output_vector_size per frame.  All frames of a scheduler call and all
 * ports in ONE launch (batched::transmitter_work). */
class hip_transmitter_cc : public gr::block
{
public:
    typedef std::shared_ptr<hip_transmitter_cc> sptr;
    static sptr make(int timeslots, int subcarriers, int active_subcarriers, int cp_len, int cs_len, int ramp_len, std::vector<int> subcarrier_map,
                     bool per_timeslot, int overlap, std::vector<gr_complex> frequency_taps, std::vector<gr_complex> window_taps,
                     std::vector<int> cyclic_shifts, std::vector<std::vector<gr_complex>> preambles, const std::string& tsb_tag_key = "")
    {
        return sptr(new hip_transmitter_cc(timeslots, subcarriers, active_subcarriers, cp_len, cs_len, ramp_len, subcarrier_map, per_timeslot,
                                           overlap, frequency_taps, window_taps, cyclic_shifts, preambles, tsb_tag_key));
    }
    void forecast(int noutput_items, gr_vector_int& ninput_items_required) override
    {
        for (auto& n : ninput_items_required) n = fixed_rate_noutput_to_ninput(noutput_items);
    }
    int fixed_rate_ninput_to_noutput(int ninput) override { return (ninput / d_kernel->input_vector_size()) * d_kernel->output_vector_size(); }
    int fixed_rate_noutput_to_ninput(int noutput) override { return (noutput / d_kernel->output_vector_size()) * d_kernel->input_vector_size(); }

    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) override
    {
        std::vector<gr_complex*> outs(output_items.size());
        for (size_t i = 0; i < outs.size(); ++i) outs[i] = static_cast<gr_complex*>(output_items[i]);
        const int n_frames = batched::transmitter_work(*d_kernel, noutput_items, ninput_items[0], static_cast<const gr_complex*>(input_items[0]),
                                                       outs.data(), static_cast<int>(outs.size()));
        const int in_len = d_kernel->input_vector_size(), out_len = d_kernel->output_vector_size();
        if (!d_length_tag_key_str.empty()) {
            std::vector<tag_t> tags;
            get_tags_in_range(tags, 0, nitems_read(0), nitems_read(0) + static_cast<uint64_t>(n_frames) * in_len, d_length_tag_key);
            for (auto& tag : tags) remove_item_tag(0, tag);
            const pmt::pmt_t value = pmt::from_long(out_len);
            for (int i = 0; i < n_frames; ++i)
                for (unsigned port = 0; port < output_items.size(); ++port)
                    add_item_tag(port, nitems_written(port) + static_cast<uint64_t>(i) * out_len, d_length_tag_key, value);
        }
        consume_each(n_frames * in_len);
        return n_frames * out_len;
    }

private:
    hip_transmitter_cc(int timeslots, int subcarriers, int active_subcarriers, int cp_len, int cs_len, int ramp_len, std::vector<int> subcarrier_map,
                       bool per_timeslot, int overlap, std::vector<gr_complex> frequency_taps, std::vector<gr_complex> window_taps,
                       std::vector<int> cyclic_shifts, std::vector<std::vector<gr_complex>> preambles, const std::string& tsb_tag_key)
        : gr::block("hip_transmitter_cc", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                    gr::io_signature::make(static_cast<int>(cyclic_shifts.size()), static_cast<int>(cyclic_shifts.size()), sizeof(gr_complex))),
          d_kernel(std::make_unique<transmitter_kernel>(timeslots, subcarriers, active_subcarriers, cp_len, cs_len, ramp_len, subcarrier_map,
                                                        per_timeslot, overlap, frequency_taps, window_taps, cyclic_shifts, preambles)),
          d_length_tag_key_str(tsb_tag_key), d_length_tag_key(pmt::string_to_symbol(tsb_tag_key))
    {
        set_relative_rate(1.0 * d_kernel->output_vector_size() / d_kernel->input_vector_size());
        set_fixed_rate(true);
        set_output_multiple(d_kernel->output_vector_size());
    }
    std::unique_ptr<transmitter_kernel> d_kernel;
    std::string d_length_tag_key_str;
    pmt::pmt_t d_length_tag_key;
};

/* channel_estimator_cc (lib/channel_estimator_cc_impl.cc:44-120): 2 * fft_len preamble samples in, timeslots * fft_len estimate bins out
 * per frame, "snr_lin" and "cnr" tags on the first item of every frame.  Two launches per scheduler call (batched::estimator_work). */
class hip_channel_estimator_cc : public gr::block
{
public:
    typedef std::shared_ptr<hip_channel_estimator_cc> sptr;
    static sptr make(int timeslots, int fft_len, int active_subcarriers, bool is_dc_free, int which_estimator, std::vector<gr_complex> preamble)
    {
        return sptr(new hip_channel_estimator_cc(timeslots, fft_len, active_subcarriers, is_dc_free, which_estimator, preamble));
    }
    void forecast(int noutput_items, gr_vector_int& ninput_items_required) override
    {
        for (auto& n : ninput_items_required) n = fixed_rate_noutput_to_ninput(noutput_items);
    }
    int fixed_rate_ninput_to_noutput(int ninput) override { return ninput * d_kernel->timeslots() / 2; }
    int fixed_rate_noutput_to_ninput(int noutput) override { return 2 * noutput / d_kernel->timeslots(); }

    int general_work(int noutput_items, gr_vector_int&, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        const int frame_len = d_kernel->frame_len();
        const uint64_t first = nitems_written(0);
        const int n_frames = batched::estimator_work(
            *d_kernel, noutput_items, static_cast<const gr_complex*>(input_items[0]), static_cast<gr_complex*>(output_items[0]),
            [&](int i, float snr_lin, const float* cnrs, int n_cnrs) {
                add_item_tag(0, first + static_cast<uint64_t>(i) * frame_len, pmt::intern("snr_lin"), pmt::from_float(snr_lin));
                add_item_tag(0, first + static_cast<uint64_t>(i) * frame_len, pmt::intern("cnr"), pmt::init_f32vector(n_cnrs, cnrs));
            });
        consume_each(n_frames * 2 * d_kernel->fft_len());
        return n_frames * frame_len;
    }

private:
    hip_channel_estimator_cc(int timeslots, int fft_len, int active_subcarriers, bool is_dc_free, int which_estimator, std::vector<gr_complex> preamble)
        : gr::block("hip_channel_estimator_cc", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(gr_complex))),
          d_kernel(std::make_unique<preamble_channel_estimator_cc>(timeslots, fft_len, active_subcarriers, is_dc_free, which_estimator, preamble))
    {
        set_relative_rate(timeslots / 2.0);
        set_fixed_rate(true);
        set_output_multiple(fft_len * timeslots);
    }
    std::unique_ptr<preamble_channel_estimator_cc> d_kernel;
};

/* resource_mapper_cc / resource_demapper_cc (lib/resource_mapper_cc_impl.cc:43-106, lib/resource_demapper_cc_impl.cc:43-105): fixed-rate
 * general blocks over one resource_mapper_kernel_cc, direction chosen at construction. */
class hip_resource_mapper_cc : public gr::block
{
public:
    typedef std::shared_ptr<hip_resource_mapper_cc> sptr;
    static sptr make(int timeslots, int subcarriers, int active_subcarriers, std::vector<int> subcarrier_map, bool per_timeslot = true,
                     bool is_mapper = true)
    {
        return sptr(new hip_resource_mapper_cc(timeslots, subcarriers, active_subcarriers, subcarrier_map, per_timeslot, is_mapper));
    }
    void forecast(int noutput_items, gr_vector_int& ninput_items_required) override
    {
        ninput_items_required[0] = fixed_rate_noutput_to_ninput(noutput_items);
    }
    int fixed_rate_ninput_to_noutput(int ninput) override
    {
        return (ninput / static_cast<int>(d_kernel->input_vector_size())) * static_cast<int>(d_kernel->output_vector_size());
    }
    int fixed_rate_noutput_to_ninput(int noutput) override
    {
        return (noutput / static_cast<int>(d_kernel->output_vector_size())) * static_cast<int>(d_kernel->input_vector_size());
    }
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) override
    {
        const int n_frames = batched::mapper_work(*d_kernel, d_is_mapper, noutput_items, ninput_items[0], static_cast<const gr_complex*>(input_items[0]),
                                                  static_cast<gr_complex*>(output_items[0]));
        consume_each(n_frames * static_cast<int>(d_kernel->input_vector_size()));
        return n_frames * static_cast<int>(d_kernel->output_vector_size());
    }

private:
    hip_resource_mapper_cc(int timeslots, int subcarriers, int active_subcarriers, std::vector<int> subcarrier_map, bool per_timeslot, bool is_mapper)
        : gr::block(is_mapper ? "hip_resource_mapper_cc" : "hip_resource_demapper_cc", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                    gr::io_signature::make(1, 1, sizeof(gr_complex))),
          d_kernel(std::make_unique<resource_mapper_kernel_cc>(timeslots, subcarriers, active_subcarriers, subcarrier_map, per_timeslot, is_mapper)),
          d_is_mapper(is_mapper)
    {
        set_relative_rate(1.0 * d_kernel->output_vector_size() / d_kernel->input_vector_size());
        set_fixed_rate(true);
        set_output_multiple(static_cast<int>(d_kernel->output_vector_size()));
    }
    std::unique_ptr<resource_mapper_kernel_cc> d_kernel;
    bool d_is_mapper;
};

/* cyclic_prefixer_cc (lib/cyclic_prefixer_cc_impl.cc:43-110) */
class hip_cyclic_prefixer_cc : public gr::block
{
public:
    typedef std::shared_ptr<hip_cyclic_prefixer_cc> sptr;
    static sptr make(int block_len, int cp_len, int cs_len, int ramp_len, std::vector<gr_complex> window_taps)
    {
        return sptr(new hip_cyclic_prefixer_cc(block_len, cp_len, cs_len, ramp_len, window_taps));
    }
    void forecast(int noutput_items, gr_vector_int& ninput_items_required) override
    {
        for (auto& n : ninput_items_required) n = fixed_rate_noutput_to_ninput(noutput_items);
    }
    int fixed_rate_ninput_to_noutput(int ninput) override { return (ninput / d_kernel->block_size()) * d_kernel->frame_size(); }
    int fixed_rate_noutput_to_ninput(int noutput) override { return (noutput / d_kernel->frame_size()) * d_kernel->block_size(); }
    int general_work(int noutput_items, gr_vector_int&, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items) override
    {
        const int n_frames = batched::prefixer_work(*d_kernel, noutput_items, static_cast<const gr_complex*>(input_items[0]),
                                                    static_cast<gr_complex*>(output_items[0]));
        consume_each(n_frames * d_kernel->block_size());
        return n_frames * d_kernel->frame_size();
    }

private:
    hip_cyclic_prefixer_cc(int block_len, int cp_len, int cs_len, int ramp_len, std::vector<gr_complex> window_taps)
        : gr::block("hip_cyclic_prefixer_cc", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(1, 1, sizeof(gr_complex))),
          d_kernel(std::make_unique<add_cyclic_prefix_cc>(block_len, cp_len, cs_len, ramp_len, window_taps))
    {
        set_relative_rate(1.0 * d_kernel->frame_size() / d_kernel->block_size());
        set_fixed_rate(true);
        set_output_multiple(d_kernel->frame_size());
    }
    std::unique_ptr<add_cyclic_prefix_cc> d_kernel;
};

} // namespace gfdm
} // namespace gr
#endif /* GFDM_HAVE_GNURADIO */

#endif /* INCLUDED_GFDM_GR_BLOCKS_H */
