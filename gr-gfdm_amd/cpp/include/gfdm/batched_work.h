/* work() bodies of gr-gfdm's GNU Radio block wrappers, with the per-block loop replaced by ONE batched launch.
 *
 * The reference wrappers hand their kernel object one block per call:
 *     lib/simple_modulator_cc_impl.cc:62-80        for (i < n_blocks) d_kernel->generic_work(out, in);  in += bs; out += bs;
 *     lib/simple_receiver_cc_impl.cc:61-77         the same around receiver_kernel_cc::generic_work
 *     lib/advanced_receiver_sb_cc_impl.cc:86-123   generic_work_equalize(out, in, in_eq) when port 1 is connected (all three
 *                                                  pointers advance by block_size), generic_work(out, in) otherwise
 *     lib/transmitter_cc_impl.cc:130-195           per frame: modulate once, add_frame per output port
 *     lib/channel_estimator_cc_impl.cc:88-120      per frame: estimate_frame + estimate_snr, two stream tags
 *     lib/resource_mapper_cc_impl.cc:86-106, lib/resource_demapper_cc_impl.cc:87-105, lib/cyclic_prefixer_cc_impl.cc:93-110
 *                                                  per frame: map / demap / add the cyclic prefix
 * With a GPU behind the kernel classes that loop would pay one host-to-device copy, one launch, one device-to-host copy and one
 * synchronisation PER BLOCK (bench.py: single_block_host_us, ~17 us against ~10 us for the CPU kernel).  The functions here take the
 * scheduler's whole `noutput_items` run in one call of the kernels' *_batch methods: same pointers, same return values, same item
 * accounting as the loops above -- a wrapper's work() becomes one line.
 *
 * Templates over the kernel type, no GNU Radio types: the arithmetic-free part of work() that can be (and is: tests/test_wrappers_gpu.py,
 * a scheduler stand-in feeding ragged noutput_items) tested without GNU Radio.  gr_blocks.h wraps them into real gr::sync_block /
 * gr::block subclasses when <gnuradio/sync_block.h> exists.
 */
#ifndef INCLUDED_GFDM_BATCHED_WORK_H
#define INCLUDED_GFDM_BATCHED_WORK_H

#include <algorithm>
#include <complex>
#include <vector>

namespace gr {
namespace gfdm {
namespace batched {

typedef std::complex<float> cfloat;

/* the modulator's batch method has no equaliser argument, the receivers' has: one spelling for sync_work */
template <class Kernel>
auto batch_call(Kernel& k, cfloat* out, const cfloat* in, long n) -> decltype(k.generic_work_batch(out, in, n), void())
{
    k.generic_work_batch(out, in, n);
}
template <class Kernel>
auto batch_call(Kernel& k, cfloat* out, const cfloat* in, long n) -> decltype(k.generic_work_batch(out, in, nullptr, n), void())
{
    k.generic_work_batch(out, in, static_cast<const cfloat*>(nullptr), n);
}

/* simple_modulator_cc_impl::work / simple_receiver_cc_impl::work.
 * Processes floor(noutput_items / block_size) blocks in one launch and returns noutput_items, as the reference does
 * (the scheduler only ever passes multiples of block_size: set_output_multiple, lib/simple_receiver_cc_impl.cc:54). */
template <class Kernel>
int sync_work(Kernel& kernel, int noutput_items, const cfloat* in, cfloat* out)
{
    const int n_blocks = noutput_items / kernel.block_size();
    if (n_blocks > 0) batch_call(kernel, out, in, n_blocks);
    return noutput_items;
}

/* advanced_receiver_sb_cc_impl::work.  in_eq == nullptr: port 1 (the per-block equaliser vector) is not connected.
 * Returns n_blocks * block_size (lib/advanced_receiver_sb_cc_impl.cc:122); tag copying stays in the wrapper. */
template <class Kernel>
int sync_work_equalize(Kernel& kernel, int noutput_items, const cfloat* in, const cfloat* in_eq, cfloat* out)
{
    const int n_blocks = noutput_items / kernel.block_size();
    if (n_blocks > 0) kernel.generic_work_batch(out, in, in_eq, n_blocks);
    return n_blocks * kernel.block_size();
}

/* transmitter_cc_impl::general_work: n_frames = min(noutput_items / output_vector_size, ninput_items / input_vector_size)
 * (lib/transmitter_cc_impl.cc:142-143); every output port gets its cyclic shift of every frame, all ports in ONE launch.
 * Returns the frames produced: the wrapper consumes n_frames * input_vector_size items and returns n_frames * output_vector_size. */
template <class Kernel>
int transmitter_work(Kernel& kernel, int noutput_items, int ninput_items, const cfloat* in, cfloat* const* outs, int n_ports)
{
    const int n_frames = std::min(noutput_items / kernel.output_vector_size(), ninput_items / kernel.input_vector_size());
    if (n_frames > 0) kernel.generic_work_batch(outs, n_ports, in, kernel.input_vector_size(), n_frames);
    return std::max(n_frames, 0);
}

/* channel_estimator_cc_impl::general_work: n_frames = noutput_items / frame_len; per frame estimate_frame + estimate_snr
 * (lib/channel_estimator_cc_impl.cc:97-116), here two launches for the whole run.  tag(i, snr_lin, cnrs, n_cnrs) is called once per
 * frame for the wrapper's "snr_lin" / "cnr" stream tags.  Returns the frames produced (consume n_frames * 2 * fft_len). */
template <class Kernel, class TagFn>
int estimator_work(Kernel& kernel, int noutput_items, const cfloat* in, cfloat* out, TagFn&& tag)
{
    const int frame_len = kernel.frame_len(), active = kernel.active_subcarriers();
    const int n_frames = noutput_items / frame_len;
    if (n_frames <= 0) return 0;
    kernel.estimate_frame_batch(out, in, n_frames);
    std::vector<float> snr(n_frames), cnrs(static_cast<size_t>(n_frames) * active);
    kernel.estimate_snr_batch(snr.data(), cnrs.data(), in, n_frames);
    for (int i = 0; i < n_frames; ++i) tag(i, snr[i], cnrs.data() + static_cast<size_t>(i) * active, active);
    return n_frames;
}

/* resource_mapper_cc_impl::general_work / resource_demapper_cc_impl::general_work (lib/resource_mapper_cc_impl.cc:86-106,
 * lib/resource_demapper_cc_impl.cc:87-105): n_frames = min(noutput_items / output_vector_size, ninput_items / input_vector_size), whole
 * vectors only (the blocks never zero-pad), one launch for the run.  The kernel object was constructed as mapper or demapper
 * (is_mapper), which fixes the direction and the two vector sizes.  Returns the frames produced (consume n_frames * input_vector_size,
 * return n_frames * output_vector_size). */
template <class Kernel>
int mapper_work(Kernel& kernel, bool is_mapper, int noutput_items, int ninput_items, const cfloat* in, cfloat* out)
{
    const int in_len = static_cast<int>(kernel.input_vector_size()), out_len = static_cast<int>(kernel.output_vector_size());
    const int n_frames = std::min(noutput_items / out_len, ninput_items / in_len);
    if (n_frames <= 0) return 0;
    if (is_mapper) kernel.map_to_resources_batch(out, in, kernel.block_size(), n_frames);
    else kernel.demap_from_resources_batch(out, in, kernel.block_size(), n_frames);
    return n_frames;
}

/* cyclic_prefixer_cc_impl::general_work (lib/cyclic_prefixer_cc_impl.cc:93-110): n_frames = noutput_items / frame_size, the constructor's
 * cyclic shift, one launch for the run.  Returns the frames produced (consume n_frames * block_size). */
template <class Kernel>
int prefixer_work(Kernel& kernel, int noutput_items, const cfloat* in, cfloat* out)
{
    const int n_frames = noutput_items / kernel.frame_size();
    if (n_frames <= 0) return 0;
    kernel.add_cyclic_prefix_batch(out, in, kernel.cyclic_shift(), n_frames);
    return n_frames;
}

} // namespace batched
} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_BATCHED_WORK_H */
