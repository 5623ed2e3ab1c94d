/* gr::gfdm::sharded_batch<Kernel> -- the batched-blocks multi-GPU mode: ONE batch of independent GFDM blocks over the GPUs of a node.
 *
 * gr-gfdm has no multi-device code (its kernels are single-threaded CPU objects); this is an addition on top of the drop-in classes.
 * GFDM blocks carry no state between generic_work calls, so a batch of nblocks blocks is split contiguously -- shard i of n covers
 * blocks [i nblocks / n, (i + 1) nblocks / n), the first nblocks % n shards one block longer -- and shard i runs on devices[i] through
 * a kernel object of its own (one handle and one private HIP stream per device, created by the ordinary constructor of `Kernel`
 * with gfdm_kernel_utils::set_default_device(devices[i]) in effect).  Nothing is exchanged between the devices: device i reads and
 * writes only its shard of the caller's buffers.
 *
 *   host batches    generic_work_batch(out, in, [f_eq,] nblocks): the shards run on one host thread per device, each through the
 *                   kernel's own *_batch entry point (copy in, launch, copy out on that device's stream); returns when all are back.
 *   device shards   generic_work_device(outs, ins, [f_eqs,] nblocks, streams): outs[i] / ins[i] live on devices[i] and hold shard i;
 *                   enqueues every shard (asynchronously, so the devices work concurrently) and returns; the caller synchronises
 *                   its streams.
 * The same split, one process per GPU under torch.distributed, is gfdm_amd.sharding.ShardedBatch (Python), which bench.py runs.
 */
#ifndef INCLUDED_GFDM_SHARDED_BATCH_H
#define INCLUDED_GFDM_SHARDED_BATCH_H

#include <gfdm/gfdm_kernel_utils.h>

#include <exception>
#include <memory>
#include <stdexcept>
#include <thread>
#include <utility>
#include <vector>

namespace gr {
namespace gfdm {

/* contiguous, balanced partition of `total` blocks into n shards: (first block, number of blocks) of shard `index` */
inline std::pair<long, long> shard_range(long total, int index, int n)
{
    const long base = total / n, extra = total % n;
    const long start = index * base + (index < extra ? index : extra);
    return { start, base + (index < extra ? 1 : 0) };
}

template <class Kernel>
class sharded_batch
{
public:
    typedef gfdm_kernel_utils::gfdm_complex gfdm_complex;

    /* one kernel object per entry of `devices` (an ordinal may appear more than once: two handles on one GPU), constructed with
     * Kernel(args...); throws what that constructor throws */
    template <class... Args>
    explicit sharded_batch(const std::vector<int>& devices, const Args&... args) : d_devices(devices)
    {
        if (devices.empty()) throw std::invalid_argument("sharded_batch: empty device list");
        const int prev = gfdm_kernel_utils::default_device();
        try {
            for (int dev : devices) {
                gfdm_kernel_utils::set_default_device(dev);
                d_kernels.emplace_back(new Kernel(args...));
            }
        } catch (...) {
            gfdm_kernel_utils::set_default_device(prev);
            throw;
        }
        gfdm_kernel_utils::set_default_device(prev);
    }

    int n_shards() const { return static_cast<int>(d_kernels.size()); }
    const std::vector<int>& devices() const { return d_devices; }
    Kernel& kernel(int i) { return *d_kernels.at(i); }
    int block_size() { return d_kernels.front()->block_size(); }
    std::pair<long, long> shard(long nblocks, int i) const { return shard_range(nblocks, i, n_shards()); }

    /* host batch through Kernel::generic_work_batch(out, in, nblocks)  (modulator) */
    void generic_work_batch(gfdm_complex* out, const gfdm_complex* in, long nblocks)
    {
        const long n = block_size();
        each_shard(nblocks, [&](int i, long start, long count) { d_kernels[i]->generic_work_batch(out + start * n, in + start * n, count); });
    }

    /* host batch through Kernel::generic_work_batch(out, in, f_eq, nblocks)  (receivers; f_eq may be nullptr) */
    void generic_work_batch(gfdm_complex* out, const gfdm_complex* in, const gfdm_complex* f_eq, long nblocks)
    {
        const long n = block_size();
        each_shard(nblocks, [&](int i, long start, long count) {
            d_kernels[i]->generic_work_batch(out + start * n, in + start * n, f_eq ? f_eq + start * n : nullptr, count);
        });
    }

    /* device-resident shards, modulator form: outs[i] / ins[i] on devices()[i] hold shard(nblocks, i).second blocks */
    void generic_work_device(void* const* outs, const void* const* ins, long nblocks, void* const* hip_streams)
    {
        for (int i = 0; i < n_shards(); ++i) {
            const long count = shard(nblocks, i).second;
            if (count > 0) d_kernels[i]->generic_work_device(outs[i], ins[i], count, hip_streams ? hip_streams[i] : nullptr);
        }
    }

    /* device-resident shards, receiver form (f_eqs or f_eqs[i] may be nullptr) */
    void generic_work_device(void* const* outs, const void* const* ins, const void* const* f_eqs, long nblocks, void* const* hip_streams)
    {
        for (int i = 0; i < n_shards(); ++i) {
            const long count = shard(nblocks, i).second;
            if (count > 0) d_kernels[i]->generic_work_device(outs[i], ins[i], f_eqs ? f_eqs[i] : nullptr, count, hip_streams ? hip_streams[i] : nullptr);
        }
    }

private:
    /* f(i, first block, number of blocks) for every non-empty shard, one host thread per shard beyond the first; the first
     * exception (if any) is rethrown after all threads have finished */
    template <class F>
    void each_shard(long nblocks, F f)
    {
        if (nblocks <= 0) return;
        const int n = n_shards();
        std::vector<std::exception_ptr> err(n);
        std::vector<std::thread> th;
        auto body = [&](int i) {
            const auto r = shard(nblocks, i);
            if (r.second <= 0) return;
            try { f(i, r.first, r.second); } catch (...) { err[i] = std::current_exception(); }
        };
        for (int i = 1; i < n; ++i) th.emplace_back(body, i);
        body(0);
        for (auto& t : th) t.join();
        for (auto& e : err) if (e) std::rethrow_exception(e);
    }

    std::vector<int> d_devices;
    std::vector<std::unique_ptr<Kernel>> d_kernels;
};

} // namespace gfdm
} // namespace gr

#endif /* INCLUDED_GFDM_SHARDED_BATCH_H */
