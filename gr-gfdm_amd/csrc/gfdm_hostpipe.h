// Host-buffer batch path of the C-ABI (*_host entry points): the path the reference's GNU Radio wrappers call with HOST pointers
// (lib/simple_receiver_cc_impl.cc:61-77, lib/advanced_receiver_sb_cc_impl.cc:86-123 -- three pointers advanced per block --,
// lib/transmitter_cc_impl.cc:165-177).  One call = nblocks blocks of every operand back to back; the call returns when the outputs are in
// the caller's memory (the reference's synchronous generic_work contract).
//
// How the bytes cross the PCIe link (measured on the MI355X box, scratch/probe/host_link.hip, profiles/r04/host_link_probe.txt):
//   * a kernel that reads / writes pinned host memory directly reaches the link rate of the copy engines (57 GB/s in, 55 GB/s out, 96 GB/s
//     both ways at once) with no copy command, no device staging buffer and no per-chunk copy granularity -- so the kernels of a host call
//     run ON host memory ("zero copy", mode 0).  Mode 1 (copy engines: H2D on one stream, kernel on the handle's, D2H on a third, events in
//     between) exists for the A/B measurement;
//   * operands the GPU can address already -- memory registered with gfdm_hip_register_host, hipHostMalloc'ed, or device memory -- are
//     handed to the kernel as they are, no copy at all: link-bound;
//   * a kernel reads its blocks and then writes them, so one stream drives the link in one direction at a time; the chunks of a call
//     alternate between two streams to keep both directions busy;
//   * ordinary (pageable) memory is bounced through pinned staging sets in CHUNKS: while the kernel of chunk c runs across the link, the
//     calling thread (plus a small pool of copy threads) copies chunk c + 1 in and chunk c - 1 out.  A call that fits one chunk is the
//     single-block path of a GNU Radio wrapper: copy in, one launch, completion ticket, copy out.
#ifndef GFDM_HOSTPIPE_H
#define GFDM_HOSTPIPE_H

#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

namespace gfdm {

struct HostOperand {
    void* host;          // the caller's pointer (host memory of any kind, or memory the GPU can address)
    size_t stride;       // bytes from one block to the next
    size_t last;         // bytes one block occupies (== stride, except preambles read at a stride): nb blocks span (nb - 1) * stride + last
    bool write;          // output of the call
};

constexpr int HOST_MAX_OPERANDS = 10;      // transmitter: TX_MAX_PORTS outputs + the symbols
constexpr int HOST_MAX_DEPTH = 4;

// launch(dev, nb, stream): enqueue the kernels for nb blocks; dev[i] = device-visible address of operand i's first block of this chunk.
// Returns a gfdm_hip_status.  (A function reference: no allocation on the one-block path.)
struct HostLaunchRef {
    void* ctx;
    int (*fn)(void*, void* const*, int64_t, hipStream_t);
    template <class F>
    HostLaunchRef(F& f) : ctx(&f), fn([](void* c, void* const* d, int64_t nb, hipStream_t s) { return (*static_cast<F*>(c))(d, nb, s); }) {}
    int operator()(void* const* d, int64_t nb, hipStream_t s) const { return fn(ctx, d, nb, s); }
};

// what the last host call of this thread did (gfdm_hip_host_call_stats; the tests read it)
struct HostCallStats {
    int64_t chunks = 0;          // kernel launches of the call
    int64_t chunk_blocks = 0;    // blocks per chunk (the last one may be shorter)
    int64_t staged_bytes = 0;    // bytes bounced through the pinned staging sets (both directions)
    unsigned direct_mask = 0;    // bit i: operand i was handed to the kernel in place (registered / pinned / device memory)
    int mode = 0;                // 0 kernel on host memory, 1 copy engines
    int depth = 0;               // staging sets in use
    int copy_threads = 0;        // pool threads that helped with the bounce copies
    // where the calling thread spent the call, nanoseconds: sorting the operands, bounce copies, enqueueing kernels (+ copy commands), posting
    // tickets, waiting for tickets
    int64_t ns_setup = 0, ns_copy = 0, ns_launch = 0, ns_post = 0, ns_wait = 0;
};

class HostPipe {
public:
    HostPipe() = default;
    HostPipe(const HostPipe&) = delete;
    HostPipe& operator=(const HostPipe&) = delete;
    ~HostPipe() { release(); }
    // `device` must be current.  `stream`: the handle's private stream (the kernels of the call run on it).
    int run(hipStream_t stream, const HostOperand* ops, int nops, int64_t nblocks, HostLaunchRef launch);
    void release();              // frees the staging sets (the owner makes its device current first)

private:
    struct Set {
        char* host = nullptr;    // pinned, GPU-mapped
        char* dev = nullptr;     // the same memory as the GPU addresses it
        char* dcopy = nullptr;   // mode 1: device staging
        size_t cap = 0, dcap = 0;
    };
    Set sets_[HOST_MAX_DEPTH];
    unsigned* ticket_ = nullptr;
    unsigned* ticket_dev_ = nullptr;
    unsigned ticket_next_ = 0;
    hipStream_t s_b_ = nullptr;                 // second kernel stream: chunk c + 1 reads across the link while chunk c writes
    hipStream_t s_in_ = nullptr, s_out_ = nullptr;
    hipEvent_t ev_in_[HOST_MAX_DEPTH] = {}, ev_k_[HOST_MAX_DEPTH] = {};      // nullptr = not created (yet)

    int ensure_set(int i, size_t host_bytes, size_t dev_bytes);
    int ensure_ticket();
    int ensure_copy_engines();
    int ensure_second_stream();
    hipError_t post_ticket(hipStream_t s, int slot, unsigned value);
    hipError_t wait_ticket(hipStream_t s, int slot, unsigned value);
};

// process-wide settings of the host path (gfdm_hip_set_host_pipeline); negative = keep
int host_pipeline_set(int mode, int64_t chunk_bytes, int depth, int copy_threads, int kernel_streams);
void host_pipeline_get(int* mode, int64_t* chunk_bytes, int* depth, int* copy_threads, int* kernel_streams);
HostCallStats& host_last_call();           // thread-local
int host_streaming_copies(int enable);     // A/B switch: 0 = plain memcpy for the bounce copies; returns the previous setting
void host_copy_pool_quiesce();             // stop the copy threads (library exit path, gfdm_hip_quiesce)
int host_register(void* p, size_t bytes);
int host_unregister(void* p);

}  // namespace gfdm

#endif
