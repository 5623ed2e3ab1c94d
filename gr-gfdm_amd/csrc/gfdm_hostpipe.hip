// Host-buffer batch path (gfdm_hostpipe.h): operand classification, chunked bounce through pinned staging sets, completion tickets,
// the copy-thread pool.  No GFDM arithmetic here and no CPU compute path: the kernels of a call are enqueued by the caller's `launch`.
#include "gfdm_hostpipe.h"
#include "../../include/gfdm_hip.h"
#include "gfdm_plan.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <unistd.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace gfdm {

namespace {

// one spin-wait step: frees the core's issue slots for its SMT sibling (which may be running a slice of the copy pool)
inline void cpu_relax()
{
#if defined(__x86_64__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    __asm__ __volatile__("yield");
#endif
}

// how the bytes cross the link: 0 under the kernels' own accesses to host memory, 1 copy engines + device staging, 2 inputs under the
// kernel's reads and outputs by copy engine, 3 the reverse (HostPipe::run)
std::atomic<int> g_mode{ 0 };
// bytes of all staged operands per chunk; 0 = automatic: one chunk up to kSingleChunkMax, else round(sqrt(1.3 x MiB staged)) chunks, at least two, of at most
// kAutoMax.  The calling thread does copies, launches and tickets one after the other, so a call takes about (copies + ~7 us per chunk + the last chunk's time on
// the GPU): the minimum over the number of chunks n lies near sqrt(GPU time / 7 us).  profiles/r04/host_chunk_sweep.txt: two chunks from 0.5 to 3.5 MB, 3-6 up to
// 20 MB, 8-10 at 40-55 MB (the round's first rule, total / 8 but at least 512 KiB, cut a 256-block call into five: 104-109 us against 77-81 us with two).
std::atomic<int64_t> g_chunk_bytes{ 0 };
std::atomic<int> g_depth{ 3 };
std::atomic<int> g_copy_threads{ 3 };
std::atomic<int> g_kernel_streams{ 2 };         // mode 0: chunks alternate between this many streams (1 or 2)

constexpr size_t kSingleChunkMax = 512u << 10;  // a call that stages at most this much is one chunk
constexpr size_t kAutoMax = 16u << 20;
constexpr size_t kAlign = 256;
constexpr size_t kSliceBytes = 256u << 10;      // unit of work of the copy pool
#ifndef GFDM_HOST_POOL_MIN_BYTES
#define GFDM_HOST_POOL_MIN_BYTES (1u << 20)
#endif
constexpr size_t kPoolMinBytes = GFDM_HOST_POOL_MIN_BYTES;    // smaller copy jobs stay on the calling thread
constexpr int64_t kMaxLaunchBlocks = 1 << 30;
constexpr int kTicketStride = 16;               // one 64-byte line per completion ticket

size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

// (a system-scope release store: everything the kernels in front of it on the stream wrote is visible to the host thread whose acquire load reads `value`)
__global__ void k_host_ticket(unsigned* ticket, unsigned value)
{
    __hip_atomic_store(ticket, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- bounce copies -------------------------------------------------------------------------------------------------------------------
// A bounce copy writes memory the CPU will not read again soon (the staging set is read by the GPU, the caller's output by whoever comes next),
// so large copies use non-temporal stores: no read-for-ownership of the destination lines, two memory operations per byte instead of three.
// glibc's memcpy does the same, but only above a threshold of several MiB; the pool cuts copies into 256 KiB slices, far below it
// (profiles/r04/host_copy_streaming_stores.txt: one copy thread 31 -> 42 GB/s, four 69 -> 105 GB/s, a 32 768-block call 3.2 -> 4.4 / 6.7 -> 7.2 M blocks/s).
// Only for calls that stage kStreamCallBytes or more in all: from there on the streaming form measured faster or equal at every size
// (profiles/r04/host_mid_size_calls.txt: 512 blocks 194-218 -> 137-156 us, 1024 blocks 224-261 -> 194-209 us, 256 blocks equal); smaller calls keep
// memcpy, which leaves their few hundred KiB of results in cache for whoever reads them next.
constexpr size_t kStreamMinBytes = 32u << 10;
#ifndef GFDM_HOST_STREAM_CALL_BYTES
#define GFDM_HOST_STREAM_CALL_BYTES (2u << 20)
#endif
constexpr size_t kStreamCallBytes = GFDM_HOST_STREAM_CALL_BYTES;

#if defined(__x86_64__)
__attribute__((target("avx2"))) void copy_streaming_avx2(char* d, const char* s, size_t n)
{
    const size_t head = (32 - (reinterpret_cast<uintptr_t>(d) & 31)) & 31;       // stores must be 32-byte aligned
    if (head) { memcpy(d, s, head); d += head; s += head; n -= head; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + 64));
        const __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(d + i + 96), e);
    }
    _mm_sfence();
    if (i < n) memcpy(d + i, s + i, n - i);
}

const bool g_have_avx2 = __builtin_cpu_supports("avx2");
#else       // other hosts: memcpy everywhere (glibc's own non-temporal path above its threshold)
inline void copy_streaming_avx2(char* d, const char* s, size_t n) { memcpy(d, s, n); }
const bool g_have_avx2 = false;
#endif
std::atomic<int> g_streaming_copies{ 1 };       // 0: plain memcpy everywhere (A/B)

inline void bounce_copy(char* d, const char* s, size_t n, bool streaming)
{
    if (streaming && n >= kStreamMinBytes) copy_streaming_avx2(d, s, n);
    else memcpy(d, s, n);
}

// ---- copy pool: the bounce copies of one chunk, cut into slices, shared between the calling thread and a few pool threads -------------
struct Slice { char* dst; const char* src; size_t n; };
struct CopyJob {
    std::vector<Slice> slices;
    bool streaming = false;
    std::atomic<size_t> next{ 0 }, done{ 0 };
    void work()
    {
        const size_t n = slices.size();
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) return;
            bounce_copy(slices[i].dst, slices[i].src, slices[i].n, streaming);
            done.fetch_add(1, std::memory_order_release);
        }
    }
};

class CopyPool {
public:
    // runs the job on the calling thread and up to `helpers` pool threads; returns when every slice is copied
    int run(const std::shared_ptr<CopyJob>& job, int helpers)
    {
        int used = 0;
        if (helpers > 0) {
            std::lock_guard<std::mutex> lk(mu_);
            if (!stopping_) {
                // a new thread's baseline generation is read HERE, under the lock: a thread that first runs after quiesce() has bumped the
                // generation would otherwise take the bumped value as its baseline, never see stopping_ and hang quiesce()'s join
                const uint64_t g0 = gen_.load(std::memory_order_acquire);
                while ((int)threads_.size() < helpers && (int)threads_.size() < 16) threads_.emplace_back([this, g0] { loop(g0); });
                used = (int)threads_.size() < helpers ? (int)threads_.size() : helpers;
                current_ = job;
                wanted_ = used;
                gen_.fetch_add(1, std::memory_order_release);
            }
        }
        if (used) cv_.notify_all();
        job->work();
        const size_t n = job->slices.size();
        while (job->done.load(std::memory_order_acquire) < n) cpu_relax();
        return used;
    }
    void quiesce()
    {
        std::vector<std::thread> th;
        {
            std::lock_guard<std::mutex> lk(mu_);
            stopping_ = true;
            gen_.fetch_add(1, std::memory_order_release);
            th.swap(threads_);
        }
        cv_.notify_all();
        for (auto& t : th) t.join();
        std::lock_guard<std::mutex> lk(mu_);
        stopping_ = false;
        current_.reset();
    }
    ~CopyPool() { quiesce(); }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::thread> threads_;
    std::shared_ptr<CopyJob> current_;
    std::atomic<uint64_t> gen_{ 0 };
    int wanted_ = 0;
    bool stopping_ = false;

    void loop(uint64_t seen)
    {
        for (;;) {
            // chunks of one call follow each other within tens of microseconds: spin briefly before going to sleep
            for (int spin = 0; spin < 20000 && gen_.load(std::memory_order_acquire) == seen; ++spin) cpu_relax();
            std::shared_ptr<CopyJob> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stopping_ || gen_.load(std::memory_order_acquire) != seen; });
                seen = gen_.load(std::memory_order_acquire);
                if (stopping_) return;
                if (wanted_ > 0) { --wanted_; job = current_; }
            }
            if (job) job->work();
        }
    }
};

CopyPool g_pool;

thread_local HostCallStats t_stats;

// the ranges pinned through gfdm_hip_register_host (whole pages each): a host call uses an operand in place only if ONE of them holds all of it
std::mutex g_reg_mu;
std::vector<std::pair<const char*, size_t>> g_registered;
bool registry_contains(const void* p, size_t bytes)
{
    const char* c = static_cast<const char*>(p);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (const auto& r : g_registered)
        if (c >= r.first && c + bytes <= r.first + r.second) return true;
    return false;
}

inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// adds the time since *t to `acc` and restarts *t
inline void lap(int64_t& acc, int64_t& t) { const int64_t n = now_ns(); acc += n - t; t = n; }

}  // namespace

int host_pipeline_set(int mode, int64_t chunk_bytes, int depth, int copy_threads, int kernel_streams)
{
    if (mode > 3 || depth == 0 || depth > HOST_MAX_DEPTH || copy_threads > 16 || kernel_streams == 0 || kernel_streams > 2)
        return api_fail(GFDM_HIP_EINVAL, "host pipeline: mode 0..3, depth 1..4, copy threads 0..16, kernel streams 1..2");
    if (kernel_streams > 0) g_kernel_streams.store(kernel_streams);
    if (mode >= 0) g_mode.store(mode);
    if (chunk_bytes >= 0) g_chunk_bytes.store(chunk_bytes);
    if (depth > 0) g_depth.store(depth);
    if (copy_threads >= 0) g_copy_threads.store(copy_threads);
    return GFDM_HIP_OK;
}

void host_pipeline_get(int* mode, int64_t* chunk_bytes, int* depth, int* copy_threads, int* kernel_streams)
{
    if (kernel_streams) *kernel_streams = g_kernel_streams.load();
    if (mode) *mode = g_mode.load();
    if (chunk_bytes) *chunk_bytes = g_chunk_bytes.load();
    if (depth) *depth = g_depth.load();
    if (copy_threads) *copy_threads = g_copy_threads.load();
}

HostCallStats& host_last_call() { return t_stats; }
int host_streaming_copies(int enable) { return g_streaming_copies.exchange(enable ? 1 : 0); }
void host_copy_pool_quiesce() { g_pool.quiesce(); }

int host_register(void* p, size_t bytes)
{
    if (!p || bytes == 0) return api_fail(GFDM_HIP_EINVAL, "register_host: NULL pointer or zero size");
    // Whole pages only.  Pinning works on pages: a range that begins or ends inside a page drags its neighbours in that page along, and when such a
    // page is unpinned again under a neighbour the runtime also pins on the fly (a pageable hipMemcpy of some other object in that page), the GPU
    // later faults on it -- seen with small heap arrays in scratch/fuzz_host_path.py (profiles/r04/host_path_fuzz.txt).  A scheduler's buffers
    // (mmap'ed, page-granular) and anything from aligned_alloc / posix_memalign with a page-multiple size qualify.
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    if (reinterpret_cast<uintptr_t>(p) % page != 0 || bytes % page != 0) {
        char buf[200];
        snprintf(buf, sizeof buf, "register_host: the range must be whole pages the caller owns (start and size multiples of %zu bytes)", page);
        return api_fail(GFDM_HIP_EINVAL, buf);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return api_fail(GFDM_HIP_ENODEV, "no HIP device available");
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped);
    if (e != hipSuccess) { (void)hipGetLastError(); return api_fail_hip(e, "hipHostRegister"); }
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_registered.emplace_back(static_cast<const char*>(p), bytes);
    return GFDM_HIP_OK;
}

int host_unregister(void* p)
{
    if (!p) return api_fail(GFDM_HIP_EINVAL, "unregister_host: NULL pointer");
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        bool found = false;
        for (size_t i = 0; i < g_registered.size(); ++i)
            if (g_registered[i].first == static_cast<const char*>(p)) { g_registered.erase(g_registered.begin() + (long)i); found = true; break; }
        if (!found) return api_fail(GFDM_HIP_EINVAL, "unregister_host: this pointer was not registered with gfdm_hip_register_host");
    }
    hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return api_fail_hip(e, "hipHostUnregister"); }
    return GFDM_HIP_OK;
}

void HostPipe::release()
{
    for (Set& s : sets_) {
        if (s.host) (void)hipHostFree(s.host);
        if (s.dcopy) (void)hipFree(s.dcopy);
        s = Set{};
    }
    if (ticket_) (void)hipHostFree(ticket_);
    ticket_ = ticket_dev_ = nullptr;
    for (int i = 0; i < HOST_MAX_DEPTH; ++i) {
        if (ev_in_[i]) (void)hipEventDestroy(ev_in_[i]);
        if (ev_k_[i]) (void)hipEventDestroy(ev_k_[i]);
        ev_in_[i] = ev_k_[i] = nullptr;
    }
    if (s_in_) (void)hipStreamDestroy(s_in_);
    if (s_out_) (void)hipStreamDestroy(s_out_);
    if (s_b_) (void)hipStreamDestroy(s_b_);
    s_in_ = s_out_ = s_b_ = nullptr;
}

int HostPipe::ensure_set(int i, size_t host_bytes, size_t dev_bytes)
{
    Set& s = sets_[i];
    if (s.cap < host_bytes) {
        if (s.host) (void)hipHostFree(s.host);
        s.host = s.dev = nullptr;
        size_t want = host_bytes < (64u << 10) ? (64u << 10) : host_bytes;
        if (want < 2 * s.cap) want = 2 * s.cap;
        s.cap = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&s.host), want, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer(reinterpret_cast<void**>(&s.dev), s.host, 0) != hipSuccess) {
            (void)hipGetLastError();
            if (s.host) (void)hipHostFree(s.host);
            s.host = s.dev = nullptr;
            return api_fail(GFDM_HIP_ENOMEM, "pinned staging buffer allocation failed");
        }
        s.cap = want;
    }
    if (s.dcap < dev_bytes) {
        if (s.dcopy) (void)hipFree(s.dcopy);
        s.dcopy = nullptr;
        size_t want = dev_bytes < (64u << 10) ? (64u << 10) : dev_bytes;
        if (want < 2 * s.dcap) want = 2 * s.dcap;
        s.dcap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&s.dcopy), want) != hipSuccess) { (void)hipGetLastError(); return api_fail(GFDM_HIP_ENOMEM, "device staging buffer allocation failed"); }
        s.dcap = want;
    }
    return GFDM_HIP_OK;
}

int HostPipe::ensure_ticket()
{
    if (ticket_) return GFDM_HIP_OK;
    if (hipHostMalloc(reinterpret_cast<void**>(&ticket_), kTicketStride * sizeof(unsigned) * HOST_MAX_DEPTH, hipHostMallocMapped) != hipSuccess ||
        hipHostGetDevicePointer(reinterpret_cast<void**>(&ticket_dev_), ticket_, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (ticket_) (void)hipHostFree(ticket_);
        ticket_ = ticket_dev_ = nullptr;
        return api_fail(GFDM_HIP_ENOMEM, "completion ticket allocation failed");
    }
    for (int i = 0; i < HOST_MAX_DEPTH; ++i) ticket_[i * kTicketStride] = 0;
    ticket_next_ = 0;
    return GFDM_HIP_OK;
}

int HostPipe::ensure_copy_engines()
{
    if (!s_in_ && hipStreamCreateWithFlags(&s_in_, hipStreamNonBlocking) != hipSuccess) { s_in_ = nullptr; (void)hipGetLastError(); return api_fail(GFDM_HIP_EHIP, "hipStreamCreate"); }
    if (!s_out_ && hipStreamCreateWithFlags(&s_out_, hipStreamNonBlocking) != hipSuccess) { s_out_ = nullptr; (void)hipGetLastError(); return api_fail(GFDM_HIP_EHIP, "hipStreamCreate"); }
    // (each event on its own: a creation that fails half way leaves the ones made so far to release() and to the next attempt -- the all-or-nothing flag this
    // replaces leaked them, found by the injected failures of tests/sanitize)
    for (int i = 0; i < HOST_MAX_DEPTH; ++i) {
        if (!ev_in_[i] && hipEventCreateWithFlags(&ev_in_[i], hipEventDisableTiming) != hipSuccess) { ev_in_[i] = nullptr; (void)hipGetLastError(); return api_fail(GFDM_HIP_EHIP, "hipEventCreate"); }
        if (!ev_k_[i] && hipEventCreateWithFlags(&ev_k_[i], hipEventDisableTiming) != hipSuccess) { ev_k_[i] = nullptr; (void)hipGetLastError(); return api_fail(GFDM_HIP_EHIP, "hipEventCreate"); }
    }
    return GFDM_HIP_OK;
}

// Completion without hipStreamSynchronize: a one-lane kernel behind the work on the same stream writes a ticket into pinned host memory and
// the host spins on it (scratch/probe/host_latency.hip: empty kernel + hipStreamSynchronize 12.2 us, + ticket kernel + spin 9.1 us).  Every
// few thousand spins the stream is queried, so a faulting kernel ends the wait with its error instead of hanging the caller.
hipError_t HostPipe::post_ticket(hipStream_t s, int slot, unsigned value)
{
    (void)hipGetLastError();          // (sticky: a stale error of an earlier call must not be read as this launch's)
    hipLaunchKernelGGL(k_host_ticket, dim3(1), dim3(1), 0, s, ticket_dev_ + slot * kTicketStride, value);
    return hipGetLastError();
}

hipError_t HostPipe::wait_ticket(hipStream_t s, int slot, unsigned value)
{
    const unsigned* t = ticket_ + slot * kTicketStride;
    auto reached = [&] { return (int)(__atomic_load_n(t, __ATOMIC_ACQUIRE) - value) >= 0; };      // acquire: the results were written before the ticket
    for (unsigned spins = 1;; ++spins) {
        if (reached()) return hipSuccess;
        if ((spins & 0xFFF) == 0) {
            hipError_t e = hipStreamQuery(s);
            if (e == hipSuccess) return reached() ? hipSuccess : hipStreamSynchronize(s);
            if (e != hipErrorNotReady) return e;
        }
        cpu_relax();
    }
}

int HostPipe::run(hipStream_t stream, const HostOperand* ops, int nops, int64_t nblocks, HostLaunchRef launch)
{
    HostCallStats& st = t_stats;
    st = HostCallStats{};
    int64_t tl = now_ns();
    if (nops < 1 || nops > HOST_MAX_OPERANDS || nblocks < 0) return api_fail(GFDM_HIP_EINVAL, "bad operand list or negative block count");
    if (nblocks == 0) return GFDM_HIP_OK;
    const int mode = g_mode.load();
    st.mode = mode;

    // ---- which operands can the GPU address as they are? ---------------------------------------------------------------------------
    bool direct[HOST_MAX_OPERANDS];
    bool host_mem[HOST_MAX_OPERANDS];                  // not device / managed memory: the copy engines can serve it
    char* direct_dev[HOST_MAX_OPERANDS];
    size_t extent[HOST_MAX_OPERANDS];
    int cur_dev = 0;
    (void)hipGetDevice(&cur_dev);
    for (int i = 0; i < nops; ++i) {
        const HostOperand& o = ops[i];
        extent[i] = o.last ? (size_t)(nblocks - 1) * o.stride + o.last : 0;
        direct[i] = false;
        host_mem[i] = true;
        direct_dev[i] = static_cast<char*>(o.host);
        if (extent[i] == 0) { direct[i] = true; continue; }           // nothing to move: the kernel never touches it
        if (!o.host) return api_fail(GFDM_HIP_EINVAL, "NULL buffer");
        hipPointerAttribute_t a0{}, a1{};
        if (hipPointerGetAttributes(&a0, o.host) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (a0.type != hipMemoryTypeHost && a0.type != hipMemoryTypeDevice && a0.type != hipMemoryTypeManaged) continue;
        // device / managed memory is never bounced (a CPU copy would dereference a device address in the caller's thread): it is used in place when the
        // runtime confirms that the WHOLE extent lies inside the one allocation the first byte belongs to, and refused otherwise -- an undersized buffer
        // (last byte unmapped: the lookup fails or reports another kind) and an extent that runs on into a neighbouring allocation alike
        const bool dev_mem = a0.type == hipMemoryTypeDevice || a0.type == hipMemoryTypeManaged;
        if (a0.type == hipMemoryTypeDevice && a0.device != cur_dev) return api_fail(GFDM_HIP_EINVAL, "buffer lives in the memory of another GPU");
        if (dev_mem) {
            hipDeviceptr_t rbase = nullptr;
            size_t rsize = 0;
            const bool last_ok = hipPointerGetAttributes(&a1, static_cast<char*>(o.host) + extent[i] - 1) == hipSuccess && a1.type == a0.type;
            const bool range_ok = last_ok && a0.devicePointer && hipMemGetAddressRange(&rbase, &rsize, a0.devicePointer) == hipSuccess &&
                                  static_cast<char*>(a0.devicePointer) >= static_cast<char*>(rbase) &&
                                  static_cast<char*>(a0.devicePointer) + extent[i] <= static_cast<char*>(rbase) + rsize;
            if (!range_ok) { (void)hipGetLastError(); return api_fail(GFDM_HIP_EINVAL, "device buffer: the extent of the call does not lie inside one device allocation"); }
            direct[i] = true;
            host_mem[i] = false;
            direct_dev[i] = static_cast<char*>(a0.devicePointer);
            continue;
        }
        if (hipPointerGetAttributes(&a1, static_cast<char*>(o.host) + extent[i] - 1) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (a1.type != a0.type) continue;                             // only partly registered: bounce it
        // two registrations side by side need not be contiguous as the GPU sees them: the last byte must sit where the first one says
        if (a1.devicePointer != static_cast<char*>(a0.devicePointer) + (extent[i] - 1)) continue;
        // ... and the WHOLE range must lie inside ONE registration / allocation: small heap buffers share pages with their neighbours, so a buffer
        // that was never registered can begin in the last page of one registered neighbour and end in the first page of another -- both ends
        // look registered, the pages in between are not, and the kernel would fault on them (found by scratch/fuzz_host_path.py)
        // Ranges registered through gfdm_hip_register_host are known exactly; for memory pinned or allocated by somebody else (hipHostMalloc, a device
        // allocation) the runtime is asked for the allocation around the pointer, and where it cannot tell the operand is bounced.
        if (!registry_contains(o.host, extent[i])) {
            hipDeviceptr_t rbase = nullptr;
            size_t rsize = 0;
            if (hipMemGetAddressRange(&rbase, &rsize, a0.devicePointer) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (static_cast<char*>(a0.devicePointer) < static_cast<char*>(rbase) ||
                static_cast<char*>(a0.devicePointer) + extent[i] > static_cast<char*>(rbase) + rsize) continue;
        }
        if (!a0.devicePointer) continue;
        direct[i] = true;
        host_mem[i] = true;
        direct_dev[i] = static_cast<char*>(a0.devicePointer);
    }
    // an output handed over in place must not overlap another in-place operand (the kernels assume distinct buffers): bounce it instead
    for (int i = 0; i < nops; ++i) {
        if (!direct[i] || !ops[i].write || extent[i] == 0) continue;
        const char* b0 = static_cast<const char*>(ops[i].host);
        for (int j = 0; j < nops; ++j) {
            if (j == i || !direct[j] || extent[j] == 0) continue;
            const char* b1 = static_cast<const char*>(ops[j].host);
            if (b0 < b1 + extent[j] && b1 < b0 + extent[i]) {
                if (!host_mem[i]) return api_fail(GFDM_HIP_EINVAL, "a device-memory output overlaps another operand: the *_host calls need distinct device buffers");
                direct[i] = false;
                break;
            }
        }
    }
    size_t per_block = 0, total_staged = 0;
    for (int i = 0; i < nops; ++i) {
        if (direct[i]) { st.direct_mask |= 1u << i; continue; }
        per_block += ops[i].stride ? ops[i].stride : ops[i].last;
        total_staged += extent[i];
    }
    st.staged_bytes = (int64_t)total_staged;

    int rc = ensure_ticket();
    if (rc != GFDM_HIP_OK) return rc;

    // ---- routes ------------------------------------------------------------------------------------------------------------------------------
    // A kernel reads its blocks and then writes them; a launch whose workgroups are all resident at once (a few thousand blocks) therefore drives
    // the link in ONE direction at a time (57 GB/s of the 96 GB/s both directions carry together), and launches on one stream do not overlap.
    // Per operand the bytes cross the link either under the kernel's own accesses ("zero copy", ZC) or on a copy engine to / from device staging
    // (CE).  mode 0: everything ZC; 1: everything CE; 2: inputs ZC, outputs CE -- the copy engine writes chunk c back while the kernel of chunk
    // c + 1 reads, i.e. both directions busy; 3: inputs CE, outputs ZC.  Device / managed memory handed to a *_host call is always used as is.
    bool ce[HOST_MAX_OPERANDS];
    bool any_ce = false;
    for (int i = 0; i < nops; ++i) {
        ce[i] = extent[i] != 0 && host_mem[i] && (mode == 1 || (mode == 2 && ops[i].write) || (mode == 3 && !ops[i].write));
        any_ce = any_ce || ce[i];
    }
    const bool all_direct = total_staged == 0;
    const bool one_launch = all_direct && !any_ce;      // kernels on the caller's memory: a single launch mixes the directions by itself
    size_t per_block_all = 0, total_all = 0;
    for (int i = 0; i < nops; ++i) { per_block_all += ops[i].stride ? ops[i].stride : ops[i].last; total_all += extent[i]; }
    const size_t plan_per_block = all_direct ? per_block_all : per_block, plan_total = all_direct ? total_all : total_staged;
    size_t chunk_bytes = (size_t)g_chunk_bytes.load();
    int64_t chunk_blocks;
    if (one_launch) chunk_blocks = nblocks;
    else if (chunk_bytes == 0) {
        if (plan_total <= kSingleChunkMax) chunk_blocks = nblocks;
        else {
            int64_t n = (int64_t)(sqrt(1.3 * (double)plan_total / 1048576.0) + 0.5);
            if (n < 2) n = 2;
            chunk_blocks = (nblocks + n - 1) / n;
            const int64_t cap = (int64_t)(kAutoMax / (plan_per_block ? plan_per_block : 1));
            if (chunk_blocks > cap) chunk_blocks = cap;
        }
    } else {
        chunk_blocks = (int64_t)(chunk_bytes / (plan_per_block ? plan_per_block : 1));
    }
    if (chunk_blocks < 1) chunk_blocks = 1;
    if (chunk_blocks > nblocks) chunk_blocks = nblocks;
    if (chunk_blocks > kMaxLaunchBlocks) chunk_blocks = kMaxLaunchBlocks;
    const int64_t nchunks = (nblocks + chunk_blocks - 1) / chunk_blocks;
    auto chunk_nb = [&](int64_t c) { const int64_t c0 = c * chunk_blocks; return nblocks - c0 < chunk_blocks ? nblocks - c0 : chunk_blocks; };
    auto chunk_size = [&](int i, int64_t nb) { return (size_t)(nb - 1) * ops[i].stride + ops[i].last; };
    st.chunk_blocks = chunk_blocks;
    const bool two_streams = !any_ce && nchunks >= 2 && g_kernel_streams.load() >= 2;
    if (two_streams && (rc = ensure_second_stream()) != GFDM_HIP_OK) return rc;
    auto kernel_stream = [&](int64_t c) { return (two_streams && (c & 1)) ? s_b_ : stream; };
    auto drain = [&]() {      // error path: nothing may touch the staging sets or the caller's memory after we return
        (void)hipStreamSynchronize(stream);
        if (s_b_) (void)hipStreamSynchronize(s_b_);
        if (s_in_) (void)hipStreamSynchronize(s_in_);
        if (s_out_) (void)hipStreamSynchronize(s_out_);
    };

    // ---- staging sets: pinned host memory for the bounced operands, device memory for the copy-engine routed ones ---------------------------
    int depth = g_depth.load();
    if (depth > nchunks) depth = (int)nchunks;
    st.depth = depth;
    size_t off_host[HOST_MAX_OPERANDS] = {}, off_dev[HOST_MAX_OPERANDS] = {};
    size_t host_bytes = 0, dev_bytes = 0;
    bool ce_in = false, ce_out = false;
    for (int i = 0; i < nops; ++i) {
        const size_t sz = extent[i] ? align_up(chunk_size(i, chunk_blocks)) : 0;
        if (!direct[i]) { off_host[i] = host_bytes; host_bytes += sz; }
        if (ce[i]) { off_dev[i] = dev_bytes; dev_bytes += sz; (ops[i].write ? ce_out : ce_in) = true; }
    }
    for (int s = 0; s < depth; ++s)
        if ((host_bytes || dev_bytes) && (rc = ensure_set(s, host_bytes, dev_bytes)) != GFDM_HIP_OK) return rc;
    if (any_ce && (rc = ensure_copy_engines()) != GFDM_HIP_OK) return rc;
    const int helpers = g_copy_threads.load();
    auto done_stream = [&](int64_t c) { return ce_out ? s_out_ : kernel_stream(c); };
    unsigned want[HOST_MAX_DEPTH] = {};

    // bounce copies: the chunks of a large call go to the copy pool as jobs (the bytes of one chunk, in and out, decide); a small call -- the
    // one-block call of a GNU Radio wrapper -- copies on the spot, no job object, no allocation
    // (copy jobs below 1 MiB stay on the calling thread: waking the helpers costs more than they take over -- 64 blocks of K=64 M=9 per call: 38-41 us pooled,
    // 34 us on the calling thread, profiles/r04/host_mid_size_calls.txt; three 590 KB chunks 95-97 us pooled, 72 us not, profiles/r04/host_chunk_sweep.txt)
    const bool pooled = helpers > 0 && host_bytes >= kPoolMinBytes && nchunks > 1;
    const bool streaming = total_staged >= kStreamCallBytes && g_have_avx2 && g_streaming_copies.load(std::memory_order_relaxed) != 0;
    std::shared_ptr<CopyJob> job;
    auto job_add = [&](char* dst, const char* src, size_t n) {
        if (!pooled) { bounce_copy(dst, src, n, streaming); return; }
        if (!job) { job = std::make_shared<CopyJob>(); job->streaming = streaming; }
        for (size_t o = 0; o < n; o += kSliceBytes) job->slices.push_back(Slice{ dst + o, src + o, n - o < kSliceBytes ? n - o : kSliceBytes });
    };
    auto user_ptr = [&](int i, int64_t c) { return static_cast<char*>(ops[i].host) + (size_t)(c * chunk_blocks) * ops[i].stride; };
    auto add_copy_out = [&](int64_t c) {
        const Set& s = sets_[c % depth];
        for (int i = 0; i < nops; ++i)
            if (!direct[i] && ops[i].write) job_add(user_ptr(i, c), s.host + off_host[i], chunk_size(i, chunk_nb(c)));
    };
    auto add_copy_in = [&](int64_t c) {
        const Set& s = sets_[c % depth];
        for (int i = 0; i < nops; ++i)
            if (!direct[i] && !ops[i].write) job_add(s.host + off_host[i], user_ptr(i, c), chunk_size(i, chunk_nb(c)));
    };
    auto run_job = [&]() {
        if (!job) return;
        const int used = g_pool.run(job, helpers);
        if (used > st.copy_threads) st.copy_threads = used;
        job.reset();
    };

    int64_t retired = 0;                       // chunks [0, retired) are back in the caller's memory
    lap(st.ns_setup, tl);
    for (int64_t c = 0; c < nchunks; ++c) {
        const int si = (int)(c % depth);
        Set& s = sets_[si];
        if (c >= depth) {                      // the set is still in use by chunk c - depth: wait for it, take its outputs with this chunk's inputs
            hipError_t e = wait_ticket(done_stream(c - depth), si, want[si]);
            if (e != hipSuccess) { drain(); return api_fail_hip(e, "host call"); }
            lap(st.ns_wait, tl);
            add_copy_out(c - depth);
            retired = c - depth + 1;
        }
        add_copy_in(c);
        run_job();
        lap(st.ns_copy, tl);
        const int64_t nb = chunk_nb(c);
        hipStream_t ks = kernel_stream(c);
        void* dev[HOST_MAX_OPERANDS];
        for (int i = 0; i < nops; ++i)
            dev[i] = ce[i] ? s.dcopy + off_dev[i] : direct[i] ? direct_dev[i] + (size_t)(c * chunk_blocks) * ops[i].stride : s.dev + off_host[i];
        hipError_t e = hipSuccess;
        if (ce_in) {
            for (int i = 0; i < nops && e == hipSuccess; ++i)
                if (ce[i] && !ops[i].write)
                    e = hipMemcpyAsync(s.dcopy + off_dev[i], direct[i] ? user_ptr(i, c) : s.host + off_host[i], chunk_size(i, nb), hipMemcpyHostToDevice, s_in_);
            if (e == hipSuccess) e = hipEventRecord(ev_in_[si], s_in_);
            if (e == hipSuccess) e = hipStreamWaitEvent(ks, ev_in_[si], 0);
            if (e != hipSuccess) { drain(); return api_fail_hip(e, "host call: H2D"); }
        }
        rc = launch(dev, nb, ks);
        if (rc != GFDM_HIP_OK) { drain(); return rc; }
        ++st.chunks;
        if (ce_out) {
            e = hipEventRecord(ev_k_[si], ks);
            if (e == hipSuccess) e = hipStreamWaitEvent(s_out_, ev_k_[si], 0);
            for (int i = 0; i < nops && e == hipSuccess; ++i)
                if (ce[i] && ops[i].write)
                    e = hipMemcpyAsync(direct[i] ? user_ptr(i, c) : s.host + off_host[i], s.dcopy + off_dev[i], chunk_size(i, nb), hipMemcpyDeviceToHost, s_out_);
            if (e != hipSuccess) { drain(); return api_fail_hip(e, "host call: D2H"); }
        }
        lap(st.ns_launch, tl);
        // (the ticket kernel's launch costs ~3 us of host time, but it runs in the shadow of the ~8 us the GPU needs from the first launch to a
        // visible completion: a ticket written by the compute kernel's last workgroup instead was built and measured -- the same 13.5 us per one-block
        // call, profiles/r04/host_call_breakdown.txt)
        want[si] = ++ticket_next_;
        e = post_ticket(done_stream(c), si, want[si]);
        if (e != hipSuccess) { drain(); return api_fail_hip(e, "host call: ticket"); }
        lap(st.ns_post, tl);
    }
    for (int64_t c = retired; c < nchunks; ++c) {
        const int si = (int)(c % depth);
        hipError_t e = wait_ticket(done_stream(c), si, want[si]);
        if (e != hipSuccess) { drain(); return api_fail_hip(e, "host call"); }
        lap(st.ns_wait, tl);
        add_copy_out(c);
        run_job();
        lap(st.ns_copy, tl);
    }
    return GFDM_HIP_OK;
}

int HostPipe::ensure_second_stream()
{
    if (!s_b_ && hipStreamCreateWithFlags(&s_b_, hipStreamNonBlocking) != hipSuccess) { s_b_ = nullptr; (void)hipGetLastError(); return api_fail(GFDM_HIP_EHIP, "hipStreamCreate"); }
    return GFDM_HIP_OK;
}

}  // namespace gfdm
