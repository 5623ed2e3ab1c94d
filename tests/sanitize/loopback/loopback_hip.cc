// TEST-ONLY loop-back HIP layer: implementation (see hip/hip_runtime.h for what it is and is not).
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include <unistd.h>

namespace loopback {
hipError_t module_launch(const std::string& lowered_name, void** args, hipStream_t s);      // loopback_kernels.cc
}

// ---- streams ---------------------------------------------------------------------------------------------------------------------------------
struct loopback_stream {
    int device = 0;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    uint64_t submitted = 0, completed = 0;
    bool closing = false;            // hipStreamDestroy: finish the queue, then the thread deletes the stream (HIP releases a busy stream when its work is done)
    void run()
    {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return closing || !q.empty(); });
                if (q.empty()) break;                // closing and drained
                fn = std::move(q.front());
                q.pop_front();
            }
            fn();
            fn = nullptr;                            // captured state dies before the completion becomes visible
            {
                std::lock_guard<std::mutex> lk(m);
                ++completed;
            }
            cv.notify_all();
        }
        delete this;
    }
};

struct loopback_event_state {
    std::mutex m;
    std::condition_variable cv;
    uint64_t recorded = 0, done = 0;
};
struct loopback_event { std::shared_ptr<loopback_event_state> st = std::make_shared<loopback_event_state>(); };

struct loopback_function { std::string name; };
struct loopback_module { std::vector<std::unique_ptr<loopback_function>> fns; std::vector<std::string> names; };

namespace {

struct MemEntry { size_t size; hipMemoryType type; int device; bool owned; };

struct Failure { int skip, count; hipError_t err; };

// process-wide state, allocated once and never destroyed: the product's static destructors (copy pool, JIT workers) may still call in at exit
struct State {
    std::mutex mem_m;
    std::map<uintptr_t, MemEntry> mem;
    std::mutex streams_m;
    std::vector<loopback_stream*> streams;              // live streams (for the implicit synchronisation of hipFree)
    std::vector<loopback_stream*> default_streams;      // one per device, created at first use
    std::mutex fail_m;
    std::map<std::string, Failure> failures;
    std::atomic<int> ndev{ 2 };
    std::atomic<bool> unknown_is_error{ false };
    std::atomic<long> launches{ 0 }, async_copies{ 0 }, streams_created{ 0 }, streams_destroyed{ 0 }, host_allocs{ 0 }, host_frees{ 0 }, dev_allocs{ 0 },
        dev_frees{ 0 }, registers{ 0 }, unregisters{ 0 }, modules_loaded{ 0 }, compiles{ 0 };
    std::atomic<int> compile_lo{ 2 }, compile_hi{ 20 };
};
State& S()
{
    static State* s = new State();
    return *s;
}

thread_local int t_device = 0;
thread_local hipError_t t_last = hipSuccess;

hipError_t ret(hipError_t e)
{
    if (e != hipSuccess && e != hipErrorNotReady) t_last = e;
    return e;
}

// an injected failure for this API?
thread_local bool t_armed = false;
hipError_t injected(const char* api)
{
    if (!t_armed) return hipSuccess;
    State& s = S();
    std::lock_guard<std::mutex> lk(s.fail_m);
    auto it = s.failures.find(api);
    if (it == s.failures.end()) return hipSuccess;
    if (it->second.skip > 0) { --it->second.skip; return hipSuccess; }
    const hipError_t e = it->second.err;
    if (--it->second.count <= 0) s.failures.erase(it);
    return e;
}
#define LB_INJECT(api) do { const hipError_t _e = injected(api); if (_e != hipSuccess) return ret(_e); } while (0)

loopback_stream* new_stream(int device)
{
    loopback_stream* st = new loopback_stream();
    st->device = device;
    std::thread([st] { st->run(); }).detach();
    return st;
}

loopback_stream* resolve(hipStream_t s)
{
    if (s) return s;
    State& g = S();
    std::lock_guard<std::mutex> lk(g.streams_m);
    if ((int)g.default_streams.size() < g.ndev.load()) g.default_streams.resize((size_t)g.ndev.load(), nullptr);
    loopback_stream*& d = g.default_streams[(size_t)t_device];
    if (!d) { d = new_stream(t_device); g.streams.push_back(d); }
    return d;
}

void stream_sync(loopback_stream* st)
{
    std::unique_lock<std::mutex> lk(st->m);
    const uint64_t upto = st->submitted;
    st->cv.wait(lk, [&] { return st->completed >= upto; });
}

void device_sync(int device)
{
    State& g = S();
    // (streams are only deleted by their own thread after hipStreamDestroy removed them from the list under streams_m, so holding the lock keeps them alive;
    // no stream function takes streams_m, hence no inversion)
    std::lock_guard<std::mutex> lk(g.streams_m);
    for (loopback_stream* st : g.streams)
        if (st->device == device) stream_sync(st);
}

std::map<uintptr_t, MemEntry>::iterator find_containing(State& s, uintptr_t a)
{
    auto it = s.mem.upper_bound(a);
    if (it == s.mem.begin()) return s.mem.end();
    --it;
    return (a < it->first + it->second.size) ? it : s.mem.end();
}

}  // namespace

namespace loopback {

hipError_t enqueue(hipStream_t s, std::function<void()> fn)
{
    const hipError_t inj = injected("launch");
    if (inj != hipSuccess) return ret(inj);
    loopback_stream* st = resolve(s);
    {
        std::lock_guard<std::mutex> lk(st->m);
        st->q.push_back(std::move(fn));
        ++st->submitted;
    }
    st->cv.notify_all();
    S().launches.fetch_add(1, std::memory_order_relaxed);
    return hipSuccess;
}

void fail_next(const char* api, int skip, int count, hipError_t err)
{
    std::lock_guard<std::mutex> lk(S().fail_m);
    S().failures[api] = Failure{ skip, count, err };
}
void arm(bool on) { t_armed = on; }
void clear_failures()
{
    std::lock_guard<std::mutex> lk(S().fail_m);
    S().failures.clear();
}
Stats stats()
{
    State& s = S();
    return Stats{ s.launches.load(), s.async_copies.load(), s.streams_created.load(), s.streams_destroyed.load(), s.host_allocs.load(), s.host_frees.load(),
                  s.dev_allocs.load(), s.dev_frees.load(), s.registers.load(), s.unregisters.load(), s.modules_loaded.load() };
}
long live_allocations()
{
    State& s = S();
    std::lock_guard<std::mutex> lk(s.mem_m);
    long n = 0;
    for (const auto& e : s.mem) n += e.second.owned ? 1 : 0;
    return n;
}
void set_device_count(int n) { S().ndev.store(n); }
void set_unknown_pointer_is_error(bool on) { S().unknown_is_error.store(on); }
void set_compile_ms(int lo, int hi) { S().compile_lo.store(lo); S().compile_hi.store(hi < lo ? lo : hi); }
long compiles() { return S().compiles.load(); }

}  // namespace loopback

extern "C" {

const char* hipGetErrorString(hipError_t e)
{
    switch (e) {
    case hipSuccess: return "no error";
    case hipErrorInvalidValue: return "invalid argument";
    case hipErrorOutOfMemory: return "out of memory";
    case hipErrorNotInitialized: return "not initialized";
    case hipErrorInvalidDevice: return "invalid device ordinal";
    case hipErrorInvalidImage: return "device kernel image is invalid";
    case hipErrorNotFound: return "named symbol not found";
    case hipErrorNotReady: return "device not ready";
    case hipErrorLaunchFailure: return "unspecified launch failure";
    case hipErrorHostMemoryAlreadyRegistered: return "part or all of the requested memory range is already mapped";
    case hipErrorHostMemoryNotRegistered: return "pointer does not correspond to a registered memory region";
    default: return "unknown error";
    }
}

hipError_t hipGetLastError(void)
{
    const hipError_t e = t_last;
    t_last = hipSuccess;
    return e;
}

hipError_t hipGetDeviceCount(int* n)
{
    LB_INJECT("hipGetDeviceCount");
    *n = S().ndev.load();
    return hipSuccess;
}
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d)
{
    LB_INJECT("hipSetDevice");
    if (d < 0 || d >= S().ndev.load()) return ret(hipErrorInvalidDevice);
    t_device = d;
    return hipSuccess;
}

hipError_t hipMalloc(void** p, size_t bytes)
{
    *p = nullptr;
    LB_INJECT("hipMalloc");
    void* m = malloc(bytes ? bytes : 1);
    if (!m) return ret(hipErrorOutOfMemory);
    memset(m, 0xD5, bytes);                          // device memory is not zero-initialised: poison it
    State& s = S();
    std::lock_guard<std::mutex> lk(s.mem_m);
    s.mem[reinterpret_cast<uintptr_t>(m)] = MemEntry{ bytes ? bytes : 1, hipMemoryTypeDevice, t_device, true };
    s.dev_allocs.fetch_add(1);
    *p = m;
    return hipSuccess;
}

hipError_t hipFree(void* p)
{
    if (!p) return hipSuccess;
    State& s = S();
    int device;
    {
        std::lock_guard<std::mutex> lk(s.mem_m);
        auto it = s.mem.find(reinterpret_cast<uintptr_t>(p));
        if (it == s.mem.end() || it->second.type != hipMemoryTypeDevice || !it->second.owned) return ret(hipErrorInvalidValue);
        device = it->second.device;
    }
    device_sync(device);                             // hipFree waits for the device's outstanding work
    {
        std::lock_guard<std::mutex> lk(s.mem_m);
        s.mem.erase(reinterpret_cast<uintptr_t>(p));
    }
    s.dev_frees.fetch_add(1);
    free(p);
    return hipSuccess;
}

hipError_t hipHostMalloc(void** p, size_t bytes, unsigned)
{
    *p = nullptr;
    LB_INJECT("hipHostMalloc");
    void* m = nullptr;
    if (posix_memalign(&m, 4096, bytes ? bytes : 1) != 0) return ret(hipErrorOutOfMemory);
    memset(m, 0xC7, bytes);
    State& s = S();
    std::lock_guard<std::mutex> lk(s.mem_m);
    s.mem[reinterpret_cast<uintptr_t>(m)] = MemEntry{ bytes ? bytes : 1, hipMemoryTypeHost, t_device, true };
    s.host_allocs.fetch_add(1);
    *p = m;
    return hipSuccess;
}

hipError_t hipHostFree(void* p)
{
    if (!p) return hipSuccess;
    State& s = S();
    {
        std::lock_guard<std::mutex> lk(s.mem_m);
        auto it = s.mem.find(reinterpret_cast<uintptr_t>(p));
        if (it == s.mem.end() || it->second.type != hipMemoryTypeHost || !it->second.owned) return ret(hipErrorInvalidValue);
        s.mem.erase(it);
    }
    s.host_frees.fetch_add(1);
    free(p);                                         // NO implicit synchronisation here: work still touching it shows up as a use-after-free
    return hipSuccess;
}

hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned)
{
    LB_INJECT("hipHostGetDevicePointer");
    State& s = S();
    std::lock_guard<std::mutex> lk(s.mem_m);
    auto it = find_containing(s, reinterpret_cast<uintptr_t>(host));
    if (it == s.mem.end() || it->second.type != hipMemoryTypeHost) return ret(hipErrorInvalidValue);
    *dev = host;
    return hipSuccess;
}

hipError_t hipHostRegister(void* p, size_t bytes, unsigned)
{
    LB_INJECT("hipHostRegister");
    if (!p || !bytes) return ret(hipErrorInvalidValue);
    State& s = S();
    std::lock_guard<std::mutex> lk(s.mem_m);
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    auto next = s.mem.lower_bound(a);
    if (next != s.mem.end() && next->first < a + bytes) return ret(hipErrorHostMemoryAlreadyRegistered);
    if (find_containing(s, a) != s.mem.end()) return ret(hipErrorHostMemoryAlreadyRegistered);
    s.mem[a] = MemEntry{ bytes, hipMemoryTypeHost, t_device, false };
    s.registers.fetch_add(1);
    return hipSuccess;
}

hipError_t hipHostUnregister(void* p)
{
    LB_INJECT("hipHostUnregister");
    State& s = S();
    std::lock_guard<std::mutex> lk(s.mem_m);
    auto it = s.mem.find(reinterpret_cast<uintptr_t>(p));
    if (it == s.mem.end() || it->second.owned) return ret(hipErrorHostMemoryNotRegistered);
    s.mem.erase(it);
    s.unregisters.fetch_add(1);
    return hipSuccess;
}

hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p)
{
    memset(a, 0, sizeof *a);
    LB_INJECT("hipPointerGetAttributes");
    State& s = S();
    std::lock_guard<std::mutex> lk(s.mem_m);
    auto it = find_containing(s, reinterpret_cast<uintptr_t>(p));
    if (it == s.mem.end()) {
        if (s.unknown_is_error.load()) return ret(hipErrorInvalidValue);
        a->type = hipMemoryTypeUnregistered;
        return hipSuccess;
    }
    a->type = it->second.type;
    a->device = it->second.device;
    a->devicePointer = const_cast<void*>(p);
    a->hostPointer = it->second.type == hipMemoryTypeHost ? const_cast<void*>(p) : nullptr;
    return hipSuccess;
}

hipError_t hipMemGetAddressRange(hipDeviceptr_t* base, size_t* size, hipDeviceptr_t p)
{
    LB_INJECT("hipMemGetAddressRange");
    State& s = S();
    std::lock_guard<std::mutex> lk(s.mem_m);
    auto it = find_containing(s, reinterpret_cast<uintptr_t>(p));
    if (it == s.mem.end()) return ret(hipErrorInvalidValue);
    *base = reinterpret_cast<void*>(it->first);
    *size = it->second.size;
    return hipSuccess;
}

hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind)
{
    LB_INJECT("hipMemcpy");
    memcpy(dst, src, bytes);
    return hipSuccess;
}

hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s)
{
    LB_INJECT("hipMemcpyAsync");
    S().async_copies.fetch_add(1, std::memory_order_relaxed);
    loopback_stream* st = resolve(s);
    {
        std::lock_guard<std::mutex> lk(st->m);
        st->q.push_back([dst, src, bytes] { memcpy(dst, src, bytes); });
        ++st->submitted;
    }
    st->cv.notify_all();
    return hipSuccess;
}

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned)
{
    *s = nullptr;
    LB_INJECT("hipStreamCreateWithFlags");
    loopback_stream* st = new_stream(t_device);
    State& g = S();
    {
        std::lock_guard<std::mutex> lk(g.streams_m);
        g.streams.push_back(st);
    }
    g.streams_created.fetch_add(1);
    *s = st;
    return hipSuccess;
}

hipError_t hipStreamDestroy(hipStream_t s)
{
    if (!s) return ret(hipErrorInvalidValue);
    State& g = S();
    {
        std::lock_guard<std::mutex> lk(g.streams_m);
        for (size_t i = 0; i < g.streams.size(); ++i)
            if (g.streams[i] == s) { g.streams.erase(g.streams.begin() + (long)i); break; }
    }
    {
        std::lock_guard<std::mutex> lk(s->m);
        s->closing = true;
        s->cv.notify_all();                          // under the lock: the stream's thread deletes the stream as soon as it sees `closing`
    }
    g.streams_destroyed.fetch_add(1);
    return hipSuccess;
}

hipError_t hipStreamSynchronize(hipStream_t s)
{
    LB_INJECT("hipStreamSynchronize");
    stream_sync(resolve(s));
    return hipSuccess;
}

hipError_t hipStreamQuery(hipStream_t s)
{
    LB_INJECT("hipStreamQuery");
    loopback_stream* st = resolve(s);
    std::lock_guard<std::mutex> lk(st->m);
    return st->completed >= st->submitted ? hipSuccess : hipErrorNotReady;
}

hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned)
{
    *e = nullptr;
    LB_INJECT("hipEventCreateWithFlags");
    *e = new loopback_event();
    return hipSuccess;
}

hipError_t hipEventDestroy(hipEvent_t e)
{
    if (!e) return ret(hipErrorInvalidValue);
    delete e;                                        // queued records / waits keep the state alive (shared_ptr), as HIP does for a busy event
    return hipSuccess;
}

hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{
    LB_INJECT("hipEventRecord");
    if (!e) return ret(hipErrorInvalidValue);
    std::shared_ptr<loopback_event_state> st = e->st;
    uint64_t g;
    {
        std::lock_guard<std::mutex> lk(st->m);
        g = ++st->recorded;
    }
    loopback_stream* q = resolve(s);
    {
        std::lock_guard<std::mutex> lk(q->m);
        q->q.push_back([st, g] {
            {
                std::lock_guard<std::mutex> lk2(st->m);
                if (st->done < g) st->done = g;
            }
            st->cv.notify_all();
        });
        ++q->submitted;
    }
    q->cv.notify_all();
    return hipSuccess;
}

hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{
    LB_INJECT("hipStreamWaitEvent");
    if (!e) return ret(hipErrorInvalidValue);
    std::shared_ptr<loopback_event_state> st = e->st;
    uint64_t g;
    {
        std::lock_guard<std::mutex> lk(st->m);
        g = st->recorded;
    }
    if (g == 0) return hipSuccess;                   // never recorded: nothing to wait for
    loopback_stream* q = resolve(s);
    {
        std::lock_guard<std::mutex> lk(q->m);
        q->q.push_back([st, g] {
            std::unique_lock<std::mutex> lk2(st->m);
            st->cv.wait(lk2, [&] { return st->done >= g; });
        });
        ++q->submitted;
    }
    q->cv.notify_all();
    return hipSuccess;
}

hipError_t hipDeviceSynchronize(void)
{
    device_sync(t_device);
    return hipSuccess;
}

// ---- modules: the blob the loop-back hiprtc emits ---------------------------------------------------------------------------------------------
static const char kMagic[] = "LOOPBACK-HSACO ";

hipError_t hipModuleLoadData(hipModule_t* m, const void* image)
{
    *m = nullptr;
    LB_INJECT("hipModuleLoadData");
    const char* c = static_cast<const char*>(image);
    if (memcmp(c, kMagic, sizeof kMagic - 1) != 0) return ret(hipErrorInvalidImage);
    char* end = nullptr;
    const unsigned long n = strtoul(c + sizeof kMagic - 1, &end, 10);
    if (!end || *end != '\n') return ret(hipErrorInvalidImage);
    const char* p = end + 1;
    const char* stop = p + n;
    auto mod = std::make_unique<loopback_module>();
    while (p < stop) {
        const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(stop - p)));
        if (!nl) return ret(hipErrorInvalidImage);
        mod->names.emplace_back(p, nl);
        p = nl + 1;
    }
    S().modules_loaded.fetch_add(1);
    *m = mod.release();
    return hipSuccess;
}

hipError_t hipModuleUnload(hipModule_t m)
{
    delete m;
    return hipSuccess;
}

hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char* name)
{
    *f = nullptr;
    LB_INJECT("hipModuleGetFunction");
    for (const auto& n : m->names)
        if (n == name) {
            m->fns.emplace_back(new loopback_function{ n });
            *f = m->fns.back().get();
            return hipSuccess;
        }
    return ret(hipErrorNotFound);
}

hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned, hipStream_t s, void** args, void**)
{
    if (!f) return ret(hipErrorInvalidValue);
    return ret(loopback::module_launch(f->name, args, s));
}

hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

}  // extern "C"

// ---- hiprtc ---------------------------------------------------------------------------------------------------------------------------------------
struct loopback_rtc_program {
    std::vector<std::string> exprs, lowered;
    std::string code, log;
    bool compiled = false;
};

namespace {
std::string lower(const std::string& expr)
{
    std::string s = "lb$";
    for (char c : expr)
        if (c != ' ') s += c;
    return s;
}
}  // namespace

extern "C" {

const char* hiprtcGetErrorString(hiprtcResult r)
{
    switch (r) {
    case HIPRTC_SUCCESS: return "HIPRTC_SUCCESS";
    case HIPRTC_ERROR_COMPILATION: return "HIPRTC_ERROR_COMPILATION";
    case HIPRTC_ERROR_INVALID_INPUT: return "HIPRTC_ERROR_INVALID_INPUT";
    default: return "HIPRTC_ERROR";
    }
}

hiprtcResult hiprtcVersion(int* major, int* minor) { *major = 0; *minor = 1; return HIPRTC_SUCCESS; }

hiprtcResult hiprtcCreateProgram(hiprtcProgram* prog, const char* src, const char*, int nheaders, const char* const* headers, const char* const* include_names)
{
    *prog = nullptr;
    if (injected("hiprtcCreateProgram") != hipSuccess) return HIPRTC_ERROR_OUT_OF_MEMORY;
    if (!src) return HIPRTC_ERROR_INVALID_INPUT;
    size_t touched = strlen(src);
    for (int i = 0; i < nheaders; ++i) touched += strlen(headers[i]) + strlen(include_names[i]);      // the sources must be readable strings
    (void)touched;
    *prog = new loopback_rtc_program();
    return HIPRTC_SUCCESS;
}

hiprtcResult hiprtcDestroyProgram(hiprtcProgram* prog)
{
    if (!prog || !*prog) return HIPRTC_ERROR_INVALID_INPUT;
    delete *prog;
    *prog = nullptr;
    return HIPRTC_SUCCESS;
}

hiprtcResult hiprtcAddNameExpression(hiprtcProgram prog, const char* expr)
{
    prog->exprs.emplace_back(expr);
    return HIPRTC_SUCCESS;
}

hiprtcResult hiprtcCompileProgram(hiprtcProgram prog, int nopts, const char** opts)
{
    for (int i = 0; i < nopts; ++i) (void)strlen(opts[i]);
    State& s = S();
    thread_local std::mt19937 rng((unsigned)std::hash<std::thread::id>()(std::this_thread::get_id()));
    const int lo = s.compile_lo.load(), hi = s.compile_hi.load();
    std::this_thread::sleep_for(std::chrono::milliseconds(lo + (int)(rng() % (unsigned)(hi - lo + 1))));
    s.compiles.fetch_add(1);
    if (injected("hiprtcCompileProgram") != hipSuccess) {
        prog->log = "loop-back hiprtc: injected compilation failure";
        return HIPRTC_ERROR_COMPILATION;
    }
    std::string names;
    prog->lowered.clear();
    for (const auto& e : prog->exprs) { prog->lowered.push_back(lower(e)); names += prog->lowered.back(); names += '\n'; }
    prog->code = std::string(kMagic) + std::to_string(names.size()) + "\n" + names;
    prog->compiled = true;
    return HIPRTC_SUCCESS;
}

hiprtcResult hiprtcGetProgramLogSize(hiprtcProgram prog, size_t* n) { *n = prog->log.size() + 1; return HIPRTC_SUCCESS; }
hiprtcResult hiprtcGetProgramLog(hiprtcProgram prog, char* log) { memcpy(log, prog->log.c_str(), prog->log.size() + 1); return HIPRTC_SUCCESS; }

hiprtcResult hiprtcGetLoweredName(hiprtcProgram prog, const char* expr, const char** lowered)
{
    for (size_t i = 0; i < prog->exprs.size() && i < prog->lowered.size(); ++i)
        if (prog->exprs[i] == expr) { *lowered = prog->lowered[i].c_str(); return HIPRTC_SUCCESS; }
    return HIPRTC_ERROR_INVALID_INPUT;
}

hiprtcResult hiprtcGetCodeSize(hiprtcProgram prog, size_t* n) { *n = prog->compiled ? prog->code.size() : 0; return HIPRTC_SUCCESS; }
hiprtcResult hiprtcGetCode(hiprtcProgram prog, char* code)
{
    if (!prog->compiled) return HIPRTC_ERROR_INVALID_INPUT;
    memcpy(code, prog->code.data(), prog->code.size());
    return HIPRTC_SUCCESS;
}

}  // extern "C"
