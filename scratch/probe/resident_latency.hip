// How fast could a one-block host call be with a RESIDENT kernel instead of a launch per call?  One wavefront (or one 256-lane workgroup) stays on the
// GPU, polls a doorbell word in pinned host memory, reads a 4608-byte block (K=64 M=9) from pinned host memory, does a token amount of work on it,
// writes 4608 bytes back to pinned host memory, and posts a completion word; the host writes the block, rings the doorbell and spins on the completion.
// Beside it: the launch-per-call form the library uses today (work kernel + flag kernel on one stream, host polls).
// The resident kernel ENDS BY ITSELF: after `iters` calls, or when no doorbell came for 20 ms, or 200 ms after it started -- whatever comes first.
// hipcc --offload-arch=gfx950 -O2 -o resident_latency resident_latency.hip && timeout 60 ./resident_latency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

constexpr int N = 576;                  // complex samples of one block
struct Mailbox {
    unsigned doorbell;  unsigned pad0[15];
    unsigned done;      unsigned pad1[15];
    unsigned exited;    unsigned pad2[15];
    float2 in[N];
    float2 out[N];
};

__device__ __forceinline__ unsigned ld_sys(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// wall_clock64(): 100 MHz, constant
__global__ void k_resident(Mailbox* mb, unsigned iters, long long idle_ticks, long long life_ticks)
{
    __shared__ unsigned go;
    const long long born = wall_clock64();
    for (unsigned seq = 1; seq <= iters; ++seq) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            unsigned ok = 0;
            for (;;) {
                if (ld_sys(&mb->doorbell) == seq) { ok = 1; break; }
                const long long t = wall_clock64();
                if (t - t0 > idle_ticks || t - born > life_ticks) break;
                __builtin_amdgcn_s_sleep(1);
            }
            go = ok;
        }
        __syncthreads();
        if (!go) break;
        // all loads of a lane in flight at once (as the block kernels do): one round trip over the link, not one per element
        float2 v[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) { const int i = threadIdx.x + j * blockDim.x; if (i < N) v[j] = mb->in[i]; }
#pragma unroll
        for (int j = 0; j < 9; ++j) { const int i = threadIdx.x + j * blockDim.x; if (i < N) mb->out[i] = make_float2(v[j].x * 2.f, v[j].y * 2.f); }
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) st_sys(&mb->done, seq);
    }
    if (threadIdx.x == 0) st_sys(&mb->exited, 1u);
}

__global__ void k_work(const float2* in, float2* out)
{
    for (int i = threadIdx.x; i < N; i += blockDim.x) { const float2 v = in[i]; out[i] = make_float2(v.x * 2.f, v.y * 2.f); }
}
__global__ void k_flag(volatile unsigned* flag, unsigned v) { __threadfence_system(); *flag = v; }

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void report(const char* what, std::vector<double>& t)
{
    std::sort(t.begin(), t.end());
    printf("%-78s median %6.2f us   p10 %6.2f   p90 %6.2f   max %7.2f\n", what, t[t.size() / 2], t[t.size() / 10], t[t.size() * 9 / 10], t.back());
}

int main()
{
    Mailbox *mb, *dmb;
    if (hipHostMalloc((void**)&mb, sizeof(Mailbox), hipHostMallocMapped) != hipSuccess) { printf("hipHostMalloc failed\n"); return 1; }
    (void)hipHostGetDevicePointer((void**)&dmb, mb, 0);
    memset(mb, 0, sizeof(Mailbox));
    std::vector<float2> user_in(N), user_out(N);
    for (int i = 0; i < N; ++i) user_in[i] = make_float2((float)i, -(float)i);
    hipStream_t s, s2;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    (void)hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const unsigned iters = 3000;

    for (int threads : { 64, 256 }) {
        memset(mb, 0, sizeof(Mailbox));
        hipLaunchKernelGGL(k_resident, dim3(1), dim3(threads), 0, s, dmb, iters, 2000000LL /* 20 ms */, 20000000LL /* 200 ms */);
        std::vector<double> t;
        bool lost = false;
        for (unsigned seq = 1; seq <= iters && !lost; ++seq) {
            const double t0 = now();
            memcpy(mb->in, user_in.data(), sizeof(float2) * N);                       // the caller's block into the pinned mailbox
            __atomic_store_n(&mb->doorbell, seq, __ATOMIC_RELEASE);
            while (__atomic_load_n(&mb->done, __ATOMIC_ACQUIRE) != seq) {
                if (__atomic_load_n(&mb->exited, __ATOMIC_ACQUIRE)) { lost = true; break; }
            }
            memcpy(user_out.data(), mb->out, sizeof(float2) * N);                     // and the result back
            if (seq > 200) t.push_back(now() - t0);
        }
        (void)hipStreamSynchronize(s);
        char what[160];
        snprintf(what, sizeof what, "resident kernel, %3d lanes: copy in + doorbell + block across the link + completion + copy out", threads);
        if (lost) printf("%s: the kernel retired before the calls were over (%zu timed)\n", what, t.size());
        if (!t.empty()) report(what, t);
        if (user_out[5].x != 10.f) printf("   WRONG RESULT %f\n", user_out[5].x);
    }
    {   // the retirement rule: a resident kernel with nobody ringing ends after its idle time
        memset(mb, 0, sizeof(Mailbox));
        const double t0 = now();
        hipLaunchKernelGGL(k_resident, dim3(1), dim3(64), 0, s, dmb, iters, 2000000LL, 20000000LL);
        (void)hipStreamSynchronize(s);
        printf("resident kernel with no caller: gone after %.1f ms (idle limit 20 ms), exited flag %u\n", (now() - t0) / 1e3, mb->exited);
    }
    {   // today's form: launch per call (work kernel + flag kernel), host polls the flag
        unsigned* flag = &mb->done; unsigned* dflag = &dmb->done;
        *flag = 0;
        std::vector<double> t;
        for (unsigned seq = 1; seq <= iters; ++seq) {
            const double t0 = now();
            memcpy(mb->in, user_in.data(), sizeof(float2) * N);
            hipLaunchKernelGGL(k_work, dim3(1), dim3(256), 0, s2, dmb->in, dmb->out);
            hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s2, dflag, seq);
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) { }
            memcpy(user_out.data(), mb->out, sizeof(float2) * N);
            if (seq > 200) t.push_back(now() - t0);
        }
        (void)hipStreamSynchronize(s2);
        report("launch per call (work kernel + flag kernel on one stream, host polls): the library's form today", t);
    }
    (void)hipHostFree(mb);
    return 0;
}
