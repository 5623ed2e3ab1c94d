// Stand-alone resource mapper / demapper and cyclic prefixer (gr-gfdm resource_mapper_kernel_cc, add_cyclic_prefix_cc) behind the
// C-ABI: the data-format stages either side of the modulator and the receivers, for callers that use them as separate objects
// (the reference's `Resource_mapper` and `Cyclic_prefixer` Python classes, its resource_mapper_cc / cyclic_prefixer_cc /
// remove_prefix_cc blocks).  The composite transmitter and the frame receivers have the same arithmetic fused into their load and
// store stages (gfdm_tx.h, RxIo); use those where the stages sit next to a modulator or receiver -- these kernels exist so that the
// separate objects keep working, batched, on device-resident data.
//
// Pure data movement (index gather, copy with cyclic shift, two ramp multiplies): HBM-bound by construction.
//   map      out[K][M] grid  <- in[ninput_size] symbols      reads 8 n_in + writes 8 K M bytes per block
//   demap    out[noutput]    <- in[K][M]                     reads 8 n_out (gathered) + writes 8 n_out
//   add cp   out[cp + N + cs] <- in[N]                       reads 8 N + writes 8 (cp + N + cs)
//   remove   out[N]          <- in[cp + N + cs]              reads 8 N + writes 8 N
// Restated from gr-gfdm: lib/resource_mapper_kernel_cc.cc:30-163, lib/add_cyclic_prefix_cc.cc:30-104.
#include "../../include/gfdm_hip.h"
#include "gfdm_plan.h"
#include "gfdm_dft.h"
#include "gfdm_hostpipe.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using gfdm::cf;
using gfdm::api_fail;
using gfdm::api_fail_hip;

namespace {

#define STAGE_TRY(expr)                                          \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return api_fail_hip(_e, #expr);    \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev)
    {
        // hipGetLastError() is sticky: it keeps the error of ANY earlier failed runtime call of this thread (ours or the application's) until somebody reads
        // it, and the launchers check their launches with it -- a call must not fail on somebody else's stale error (found by tests/sanitize: an allocation
        // failure in one constructor failed the next handle's first launch).  Every entry point that launches builds a DeviceGuard first.
        (void)hipGetLastError();
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// device, private stream and the host-buffer path (gfdm_hostpipe.h) of the *_host entry points
struct StageCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    gfdm::HostPipe pipe;
    void* d_tables = nullptr;

    int open(int dev)
    {
        int count = 0;
        if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return api_fail(GFDM_HIP_ENODEV, "no HIP device available (this library has no CPU path)");
        if (dev < 0 || dev >= count) return api_fail(GFDM_HIP_ENODEV, "HIP device ordinal out of range");
        device = dev;
        DeviceGuard guard(dev);
        if (!guard.ok) return api_fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
        STAGE_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        return GFDM_HIP_OK;
    }
    int upload(const void* host, size_t bytes)
    {
        DeviceGuard guard(device);
        STAGE_TRY(hipMalloc(&d_tables, bytes > 0 ? bytes : 16));
        if (bytes) STAGE_TRY(hipMemcpy(d_tables, host, bytes, hipMemcpyHostToDevice));
        return GFDM_HIP_OK;
    }
    ~StageCtx()
    {
        DeviceGuard guard(device);
        pipe.release();
        if (d_tables) (void)hipFree(d_tables);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

// the host-pointer form of a device entry point: out_pb / in_pb complex samples per block, enqueue(out, in, nb, stream) -> status
template <class Enqueue>
int run_host(StageCtx& c, float* out, size_t out_pb, const float* in, size_t in_pb, int64_t nblocks, Enqueue&& enqueue)
{
    if (nblocks < 0 || !out || !in) return api_fail(GFDM_HIP_EINVAL, "NULL buffer or negative block count");
    if (nblocks == 0) return GFDM_HIP_OK;
    DeviceGuard guard(c.device);
    if (!guard.ok) return api_fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    const gfdm::HostOperand ops[2] = { { out, out_pb * sizeof(cf), out_pb * sizeof(cf), true }, { const_cast<float*>(in), in_pb * sizeof(cf), in_pb * sizeof(cf), false } };
    auto fn = [&](void* const* d, int64_t nb, hipStream_t s) { return enqueue(static_cast<cf*>(d[0]), static_cast<const cf*>(d[1]), nb, s); };
    return c.pipe.run(c.stream, ops, 2, nblocks, fn);
}

constexpr int kThreads = 256;
constexpr int kMaxThreads = 1024;   // the LDS-staged kernels take one GFDM block per workgroup: large blocks get more lanes
constexpr unsigned kMaxGridY = 32768;

// ---- kernels: blockIdx.y strides over the GFDM blocks, blockIdx.x * 256 + threadIdx.x over the elements of one block's OUTPUT, so
// ---- every store instruction of a wave writes 512 contiguous bytes; loads are contiguous runs where the mapping allows it

// out[k M + t] = symbol of (subcarrier k, timeslot t), zero where k is inactive or the block has fewer symbols   mapper:74-89,108-134
__global__ __launch_bounds__(kThreads) void k_map_to_resources(cf* __restrict__ out, const cf* __restrict__ in, const short* __restrict__ rank,
                                                              int M, int N, int A, int per_timeslot, int nin, int64_t nblocks)
{
    const int f = blockIdx.x * kThreads + threadIdx.x;
    if (f >= N) return;
    const int k = f / M, t = f - k * M;
    const int a = rank[k];
    const int idx = per_timeslot ? (t * A + a) : (a * M + t);
    const bool live = (a >= 0) && (idx < nin);
    for (int64_t b = blockIdx.y; b < nblocks; b += gridDim.y) {
        const cf v = live ? gfdm::dft::ld_stream(in + b * nin + idx) : make_float2(0.f, 0.f);
        gfdm::dft::st_stream(out, b * N + f, v);
    }
}

// out[i] = grid value of the i-th data symbol                                                                  mapper:91-106,136-163
__global__ __launch_bounds__(kThreads) void k_demap_from_resources(cf* __restrict__ out, const cf* __restrict__ in, const int* __restrict__ smap,
                                                                  int M, int N, int A, int per_timeslot, int nout, int64_t nblocks)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= nout) return;
    int s, t;
    if (per_timeslot) { t = i / A; s = smap[i - t * A]; }
    else { const int a = i / M; t = i - a * M; s = smap[a]; }
    const int src = M * s + t;
    for (int64_t b = blockIdx.y; b < nblocks; b += gridDim.y)
        gfdm::dft::st_stream(out, b * nout + i, gfdm::dft::ld_stream(in + b * N + src));
}

// out[j] = in[(j - cp - shift) mod N], first and last `ramp` samples times the window ramps                    prefixer:66-98
__global__ __launch_bounds__(kThreads) void k_add_cyclic_prefix(cf* __restrict__ out, const cf* __restrict__ in, const cf* __restrict__ front,
                                                               const cf* __restrict__ back, int N, int cp, int cs, int ramp, int shift,
                                                               int64_t nblocks)
{
    const int F = cp + N + cs;
    const int j = blockIdx.x * kThreads + threadIdx.x;
    if (j >= F) return;
    int src = j - cp - shift;
    if (src < 0) src += N;
    if (src >= N) src -= N;
    cf w = make_float2(1.f, 0.f);
    const bool ramped = (j < ramp) || (j >= F - ramp);
    if (j < ramp) w = front[j];
    else if (j >= F - ramp) w = back[j - (F - ramp)];
    for (int64_t b = blockIdx.y; b < nblocks; b += gridDim.y) {
        cf x = gfdm::dft::ld_stream(in + b * N + src);
        if (ramped) x = make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x);
        gfdm::dft::st_stream(out, b * (int64_t)F + j, x);
    }
}

// out[i] = in[cp + i]                                                                                           prefixer:100-104
__global__ __launch_bounds__(kThreads) void k_remove_cyclic_prefix(cf* __restrict__ out, const cf* __restrict__ in, int N, int F, int cp,
                                                                  int64_t nblocks)
{
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= N) return;
    for (int64_t b = blockIdx.y; b < nblocks; b += gridDim.y)
        gfdm::dft::st_stream(out, b * N + i, gfdm::dft::ld_stream(in + b * (int64_t)F + cp + i));
}

// (q, r) = (i / d, i % d) for i = first, first + step, first + 2 step, ...: one division up front, carries afterwards
struct DivWalk {
    int q, r, dq, dr, d;
    __device__ DivWalk(int first, int step, int d_) : q(first / d_), r(first - (first / d_) * d_), dq(step / d_), dr(step - (step / d_) * d_), d(d_) {}
    __device__ void next()
    {
        q += dq;
        r += dr;
        if (r >= d) { r -= d; ++q; }
    }
};

// Per-timeslot symbol order is a transpose of the grid ([timeslot][active] <-> [subcarrier][timeslot]): gathered straight from
// memory, neighbouring lanes touch addresses 8 M (8 A) bytes apart, one cache line each.  These two kernels stage one GFDM block per
// workgroup through LDS instead -- linear (coalesced) on both memory sides, the transpose happens in LDS with an odd row stride.
//   map:   symbols [t][a] -> LDS [t][AP]; grid element (k, t) = LDS[t AP + rank[k]]
//   demap: grid [k][t] -> LDS [k][MP];    symbol (t, a) = LDS[smap[a] MP + t]
__global__ __launch_bounds__(kMaxThreads) void k_map_per_timeslot_lds(cf* __restrict__ out, const cf* __restrict__ in, const short* __restrict__ rank,
                                                                     int M, int N, int A, int AP, int nin, int64_t nblocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* sh = reinterpret_cast<cf*>(smem_raw);
    const int T = blockDim.x;
    for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        DivWalk w(threadIdx.x, T, A);                                          // i -> (t, a)
        for (int i = threadIdx.x; i < nin; i += T, w.next()) sh[w.q * AP + w.r] = gfdm::dft::ld_stream(in + b * nin + i);
        __syncthreads();
        DivWalk g(threadIdx.x, T, M);                                          // f -> (k, t)
        for (int f = threadIdx.x; f < N; f += T, g.next()) {
            const int a = rank[g.q];
            const bool live = (a >= 0) && (g.r * A + a < nin);
            gfdm::dft::st_stream(out, b * N + f, live ? sh[g.r * AP + a] : make_float2(0.f, 0.f));
        }
        __syncthreads();
    }
}

// only the active rows are read: A runs of M contiguous values
__global__ __launch_bounds__(kMaxThreads) void k_demap_per_timeslot_lds(cf* __restrict__ out, const cf* __restrict__ in, const int* __restrict__ smap,
                                                                       int M, int MP, int N, int A, int nout, int64_t nblocks)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf* sh = reinterpret_cast<cf*>(smem_raw);
    const int T = blockDim.x;
    for (int64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        DivWalk g(threadIdx.x, T, M);                                          // idx -> (a, t)
        for (int idx = threadIdx.x; idx < A * M; idx += T, g.next()) sh[g.q * MP + g.r] = gfdm::dft::ld_stream(in + b * N + smap[g.q] * M + g.r);
        __syncthreads();
        DivWalk w(threadIdx.x, T, A);                                          // i -> (t, a)
        for (int i = threadIdx.x; i < nout; i += T, w.next()) gfdm::dft::st_stream(out, b * nout + i, sh[w.r * MP + w.q]);
        __syncthreads();
    }
}

constexpr size_t kLdsLimit = 64 * 1024;          // default dynamic LDS per workgroup: larger blocks take the gather kernels

int staged_threads(int elems) { return elems <= 1024 ? 256 : elems <= 4096 ? 512 : kMaxThreads; }

dim3 grid_for(int elems, int64_t nblocks)
{
    return dim3((unsigned)((elems + kThreads - 1) / kThreads), (unsigned)std::min<int64_t>(nblocks, kMaxGridY));
}

}  // namespace

struct gfdm_hip_resource_mapper {
    StageCtx ctx;
    int M = 0, K = 0, A = 0, per_timeslot = 1;
    const short* d_rank = nullptr;      // [K] position of subcarrier k in the SORTED map, -1 when inactive
    const int* d_smap = nullptr;        // [A] sorted subcarrier map
};

struct gfdm_hip_cyclic_prefixer {
    StageCtx ctx;
    int N = 0, cp = 0, cs = 0, ramp = 0, shift = 0;
    const cf* d_front = nullptr;
    const cf* d_back = nullptr;
};

extern "C" {

int gfdm_hip_resource_mapper_create(gfdm_hip_resource_mapper** out, int timeslots, int subcarriers, int active_subcarriers,
                                    const int* subcarrier_map, int n_subcarrier_map, int per_timeslot, int device)
{
    if (!out) return api_fail(GFDM_HIP_EINVAL, "NULL handle pointer");
    *out = nullptr;
    const int M = timeslots, K = subcarriers, A = active_subcarriers;
    char buf[256];
    if (M < 1 || K < 1 || A < 1) return api_fail(GFDM_HIP_EINVAL, "timeslots, subcarriers and active_subcarriers must be >= 1");
    if (A > K) {                                                                                                  // mapper:44-49
        snprintf(buf, sizeof(buf), "active_subcarriers(%d) MUST be smaller or equal to subcarriers(%d)!", A, K);
        return api_fail(GFDM_HIP_EINVAL, buf);
    }
    if (!subcarrier_map || n_subcarrier_map != A) {                                                               // mapper:50-55
        snprintf(buf, sizeof(buf), "number of subcarrier_map entries(%d) MUST be equal to active_subcarriers(%d)!", n_subcarrier_map, A);
        return api_fail(GFDM_HIP_EINVAL, buf);
    }
    std::vector<int> smap(subcarrier_map, subcarrier_map + A);
    std::sort(smap.begin(), smap.end());                                                                          // mapper:56
    if (std::adjacent_find(smap.begin(), smap.end()) != smap.end()) return api_fail(GFDM_HIP_EINVAL, "All entries in subcarrier_map MUST be unique!");
    if (smap.front() < 0) return api_fail(GFDM_HIP_EINVAL, "All subcarrier indices MUST be greater or equal to ZERO!");
    // the reference accepts an index EQUAL to `subcarriers` (mapper:65: '>' where '>=' is meant) and then writes one row past the grid
    if (smap.back() >= K) return api_fail(GFDM_HIP_EINVAL, "All subcarrier indices MUST be smaller than subcarriers!");
    if ((int64_t)M * K > (int64_t)1 << 30) return api_fail(GFDM_HIP_EINVAL, "frame too large");
    if (A > 32767) return api_fail(GFDM_HIP_EUNSUPPORTED, "more than 32767 active subcarriers (the rank table holds 16-bit positions)");

    gfdm_hip_resource_mapper* m = new (std::nothrow) gfdm_hip_resource_mapper();
    if (!m) return api_fail(GFDM_HIP_ENOMEM, "out of host memory");
    int rc = m->ctx.open(device);
    if (rc != GFDM_HIP_OK) { delete m; return rc; }
    const size_t rank_bytes = ((size_t)K * sizeof(short) + 15) / 16 * 16;
    std::vector<unsigned char> blob(rank_bytes + (size_t)A * sizeof(int), 0);
    short* rank = reinterpret_cast<short*>(blob.data());
    for (int k = 0; k < K; ++k) rank[k] = -1;
    for (int a = 0; a < A; ++a) rank[smap[a]] = (short)a;
    memcpy(blob.data() + rank_bytes, smap.data(), (size_t)A * sizeof(int));
    rc = m->ctx.upload(blob.data(), blob.size());
    if (rc != GFDM_HIP_OK) { delete m; return rc; }
    m->M = M; m->K = K; m->A = A; m->per_timeslot = per_timeslot ? 1 : 0;
    m->d_rank = reinterpret_cast<const short*>(m->ctx.d_tables);
    m->d_smap = reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(m->ctx.d_tables) + rank_bytes);
    *out = m;
    return GFDM_HIP_OK;
}

int gfdm_hip_resource_mapper_destroy(gfdm_hip_resource_mapper* m) { delete m; return GFDM_HIP_OK; }
int gfdm_hip_resource_mapper_block_size(const gfdm_hip_resource_mapper* m) { return m ? m->A * m->M : GFDM_HIP_EINVAL; }
int gfdm_hip_resource_mapper_frame_size(const gfdm_hip_resource_mapper* m) { return m ? m->K * m->M : GFDM_HIP_EINVAL; }

static int mapper_check(const gfdm_hip_resource_mapper* m, int n, const char* what)
{
    if (n < 0 || n > m->A * m->M) {                                                                               // mapper:78-82, 95-99
        char buf[200];
        snprintf(buf, sizeof(buf), "%s vector size(%d) MUST not exceed active_subcarriers * timeslots(%d)!", what, n, m->A * m->M);
        return api_fail(GFDM_HIP_EINVAL, buf);
    }
    return GFDM_HIP_OK;
}

int gfdm_hip_resource_mapper_map_device(gfdm_hip_resource_mapper* m, void* out, const void* in, int ninput_size, int64_t nblocks, void* stream)
{
    if (!m) return api_fail(GFDM_HIP_EINVAL, "NULL handle");
    int rc = mapper_check(m, ninput_size, "input");
    if (rc != GFDM_HIP_OK) return rc;
    if (nblocks < 0 || !out || (!in && ninput_size > 0)) return api_fail(GFDM_HIP_EINVAL, "NULL buffer or negative block count");
    if (nblocks == 0) return GFDM_HIP_OK;
    DeviceGuard guard(m->ctx.device);
    if (!guard.ok) return api_fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    const int N = m->K * m->M;
    const int AP = m->A | 1;
    const size_t lds = (size_t)m->M * AP * sizeof(cf);
    if (m->per_timeslot && m->M > 1 && lds <= kLdsLimit)
        hipLaunchKernelGGL(k_map_per_timeslot_lds, dim3((unsigned)std::min<int64_t>(nblocks, 1 << 20)), dim3(staged_threads(N)), lds, (hipStream_t)stream, (cf*)out,
                           (const cf*)in, m->d_rank, m->M, N, m->A, AP, ninput_size, nblocks);
    else
        hipLaunchKernelGGL(k_map_to_resources, grid_for(N, nblocks), dim3(kThreads), 0, (hipStream_t)stream, (cf*)out, (const cf*)in, m->d_rank, m->M, N,
                           m->A, m->per_timeslot, ninput_size, nblocks);
    STAGE_TRY(hipGetLastError());
    return GFDM_HIP_OK;
}

int gfdm_hip_resource_mapper_demap_device(gfdm_hip_resource_mapper* m, void* out, const void* in, int noutput_size, int64_t nblocks, void* stream)
{
    if (!m) return api_fail(GFDM_HIP_EINVAL, "NULL handle");
    int rc = mapper_check(m, noutput_size, "output");
    if (rc != GFDM_HIP_OK) return rc;
    if (nblocks < 0 || !in || (!out && noutput_size > 0)) return api_fail(GFDM_HIP_EINVAL, "NULL buffer or negative block count");
    if (nblocks == 0 || noutput_size == 0) return GFDM_HIP_OK;
    DeviceGuard guard(m->ctx.device);
    if (!guard.ok) return api_fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    const int MP = m->M | 1;
    const size_t lds = (size_t)m->A * MP * sizeof(cf);
    if (m->per_timeslot && m->M > 1 && lds <= kLdsLimit)
        hipLaunchKernelGGL(k_demap_per_timeslot_lds, dim3((unsigned)std::min<int64_t>(nblocks, 1 << 20)), dim3(staged_threads(m->A * m->M)), lds, (hipStream_t)stream,
                           (cf*)out, (const cf*)in, m->d_smap, m->M, MP, m->K * m->M, m->A, noutput_size, nblocks);
    else
        hipLaunchKernelGGL(k_demap_from_resources, grid_for(noutput_size, nblocks), dim3(kThreads), 0, (hipStream_t)stream, (cf*)out, (const cf*)in,
                           m->d_smap, m->M, m->K * m->M, m->A, m->per_timeslot, noutput_size, nblocks);
    STAGE_TRY(hipGetLastError());
    return GFDM_HIP_OK;
}

int gfdm_hip_resource_mapper_map_host(gfdm_hip_resource_mapper* m, float* out, const float* in, int ninput_size, int64_t nblocks)
{
    if (!m) return api_fail(GFDM_HIP_EINVAL, "NULL handle");
    int rc = mapper_check(m, ninput_size, "input");
    if (rc != GFDM_HIP_OK) return rc;
    return run_host(m->ctx, out, (size_t)(m->K * m->M), in, (size_t)ninput_size, nblocks, [&](cf* o, const cf* i, int64_t nb, hipStream_t s) {
        return gfdm_hip_resource_mapper_map_device(m, o, i, ninput_size, nb, (void*)s);
    });
}

int gfdm_hip_resource_mapper_demap_host(gfdm_hip_resource_mapper* m, float* out, const float* in, int noutput_size, int64_t nblocks)
{
    if (!m) return api_fail(GFDM_HIP_EINVAL, "NULL handle");
    int rc = mapper_check(m, noutput_size, "output");
    if (rc != GFDM_HIP_OK) return rc;
    return run_host(m->ctx, out, (size_t)noutput_size, in, (size_t)(m->K * m->M), nblocks, [&](cf* o, const cf* i, int64_t nb, hipStream_t s) {
        return gfdm_hip_resource_mapper_demap_device(m, o, i, noutput_size, nb, (void*)s);
    });
}

// ---- cyclic prefixer ------------------------------------------------------------------------------------------------------------

int gfdm_hip_cyclic_prefixer_create(gfdm_hip_cyclic_prefixer** out, int block_len, int cp_len, int cs_len, int ramp_len, const float* window_taps,
                                    int n_window_taps, int cyclic_shift, int device)
{
    if (!out) return api_fail(GFDM_HIP_EINVAL, "NULL handle pointer");
    *out = nullptr;
    char buf[256];
    if (block_len < 1 || cp_len < 0 || cs_len < 0 || ramp_len < 0) return api_fail(GFDM_HIP_EINVAL, "block_len must be >= 1, cp_len / cs_len / ramp_len >= 0");
    const int64_t window_len = (int64_t)block_len + cp_len + cs_len;
    if (window_len > (int64_t)1 << 30) return api_fail(GFDM_HIP_EINVAL, "frame too large");
    if (n_window_taps < 0 || (n_window_taps > 0 && !window_taps) || (n_window_taps != window_len && n_window_taps != 2 * ramp_len)) {   // prefixer:42-50
        snprintf(buf, sizeof(buf), "number of window taps(%d) MUST be equal to 2*ramp_len(%d) OR block_len+cp_len (%d)!", n_window_taps, 2 * ramp_len,
                 (int)window_len);
        return api_fail(GFDM_HIP_EINVAL, buf);
    }
    if (2 * (int64_t)ramp_len > window_len) return api_fail(GFDM_HIP_EINVAL, "ramp_len too large for the frame");
    // the reference copies in[block - cp - shift ...) and in[0, cs - shift): shifts outside [0, cs] or cp + shift > block read out of bounds there
    // ... and so does a suffix longer than the block (in[0, cs - shift) with cs - shift > block)
    if (cyclic_shift < 0 || cyclic_shift > cs_len || (int64_t)cp_len + cyclic_shift > block_len || (int64_t)cs_len - cyclic_shift > block_len)
        return api_fail(GFDM_HIP_EINVAL, "cyclic shift must lie in [0, cs_len], cp_len + shift and cs_len - shift must not exceed the block");
    gfdm_hip_cyclic_prefixer* c = new (std::nothrow) gfdm_hip_cyclic_prefixer();
    if (!c) return api_fail(GFDM_HIP_ENOMEM, "out of host memory");
    int rc = c->ctx.open(device);
    if (rc != GFDM_HIP_OK) { delete c; return rc; }
    std::vector<cf> ramps((size_t)2 * (ramp_len > 0 ? ramp_len : 1), make_float2(1.f, 0.f));
    const cf* w = reinterpret_cast<const cf*>(window_taps);
    for (int i = 0; i < ramp_len; ++i) {
        ramps[i] = w[i];                                                  // front ramp  :51-53
        ramps[ramp_len + i] = w[n_window_taps - ramp_len + i];            // back ramp   :54-56
    }
    rc = c->ctx.upload(ramps.data(), ramps.size() * sizeof(cf));
    if (rc != GFDM_HIP_OK) { delete c; return rc; }
    c->N = block_len; c->cp = cp_len; c->cs = cs_len; c->ramp = ramp_len; c->shift = cyclic_shift;
    c->d_front = reinterpret_cast<const cf*>(c->ctx.d_tables);
    c->d_back = c->d_front + (ramp_len > 0 ? ramp_len : 1);
    *out = c;
    return GFDM_HIP_OK;
}

int gfdm_hip_cyclic_prefixer_destroy(gfdm_hip_cyclic_prefixer* c) { delete c; return GFDM_HIP_OK; }
int gfdm_hip_cyclic_prefixer_block_size(const gfdm_hip_cyclic_prefixer* c) { return c ? c->N : GFDM_HIP_EINVAL; }
int gfdm_hip_cyclic_prefixer_frame_size(const gfdm_hip_cyclic_prefixer* c) { return c ? c->N + c->cp + c->cs : GFDM_HIP_EINVAL; }
int gfdm_hip_cyclic_prefixer_cyclic_shift(const gfdm_hip_cyclic_prefixer* c) { return c ? c->shift : GFDM_HIP_EINVAL; }

int gfdm_hip_cyclic_prefixer_add_device(gfdm_hip_cyclic_prefixer* c, void* out, const void* in, int cyclic_shift, int64_t nblocks, void* stream)
{
    if (!c) return api_fail(GFDM_HIP_EINVAL, "NULL handle");
    if (cyclic_shift < 0 || cyclic_shift > c->cs || c->cp + cyclic_shift > c->N || c->cs - cyclic_shift > c->N)
        return api_fail(GFDM_HIP_EINVAL, "cyclic shift must lie in [0, cs_len], cp_len + shift and cs_len - shift must not exceed the block");
    if (nblocks < 0 || !out || !in) return api_fail(GFDM_HIP_EINVAL, "NULL buffer or negative block count");
    if (nblocks == 0) return GFDM_HIP_OK;
    DeviceGuard guard(c->ctx.device);
    if (!guard.ok) return api_fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    hipLaunchKernelGGL(k_add_cyclic_prefix, grid_for(c->N + c->cp + c->cs, nblocks), dim3(kThreads), 0, (hipStream_t)stream, (cf*)out, (const cf*)in,
                       c->d_front, c->d_back, c->N, c->cp, c->cs, c->ramp, cyclic_shift, nblocks);
    STAGE_TRY(hipGetLastError());
    return GFDM_HIP_OK;
}

int gfdm_hip_cyclic_prefixer_remove_device(gfdm_hip_cyclic_prefixer* c, void* out, const void* in, int64_t nblocks, void* stream)
{
    if (!c) return api_fail(GFDM_HIP_EINVAL, "NULL handle");
    if (nblocks < 0 || !out || !in) return api_fail(GFDM_HIP_EINVAL, "NULL buffer or negative block count");
    if (nblocks == 0) return GFDM_HIP_OK;
    DeviceGuard guard(c->ctx.device);
    if (!guard.ok) return api_fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    hipLaunchKernelGGL(k_remove_cyclic_prefix, grid_for(c->N, nblocks), dim3(kThreads), 0, (hipStream_t)stream, (cf*)out, (const cf*)in, c->N,
                       c->N + c->cp + c->cs, c->cp, nblocks);
    STAGE_TRY(hipGetLastError());
    return GFDM_HIP_OK;
}

int gfdm_hip_cyclic_prefixer_add_host(gfdm_hip_cyclic_prefixer* c, float* out, const float* in, int cyclic_shift, int64_t nblocks)
{
    if (!c) return api_fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_host(c->ctx, out, (size_t)(c->N + c->cp + c->cs), in, (size_t)c->N, nblocks, [&](cf* o, const cf* i, int64_t nb, hipStream_t s) {
        return gfdm_hip_cyclic_prefixer_add_device(c, o, i, cyclic_shift, nb, (void*)s);
    });
}

int gfdm_hip_cyclic_prefixer_remove_host(gfdm_hip_cyclic_prefixer* c, float* out, const float* in, int64_t nblocks)
{
    if (!c) return api_fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_host(c->ctx, out, (size_t)c->N, in, (size_t)(c->N + c->cp + c->cs), nblocks, [&](cf* o, const cf* i, int64_t nb, hipStream_t s) {
        return gfdm_hip_cyclic_prefixer_remove_device(c, o, i, nb, (void*)s);
    });
}

}  // extern "C"
