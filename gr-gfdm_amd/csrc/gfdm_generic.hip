// Generic HIP kernel family: any (timeslots M, subcarriers K, overlap L) whose block fits LDS.
//
// One 256-thread workgroup per GFDM block.  The K x M block lives in LDS tiles for the whole
// computation, so HBM sees exactly one read of the inputs and one write of the output.
// The N = K*M point transform is factored along GFDM's own structure (n = K p + q, f = M j + m):
//     X[M j + m] = sum_q W_K^{q j} * W_N^{q m} * ( sum_p x[K p + q] W_M^{p m} )
// i.e. M-point DFTs over the timeslot axis, a twiddle, K-point FFTs over the subcarrier axis.
// This family uses table-driven direct DFTs for the M axis and a mixed-radix Stockham FFT (any K; a
// prime K degenerates to the direct DFT) for the K axis: it is the shape-agnostic path, the
// tuned family for the benchmark shapes is gfdm_rowlane_impl.h.
//
// Algorithm restated from (gr-gfdm checkout):
//   modulator  lib/modulator_kernel_cc.cc:98-141
//   receiver   lib/receiver_kernel_cc.cc:165-192, 211-225, 274-334
//   IC loop    lib/advanced_receiver_kernel_cc.cc:56-123
#include "gfdm_plan.h"
#include "gfdm_tx.h"
#include "gfdm_est.h"

namespace gfdm {
namespace {

constexpr int GT = 256;           // threads per workgroup
constexpr int RED_BYTES = 1024;   // reduction scratch carved from the dynamic LDS region

// Diagnostic build only (-DGFDM_STAMPS, scratch/stamps_generic.py): timestamps of the phase boundaries, one set per wavefront, written to a
// side buffer that nothing else reads.  The product library is built without it.
#ifdef GFDM_STAMPS
__device__ unsigned long long* g_gstamp_buf = nullptr;
#define GFDM_GSTAMP(slot)                                                                                        \
    do {                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                              \
        if (g_gstamp_buf && (threadIdx.x & 63) == 0)                                                              \
            g_gstamp_buf[((size_t)blockIdx.x * (GT / 64) + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define GFDM_GSTAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf cmulj(cf a, cf b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a * conj(b)
__device__ __forceinline__ cf cadd(cf a, cf b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cf csub(cf a, cf b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cf cfma(cf a, cf b, cf c) { return make_float2(c.x + a.x * b.x - a.y * b.y, c.y + a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf cfmaj(cf a, cf b, cf c) { return make_float2(c.x + a.x * b.x + a.y * b.y, c.y + a.y * b.x - a.x * b.y); }
__device__ __forceinline__ cf cdiv(cf a, cf b)
{
    const float d = b.x * b.x + b.y * b.y;
    return make_float2((a.x * b.x + a.y * b.y) / d, (a.y * b.x - a.x * b.y) / d);
}

// (idx / d, idx % d) for idx = start, start + step, ...: one division at the start, then increments (the generic kernels walk
// every tile with idx = threadIdx.x + i * GT and need row / column of each element; d is a run-time value)
struct DivStep {
    int q, r, dq, dr, d;
    __device__ __forceinline__ DivStep(int start, int step, int den) : d(den)
    {
        q = start / den; r = start - q * den;
        dq = step / den; dr = step - dq * den;
    }
    __device__ __forceinline__ void next()
    {
        q += dq; r += dr;
        if (r >= d) { r -= d; ++q; }
    }
};

// The roots W_M and W_K are read once per multiply-add of the table-driven transforms: every kernel copies them from global
// memory into LDS first (behind its tiles, at byte offset tab_off of the dynamic region) and works on a plan that points there.
__device__ __forceinline__ DevicePlan stage_tables(const DevicePlan& p, unsigned char* smem, int tab_off)
{
    cf* lwM = reinterpret_cast<cf*>(smem + tab_off);
    cf* lwK = lwM + p.M;
    cf* lg = lwK + p.K;
    for (int i = threadIdx.x; i < p.M; i += GT) { lwM[i] = p.wM[i]; lg[i] = p.icg[i]; }
    for (int i = threadIdx.x; i < p.K; i += GT) lwK[i] = p.wK[i];
    DevicePlan q = p;
    q.wM = lwM;
    q.wK = lwK;
    q.icg = lg;
    return q;                                   // the caller's first __syncthreads() publishes the tables
}

// Global -> LDS copy of a block with up to EIGHT loads of a thread in flight (a plain `dst[i] = src[i]` loop of unknown length waits for every load before
// it issues the next one: eight DRAM round trips per thread for a 2000-sample block).  op(value, index) is applied on the way.
template <class Op>
__device__ __forceinline__ void stream_in(cf* dst, const cf* __restrict__ src, int n, Op op)
{
    int idx = threadIdx.x;
    for (; idx + 7 * GT < n; idx += 8 * GT) {
        cf v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[idx + j * GT];
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[idx + j * GT] = op(v[j], idx + j * GT);
    }
    for (; idx + 3 * GT < n; idx += 4 * GT) {
        cf v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = src[idx + j * GT];
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[idx + j * GT] = op(v[j], idx + j * GT);
    }
    for (; idx < n; idx += GT) dst[idx] = op(src[idx], idx);
}
__device__ __forceinline__ void stream_in(cf* dst, const cf* __restrict__ src, int n)
{
    stream_in(dst, src, n, [](cf v, int) { return v; });
}

// Threads own COLUMNS of a [rows][M] tile: thread t takes column t % M of the rows t / M, t / M + G, ... (G = GT / M row groups; M > GT: columns
// t, t + GT, ... of every row), so that whatever depends on the column only -- the filter taps, which live in global memory: an L2 round trip of
// ~0.5 us each -- is fetched once per thread instead of once per element.  body(m, first row, row step).
template <class Body>
__device__ __forceinline__ void for_columns(int M, Body body)
{
    if (M <= GT) {
        const int G = GT / M, g = threadIdx.x / M, m = threadIdx.x - g * M;
        if (g < G) body(m, g, G);
    } else {
        for (int m = threadIdx.x; m < M; m += GT) body(m, 0, 1);
    }
}
constexpr int TAPS_IN_REGS = 4;    // overlap factors up to this keep a column's taps in registers (the reference's flowgraphs use 2; longer filters take the loop)

// One pass over the samples v_p and the roots w_p = W_M^{p m} yields TWO outputs of the direct DFT, m and M - m (W_M^{p (M - m)} = conj(w_p)),
// and it takes the samples in PAIRS as well: w_{M - p} = conj(w_p), so with a = v_p + v_{M-p}, b = v_p - v_{M-p} and w_p = (c, s)
//     v_p w_p + v_{M-p} conj(w_p) = a c + j b s          v_p conj(w_p) + v_{M-p} w_p = a c - j b s
// i.e. the four sums Q1 = sum a.x c, Q2 = sum a.y c, Q3 = sum b.x s, Q4 = sum b.y s give  sum v w = (Q1 - Q4, Q2 + Q3)  and
// sum v conj(w) = (Q1 + Q4, Q2 - Q3): four multiply-adds and ONE root per two samples and two outputs (the symmetric form of gfdm_dft.h's
// prime codelet, table driven).  The sample p = 0 enters through the start values, the middle sample of an even M through add(v, 0, w).
struct DftPair {
    float q1 = 0.f, q2 = 0.f, q3 = 0.f, q4 = 0.f;
    __device__ __forceinline__ void start(cf v0) { q1 = v0.x; q2 = v0.y; }
    __device__ __forceinline__ void add(cf a, cf b, cf w)
    {
        q1 = fmaf(a.x, w.x, q1);
        q2 = fmaf(a.y, w.x, q2);
        q3 = fmaf(b.x, w.y, q3);
        q4 = fmaf(b.y, w.y, q4);
    }
    __device__ __forceinline__ cf with_root() const { return make_float2(q1 - q4, q2 + q3); }        // sum v w
    __device__ __forceinline__ cf with_conj() const { return make_float2(q1 + q4, q2 - q3); }        // sum v conj(w)
};

// The direct timeslot transforms of this family, one loop for all of them: work items (row r < rows, output pair m < H = M/2 + 1), each the sum over
// p < M of src(r, p) W_M^{p m} in the paired form above, handed to fin(r, m, acc).  These loops are LDS-bound -- two LDS reads (sample, root) per
// four multiply-adds -- so a thread takes NB CONSECUTIVE pairs of one row at a time where the block has enough work for that (register blocking:
// one sample pair feeds NB output pairs).  ROWFAST: neighbouring threads
// take neighbouring rows of one pair block (the modulator's last stage, whose stores want the row index fastest), else neighbouring pair blocks.
template <int NB, bool ROWFAST, class Src, class Fin>
__device__ __forceinline__ void paired_dft_blocked(int rows, int M, const cf* __restrict__ wM, Src src, Fin fin)
{
    const int H = M / 2 + 1, HB = (H + NB - 1) / NB, HP = (M - 1) / 2;
    DivStep ix(threadIdx.x, GT, ROWFAST ? rows : HB);
    for (int idx = threadIdx.x; idx < rows * HB; idx += GT, ix.next()) {
        const int r = ROWFAST ? ix.r : ix.q, m0 = (ROWFAST ? ix.q : ix.r) * NB;
        const cf v0 = src(r, 0);
        DftPair acc[NB];
        int e[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) { acc[j].start(v0); e[j] = m0 + j; }     // p m mod M for p = 1 (a pair index past H - 1 in the last block is still < M)
        for (int p = 1; p <= HP; ++p) {
            const cf vp = src(r, p), vq = src(r, M - p);
            const cf a = cadd(vp, vq), b = csub(vp, vq);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                acc[j].add(a, b, wM[e[j]]);
                e[j] += m0 + j;
                if (e[j] >= M) e[j] -= M;
            }
        }
        if ((M & 1) == 0) {                               // the middle sample of an even M: W_M^{(M/2) m} = +-1
            const cf v = src(r, M / 2);
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j].add(v, make_float2(0.f, 0.f), wM[e[j]]);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j)
            if (m0 + j < H) fin(r, m0 + j, acc[j]);
    }
}

// The same transforms on the matrix cores (from MX_DFT_MIN_M timeslots on): the four sums above are products of CONSTANT matrices with the block's
// samples,  [Q1 | Q2] = C [a.x | a.y],  [Q3 | Q4] = S [b.x | b.y],  C[m][p] = Re W_M^{m p}, S[m][p] = Im W_M^{m p}  (rows: the H = M/2 + 1 output
// pairs; columns: p = 0 -- the sample v_0 itself, C = 1 --, the sample pairs p = 1..(M-1)/2, the middle sample of an even M), i.e. real GEMMs of
// H x ~M/2 x (2 x rows) per transform -- a dense contraction, f32 in, f32 sums (v_mfma_f32_16x16x4_f32), nothing reduced in precision.
//   A operand: the handle's table DevicePlan::dftA (gfdm_hip_api.hip), [output tile][k-step][cos, sin][lane], read from global memory (L1 / L2 hits)
//   B operand: the sums a and differences b of one 16-row group, written once into the LDS scratch xs as four planes [a.x][a.y][b.x][b.y] of
//              [4 KS][16 rows] floats: the operand of k-step ks is the 64 consecutive floats at 64 ks (lane l: k = l >> 4, row l & 15, conflict-free)
//   D: lane l holds the outputs m = 16 mt + 4 (l >> 4) + i, i < 4, of row l & 15: Q1..Q4 of one (row, output pair) end up in ONE lane, so the epilogue is
//      the same fin(r, m, DftPair) as in the vector-ALU form.
// A wavefront takes (output tile, row group) units.  The operands sit in whichever LDS tile is free at that point (modulator / receiver kernels: MxDft::at,
// all row groups at once) or in a scratch region of rtc row groups (stand-alone and global-scratch kernels); launch code: mx_alias / mx_lds.

// A operands of one output tile, k-steps [ks0, ks0 + 16): 32 registers.  The table does not depend on the samples: a kernel fetches the operands of the
// output tile its wavefront starts with at ENTRY (MxDft::preload) and keeps them -- every transform of the kernel uses the same table, and with up
// to 16 k-steps and one unit per wavefront (M = 127, K = 16) no transform waits for the L2 again.
struct MxA {
    float c[16], s[16];
    __device__ __forceinline__ void load(const float* __restrict__ A, int ks0, int KS)
    {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool in = ks0 + j < KS;                          // uniform
            c[j] = in ? A[(ks0 + j) * 128] : 0.f;
            s[j] = in ? A[(ks0 + j) * 128 + 64] : 0.f;
        }
    }
};

struct MxDft {
    const float* A;
    float* xs;
    int rtc, MT, KS;
    int alias;             // 1: no scratch of its own -- a transform borrows the tile that is free at that point (`at`) and takes all row groups at once
    // the transform of `rows` rows whose operands go to the free tile `tile` (alias form; else the scratch region as it is)
    __device__ __forceinline__ MxDft at(cf* tile, int rows) const
    {
        MxDft m = *this;
        if (alias) { m.xs = reinterpret_cast<float*>(tile); m.rtc = (rows + 15) / 16; }
        return m;
    }
    __device__ __forceinline__ bool in_place() const { return alias != 0; }
    MxA a;                 // operands of output tile a_mt, first 16 k-steps
    int a_mt;
    __device__ __forceinline__ void preload()
    {
        a_mt = (int)(threadIdx.x >> 6) % MT;
        a.load(A + (size_t)a_mt * KS * 128 + (threadIdx.x & 63), 0, KS);
    }
};

typedef float mx_f4 __attribute__((ext_vector_type(4)));

template <class Src, class Fin>
__device__ __forceinline__ void mx_dft(const MxDft& mx, int rows, int M, Src src, Fin fin)
{
    const int H = M / 2 + 1, HP = (M - 1) / 2, KD = HP + 1 + ((M & 1) == 0 ? 1 : 0), KDp = 4 * mx.KS, RT = (rows + 15) / 16;
    const int plane = KDp * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    MxA a = mx.a;                                                  // (fetched at kernel entry)
    int a_mt = mx.a_mt;                                            // the output tile whose first 16 k-steps sit in a
    for (int rt0 = 0; rt0 < RT; rt0 += mx.rtc) {
        const int nrt = (RT - rt0 < mx.rtc) ? RT - rt0 : mx.rtc;
        if (rt0) __syncthreads();                                  // the previous chunk's operands are spent
        DivStep ix(threadIdx.x >> 4, GT >> 4, KDp);
        for (int idx = threadIdx.x; idx < nrt * plane; idx += GT, ix.next()) {
            const int n = idx & 15, kk = ix.r, rt = ix.q;
            const int r = 16 * (rt0 + rt) + n;
            cf va = make_float2(0.f, 0.f), vb = va;
            if (r < rows && kk < KD) {
                if (kk == 0) va = src(r, 0);
                else if (kk <= HP) { const cf vp = src(r, kk), vq = src(r, M - kk); va = cadd(vp, vq); vb = csub(vp, vq); }
                else va = src(r, M / 2);
            }
            float* x = mx.xs + (rt * 4) * plane + kk * 16 + n;
            x[0] = va.x; x[plane] = va.y; x[2 * plane] = vb.x; x[3 * plane] = vb.y;
        }
        __syncthreads();
        for (int u = wave; u < mx.MT * nrt; u += GT / 64) {
            const int rt = u / mx.MT, mt = u - rt * mx.MT;
            const float* __restrict__ A = mx.A + (size_t)mt * mx.KS * 128 + lane;
            const float* X = mx.xs + (rt * 4) * plane + lane;
            mx_f4 q1 = { 0.f, 0.f, 0.f, 0.f }, q2 = q1, q3 = q1, q4 = q1;
            for (int ks0 = 0; ks0 < mx.KS; ks0 += 16) {
                if (ks0 || mt != a_mt) {                          // (uniform) not the operands this wavefront holds
                    a.load(A, ks0, mx.KS);
                    a_mt = ks0 ? -1 : mt;
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    if (ks0 + j < mx.KS) {                         // uniform
                        const int ks = ks0 + j;
                        const float ax = X[ks * 64], ay = X[plane + ks * 64], bx = X[2 * plane + ks * 64], by = X[3 * plane + ks * 64];
                        q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.c[j], ax, q1, 0, 0, 0);
                        q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.c[j], ay, q2, 0, 0, 0);
                        q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.s[j], bx, q3, 0, 0, 0);
                        q4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.s[j], by, q4, 0, 0, 0);
                    }
                }
            }
            const int r = 16 * (rt0 + rt) + (lane & 15);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 16 * mt + 4 * (lane >> 4) + i;
                if (m < H && r < rows) {
                    DftPair acc;
                    acc.q1 = q1[i]; acc.q2 = q2[i]; acc.q3 = q3[i]; acc.q4 = q4[i];
                    fin(r, m, acc);
                }
            }
        }
    }
}

// kernels come in two instantiations: with the matrix-core transforms (an MxDft in hand) and without (NoMx: the vector-ALU loops; these do not
// carry the accumulator registers of the other form, which would cost the small shapes their occupancy)
struct NoMx {
    __device__ __forceinline__ NoMx at(cf*, int) const { return NoMx{}; }
    __device__ __forceinline__ bool in_place() const { return false; }
};

template <bool ROWFAST, class Src, class Fin>
__device__ __forceinline__ void paired_dft(const MxDft& mx, int rows, int M, const cf* __restrict__, Src src, Fin fin)
{
    mx_dft(mx, rows, M, src, fin);
}

template <bool ROWFAST, class Src, class Fin>
__device__ __forceinline__ void paired_dft(const NoMx&, int rows, int M, const cf* __restrict__ wM, Src src, Fin fin)
{
    if (rows * (M / 2 + 1) >= 4 * GT) paired_dft_blocked<4, ROWFAST>(rows, M, wM, src, fin);
    else paired_dft_blocked<1, ROWFAST>(rows, M, wM, src, fin);
}

// dst[r*M + m] = scale * sum_p src[r*rs + p*ps] * W_M^{+-(p m)}      (dst may be LDS or global); one work item per pair (m, M - m)
template <bool INV, class Mx>
__device__ __forceinline__ void row_dft(const Mx& mx, cf* dst, const cf* src, int rows, int M, int rs, int ps, const cf* __restrict__ wM, float scale)
{
    paired_dft<false>(mx, rows, M, wM, [&](int r, int p) { return src[r * rs + p * ps]; },
                      [&](int r, int m, const DftPair& acc) {
                          const int m2 = (m == 0) ? 0 : M - m;
                          const cf ya = INV ? acc.with_conj() : acc.with_root(), yb = INV ? acc.with_root() : acc.with_conj();
                          dst[r * M + m] = make_float2(ya.x * scale, ya.y * scale);
                          if (m2 != m) dst[r * M + m2] = make_float2(yb.x * scale, yb.y * scale);
                      });
}

// K-point transform along the subcarrier axis of a [K][M] LDS tile, all M columns at once: mixed-radix Stockham autosort
// (decimation in frequency) for ANY K.  Each pass takes the radix r = 4 while 4 divides the remaining length n, else its
// smallest prime factor: with stride s (product of the radices so far) and m = n / r,
//     y[q + s (r p + j)] = W_n^{p j} * sum_k x[q + s (p + m k)] W_r^{j k},     q < s, p < m, j < r
// (a prime K is one pass of radix K, i.e. the direct DFT).  All roots come from the table wK[i] = exp(-2 pi i / K).
// Data in `a`, scratch `b`; returns the tile that holds the result.  Caller syncs before.
__device__ __forceinline__ int next_radix(int n)
{
    if ((n & 3) == 0) return 4;
    if ((n & 1) == 0) return 2;
    for (int f = 3; f * f <= n; f += 2)
        if (n % f == 0) return f;
    return n;
}

template <bool INV>
__device__ __forceinline__ cf* col_fft(cf* a, cf* b, const DevicePlan& p)
{
    const int M = p.M, K = p.K;
    const cf* __restrict__ wK = p.wK;
    cf* x = a;
    cf* y = b;
    int n = K, s = 1;
    while (n > 1) {
        const int r = next_radix(n), m = n / r;
        const int rstep = K / r;                                  // W_r^1 = wK[K / r]
        const int total = m * s * M;                              // (butterfly, column) pairs of this pass
        const int in_step = s * m * M, out_step = s * M;
        if (r == 4 || r == 2) {
            DivStep bx(threadIdx.x, GT, M);
            for (int idx = threadIdx.x; idx < total; idx += GT, bx.next()) {
                const int bf = bx.q, col = bx.r;
                const int pp = (s == 1) ? bf : bf / s, q = bf - pp * s;
                const cf* xin = x + (q + s * pp) * M + col;           // element k at xin[s m k M]
                cf* yout = y + (q + s * r * pp) * M + col;            // element j at yout[s j M]
                if (r == 4) {
                    const cf x0 = xin[0], x1 = xin[in_step], x2 = xin[2 * in_step], x3 = xin[3 * in_step];
                    const cf apc = cadd(x0, x2), amc = csub(x0, x2), bpd = cadd(x1, x3), t = csub(x1, x3);
                    const cf bmd = INV ? make_float2(-t.y, t.x) : make_float2(t.y, -t.x);        // -+ j (x1 - x3)
                    const cf y0 = cadd(apc, bpd), y1 = cadd(amc, bmd), y2 = csub(apc, bpd), y3 = csub(amc, bmd);
                    const int e = pp * s;                             // W_n^{p j} = wK[p j s]
                    yout[0] = y0;
                    const cf w1 = wK[e], w2 = wK[2 * e], w3 = wK[3 * e];
                    yout[out_step] = INV ? cmulj(y1, w1) : cmul(y1, w1);
                    yout[2 * out_step] = INV ? cmulj(y2, w2) : cmul(y2, w2);
                    yout[3 * out_step] = INV ? cmulj(y3, w3) : cmul(y3, w3);
                } else {
                    const cf u = xin[0], v = xin[in_step];
                    const cf w = wK[pp * s], d = csub(u, v);
                    yout[0] = cadd(u, v);
                    yout[out_step] = INV ? cmulj(d, w) : cmul(d, w);
                }
            }
        } else {
            // an odd prime radix (a prime K is ONE such pass, i.e. the direct DFT): one work item per (output pair j / r - j, butterfly, column),
            // not per butterfly -- a prime K has only M butterflies -- in the paired form of DftPair: r / 2 steps of two samples and one root
            const int JH = r / 2 + 1, HP = (r - 1) / 2, nbf = m * s;
            DivStep bx(threadIdx.x, GT, M);
            for (int idx = threadIdx.x; idx < total * JH; idx += GT, bx.next()) {
                const int jb = bx.q, col = bx.r;
                const int j = (nbf == 1) ? jb : jb / nbf, bf = jb - j * nbf;
                const int pp = (s == 1) ? bf : bf / s, q = bf - pp * s;
                const cf* xin = x + (q + s * pp) * M + col;
                cf* yout = y + (q + s * r * pp) * M + col;
                DftPair acc;
                acc.start(xin[0]);
                const int de = j * rstep;                             // W_r^{j k} = wK[j k K / r mod K]
                int e = de;
                for (int k = 1; k <= HP; ++k) {
                    const cf vp = xin[k * in_step], vq = xin[(r - k) * in_step];
                    acc.add(cadd(vp, vq), csub(vp, vq), wK[e]);
                    e += de;
                    if (e >= K) e -= K;
                }
                const int j2 = (j == 0) ? 0 : r - j;
                const cf ya = INV ? acc.with_conj() : acc.with_root(), yb = INV ? acc.with_root() : acc.with_conj();
                const cf wa = wK[(int)(((long long)pp * j * s) % K)];
                yout[j * out_step] = INV ? cmulj(ya, wa) : cmul(ya, wa);
                if (j2 != j) {
                    const cf wb = wK[(int)(((long long)pp * j2 * s) % K)];
                    yout[j2 * out_step] = INV ? cmulj(yb, wb) : cmul(yb, wb);
                }
            }
        }
        __syncthreads();
        cf* t = x; x = y; y = t;
        n = m;
        s *= r;
    }
    return x;
}

// estimate_preamble_channel :118-145 -- K-point FFT of both preamble halves, times 0.5 / FFT(known half), summed.
// a, b: LDS scratch of 2K each; dst: K bins (LDS or global).  Caller syncs afterwards.
__device__ __forceinline__ void estimate_preamble_bins(const EstPlan& e, const cf* __restrict__ rx, cf* a, cf* b, cf* dst)
{
    const int K = e.K;
    for (int i = threadIdx.x; i < 2 * K; i += GT) {
        const int h = i >= K, q = i - h * K;
        a[q * 2 + h] = rx[i];                        // [q][half]: both halves go through one column FFT
    }
    __syncthreads();
    DevicePlan p{};
    p.M = 2; p.K = K; p.log2K = e.log2K; p.wK = e.wK;
    const cf* E = col_fft<false>(a, b, p);
    for (int j = threadIdx.x; j < K; j += GT) dst[j] = cfma(E[2 * j], e.inv0[j], cmul(E[2 * j + 1], e.inv1[j]));
}

__device__ __forceinline__ cf decide(cf x, const IcParams& ic)
{
    int idx;
    if (ic.decision == 1) {
        idx = 2 * (x.y > 0.f) + (x.x > 0.f);
    } else if (ic.decision == 2) {
        idx = (x.x > 0.f);
    } else {
        idx = 0;
        float best = INFINITY;
        for (int i = 0; i < ic.npoints; ++i) {
            const cf pt = ic.points[i];
            const float dr = x.x - pt.x, di = x.y - pt.y, d = dr * dr + di * di;
            if (d < best) { best = d; idx = i; }
        }
    }
    return ic.points[idx];
}

// calculate_phase_offset (adv:78-91): exp(j phi), phi = mean over the active symbols of arg(decision) - arg(symbol); red: GT floats of LDS
__device__ __forceinline__ cf phase_rotation(const cf* D, const IcParams& ic, float* red, int M)
{
    float acc = 0.f;
    for (int idx = threadIdx.x; idx < ic.n_active * M; idx += GT) {
        const int a = idx / M, m = idx - a * M;
        const cf v = D[ic.smap[a] * M + m];
        const cf d = decide(v, ic);
        acc += atan2f(d.y, d.x) - atan2f(v.y, v.x);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = GT / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float phi = red[0] / (float)(ic.n_active * M);
    float sn, cs;
    sincosf(phi, &sn, &cs);
    return make_float2(cs, sn);
}

// out[k][m] = fd[k][m] - ic[m] * sum_p (td[k-1][p] + td[k+1][p]) W_M^{p m}     (receiver_kernel_cc.cc:274-299)
template <class Mx>
__device__ __forceinline__ void cancel_rows(const Mx& mx, cf* dst, const cf* td, const cf* fd, const DevicePlan& p)
{
    const int M = p.M, K = p.K;
    paired_dft<false>(mx, K, M, p.wM,
                      [&](int k, int q) { return cadd(td[(k == 0 ? K - 1 : k - 1) * M + q], td[(k == K - 1 ? 0 : k + 1) * M + q]); },
                      [&](int k, int m, const DftPair& acc) {
                          const int m2 = (m == 0) ? 0 : M - m;
                          dst[k * M + m] = csub(fd[k * M + m], cmul(p.ictaps[m], acc.with_root()));
                          if (m2 != m) dst[k * M + m2] = csub(fd[k * M + m2], cmul(p.ictaps[m2], acc.with_conj()));
                      });
}

// resource demapper in the store stage: active subcarriers only, mapper order (resource_mapper_kernel_cc.cc:91-106,136-163)
__device__ __forceinline__ void emit_demapped(cf* o, const cf* tile, const RxIo& io, int K, int M)
{
    for (int idx = threadIdx.x; idx < K * M; idx += GT) {
        const int k = idx / M, m = idx - k * M;
        const int a = io.rank[k];
        if (a < 0) continue;
        const int dst = io.per_timeslot ? (m * io.A + a) : (a * M + m);
        if (dst < io.nout) o[dst] = tile[idx];
    }
}

// Blocks too large for the tiles to sit in LDS (16 N bytes above the CU's 160 KiB, i.e. N > ~10 000) run the SAME kernels with the
// tiles in a global scratch buffer, one slice per workgroup (GLOBAL = true; only the reduction scratch and the root tables stay in
// LDS).  The tiles are private to the workgroup, so the workgroup barrier orders them exactly as it orders LDS (all waves of a
// workgroup share the CU's vector L1).  Slow, but every shape the reference accepts works.  TileArgs: slice base, slice length,
// first GFDM block of this launch (a long batch goes down in chunks that share one scratch buffer).
struct TileArgs {
    cf* gtiles;
    int64_t tile_elems;
    int64_t blk0;
    int xs_off, xs_rtc;            // matrix-core transforms: byte offset of the operand scratch in the dynamic LDS region, row groups it holds (0: vector ALU)
    int xs_alias;                  // ... or no scratch region: the operands go to whichever tile is free (MxDft::at); the LDS tiles are then tile_stride apart
    int tile_stride;               // elements between the tiles in LDS (>= N: a tile must also hold the operands of a whole block)
};

template <bool MX>
__device__ __forceinline__ auto mx_of(const DevicePlan& p, const TileArgs& ta, unsigned char* smem)
{
    if constexpr (MX) {
        MxDft m{ p.dftA, reinterpret_cast<float*>(smem + ta.xs_off), ta.xs_rtc, p.dft_mt, p.dft_ks, ta.xs_alias, {}, -1 };
        m.preload();
        return m;
    }
    else return NoMx{};
}

template <bool GLOBAL, bool MX>
__global__ __launch_bounds__(GT, MX ? 4 : 8) void k_generic_modulate(DevicePlan pg, TxParams tx, int tab_off, TileArgs ta, cf* __restrict__ out, const cf* __restrict__ in)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevicePlan p = stage_tables(pg, smem, tab_off);
    const auto mx = mx_of<MX>(p, ta, smem);
    cf* t0;
    if constexpr (GLOBAL) t0 = ta.gtiles + (int64_t)blockIdx.x * ta.tile_elems; else t0 = reinterpret_cast<cf*>(smem);
    const int M = p.M, K = p.K, L = p.L, N = p.N;
    cf* t1 = t0 + (GLOBAL ? N : ta.tile_stride);
    const int64_t blk = ta.blk0 + blockIdx.x;
    const cf* x = in + blk * (tx.mapped ? tx.nin : N);
    cf* o = out + blk * N;
    GFDM_GSTAMP(0);

    if (tx.mapped) {                                               // resource mapper fused into the load (gfdm_tx.h)
        for (int idx = threadIdx.x; idx < N; idx += GT) t1[idx] = tx_symbol(tx, x, M, idx / M, idx % M);
    } else {
        stream_in(t1, x, N);
    }
    __syncthreads();
    GFDM_GSTAMP(1);
    // (matrix-core form without a scratch of its own: the operands go to the free tile t0, the spectra replace the samples in t1)
    cf* D = mx.in_place() ? t1 : t0;
    cf* Y = mx.in_place() ? t0 : t1;
    row_dft<false>(mx.at(t0, K), D, t1, K, M, M, 1, p.wM, 1.f);       // D_k = FFT_M(d_k)                 :109-110
    __syncthreads();
    GFDM_GSTAMP(2);
    // gather form of the filter + overlap-add scatter (:116-132):
    //   Y[j][m] = sum_i D[(j - i + L/2) mod K][m] * taps[((i + L/2) % L) M + m],  m < part_len
    if (L <= TAPS_IN_REGS) {
        for_columns(M, [&](int m, int j0, int js) {
            cf tp[TAPS_IN_REGS];
#pragma unroll
            for (int i = 0; i < TAPS_IN_REGS; ++i) tp[i] = (i < L && m < p.part_len) ? p.taps[((i + L / 2) % L) * M + m] : make_float2(0.f, 0.f);
            int k0 = (j0 + L / 2) % K;
            const int ks = js % K;
            for (int j = j0; j < K; j += js) {
                cf acc = make_float2(0.f, 0.f);
                int k = k0;
#pragma unroll
                for (int i = 0; i < TAPS_IN_REGS; ++i) {
                    if (i < L) {
                        acc = cfma(D[k * M + m], tp[i], acc);
                        if (--k < 0) k = K - 1;
                    }
                }
                Y[j * M + m] = acc;
                k0 += ks;
                if (k0 >= K) k0 -= K;
            }
        });
    } else {
        DivStep jx(threadIdx.x, GT, M);
        for (int idx = threadIdx.x; idx < N; idx += GT, jx.next()) {
            const int j = jx.q, m = jx.r;
            cf acc = make_float2(0.f, 0.f);
            if (m < p.part_len) {
                for (int i = 0; i < L; ++i) {
                    const int k = ((j - i + L / 2) % K + K) % K;
                    acc = cfma(D[k * M + m], p.taps[((i + L / 2) % L) * M + m], acc);
                }
            }
            Y[idx] = acc;
        }
    }
    __syncthreads();
    GFDM_GSTAMP(3);
    cf* z = col_fft<true>(Y, D, p);                                // K-point inverse over j
    GFDM_GSTAMP(4);
    cf* u = (z == Y) ? D : Y;
    DivStep qx(threadIdx.x, GT, M);
    for (int idx0 = threadIdx.x; idx0 < N; idx0 += 8 * GT) {       // twiddle conj(W_N^{q m}): eight roots (global memory) in flight per thread
        cf w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (idx0 + j * GT < N) w[j] = p.wN[qx.q * qx.r];
            qx.next();
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = idx0 + j * GT;
            if (idx < N) u[idx] = cmulj(z[idx], w[j]);
        }
    }
    __syncthreads();
    GFDM_GSTAMP(5);
    // x[K p + q] = (1/N) sum_m u[q][m] conj(W_M^{p m});  q fastest so the global store is coalesced   :137-140
    const float scale = 1.f / (float)N;
    paired_dft<true>(mx.at(z, K), K, M, p.wM, [&](int q, int m) { return u[q * M + m]; },
                     [&](int q, int pp, const DftPair& acc) {                    // time slots pp and M - pp from one pass
                         const int pp2 = (pp == 0) ? 0 : M - pp;
                         const cf ya = acc.with_conj(), yb = acc.with_root();    // inverse transform: conj(W_M^{p m}) for pp, the root itself for M - pp
                         const cf y = make_float2(ya.x * scale, ya.y * scale);
                         if (tx.framed) tx_store_sample(tx, blk, N, K * pp + q, y);      // cyclic prefix / suffix + ramp, every port
                         else o[K * pp + q] = y;
                         if (pp2 != pp) {
                             const cf y2 = make_float2(yb.x * scale, yb.y * scale);
                             if (tx.framed) tx_store_sample(tx, blk, N, K * pp2 + q, y2);
                             else o[K * pp2 + q] = y2;
                         }
                     });
    if (tx.framed) tx_store_preamble(tx, blk, threadIdx.x, GT);
    GFDM_GSTAMP(6);
}

// transmitter_kernel::add_frame on an already modulated block
__global__ __launch_bounds__(GT) void k_add_frame(DevicePlan p, TxParams tx, const cf* __restrict__ in)
{
    const cf* x = in + (int64_t)blockIdx.x * p.N;
    for (int idx = threadIdx.x; idx < p.N; idx += GT) tx_store_sample(tx, blockIdx.x, p.N, idx, x[idx]);
    tx_store_preamble(tx, blockIdx.x, threadIdx.x, GT);
}

template <bool GLOBAL, bool MX>
__global__ __launch_bounds__(GT, MX ? 4 : 8) void k_generic_receive(DevicePlan pg, IcParams ic, EstPlan est, int eq_source, int ntiles, int mode, int s_in_global,
                                                        int tab_off, TileArgs ta, cf* __restrict__ out, const cf* __restrict__ in,
                                                        const cf* __restrict__ f_eq)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevicePlan p = stage_tables(pg, smem, tab_off);
    const auto mx = mx_of<MX>(p, ta, smem);
    float* red = reinterpret_cast<float*>(smem);
    cf* t0;
    if constexpr (GLOBAL) t0 = ta.gtiles + (int64_t)blockIdx.x * ta.tile_elems; else t0 = reinterpret_cast<cf*>(smem + RED_BYTES);
    const int M = p.M, K = p.K, L = p.L, N = p.N;
    const int TS = GLOBAL ? N : ta.tile_stride;
    cf* t1 = t0 + TS;
    cf* t2 = t1 + TS;                                              // only valid when 3 tiles were requested
    const int64_t blk = ta.blk0 + blockIdx.x;
    const cf* x = in + blk * (int64_t)(ic.io.in_stride ? ic.io.in_stride : N) + ic.io.in_offset;   // frame -> block
    const bool demap = ic.io.demap && mode != RX_FD;
    cf* o = out + blk * (demap ? ic.io.nout : N);
    const cf* eq = (eq_source == EQ_VECTOR) ? f_eq + blk * N : nullptr;
    cf* filt = t0 + (size_t)ntiles * TS + K;                      // EQ_PREAMBLE: smoothed channel estimate, behind the tiles
    if (eq_source == EQ_PREAMBLE) {                                // channel estimator in front, the tiles are its scratch
        cf* bins = t0 + (size_t)ntiles * TS;
        estimate_preamble_bins(est, f_eq + blk * (est.pre_stride ? est.pre_stride : 2 * K), t0, t1, bins);
        __syncthreads();
        for (int i = threadIdx.x; i < est.n_est; i += GT) filt[i] = est_filter_bin(bins, i, est);
        __syncthreads();
    }

    GFDM_GSTAMP(0);
    stream_in(t1, x, N);
    __syncthreads();
    GFDM_GSTAMP(1);
    // A[q][m] = W_N^{q m} * sum_p x[K p + q] W_M^{p m}
    // (matrix-core form without a scratch of its own: the operands go to the free tile t0, the result replaces the samples in t1)
    cf* A0 = mx.in_place() ? t1 : t0;
    cf* A1 = mx.in_place() ? t0 : t1;
    paired_dft<false>(mx.at(t0, K), K, M, p.wM, [&](int q, int pp) { return t1[K * pp + q]; },       // outputs m and M - m from one pass (DftPair)
                      [&](int q, int m, const DftPair& acc) {
                          const int m2 = (m == 0) ? 0 : M - m;
                          A0[q * M + m] = cmul(acc.with_root(), p.wN[q * m]);
                          if (m2 != m) A0[q * M + m2] = cmul(acc.with_conj(), p.wN[q * m2]);
                      });
    __syncthreads();
    GFDM_GSTAMP(2);
    cf* X = col_fft<false>(A0, A1, p);                             // X[j][m] = FFT_N(x)[M j + m]       :304-305
    cf* U = (X == A0) ? A1 : A0;
    GFDM_GSTAMP(3);
    if (eq) {                                                      // one-tap equaliser                 :315-316
        stream_in(X, eq, N, [&](cf e, int idx) { return cdiv(X[idx], e); });
        __syncthreads();
    } else if (eq_source == EQ_PREAMBLE) {                         // same, the estimate interpolated on the fly
        for (int idx = threadIdx.x; idx < N; idx += GT) X[idx] = cdiv(X[idx], est_frame_bin<0>(filt, idx, est));
        __syncthreads();
    }
    // S[k][m] = sum_i taps[((i + L/2) % L) M + m] * X[((k + i + K - L/2) % K) M + m]                    :165-192
    cf* Sdst = (mode == RX_FD) ? o : U;
    if (L <= TAPS_IN_REGS) {
        for_columns(M, [&](int m, int k0, int ks) {
            cf tp[TAPS_IN_REGS];
#pragma unroll
            for (int i = 0; i < TAPS_IN_REGS; ++i) tp[i] = (i < L) ? p.taps[((i + L / 2) % L) * M + m] : make_float2(0.f, 0.f);
            int r0 = ((k0 - L / 2) % K + K) % K;                    // row (k + i - L/2) mod K, tap part (i + L/2) mod L
            const int rs = ks % K;
            for (int k = k0; k < K; k += ks) {
                cf acc = make_float2(0.f, 0.f);
                int row = r0;
#pragma unroll
                for (int i = 0; i < TAPS_IN_REGS; ++i) {
                    if (i < L) {
                        acc = cfma(tp[i], X[row * M + m], acc);
                        if (++row == K) row = 0;
                    }
                }
                Sdst[k * M + m] = acc;
                r0 += rs;
                if (r0 >= K) r0 -= K;
            }
        });
    } else {
        DivStep fx(threadIdx.x, GT, M);
        for (int idx = threadIdx.x; idx < N; idx += GT, fx.next()) {
            const int k = fx.q, m = fx.r;
            cf acc = make_float2(0.f, 0.f);
            int row = k - L / 2, part = L / 2;
            if (row < 0) row += K;
            for (int i = 0; i < L; ++i) {
                acc = cfma(p.taps[part * M + m], X[row * M + m], acc);
                if (++row == K) row = 0;
                if (++part == L) part = 0;
            }
            Sdst[idx] = acc;
        }
    }
    if (mode == RX_FD) return;
    __syncthreads();
    GFDM_GSTAMP(4);
    const float invM = 1.f / (float)M;
    if (mode == RX_DEMOD || ic.ic_iter <= 0) {
        if (!demap) {
            row_dft<true>(mx.at(X, K), o, U, K, M, M, 1, p.wM, invM);      // d = IFFT_M(S_k) / M                :211-225
            GFDM_GSTAMP(5);
        } else {
            cf* d = mx.in_place() ? U : X;                              // (X holds the operands then)
            row_dft<true>(mx.at(X, K), d, U, K, M, M, 1, p.wM, invM);
            __syncthreads();
            emit_demapped(o, d, ic.io, K, M);
        }
        return;
    }
    // One cancellation round of the reference is  d_new = IDFT_M(S - ic (.) DFT_M(nb)) / M  with nb = dec_{k-1} + dec_{k+1}
    // (receiver_kernel_cc.cc:274-299 + :211-225).  Both transforms are linear, so  d_new = d0 - g (*) nb  with d0 = IDFT_M(S) / M
    // and the M-tap circular kernel g = IDFT_M(ic) / M (p.icg): one table-driven pass per round instead of two, S is not needed
    // again (a rotation of S by the phase compensation is the same rotation of d0).
    cf* D = mx.in_place() ? t2 : X;                                // (in place: X takes the operands of every transform from here on)
    row_dft<true>(mx.at(X, K), D, U, K, M, M, 1, p.wM, invM);
    __syncthreads();
    if constexpr (MX) {
        // With the transforms on the matrix cores the rounds keep the reference's own form, S' = S - ic (.) DFT_M(nb), d = IDFT_M(S') / M: two constant-
        // matrix products per round instead of the O(M^2) convolution on the vector ALU.  S stays in its tile, S' goes to the third one (or the output block).
        cf* S = U;
        cf* V = mx.in_place() ? D : s_in_global ? o : t2;             // (in place: S' replaces the decisions once all of them are operands)
        const auto mxs = mx.at(X, K);
        for (int j = 0; j < ic.ic_iter; ++j) {
            if (ic.do_phase_compensation > 0 && j == 0) {
                const cf rot = phase_rotation(D, ic, red, M);
                for (int idx = threadIdx.x; idx < N; idx += GT) S[idx] = cmul(S[idx], rot);     // adv:63-70: the rotation of S persists
                __syncthreads();
            }
            {
                DivStep dx(threadIdx.x, GT, M);
                for (int idx = threadIdx.x; idx < N; idx += GT, dx.next())
                    D[idx] = ic.active[dx.q] ? decide(D[idx], ic) : make_float2(0.f, 0.f);
            }
            __syncthreads();
            cancel_rows(mxs, V, D, S, p);
            __syncthreads();
            const bool last = (j == ic.ic_iter - 1);
            row_dft<true>(mxs, (last && !demap) ? o : D, V, K, M, M, 1, p.wM, invM);
            __syncthreads();
            if (last && demap) emit_demapped(o, D, ic.io, K, M);
        }
        return;
    }
    cf* D0 = U;
    cf* V = t2;
    if (s_in_global) {                                            // third tile does not fit: d0 lives in the output block
        D0 = o;
        V = U;
    }
    for (int idx = threadIdx.x; idx < N; idx += GT) D0[idx] = D[idx];
    __syncthreads();
    for (int j = 0; j < ic.ic_iter; ++j) {                        // perform_ic_iterations            adv:56-76
        if (ic.do_phase_compensation > 0 && j == 0) {
            const cf rot = phase_rotation(D, ic, red, M);
            for (int idx = threadIdx.x; idx < N; idx += GT) D0[idx] = cmul(D0[idx], rot);   // rotating S rotates d0; persists  adv:63-70
            __syncthreads();
        }
        {
            DivStep dx(threadIdx.x, GT, M);
            for (int idx = threadIdx.x; idx < N; idx += GT, dx.next())   // map_symbols_to_constellation_points  adv:109-123
                D[idx] = ic.active[dx.q] ? decide(D[idx], ic) : make_float2(0.f, 0.f);
        }
        __syncthreads();
        {                                                         // nb = dec_{k-1} + dec_{k+1} (wraps mod K)  rx:279-284
            DivStep nx(threadIdx.x, GT, M);
            for (int idx = threadIdx.x; idx < N; idx += GT, nx.next()) {
                const int k = nx.q, pp = nx.r;
                V[idx] = cadd(D[(k == 0 ? K - 1 : k - 1) * M + pp], D[(k == K - 1 ? 0 : k + 1) * M + pp]);
            }
        }
        __syncthreads();
        const bool last = (j == ic.ic_iter - 1);
        cf* dst = (last && !demap) ? o : D;                        // the decisions are spent: the new symbols replace them
        DivStep cx(threadIdx.x, GT, M);
        for (int idx = threadIdx.x; idx < N; idx += GT, cx.next()) {
            const int pp = cx.r;
            const cf* nb = V + cx.q * M;
            cf acc = D0[idx];
            if (p.ic_real_sym) {                                   // g real and even: g_r (nb[p - r] + nb[p + r])
                const float g0 = p.icg[0].x;
                acc = make_float2(acc.x - g0 * nb[pp].x, acc.y - g0 * nb[pp].y);
                int lo = pp, hi = pp;
                const int H = (M - 1) / 2;
                for (int r = 1; r <= H; ++r) {
                    if (--lo < 0) lo = M - 1;
                    if (++hi == M) hi = 0;
                    const float g = p.icg[r].x;
                    acc = make_float2(acc.x - g * (nb[lo].x + nb[hi].x), acc.y - g * (nb[lo].y + nb[hi].y));
                }
                if ((M & 1) == 0) {                                // the middle tap of an even length
                    if (--lo < 0) lo = M - 1;
                    const float g = p.icg[M / 2].x;
                    acc = make_float2(acc.x - g * nb[lo].x, acc.y - g * nb[lo].y);
                }
            } else {
                int src = pp;                                     // (pp - r) mod M
                for (int r = 0; r < M; ++r) {
                    const cf g = p.icg[r], x = nb[src];
                    acc = make_float2(acc.x - g.x * x.x + g.y * x.y, acc.y - g.x * x.y - g.y * x.x);
                    if (--src < 0) src = M - 1;
                }
            }
            dst[idx] = acc;
        }
        __syncthreads();
        if (last && demap) emit_demapped(o, D, ic.io, K, M);
    }
}

template <bool GLOBAL, bool MX>
__global__ __launch_bounds__(GT, MX ? 4 : 8) void k_generic_to_td(DevicePlan pg, int tab_off, TileArgs ta, cf* __restrict__ out, const cf* __restrict__ in)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevicePlan p = stage_tables(pg, smem, tab_off);
    const auto mx = mx_of<MX>(p, ta, smem);
    cf* t0;
    if constexpr (GLOBAL) t0 = ta.gtiles + (int64_t)blockIdx.x * ta.tile_elems; else t0 = reinterpret_cast<cf*>(smem);
    const int64_t blk = ta.blk0 + blockIdx.x;
    const cf* x = in + blk * p.N;
    stream_in(t0, x, p.N);
    __syncthreads();
    row_dft<true>(mx, out + blk * p.N, t0, p.K, p.M, p.M, 1, p.wM, 1.f / (float)p.M);
}

template <bool GLOBAL, bool MX>
__global__ __launch_bounds__(GT, MX ? 4 : 8) void k_generic_cancel(DevicePlan pg, int tab_off, TileArgs ta, cf* __restrict__ out, const cf* __restrict__ td,
                                                       const cf* __restrict__ fd)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevicePlan p = stage_tables(pg, smem, tab_off);
    const auto mx = mx_of<MX>(p, ta, smem);
    cf* t0;
    if constexpr (GLOBAL) t0 = ta.gtiles + (int64_t)blockIdx.x * ta.tile_elems; else t0 = reinterpret_cast<cf*>(smem);
    const int64_t blk = ta.blk0 + blockIdx.x;
    const cf* x = td + blk * p.N;
    stream_in(t0, x, p.N);
    __syncthreads();
    cancel_rows(mx, out + blk * p.N, t0, fd + blk * p.N, p);
}


// ---------------------------------------------------------------------------------------------------------------------
// Preamble channel estimator, one workgroup per received preamble (lib/preamble_channel_estimator_cc.cc):
//   estimate_preamble_channel :118-145   K-point FFT of both preamble halves, times 0.5 / FFT(known half), summed
//   filter_preamble_estimate  :147-187   active bins in fftshift order, DC interpolated when dc-free, 9-tap Gaussian
//   interpolate_frame         :238-273   linear interpolation K bins -> M*K bins, written as a gather over the output
//   prepare_for_zf            :275-281   conj(1 / x)
// The whole chain stays in LDS: HBM sees the 2K input samples and the M*K output bins once.


__global__ __launch_bounds__(GT) void k_estimate(EstPlan e, int in_stage, int out_stage, cf* __restrict__ out,
                                                 const cf* __restrict__ in)
{
    extern __shared__ cf lds[];
    const int K = e.K, n_est = e.n_est, M = e.M;
    cf* a = lds;
    cf* b = a + 2 * K;
    cf* est = b + 2 * K;
    cf* filt = est + K;
    const int64_t f = blockIdx.x;
    if (in_stage == EST_RX_PREAMBLE) {
        estimate_preamble_bins(e, in + f * 2 * K, a, b, out_stage == EST_PREAMBLE_CHANNEL ? out + f * K : est);
        if (out_stage == EST_PREAMBLE_CHANNEL) return;
        __syncthreads();
    } else if (in_stage == EST_PREAMBLE_CHANNEL) {
        for (int j = threadIdx.x; j < K; j += GT) est[j] = in[f * K + j];
        __syncthreads();
    }
    if (in_stage <= EST_PREAMBLE_CHANNEL) {
        for (int i = threadIdx.x; i < n_est; i += GT) {
            const cf acc = est_filter_bin(est, i, e);
            if (out_stage == EST_FILTERED) out[f * n_est + i] = acc; else filt[i] = acc;
        }
        if (out_stage == EST_FILTERED) return;
        __syncthreads();
    } else {
        for (int i = threadIdx.x; i < n_est; i += GT) filt[i] = in[f * n_est + i];
        __syncthreads();
    }
    const int N = M * K;
    cf* o = out + f * N;
    for (int n = threadIdx.x; n < N; n += GT) {
        o[n] = est_frame_bin<0>(filt, n, e);
    }
}

// estimate_snr :189-227 -- 2K-point FFT of the two-fold repeated preamble; even bins = symbol + noise, odd bins = noise.
__global__ __launch_bounds__(GT) void k_estimate_snr(EstPlan e, float* __restrict__ snr, float* __restrict__ cnrs,
                                                     const cf* __restrict__ in)
{
    extern __shared__ cf lds[];
    __shared__ float red[2][GT];
    const int K = e.K, A = e.A, half = e.A >> 1;
    cf* a = lds;
    cf* b = a + 2 * K;
    const int64_t f = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * K; i += GT) a[i] = in[f * 2 * K + i];
    __syncthreads();
    DevicePlan p{};
    p.M = 1; p.K = 2 * K; p.log2K = e.log2K2; p.wK = e.w2K;
    const cf* S = col_fft<false>(a, b, p);
    float* se_all = reinterpret_cast<float*>(S == a ? b : a);      // the idle tile keeps the per-bin symbol energies
    float sym = 0.f, noise = 0.f;
    for (int i = threadIdx.x; i < A; i += GT) {
        const int bin = i < half ? i + (e.dc_free ? 1 : 0) : (i - half) + (K - A) / 2 + K / 2;
        const cf s0 = S[2 * bin], s1 = S[2 * bin + 1];
        const float se = s0.x * s0.x + s0.y * s0.y;
        se_all[i] = se;
        sym += se;
        noise += s1.x * s1.x + s1.y * s1.y;
    }
    red[0][threadIdx.x] = sym;
    red[1][threadIdx.x] = noise;
    __syncthreads();
    for (int w = GT / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    const float snr_lin = (red[0][0] - red[1][0]) / red[1][0];
    const float scale = snr_lin / (red[0][0] / (float)A);
    if (threadIdx.x == 0) snr[f] = snr_lin;
    for (int i = threadIdx.x; i < A; i += GT) cnrs[f * A + i] = se_all[i] * scale;
}

__global__ __launch_bounds__(GT) void k_prepare_for_zf(cf* __restrict__ out, const cf* __restrict__ in, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * GT + threadIdx.x;
    if (i < n) {
        cf v = cdiv(make_float2(1.f, 0.f), in[i]);
        v.y = -v.y;
        out[i] = v;
    }
}

constexpr size_t LDS_MAX = 160 * 1024;

template <typename KernelT>
hipError_t allow_lds(KernelT kernel, size_t bytes)
{
    if (bytes <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

size_t generic_lds_bytes(int N, int ntiles) { return (size_t)ntiles * (size_t)N * sizeof(cf) + RED_BYTES; }
// ... plus the W_M and W_K tables behind everything else
static size_t table_bytes(const DevicePlan& p) { return (((size_t)(2 * p.M + p.K) * sizeof(cf)) + 15) & ~(size_t)15; }

// Blocks whose tiles fit the LDS of a CU use it; larger ones take the global-scratch form of the same kernels (GLOBAL), bounded only by
// the root tables that always sit in LDS.  one_tile is kept for the callers' sake: both forms exist for every kernel.
bool generic_supports(int M, int K, bool one_tile)
{
    (void)one_tile;
    return M >= 1 && K >= 1 && (int64_t)M * K <= ((int64_t)1 << 24) && RED_BYTES + (size_t)(2 * M + K + 2) * sizeof(cf) + 64 <= LDS_MAX;
}

namespace {

// Where the matrix-core form of the timeslot transforms (mx_dft) pays -- measured: where the 16 x 16 x 4 operand tiles are well filled (M = 127:
// 64 output pairs x 64 sample pairs = 4 x 16 full tiles; M = 33: 17 x 17 in 32 x 20) and the block has a unit of work for each of its four wavefronts
// (K=61 M=33: 147 -> 180 us per 4096 blocks, vector ALU kept).  A handle created in mode 2 (gfdm_hip_set_dft_matrix_cores) takes it wherever it fits.
bool mx_pays(const DevicePlan& p)
{
    if (p.dft_always) return true;
    const int RT = (p.K + 15) / 16, H = p.M / 2 + 1, KD = (p.M - 1) / 2 + 1 + ((p.M & 1) == 0 ? 1 : 0);
    const double fill = (double)H * KD * p.K / ((double)(16 * p.dft_mt) * (4 * p.dft_ks) * (16 * RT));
    return fill >= 0.7 && p.dft_mt * RT >= GT / 64;
}
size_t cu_workgroups(size_t lds_bytes) { return std::min<size_t>(8, LDS_MAX / lds_bytes); }

// The operands of a 16-row group take 1024 KS bytes of LDS (four planes of [4 KS][16] floats).  The modulator and the receiver keep them in whichever
// of their tiles is free at that point (MxDft::at): no LDS beyond the tiles, which are spaced so that one holds the operands of a whole block
// (tile_stride >= N elements; K = 16, M = 127: 16 384 bytes of operands, 16 256 of samples).  These kernels are latency-bound -- a block alone on a CU
// takes 18 us, four of them 24 us each (profiles/r03/stamps_generic_16_127.txt) -- so workgroups per CU decide: the form is not taken where the
// wider spacing would cost the CU a workgroup and leave it fewer than two.  ntiles tiles + `other` bytes of LDS.
struct MxAlias {
    bool on;
    int tile_stride;
};
MxAlias mx_alias(const DevicePlan& p, int ntiles, size_t other)
{
    MxAlias r{ false, p.N };
    if (!p.dftA || !mx_pays(p)) return r;
    const size_t xbytes = (size_t)((p.K + 15) / 16) * 1024 * (size_t)p.dft_ks;
    const size_t ts = (std::max<size_t>((size_t)p.N, xbytes / sizeof(cf)) + 1) & ~(size_t)1;        // (even: the operand reads are 16 bytes wide)
    const size_t lds = (size_t)ntiles * ts * sizeof(cf) + other, plain = (size_t)ntiles * (size_t)p.N * sizeof(cf) + other;
    if (lds > LDS_MAX) return r;
    if (!p.dft_always && cu_workgroups(lds) < 2 && cu_workgroups(lds) < cu_workgroups(plain)) return r;
    r.on = true;
    r.tile_stride = (int)ts;
    return r;
}

// The stand-alone transform / cancellation kernels (one tile) and the global-scratch kernels keep a scratch region of their own behind everything
// else in LDS, for as many row groups as do not lower the number of workgroups a CU holds; rtc = 0: the vector-ALU instantiation.
struct MxLds {
    int off, rtc;
    size_t total;
};
MxLds mx_lds(const DevicePlan& p, size_t used)
{
    MxLds r{ 0, 0, used };
    if (!p.dftA || !mx_pays(p)) return r;
    const size_t per_rt = (size_t)1024 * (size_t)p.dft_ks, off = (used + 255) & ~(size_t)255;
    if (off + per_rt > LDS_MAX) return r;
    const int RT = (p.K + 15) / 16;
    if (!p.dft_always && cu_workgroups(off + per_rt) < 2 && cu_workgroups(off + per_rt) < cu_workgroups(used)) return r;
    int rtc = 1;
    while (rtc < RT && cu_workgroups(off + (size_t)(rtc + 1) * per_rt) == cu_workgroups(off + per_rt)) ++rtc;
    r.off = (int)off;
    r.rtc = rtc;
    r.total = off + (size_t)rtc * per_rt;
    return r;
}

// scratch slices for a launch of the GLOBAL kernels: at most ~256 MiB, so a long batch goes down in chunks; allocated and freed in
// stream order (no handle state: calls on different streams do not share it)
struct Scratch {
    cf* base = nullptr;
    int64_t chunk = 0;
    hipStream_t stream = nullptr;
    hipError_t open(int64_t tile_elems, int64_t nblocks, hipStream_t s)
    {
        stream = s;
        const int64_t bytes_per_block = tile_elems * (int64_t)sizeof(cf);
        chunk = std::max<int64_t>(64, ((int64_t)256 << 20) / bytes_per_block);
        if (chunk > nblocks) chunk = nblocks;
        return hipMallocAsync(reinterpret_cast<void**>(&base), (size_t)(chunk * bytes_per_block), s);
    }
    hipError_t close() { return base ? hipFreeAsync(base, stream) : hipSuccess; }
};

}  // namespace

hipError_t launch_generic_modulate(const DevicePlan& p, const TxParams& tx, cf* out, const cf* in, int64_t nblocks, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    if (rader_applies_modulate(p, tx)) return launch_rader_modulate(p, out, in, nblocks, s);
    if (generic_lds_bytes(p.N, 2) + table_bytes(p) <= LDS_MAX) {
        const MxAlias al = mx_alias(p, 2, RED_BYTES + table_bytes(p));
        const size_t tab = generic_lds_bytes(al.tile_stride, 2), lds = tab + table_bytes(p);
        auto kern = al.on ? k_generic_modulate<false, true> : k_generic_modulate<false, false>;
        hipError_t e = allow_lds(kern, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(GT), lds, s, p, tx, (int)tab, TileArgs{ nullptr, 0, 0, 0, 0, al.on ? 1 : 0, al.tile_stride }, out, in);
        return hipGetLastError();
    }
    Scratch sc;
    const int64_t tile_elems = 2 * (int64_t)p.N;
    hipError_t e = sc.open(tile_elems, nblocks, s);
    if (e != hipSuccess) return e;
    const MxLds mx = mx_lds(p, table_bytes(p));
    auto kern = mx.rtc ? k_generic_modulate<true, true> : k_generic_modulate<true, false>;
    if ((e = allow_lds(kern, mx.total)) != hipSuccess) return e;
    for (int64_t b0 = 0; b0 < nblocks && e == hipSuccess; b0 += sc.chunk) {
        const int64_t n = std::min(sc.chunk, nblocks - b0);
        hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(GT), mx.total, s, p, tx, 0, TileArgs{ sc.base, tile_elems, b0, mx.off, mx.rtc, 0, p.N }, out, in);
        e = hipGetLastError();
    }
    const hipError_t e2 = sc.close();
    return e != hipSuccess ? e : e2;
}

hipError_t launch_add_frame(const DevicePlan& p, const TxParams& tx, const cf* in, int64_t nblocks, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_frame, dim3((unsigned)nblocks), dim3(GT), 0, s, p, tx, in);
    return hipGetLastError();
}

hipError_t launch_generic_receive(const DevicePlan& p, const IcParams& ic, const EstPlan* est, int mode, cf* out, const cf* in, const cf* f_eq,
                                  int64_t nblocks, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    if (rader_applies_receive(p, ic, est, mode)) return launch_rader_receive(p, ic, mode, out, in, f_eq, nblocks, s);
    const int eq_source = est ? EQ_PREAMBLE : f_eq ? EQ_VECTOR : EQ_NONE;
    if (est && p.M < 2) return hipErrorInvalidConfiguration;               // the tiles double as the estimator's 2K scratch
    const size_t extra = est ? (size_t)(2 * p.K + 2) * sizeof(cf) : 0;     // K-bin estimate + smoothed estimate
    static const EstPlan kNoEst = {};
    const bool ic_rounds = (mode == RX_IC && ic.ic_iter > 0);
    if (generic_lds_bytes(p.N, 2) + extra + table_bytes(p) <= LDS_MAX) {
        int ntiles = 2, s_in_global = 0;
        if (ic_rounds) {
            if (generic_lds_bytes(p.N, 3) + extra + table_bytes(p) <= LDS_MAX) ntiles = 3; else s_in_global = 1;
        }
        if (!(s_in_global && ic.io.demap)) {                               // (a demapped output block is too small to park S in: GLOBAL below)
            // (the rounds of the matrix-core form keep S, the decisions and the operands in three LDS tiles)
            const MxAlias al = s_in_global ? MxAlias{ false, p.N } : mx_alias(p, ntiles, RED_BYTES + extra + 16 + table_bytes(p));
            const size_t tab = (generic_lds_bytes(al.tile_stride, ntiles) + extra + 15) & ~(size_t)15, lds = tab + table_bytes(p);
            auto kern = al.on ? k_generic_receive<false, true> : k_generic_receive<false, false>;
            hipError_t e = allow_lds(kern, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(GT), lds, s, p, ic, est ? *est : kNoEst, eq_source, ntiles, mode,
                               s_in_global, (int)tab, TileArgs{ nullptr, 0, 0, 0, 0, al.on ? 1 : 0, al.tile_stride }, out, in, f_eq);
            return hipGetLastError();
        }
    }
    Scratch sc;
    const int ntiles = ic_rounds ? 3 : 2;
    const int64_t tile_elems = (int64_t)ntiles * p.N + 2 * p.K + 2;
    hipError_t e = sc.open(tile_elems, nblocks, s);
    if (e != hipSuccess) return e;
    const size_t tab = (RED_BYTES + 15) & ~(size_t)15;
    const MxLds mx = mx_lds(p, tab + table_bytes(p));
    auto kern = mx.rtc ? k_generic_receive<true, true> : k_generic_receive<true, false>;
    if ((e = allow_lds(kern, mx.total)) != hipSuccess) return e;
    for (int64_t b0 = 0; b0 < nblocks && e == hipSuccess; b0 += sc.chunk) {
        const int64_t n = std::min(sc.chunk, nblocks - b0);
        hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(GT), mx.total, s, p, ic, est ? *est : kNoEst, eq_source, ntiles,
                           mode, 0, (int)tab, TileArgs{ sc.base, tile_elems, b0, mx.off, mx.rtc, 0, p.N }, out, in, f_eq);
        e = hipGetLastError();
    }
    const hipError_t e2 = sc.close();
    return e != hipSuccess ? e : e2;
}

hipError_t launch_generic_to_td(const DevicePlan& p, cf* out, const cf* in, int64_t nblocks, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    const size_t tab = generic_lds_bytes(p.N, 1), lds = tab + table_bytes(p);
    if (lds <= LDS_MAX) {
        const MxLds mx = mx_lds(p, lds);
        auto kern = mx.rtc ? k_generic_to_td<false, true> : k_generic_to_td<false, false>;
        hipError_t e = allow_lds(kern, mx.total);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(GT), mx.total, s, p, (int)tab, TileArgs{ nullptr, 0, 0, mx.off, mx.rtc, 0, p.N }, out, in);
        return hipGetLastError();
    }
    Scratch sc;
    hipError_t e = sc.open(p.N, nblocks, s);
    if (e != hipSuccess) return e;
    const MxLds mx = mx_lds(p, table_bytes(p));
    auto kern = mx.rtc ? k_generic_to_td<true, true> : k_generic_to_td<true, false>;
    if ((e = allow_lds(kern, mx.total)) != hipSuccess) return e;
    for (int64_t b0 = 0; b0 < nblocks && e == hipSuccess; b0 += sc.chunk) {
        const int64_t n = std::min(sc.chunk, nblocks - b0);
        hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(GT), mx.total, s, p, 0, TileArgs{ sc.base, (int64_t)p.N, b0, mx.off, mx.rtc, 0, p.N }, out, in);
        e = hipGetLastError();
    }
    const hipError_t e2 = sc.close();
    return e != hipSuccess ? e : e2;
}

hipError_t launch_generic_cancel(const DevicePlan& p, cf* out, const cf* td, const cf* fd, int64_t nblocks, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    const size_t tab = generic_lds_bytes(p.N, 1), lds = tab + table_bytes(p);
    if (lds <= LDS_MAX) {
        const MxLds mx = mx_lds(p, lds);
        auto kern = mx.rtc ? k_generic_cancel<false, true> : k_generic_cancel<false, false>;
        hipError_t e = allow_lds(kern, mx.total);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(GT), mx.total, s, p, (int)tab, TileArgs{ nullptr, 0, 0, mx.off, mx.rtc, 0, p.N }, out, td, fd);
        return hipGetLastError();
    }
    Scratch sc;
    hipError_t e = sc.open(p.N, nblocks, s);
    if (e != hipSuccess) return e;
    const MxLds mx = mx_lds(p, table_bytes(p));
    auto kern = mx.rtc ? k_generic_cancel<true, true> : k_generic_cancel<true, false>;
    if ((e = allow_lds(kern, mx.total)) != hipSuccess) return e;
    for (int64_t b0 = 0; b0 < nblocks && e == hipSuccess; b0 += sc.chunk) {
        const int64_t n = std::min(sc.chunk, nblocks - b0);
        hipLaunchKernelGGL(kern, dim3((unsigned)n), dim3(GT), mx.total, s, p, 0, TileArgs{ sc.base, (int64_t)p.N, b0, mx.off, mx.rtc, 0, p.N }, out, td, fd);
        e = hipGetLastError();
    }
    const hipError_t e2 = sc.close();
    return e != hipSuccess ? e : e2;
}

size_t estimator_lds_bytes(int K) { return (size_t)(6 * K + 1) * sizeof(cf); }   // two [K][2] tiles, K-bin estimate, smoothed estimate

bool estimator_supports(int K) { return estimator_lds_bytes(K) <= LDS_MAX; }

hipError_t launch_estimate(const EstPlan& e, int in_stage, int out_stage, cf* out, const cf* in, int64_t nframes, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    const size_t lds = estimator_lds_bytes(e.K);
    hipError_t err = allow_lds(k_estimate, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_estimate, dim3((unsigned)nframes), dim3(GT), lds, s, e, in_stage, out_stage, out, in);
    return hipGetLastError();
}

hipError_t launch_estimate_snr(const EstPlan& e, float* snr, float* cnrs, const cf* in, int64_t nframes, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
    const size_t lds = (size_t)4 * e.K * sizeof(cf);
    hipError_t err = allow_lds(k_estimate_snr, lds);
    if (err != hipSuccess) return err;
    hipLaunchKernelGGL(k_estimate_snr, dim3((unsigned)nframes), dim3(GT), lds, s, e, snr, cnrs, in);
    return hipGetLastError();
}

hipError_t launch_prepare_for_zf(cf* out, const cf* in, int64_t n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_prepare_for_zf, dim3((unsigned)((n + GT - 1) / GT)), dim3(GT), 0, s, out, in, n);
    return hipGetLastError();
}

}  // namespace gfdm

#ifdef GFDM_STAMPS
extern "C" int gfdm_debug_set_stamp_buffer_generic(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(gfdm::g_gstamp_buf), &p, sizeof(p)); }
#endif
