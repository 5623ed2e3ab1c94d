// Prime timeslot counts above the register codelets (M > 48): the reference's own QA shape M = 127, K = 16
// (python/qa_simple_receiver_cc.py:58-83, python/qa_simple_modulator_cc.py) without O(M^2) transforms.
//
// The generic family computes the M-point transforms as dense products (matrix cores, 2 x 16 x 127^2 complex multiply-adds per block: a FLOP roof of
// 16-31 % of the HBM peak).  Here they are Rader transforms: for a prime P with generator g
//     X[g^-q] = x[0] + sum_p x[g^p] W^(g^(p-q)),      X[0] = sum_n x[n]
// is a cyclic convolution of length n = P - 1 of a[p] = x[g^p] with the constant kernel b[p] = W^(g^-p), done as  IFFT_n(FFT_n(a) . FFT_n(b) / n)
// with FFT_n(b) / n a per-handle table (host, double).  n = A x B with coprime A, B (126 = 9 x 14), so both FFT_n are prime-factor transforms without
// twiddles: input map p = (B p1 + A p2) mod n, output map j = j1 mod A, j2 mod B.  The forward transform runs A-point stage then B-point stage, the
// inverse B-point stage then A-point stage -- the B-point stage of the forward transform, the product with the kernel spectrum and the B-point stage of
// the inverse all work on the SAME B values, i.e. stay in one thread's registers:
//     stage 1   (row, p2):  A samples gathered by the table g^((B p1 + A p2) mod n), Dft<A>                         -> work tile
//     stage 2   (row, j1):  Dft<B>, x FFT_n(b)/n, (+ x[0] on bin 0: adds x[0] to every output), inverse Dft<B>     in place in the work tile
//     stage 3   (row, q2):  inverse Dft<A>, outputs scattered to m = g^-((B q1 + A q2) mod n); X[0] = x[0] + FFT_n(a)[0] from stage 2
// two barriers per transform, every small transform a compile-time codelet of gfdm_dft.h.  The inverse M-point transform is the forward one with
// its output index negated (a second scatter map).
// Around the two row transforms of a block the subcarrier axis (K = 16) is ONE in-register 16-point codelet per column together with the
// equaliser and the filter: load -> Rader rows -> columns -> Rader rows -> store, 8 barriers per block, two tiles of LDS (33.4 KB: four blocks per CU).
// Limited by its vector work.  SQ counters (profiles/r06/pmc_sq_counters_summary.csv, 4096 blocks): 1510-1590 vector instructions per wave for the plain kernels, 4200-4290
// with two cancellation rounds, three of the four possible waves per SIMD resident on average; a wave issues one vector instruction per four clocks (SQ_ACTIVE_INST_VALU x 4 =
// 97 k / 269 k clocks of wave issue per SIMD in a 54 / 157 us launch of 131 k / 377 k clocks), the SIMD retires a plain one in two (MI355X_MICROARCH.md) -- the pipe is a third
// to a half busy, the three resident waves' dependent chains and their 8 barriers per block leave it idle the rest; LDS bank conflicts add a third to the busy LDS cycles.
//
// Serves plain blocks: modulate, fft_[equalize_]filter_downsample, generic_work[_equalize] and the advanced receiver's cancellation rounds (incl. phase
// compensation).  Everything else of this shape -- frames / demapper, the self-estimating receivers, the fused transmitter -- stays on the generic
// kernels (launch_generic_* falls through).
#include "gfdm_plan.h"
#include "gfdm_dft.h"
#include "gfdm_tx.h"

#include <cmath>
#include <vector>

namespace gfdm {
namespace {

using dft::Dft;
using dft::static_for;

constexpr int RT = 256;          // threads per workgroup = one block
// 126 = 9 x 14: the gather / scatter stages run 9-point transforms on 14 threads per row (224 of the 256 threads), the middle stage 14-point
// transforms on 9 threads per row; the other way round (14 x 9) measured the same within 2 % (profiles/EXPERIMENTS.md, round 5)
constexpr int RA = 9, RB = 14;

constexpr int pow_mod(int b, int e, int m)
{
    long r = 1, x = b % m;
    while (e > 0) {
        if (e & 1) r = (r * x) % m;
        x = (x * x) % m;
        e >>= 1;
    }
    return (int)r;
}
constexpr int primitive_root(int P)
{
    for (int g = 2; g < P; ++g) {
        bool ok = true;
        int n = P - 1;
        for (int f = 2; f <= n; ++f)
            if (n % f == 0) {
                if (pow_mod(g, (P - 1) / f, P) == 1) ok = false;
                while (n % f == 0) n /= f;
            }
        if (ok) return g;
    }
    return 0;
}
constexpr int inv_mod(int a, int m)
{
    for (int i = 1; i < m; ++i)
        if ((a * i) % m == 1) return i;
    return 0;
}

// index maps of one (P, A, B): everything a constant expression.  A row holds the A positions of one thread as bytes, padded to 16 so that a thread
// fetches its row with one 16-byte load at kernel entry and keeps it in four registers for both transforms of the block.
// (alignas(16): rader_prologue fetches a row through a uint4 pointer; an array of `unsigned` alone would only promise 4-byte alignment -- round-5 advisor)
template <int P, int A, int B>
struct alignas(16) RaderMaps {
    static constexpr int n = P - 1;
    static constexpr int G = primitive_root(P);
    static_assert(A * B == n && A <= 16 && P <= 256 && dft::gcd_of(A, B) == 1, "P - 1 = A x B with coprime factors, A values per map row");
    unsigned in[B][4];            // stage 1: byte p1 = position g^((B p1 + A p2) mod n) mod P of the sample thread (.., p2) takes as its p1-th
    unsigned out[B][4];           // stage 3: byte q1 = output index m = g^-((B q1 + A q2) mod n) mod P
    unsigned neg[B][4];           // ... and P - m: where output m of the forward transform lands when it serves as the inverse transform
    constexpr RaderMaps() : in{}, out{}, neg{}
    {
        const int ginv = pow_mod(G, P - 2, P);
        for (int p2 = 0; p2 < B; ++p2)
            for (int p1 = 0; p1 < A; ++p1) {
                const int e = (B * p1 + A * p2) % n;
                in[p2][p1 / 4] |= (unsigned)pow_mod(G, e, P) << (8 * (p1 % 4));
                out[p2][p1 / 4] |= (unsigned)pow_mod(ginv, e, P) << (8 * (p1 % 4));
                neg[p2][p1 / 4] |= (unsigned)(P - pow_mod(ginv, e, P)) << (8 * (p1 % 4));
            }
    }
};
template <int P, int A, int B>
__device__ constexpr RaderMaps<P, A, B> k_rader_maps{};

struct MapRow {
    unsigned w[4];
    template <int I> __device__ __forceinline__ int at() const { return (int)((w[I / 4] >> (8 * (I % 4))) & 0xffu); }
};

// The stages of a block do not fill the workgroup evenly: 224 / 144 / 224 threads of the row transforms, 127 of the column stage -- wavefront 0 always
// works, wavefront 3 mostly waits, and wavefront w of every resident workgroup sits on SIMD w.  Each workgroup therefore ROTATES the roles of its
// wavefronts by a number taken from its block index (the workgroups resident on one CU differ in bits 8 and up of it: 8 XCDs x 32 CUs take
// consecutive indices), so that the busy roles of the four resident workgroups land on different SIMDs.  Everything below indexes by this role id.
__device__ __forceinline__ int role_id()
{
#ifndef GFDM_RADER_NO_ROTATION
    const unsigned b = blockIdx.x;
    return (int)((threadIdx.x + 64u * ((b >> 8) + (b >> 10) + b)) & (RT - 1));
#else
    return (int)threadIdx.x;
#endif
}

__device__ __forceinline__ cf cmul(cf a, cf b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf cmulj(cf a, cf b) { return make_float2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a * conj(b)
__device__ __forceinline__ cf cfma(cf a, cf b, cf c) { return make_float2(c.x + a.x * b.x - a.y * b.y, c.y + a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cf cdiv(cf a, cf b)                   // (v_rcp_f32: 1 ulp, as the row-lane kernels' equaliser)
{
    const float d = __builtin_amdgcn_rcpf(b.x * b.x + b.y * b.y);
    return make_float2((a.x * b.x + a.y * b.y) * d, (a.y * b.x - a.x * b.y) * d);
}

// LDS of one block.  T: the block, ALWAYS rows of P values at stride TS (132 = 4 mod 32: the transposed write of the receiver's load and the transposed
// read of the modulator's store take the minimum of two LDS cycles per wavefront).  W: work tile of a row transform as A planes [j1][row * B + p2]
// at plane stride PL -- stage 1 writes and stage 3 reads it with consecutive lanes on consecutive elements, stage 2's accesses (lane = (row, j1)) spread
// over the banks for an odd PL (simulated: 2.3 cycles per access against the minimum of 2).  X0: output 0 of every row.  BS: the kernel spectrum.
// S2 (receivers with cancellation rounds only): the filtered spectra S, which every round cancels against.
template <int K, int P, int A, int B, bool IC = false>
struct RaderLds {
    static constexpr int TS = 132, PL = K * B + 1;
    static_assert(TS >= P && PL >= K * B && (PL & 1) == 1 && A * PL * 2 >= RT, "tile strides");
    cf T[K * TS];
    cf W[A * PL];
    cf BS[A * B];
    cf X0[K];
    cf S2[IC ? K * TS : 1];
};

// Forward P-point transforms of ROWS rows at once (all RT threads call; the caller has made the sources visible and synchronises before it reads
// what dst wrote).  src(row, pos) reads sample pos of a row, dst(row, m, value) takes output m, pre3() runs in front of stage 3 (loads a caller wants
// in flight early).  dst may overwrite the sources: every source read happens before the second barrier.
template <int ROWS, int P, int A, int B, class Lds, class Src, class Dst, class Pre3>
__device__ __forceinline__ void rader_rows(Lds& lds, const MapRow& min, const MapRow& mout, Src src, Dst dst, Pre3 pre3)
{
    constexpr int PL = Lds::PL;
    const int t = role_id();
    static_assert(ROWS * A <= RT && ROWS * B <= RT, "one pass per stage");
    if (t < ROWS * B) {
        const int row = t / B;
        cf v[A];
        static_for<0, A>([&](auto i) { constexpr int p1 = decltype(i)::value; v[p1] = src(row, min.template at<p1>()); });
        Dft<A, false>::run(v);
        static_for<0, A>([&](auto i) { constexpr int j1 = decltype(i)::value; lds.W[j1 * PL + t] = v[j1]; });
    }
    __syncthreads();
    if (t < ROWS * A) {
        const int row = t / A, j1 = t - row * A;
        cf* w = lds.W + j1 * PL + row * B;
        cf u[B];
        static_for<0, B>([&](auto i) { constexpr int k = decltype(i)::value; u[k] = w[k]; });
        Dft<B, false>::run(u);
        cf x0 = make_float2(0.f, 0.f);
        if (j1 == 0) {
            x0 = src(row, 0);
            lds.X0[row] = make_float2(x0.x + u[0].x, x0.y + u[0].y);          // X[0] = x[0] + sum of the others
        }
        static_for<0, B>([&](auto i) { constexpr int k = decltype(i)::value; u[k] = cmul(u[k], lds.BS[j1 * B + k]); });
        u[0] = make_float2(u[0].x + x0.x, u[0].y + x0.y);                     // on bin 0 of the convolution: + x[0] on every output
        Dft<B, true>::run(u);
        static_for<0, B>([&](auto i) { constexpr int k = decltype(i)::value; w[k] = u[k]; });
    }
    __syncthreads();
    pre3();
    if (t < ROWS * B) {
        const int row = t / B, q2 = t - row * B;
        cf v[A];
        static_for<0, A>([&](auto i) { constexpr int j1 = decltype(i)::value; v[j1] = lds.W[j1 * PL + t]; });
        Dft<A, true>::run(v);
        static_for<0, A>([&](auto i) { constexpr int q1 = decltype(i)::value; dst(row, mout.template at<q1>(), v[q1]); });
        if (q2 == 0) dst(row, 0, lds.X0[row]);
    }
}

// what every kernel does first: the thread's two map rows (stage 1 / 3 thread t = (row, t % B)) and the kernel spectrum into LDS
template <int K, int P, int A, int B, class Lds>
__device__ __forceinline__ void rader_prologue(Lds& lds, const cf* __restrict__ bs, MapRow& min, MapRow& mout, MapRow& mneg)
{
    const auto& maps = k_rader_maps<P, A, B>;
    static_assert(alignof(RaderMaps<P, A, B>) >= 16 && sizeof(maps.in[0]) == 16 && sizeof(maps.in) % 16 == 0, "a map row is one aligned 16-byte load");
    const int t = role_id(), p2 = t % B;
    const uint4 a = *reinterpret_cast<const uint4*>(maps.in[p2]), b = *reinterpret_cast<const uint4*>(maps.out[p2]),
                c = *reinterpret_cast<const uint4*>(maps.neg[p2]);
    min.w[0] = a.x; min.w[1] = a.y; min.w[2] = a.z; min.w[3] = a.w;
    mout.w[0] = b.x; mout.w[1] = b.y; mout.w[2] = b.z; mout.w[3] = b.w;
    mneg.w[0] = c.x; mneg.w[1] = c.y; mneg.w[2] = c.z; mneg.w[3] = c.w;
    if (t < A * B) lds.BS[t] = bs[t];
}

// hard decision of the cancellation rounds (gr::digital::constellation::decision_maker at lib/advanced_receiver_kernel_cc.cc:119): QPSK / BPSK sign tests
// (zero -> the negative point) or the nearest of the points, first minimum
__device__ __forceinline__ cf rader_decide(cf x, const IcParams& ic)
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

// ---------------------------------------------------------------------------------------------------------------------
// receiver: fft_[equalize_]filter_downsample (RX_FD) / generic_work[_equalize] (RX_DEMOD) / advanced receiver (IC: RX_DEMOD + cancellation rounds)
//   lib/receiver_kernel_cc.cc:165-192, 211-225, 301-334
// LT: the overlap when it is 2 or 4 (the filter then runs on the column in registers), 0 = any overlap (run-time loop through the tile)
// IC: cancellation rounds of advanced_receiver_kernel_cc::perform_ic_iterations (lib/advanced_receiver_kernel_cc.cc:56-76) in the reference's own
// frequency-domain form, S' = S - ic (.) DFT_M(dec_{k-1} + dec_{k+1}), d = IDFT_M(S') / M: two more row transforms per round, S kept in a third tile
// (three blocks per CU instead of four, which is why the plain receivers are an instantiation of their own).
template <int K, int P, int A, int B, int LT, bool IC>
__global__ __launch_bounds__(RT, IC ? 3 : 4) void k_rader_receive(DevicePlan p, IcParams ic, int mode, cf* __restrict__ out, const cf* __restrict__ in,
                                                                 const cf* __restrict__ f_eq)
{
    constexpr int N = K * P;
    typedef RaderLds<K, P, A, B, IC> Lds;
    constexpr int TS = Lds::TS;
    __shared__ __attribute__((aligned(16))) Lds lds;
    cf* T = lds.T;
    const int t = role_id(), L = p.L;
    const cf* x = in + (int64_t)blockIdx.x * N;
    cf* o = out + (int64_t)blockIdx.x * N;
    static_assert((K & (K - 1)) == 0, "subcarriers: a power of two");
    for (int i = t; i < N; i += RT) T[(i & (K - 1)) * TS + i / K] = dft::ld_stream(x + i);       // x[K p + q] -> row q, position p
    cf e[K];                                                       // the column's equaliser bins, requested before anything waits
    if (f_eq != nullptr && t < P) {
        const cf* eq = f_eq + (int64_t)blockIdx.x * N + t;
        static_for<0, K>([&](auto i) { constexpr int j = decltype(i)::value; e[j] = dft::ld_stream(eq + j * P); });
    }
    MapRow min, mout, mneg;
    rader_prologue<K, P, A, B>(lds, p.raderB, min, mout, mneg);
    __syncthreads();
    // A[q][m] = W_N^(q m) sum_p x[K p + q] W_M^(p m); the column's twiddles W_N^(q m), q < K, are requested in front of stage 3 and used after the barrier
    cf tw[K];
    rader_rows<K, P, A, B>(lds, min, mout, [&](int q, int pos) { return T[q * TS + pos]; }, [&](int q, int m, cf v) { T[q * TS + m] = v; },
                           [&]() { if (t < P) static_for<0, K>([&](auto i) { constexpr int q = decltype(i)::value; tw[q] = p.wN[q * t]; }); });
    __syncthreads();
    if (t < P) {                                                   // column m = t: X[j][m] = FFT_N(x)[M j + m], equaliser, filter + fold
        cf y[K];
        static_for<0, K>([&](auto i) { constexpr int q = decltype(i)::value; y[q] = cmul(T[q * TS + t], tw[q]); });
        Dft<K, false>::run(y);
        if (f_eq != nullptr) static_for<0, K>([&](auto i) { constexpr int j = decltype(i)::value; y[j] = cdiv(y[j], e[j]); });
        // S[k][m] = sum_i taps[((i + L/2) % L) M + m] X[(k + i - L/2) mod K][m]
        cf s[K];
        static_for<0, K>([&](auto i) { s[decltype(i)::value] = make_float2(0.f, 0.f); });
        if constexpr (LT > 0) {                                    // overlap known at compile time: the column never leaves the registers
            static_for<0, LT>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                const cf tp = p.taps[((i + LT / 2) % LT) * P + t];
                static_for<0, K>([&](auto ki) { constexpr int k = decltype(ki)::value; s[k] = cfma(tp, y[(((k + i - LT / 2) % K) + K) % K], s[k]); });
            });
        } else {
            static_for<0, K>([&](auto i) { constexpr int j = decltype(i)::value; T[j * TS + t] = y[j]; });     // (the thread's own column: no barrier)
            int row0 = (K - (L / 2) % K) % K, part = (L / 2) % L;
            for (int i = 0; i < L; ++i) {
                const cf tp = p.taps[part * P + t];
                static_for<0, K>([&](auto ki) {
                    constexpr int k = decltype(ki)::value;
                    int r = row0 + k;
                    if (r >= K) r -= K;
                    s[k] = cfma(tp, T[r * TS + t], s[k]);
                });
                if (++row0 == K) row0 = 0;
                if (++part == L) part = 0;
            }
        }
        if (mode == RX_FD) static_for<0, K>([&](auto i) { constexpr int k = decltype(i)::value; dft::st_stream(o, k * P + t, s[k]); });
        else static_for<0, K>([&](auto i) { constexpr int k = decltype(i)::value; T[k * TS + t] = s[k]; });
        if constexpr (IC) static_for<0, K>([&](auto i) { constexpr int k = decltype(i)::value; lds.S2[k * TS + t] = s[k]; });
    }
    if (mode == RX_FD) return;
    __syncthreads();
    // d[k][p] = (1/M) sum_m S[k][m] W_M^(-p m): the forward transform with its output index negated
    const float invM = 1.f / (float)P;
    rader_rows<K, P, A, B>(lds, min, mneg, [&](int k, int pos) { return T[k * TS + pos]; },        // (mneg: output m is written at P - m; m = 0 stays)
                           [&](int k, int m, cf v) { T[k * TS + m] = make_float2(v.x * invM, v.y * invM); }, [] {});
    __syncthreads();
    if constexpr (IC) {
        cf* S2 = lds.S2;
        for (int j = 0; j < ic.ic_iter; ++j) {
            if (ic.do_phase_compensation > 0 && j == 0) {
                // calculate_phase_offset (adv:78-91): phi = mean over the MAP's symbols of arg(decision) - arg(symbol), no unwrapping; S turns by it and keeps
                // the turn (adv:63-70).  The decisions below are taken on the symbols as they are (decision before rotation).
                float acc = 0.f;
                for (int idx = t; idx < ic.n_active * P; idx += RT) {
                    const int a = idx / P, m = idx - a * P;
                    const cf v = T[ic.smap[a] * TS + m];
                    const cf d = rader_decide(v, ic);
                    acc += atan2f(d.y, d.x) - atan2f(v.y, v.x);
                }
                float* red = reinterpret_cast<float*>(lds.W);                      // (the work tile is free between transforms)
                red[t] = acc;
                __syncthreads();
                for (int sft = RT / 2; sft > 0; sft >>= 1) {
                    if (t < sft) red[t] += red[t + sft];
                    __syncthreads();
                }
                const float phi = red[0] / (float)(ic.n_active * P);
                float sn, cs;
                sincosf(phi, &sn, &cs);
                const cf rot = make_float2(cs, sn);
                __syncthreads();                                                   // (everybody has read red[0] before the next transform writes the tile)
                for (int i = t; i < N; i += RT) { const int k = i / P, m = i - k * P; S2[k * TS + m] = cmul(S2[k * TS + m], rot); }
            }
            // map_symbols_to_constellation_points (adv:109-123): block zeroed, the map's subcarriers decided -- in place
            for (int i = t; i < N; i += RT) {
                const int k = i / P, m = i - k * P;
                T[k * TS + m] = ic.active[k] ? rader_decide(T[k * TS + m], ic) : make_float2(0.f, 0.f);
            }
            __syncthreads();
            // S'[k][m] = S[k][m] - ic[m] DFT_M(dec_{k-1} + dec_{k+1})[m]     (receiver_kernel_cc.cc:274-299; neighbours wrap mod K)
            rader_rows<K, P, A, B>(lds, min, mout,
                                   [&](int k, int pos) {
                                       const cf a = T[((k + K - 1) & (K - 1)) * TS + pos], b = T[((k + 1) & (K - 1)) * TS + pos];
                                       return make_float2(a.x + b.x, a.y + b.y);
                                   },
                                   [&](int k, int m, cf v) {
                                       const cf c = cmul(p.ictaps[m], v), sv = S2[k * TS + m];
                                       T[k * TS + m] = make_float2(sv.x - c.x, sv.y - c.y);
                                   }, [] {});
            __syncthreads();
            rader_rows<K, P, A, B>(lds, min, mneg, [&](int k, int pos) { return T[k * TS + pos]; },
                                   [&](int k, int m, cf v) { T[k * TS + m] = make_float2(v.x * invM, v.y * invM); }, [] {});
            __syncthreads();
        }
    }
    for (int i = t; i < N; i += RT) dft::st_stream(o, i, T[(i / P) * TS + i % P]);
}

// modulator, lib/modulator_kernel_cc.cc:98-141
template <int K, int P, int A, int B, int LT>
__global__ __launch_bounds__(RT, 4) void k_rader_modulate(DevicePlan p, cf* __restrict__ out, const cf* __restrict__ in)
{
    constexpr int N = K * P;
    typedef RaderLds<K, P, A, B> Lds;
    constexpr int TS = Lds::TS;
    __shared__ __attribute__((aligned(16))) Lds lds;
    cf* T = lds.T;
    const int t = role_id(), L = p.L;
    const cf* x = in + (int64_t)blockIdx.x * N;
    cf* o = out + (int64_t)blockIdx.x * N;
    for (int i = t; i < N; i += RT) T[(i / P) * TS + i % P] = dft::ld_stream(x + i);
    MapRow min, mout, mneg;
    rader_prologue<K, P, A, B>(lds, p.raderB, min, mout, mneg);
    __syncthreads();
    cf tw[K];
    rader_rows<K, P, A, B>(lds, min, mout, [&](int k, int pos) { return T[k * TS + pos]; }, [&](int k, int m, cf v) { T[k * TS + m] = v; },     // D_k = FFT_M(d_k)  :109-110
                           [&]() { if (t < P) static_for<0, K>([&](auto i) { constexpr int q = decltype(i)::value; tw[q] = p.wN[q * t]; }); });
    __syncthreads();
    if (t < P) {
        // Y[j][m] = sum_i D[(j - i + L/2) mod K][m] taps[((i + L/2) % L) M + m], m < part_len   (gather form of :116-132), then the K-point inverse over j
        cf y[K];
        static_for<0, K>([&](auto i) { y[decltype(i)::value] = make_float2(0.f, 0.f); });
        if (t < p.part_len) {
            if constexpr (LT > 0) {
                cf d[K];
                static_for<0, K>([&](auto i) { constexpr int k = decltype(i)::value; d[k] = T[k * TS + t]; });
                static_for<0, LT>([&](auto ii) {
                    constexpr int i = decltype(ii)::value;
                    const cf tp = p.taps[((i + LT / 2) % LT) * P + t];
                    static_for<0, K>([&](auto ji) { constexpr int j = decltype(ji)::value; y[j] = cfma(d[(((j - i + LT / 2) % K) + K) % K], tp, y[j]); });
                });
            } else {
                int row0 = (L / 2) % K, part = (L / 2) % L;
                for (int i = 0; i < L; ++i) {
                    const cf tp = p.taps[part * P + t];
                    static_for<0, K>([&](auto ji) {
                        constexpr int j = decltype(ji)::value;
                        int r = row0 + j;
                        if (r >= K) r -= K;
                        y[j] = cfma(T[r * TS + t], tp, y[j]);
                    });
                    if (--row0 < 0) row0 = K - 1;
                    if (++part == L) part = 0;
                }
            }
        }
        Dft<K, true>::run(y);
        static_for<0, K>([&](auto i) { constexpr int q = decltype(i)::value; T[q * TS + t] = cmulj(y[q], tw[q]); });       // conj(W_N^(q m))
    }
    __syncthreads();
    // x[K p + q] = (1/N) sum_m u[q][m] W_M^(-p m): row q, position p                                              :137-140
    const float scale = 1.f / (float)N;
    rader_rows<K, P, A, B>(lds, min, mneg, [&](int q, int pos) { return T[q * TS + pos]; },
                           [&](int q, int m, cf v) { T[q * TS + m] = make_float2(v.x * scale, v.y * scale); }, [] {});
    __syncthreads();
    for (int i = t; i < N; i += RT) dft::st_stream(o, i, T[(i & (K - 1)) * TS + i / K]);         // x[K p + q] <- row q, position p
}

}  // namespace

// the shapes compiled in: (timeslots, subcarriers) -> (A, B).  A handle carries the kernel spectrum (DevicePlan::raderB) when it was created under
// gfdm_hip_set_dft_matrix_cores(1), the default; modes 0 / 2 keep the dense transforms of the generic kernels (vector ALU / matrix cores) for A/B.
bool rader_supports(int M, int K) { return M == 127 && K == 16; }

// FFT_n(b) / n with b[p] = exp(-2 pi j (g^-p mod P) / P), in the order stage 2 reads it: [j1][j2] = bin j with j = j1 mod A, j = j2 mod B
void rader_host_table(int M, std::vector<cf>& tab)
{
    tab.clear();
    if (M != 127) return;
    constexpr int P = 127, A = RA, B = RB, n = P - 1;
    constexpr int G = primitive_root(P);
    const int ginv = pow_mod(G, P - 2, P);
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<double> br(n), bi(n);
    for (int q = 0; q < n; ++q) {
        const double a = -two_pi * (double)pow_mod(ginv, q, P) / (double)P;
        br[q] = std::cos(a);
        bi[q] = std::sin(a);
    }
    const int Bi = inv_mod(B % A, A), Ai = inv_mod(A % B, B);
    tab.resize((size_t)n);
    for (int j1 = 0; j1 < A; ++j1)
        for (int j2 = 0; j2 < B; ++j2) {
            const int j = (B * Bi * j1 + A * Ai * j2) % n;
            double sr = 0.0, si = 0.0;
            for (int q = 0; q < n; ++q) {
                const double a = -two_pi * (double)((j * q) % n) / (double)n, c = std::cos(a), s = std::sin(a);
                sr += br[q] * c - bi[q] * s;
                si += br[q] * s + bi[q] * c;
            }
            tab[(size_t)j1 * B + j2] = make_float2((float)(sr / n), (float)(si / n));
        }
}

bool rader_applies_modulate(const DevicePlan& p, const TxParams& tx)
{
    return p.raderB != nullptr && rader_supports(p.M, p.K) && !tx.mapped && !tx.framed;
}

bool rader_applies_receive(const DevicePlan& p, const IcParams& ic, const EstPlan* est, int mode)
{
    if (p.raderB == nullptr || !rader_supports(p.M, p.K) || est != nullptr) return false;
    if (ic.io.in_stride != 0 || ic.io.in_offset != 0 || ic.io.demap) return false;
    return mode == RX_FD || mode == RX_DEMOD || mode == RX_IC;
}

hipError_t launch_rader_modulate(const DevicePlan& p, cf* out, const cf* in, int64_t nblocks, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    auto kern = p.L == 2 ? k_rader_modulate<16, 127, RA, RB, 2> : p.L == 4 ? k_rader_modulate<16, 127, RA, RB, 4> : k_rader_modulate<16, 127, RA, RB, 0>;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(RT), 0, s, p, out, in);
    return hipGetLastError();
}

hipError_t launch_rader_receive(const DevicePlan& p, const IcParams& ic, int mode, cf* out, const cf* in, const cf* f_eq, int64_t nblocks, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    if (mode == RX_IC && ic.ic_iter > 0) {
        auto kern = p.L == 2 ? k_rader_receive<16, 127, RA, RB, 2, true> : p.L == 4 ? k_rader_receive<16, 127, RA, RB, 4, true> : k_rader_receive<16, 127, RA, RB, 0, true>;
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(RT), 0, s, p, ic, (int)RX_DEMOD, out, in, f_eq);
        return hipGetLastError();
    }
    auto kern = p.L == 2 ? k_rader_receive<16, 127, RA, RB, 2, false> : p.L == 4 ? k_rader_receive<16, 127, RA, RB, 4, false> : k_rader_receive<16, 127, RA, RB, 0, false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(RT), 0, s, p, ic, mode == RX_FD ? (int)RX_FD : (int)RX_DEMOD, out, in, f_eq);
    return hipGetLastError();
}

}  // namespace gfdm
