// "Row-lane" HIP kernel family for gfx950: one lane per subcarrier row, one GFDM block per K lanes.
//
// Why this layout: the benchmark batch (4096 blocks of K = 64, M = 9) is only 2.4 M symbols.  Spread over 256 CUs x 4 SIMDs
// that is four blocks per SIMD, so a layout that packs several blocks into one wavefront (e.g. four rows per lane: built and
// measured in round 1, 17 vs 11 us) leaves one wavefront per SIMD and the launch runs at the latency of a single wave.  Here a block occupies K lanes (a whole
// wavefront at K = 64; two at K = 128; four at K = 256): 4096 blocks become 4096 waves = 16 per CU.
//
// Decomposition (n = K p + q, f = M j + m):  X[M j + m] = sum_q W_K^{q j} W_N^{q m} (sum_p x[K p + q] W_M^{p m})
//   phase A  lane q: loads x[K p + q] (a 512-byte contiguous segment per wave instruction), M-point DFT codelet
//            (gfdm_dft.h), twiddle W_N^{q m} from a [M][K] table (coalesced), row -> the block's single LDS tile
//   phase B  K-point FFT over q for all M columns: Stockham passes IN PLACE in the tile, as few and as wide as the lane count
//            allows -- K = 64: radix 4 straight from the registers (lane-row transposes) + ONE radix-16 pass; K = 128: radix 8 +
//            radix 16; K = 256: radix 16 + radix 16; K not a power of two: radix R0 + radix R1, K = R0 R1, both <= 16 (96 = 6 x 16,
//            12 = one radix-12 pass); otherwise radix-4 passes (+ one radix-2).  Lane (tq, cg) owns the rows
//            tq + (K / R) r of column group cg: every pass reads exactly those rows, so addresses are base + immediates.  During
//            the passes the rows sit at a slot permutation (FftLayout) chosen so that no pass access has an LDS bank conflict;
//            the last pass writes natural order.
//   phase C  one-tap equaliser X[f] /= f_eq[f] in linear order on the tile (f_eq requested after phase A, coalesced)
//   phase D  lane k: L-tap filter + fold over rows k - L/2 .. k + L/2 - 1 (+wrap), 1/M folded in, M-point inverse DFT
//   IC       d_new = d0 - g (*) (dec_{k-1} + dec_{k+1}) with the M-tap circular kernel g = IDFT_M(ic)/M; for K = 64 the
//            neighbour rows come by DPP wave rotate (no LDS, no ordering point), otherwise through the tile
//   output   row -> tile, linear read, coalesced store
// Blocks of K <= 64 lanes, K a power of two, live inside one wavefront: their LDS accesses are ordered by the wave's program
// order, so they are packed four waves to a workgroup and never wait on s_barrier (block_sync).  Other K below 128 are packed back
// to back (floor(256 / K) blocks per workgroup) and ordered by the workgroup barrier.
// HBM traffic: x (+ f_eq) in, out out (streamed: non-temporal loads and stores, gfdm_dft.h); nothing else leaves the CU.
// The modulator is the transposed flow; with TXMODE the resource mapper becomes its load stage and cyclic prefix / ramp / preamble
// its store stage (gfdm_tx.h).
//
// Algorithm restated from gr-gfdm: lib/modulator_kernel_cc.cc:98-141, lib/receiver_kernel_cc.cc:165-334,
// lib/advanced_receiver_kernel_cc.cc:56-123, lib/transmitter_kernel.cc:78-107.
#pragma once
#include "gfdm_dft.h"
#include "gfdm_plan.h"
#include "gfdm_tx.h"
#include "gfdm_est.h"
#include "gfdm_rowgeom.h"

#ifndef __HIPCC_RTC__
#include <cstdlib>
#endif

// The kernels live in an anonymous namespace when they are compiled into the library (one translation unit per shape and part).
// Instantiated at run time through hiprtc (gfdm_jit.hip) they need names that can be looked up: gfdm::jit::k_row_receive<...>.
#ifdef __HIPCC_RTC__
#define GFDM_ROWLANE_NS jit
#else
#define GFDM_ROWLANE_NS
#endif

#ifndef GFDM_ROW_WG
#define GFDM_ROW_WG 256
#endif

namespace gfdm {
namespace GFDM_ROWLANE_NS {

using namespace dft;

// Diagnostic build only (-DGFDM_STAMPS, scratch/stamps.py): per-wave timestamps of the phase boundaries, written to a
// side buffer that nothing else reads.  The product library is built without it.
#ifdef GFDM_STAMPS
__device__ unsigned long long* g_stamp_buf = nullptr;
#define GFDM_STAMP(slot)                                                                                         \
    do {                                                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                              \
        if (g_stamp_buf && (threadIdx.x & 63) == 0)                                                               \
            g_stamp_buf[((size_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define GFDM_STAMP(slot) do { } while (0)
#endif

template <int K> struct RowShape {
    static_assert(rowgeom::supported(K), "row-lane family: K a power of two 4 .. 1024, or K <= 1024 a product of two or three factors <= 32 (gfdm_rowgeom.h)");
    static constexpr bool MIXED3 = rowgeom::mixed3(K);     // no two-factor plan: three passes
    // K not a power of two: two Stockham passes of radix R0 and R1 (R0 = 1: one pass), lds_subcarrier_fft2
    static constexpr bool MIXED = rowgeom::mixed(K);
    static constexpr int R0 = MIXED ? rowgeom::mixed_r0(K) : 1, R1 = MIXED ? rowgeom::mixed_r1(K) : 1;
    static constexpr int log2K = (K == 4) ? 2 : (K == 8) ? 3 : (K == 16) ? 4 : (K == 32) ? 5 : (K == 64) ? 6 : (K == 128) ? 7 : (K == 256) ? 8 : (K == 512) ? 9 : (K == 1024) ? 10 : 0;
    static constexpr int NP4 = log2K / 2;
    static constexpr bool HAS2 = (log2K & 1) != 0;
    static constexpr int WG = rowgeom::wg(K);              // threads per workgroup (blocks of K <= 64 lanes are packed)
    static constexpr int BPW = rowgeom::bpw(K);            // blocks per workgroup
    static constexpr int RG = K / 4;                       // row groups of the FFT passes
    // K = 256 = 16 x 16: TWO radix-16 passes instead of four radix-4 passes (lds_subcarrier_fft16): half the LDS traffic and half
    // the ordering points of the subcarrier FFT, the same butterfly arithmetic.  This shape is LDS-limited to two blocks per CU,
    // so its launch time is close to the SUM of its HBM, LDS and VALU time rather than their maximum.
    static constexpr bool RADIX16 = (K == 256);
    // K = 128 = 8 x 16: one radix-8 and one radix-16 pass instead of three radix-4 passes and a radix-2 pass (lds_subcarrier_fft8x16)
    static constexpr bool RADIX8X16 = (K == 128);
    // K = 512 = 8 x 8 x 8, K = 1024 = 8 x 8 x 16: THREE wide passes (lds_subcarrier_fft3) instead of four or five radix-4 / radix-2 passes
    static constexpr bool WIDE3 = (K == 512 || K == 1024) || MIXED3;
    static constexpr bool WIDE = RADIX16 || RADIX8X16 || MIXED || WIDE3;   // FftTwiddles holds the twiddles of the wide passes
    static constexpr int WIDE_R0 = (RADIX16 ? 16 : RADIX8X16 ? 8 : MIXED ? R0 : MIXED3 ? rowgeom::mixed3_r0(K) : WIDE3 ? 8 : 1);
    static constexpr int WIDE_R1 = (RADIX16 || RADIX8X16) ? 16 : MIXED ? R1 : MIXED3 ? rowgeom::mixed3_r1(K) : WIDE3 ? 8 : 1;
    static constexpr int WIDE_R2 = MIXED3 ? rowgeom::mixed3_r2(K) : WIDE3 ? K / 64 : 1;
    static constexpr bool LDS_REDUCE = MIXED || MIXED3;    // lanes of a block not aligned to wavefronts: block-wide sums go through the tile
};

// x mod K for 0 <= x (row and twiddle indices): a mask where K is a power of two, a constant division otherwise
template <int K>
__device__ __forceinline__ int wrap_k(int x)
{
    if constexpr (rowgeom::pow2(K)) return x & (K - 1); else return x % K;
}

constexpr int pow4(int s) { return 1 << (2 * s); }

// Ordering point for the block's LDS tile.  A block of K <= 64 lanes, K a power of two, lives inside ONE wavefront, whose LDS
// instructions execute in program order: only the compiler has to be kept from reordering them (the other waves of the workgroup
// work on other blocks and are never waited for).  Larger blocks span several waves and need the workgroup barrier, and so do
// blocks whose K does not divide 64 (they are packed back to back and straddle wavefronts).
template <int K>
__device__ __forceinline__ void block_sync()
{
    if constexpr (rowgeom::wave_local(K)) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// LDS tile: [row][M] complex, rows contiguous (M odd => conflict-free row access); tiles of one block are TS apart
template <int K, int M> struct RowTile {
    static constexpr int N = K * M;
    static constexpr int TS = rowgeom::tile_stride(K, M);
};

template <int K, int M> constexpr size_t row_lds_bytes() { return rowgeom::lds_bytes(K, M); }

// EQ_PREAMBLE: per block, behind the tiles, the edge-extended active estimate bins (<= K + 8)
template <int K> struct EstTile {
    static constexpr int FS = rowgeom::est_stride(K);
    static constexpr size_t bytes = rowgeom::est_bytes(K);
};

// Row placement inside the tile WHILE the subcarrier FFT runs.  In natural order the autosort writes of the first radix-4
// passes hit rows 4 tq + u resp. j + 16 qq + 4 u, whose b64 slots repeat every 4 lanes (4-way LDS bank conflict).  With the
// low six row bits read as three radix-4 digits, row = 64 e + 16 b + 4 c + d, a group of 16 neighbouring lanes varies
//   (c, d) in every pass read and in the row-per-lane write of phase A,   (b, c) in the pass-0 writes,   (b, d) in the pass-1 writes
// (later passes vary (c, d) again).  The slot  64 e + 16 b + 4 ((c + b) & 3) + ((d + b) & 3)  is a bijection that stays
// injective mod 16 on each of those digit pairs (and mod 32 on 32 consecutive rows), so none of these accesses has a bank
// conflict.  The LAST pass writes natural order, which is what the row-per-lane phases (equaliser, filter, IC, output)
// want.  K < 64 keeps the natural order (correct; several blocks share a wavefront there and the pattern differs).
template <int K> struct FftLayout {
    static __device__ __forceinline__ int slot(int row)
    {
        if constexpr (RowShape<K>::RADIX8X16) {
            // row = 32 a + 8 b + c: the radix-8 pass writes rows 8 tq + u (16 lanes vary a and b), everything else touches 8 or 16
            // consecutive rows; rotating c by 2 a keeps all of them distinct mod 32
            const int a = (row >> 5) & 3, c = row & 7;
            return (row & ~7) | ((c + 2 * a) & 7);
        } else if constexpr (RowShape<K>::RADIX16) {
            // radix-16 passes: 16 neighbouring lanes touch rows tq + 16 r (reads, phase A), 16 tq + u (pass-0 writes); rotating the
            // low four row bits by the next four keeps the slots of both patterns distinct mod 32 (b64 slots, odd row stride)
            return (row & ~15) | ((row + (row >> 4)) & 15);
        } else if constexpr (K >= 64 && rowgeom::pow2(K) && !RowShape<K>::WIDE3) {
            const int b = (row >> 4) & 3, c = (row >> 2) & 3, d = row & 3;
            return (row & ~63) | (16 * b + 4 * ((c + b) & 3) + ((d + b) & 3));
        } else {
            return row;
        }
    }
};

// Radix-4 (and trailing radix-2) Stockham passes over the subcarrier axis, IN PLACE in one LDS tile: every lane first
// pulls its 4 x CMAX inputs into registers, a barrier separates the reads from the (autosort-permuted) writes.
// One tile instead of a ping-pong pair halves the LDS footprint, i.e. doubles the resident waves per CU.
// Input rows are expected at FftLayout<K>::slot(row); the result is in natural row order.
// Twiddles of the radix-4 passes, per lane: three roots for every pass but the last.  They are fetched ONCE, early in the kernel
// (next to the sample loads): loaded inside the passes, each pass would wait for a vector-memory round trip between its LDS reads
// and writes, because loads cannot be hoisted across the ordering points.
template <int K> struct FftTwiddles {
    cf w[RowShape<K>::WIDE ? 1 : RowShape<K>::NP4][3];
    cf w16[RowShape<K>::WIDE_R0 > 1 ? RowShape<K>::WIDE_R0 - 1 : 1];       // wide first pass of radix R0: W_K^{tq u}, u = 1 .. R0 - 1
    cf wmid[RowShape<K>::WIDE3 ? RowShape<K>::WIDE_R1 - 1 : 1];            // three-pass plans, middle pass: W_K^{qq R0 u}, u = 1 .. R1 - 1
};

template <int K>
__device__ __forceinline__ void load_fft_twiddles(FftTwiddles<K>& t, int lane, const cf* __restrict__ wK)
{
    using S = RowShape<K>;
    if constexpr (S::WIDE) {
        const int tq0 = lane % (K / S::WIDE_R0);
        static_for<1, S::WIDE_R0>([&](auto ui) { constexpr int u = decltype(ui)::value; t.w16[u - 1] = wK[wrap_k<K>(tq0 * u)]; });
        if constexpr (S::WIDE3) {
            const int qq = (lane % (K / S::WIDE_R1)) / S::WIDE_R0;
            static_for<1, S::WIDE_R1>([&](auto ui) { constexpr int u = decltype(ui)::value; t.wmid[u - 1] = wK[wrap_k<K>(qq * S::WIDE_R0 * u)]; });
        }
        return;
    }
    const int tq = lane % S::RG;
    static_for<0, S::WIDE ? 0 : S::NP4>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int str = pow4(s), ms = K / str / 4;
        if constexpr (ms > 1) {
            const int qq = tq / str;
            t.w[s][0] = wK[(qq * str) & (K - 1)];
            t.w[s][1] = wK[(qq * 2 * str) & (K - 1)];
            t.w[s][2] = wK[(qq * 3 * str) & (K - 1)];
        }
    });
}

// Two Stockham passes of radix R0 and R1 (R0 R1 = K), in place.  Pass 0: lane (tq, cg), tq = lane % (K / R0), owns rows
// tq + (K / R0) r (r < R0) of column group cg (ceil(M / R0) columns) and writes rows R0 tq + u with twiddle W_K^{tq u}.  Pass 1:
// tq = lane % R0 reads and writes the lane's own rows tq + R0 u (u < R1) and leaves natural order.  The butterflies are the
// compile-time codelets Dft<R0>, Dft<R1>.  K = 256 = 16 x 16 and K = 128 = 8 x 16 replace four radix-4 passes; subcarrier counts
// that are not a power of two (K = 96 = 6 x 16, 12 = 1 x 12, 80 = 5 x 16 ...) have no other form.  R0 = 1: one pass.
template <int K, int M, bool INV, int R0, int R1>
__device__ __forceinline__ void lds_subcarrier_fft2(cf* tile, int lane, const FftTwiddles<K>& twd)
{
    static_assert(R0 * R1 == K, "two-pass plan");
    using LY = FftLayout<K>;
    if constexpr (R0 > 1) {
        constexpr int RG = K / R0, CMAX = (M + R0 - 1) / R0;
        const int tq = lane % RG, cg = lane / RG, c0 = cg * CMAX;
        cf x[CMAX][R0];
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) static_for<0, R0>([&](auto ri) { constexpr int r = decltype(ri)::value; x[c][r] = tile[LY::slot(tq + RG * r) * M + c0 + c]; });
        });
        block_sync<K>();                                          // everyone has its inputs in registers
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) {
                Dft<R0, INV>::run(x[c]);
                static_for<0, R0>([&](auto ui) {
                    constexpr int u = decltype(ui)::value;
                    cf y = x[c][u];
                    if constexpr (u > 0) y = cmul_dir<INV>(y, twd.w16[u - 1]);
                    tile[LY::slot(R0 * tq + u) * M + c0 + c] = y;
                });
            }
        });
        block_sync<K>();
    }
    {
        constexpr int RG = K / R1, CMAX = (M + R1 - 1) / R1;
        const int tq = lane % RG, cg = lane / RG, c0 = cg * CMAX;
        cf x[CMAX][R1];
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) static_for<0, R1>([&](auto ri) { constexpr int r = decltype(ri)::value; x[c][r] = tile[LY::slot(tq + RG * r) * M + c0 + c]; });
        });
        block_sync<K>();
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) {
                Dft<R1, INV>::run(x[c]);
                static_for<0, R1>([&](auto ui) { constexpr int u = decltype(ui)::value; tile[(tq + RG * u) * M + c0 + c] = x[c][u]; });      // natural order
            }
        });
        block_sync<K>();
    }
}

// Three Stockham passes of radix R0, R1, R2 (R0 R1 R2 = K), in place: pass 0 and the last pass as in lds_subcarrier_fft2; the middle
// pass (stride R0) reads rows tq + (K / R1) r and writes rows j + R0 (R1 qq + u), j = tq % R0, qq = tq / R0, with twiddle
// W_K^{qq R0 u}.  K = 512 = 8 x 8 x 8 and K = 1024 = 8 x 8 x 16.
template <int K, int M, bool INV, int R0, int R1, int R2>
__device__ __forceinline__ void lds_subcarrier_fft3(cf* tile, int lane, const FftTwiddles<K>& twd)
{
    static_assert(R0 * R1 * R2 == K, "three-pass plan");
    using LY = FftLayout<K>;
    {
        constexpr int RG = K / R0, CMAX = (M + R0 - 1) / R0;
        const int tq = lane % RG, cg = lane / RG, c0 = cg * CMAX;
        cf x[CMAX][R0];
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) static_for<0, R0>([&](auto ri) { constexpr int r = decltype(ri)::value; x[c][r] = tile[LY::slot(tq + RG * r) * M + c0 + c]; });
        });
        block_sync<K>();
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) {
                Dft<R0, INV>::run(x[c]);
                static_for<0, R0>([&](auto ui) {
                    constexpr int u = decltype(ui)::value;
                    cf y = x[c][u];
                    if constexpr (u > 0) y = cmul_dir<INV>(y, twd.w16[u - 1]);
                    tile[LY::slot(R0 * tq + u) * M + c0 + c] = y;
                });
            }
        });
        block_sync<K>();
    }
    {
        constexpr int RG = K / R1, CMAX = (M + R1 - 1) / R1;
        const int tq = lane % RG, cg = lane / RG, c0 = cg * CMAX;
        const int j = tq % R0, qq = tq / R0;
        cf x[CMAX][R1];
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) static_for<0, R1>([&](auto ri) { constexpr int r = decltype(ri)::value; x[c][r] = tile[LY::slot(tq + RG * r) * M + c0 + c]; });
        });
        block_sync<K>();
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) {
                Dft<R1, INV>::run(x[c]);
                static_for<0, R1>([&](auto ui) {
                    constexpr int u = decltype(ui)::value;
                    cf y = x[c][u];
                    if constexpr (u > 0) y = cmul_dir<INV>(y, twd.wmid[u - 1]);
                    tile[LY::slot(j + R0 * (R1 * qq + u)) * M + c0 + c] = y;
                });
            }
        });
        block_sync<K>();
    }
    {
        constexpr int RG = K / R2, CMAX = (M + R2 - 1) / R2;
        const int tq = lane % RG, cg = lane / RG, c0 = cg * CMAX;
        cf x[CMAX][R2];
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) static_for<0, R2>([&](auto ri) { constexpr int r = decltype(ri)::value; x[c][r] = tile[LY::slot(tq + RG * r) * M + c0 + c]; });
        });
        block_sync<K>();
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) {
                Dft<R2, INV>::run(x[c]);
                static_for<0, R2>([&](auto ui) { constexpr int u = decltype(ui)::value; tile[(tq + RG * u) * M + c0 + c] = x[c][u]; });      // natural order
            }
        });
        block_sync<K>();
    }
}

template <int K, int M, bool INV, int FIRST = 0>
__device__ __forceinline__ void lds_subcarrier_fft(cf* tile, int lane, const FftTwiddles<K>& twd)
{
    using S = RowShape<K>;
    using LY = FftLayout<K>;
    if constexpr (S::WIDE3) {
        lds_subcarrier_fft3<K, M, INV, S::WIDE_R0, S::WIDE_R1, S::WIDE_R2>(tile, lane, twd);
        return;
    } else if constexpr (S::WIDE) {
        lds_subcarrier_fft2<K, M, INV, S::WIDE_R0, S::WIDE_R1>(tile, lane, twd);
        return;
    }
    if constexpr (K == 64 && FIRST == 1) {
        // K = 64 behind the register first pass: the remaining 16-point transforms (rows tq + 4 r) as ONE radix-16 pass instead of
        // two radix-4 passes.  Lane (tq, cg) = (lane % 4, lane / 4) reads and writes its own rows; the same butterfly instructions
        // per wave, a third fewer LDS operations and two ordering points fewer.
        constexpr int CMAX16 = (M + 15) / 16;
        const int tq16 = lane % 4, cg16 = lane / 4, c16 = cg16 * CMAX16;
        cf x[CMAX16][16];
        static_for<0, CMAX16>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c16 + c < M) static_for<0, 16>([&](auto ri) { constexpr int r = decltype(ri)::value; x[c][r] = tile[LY::slot(tq16 + 4 * r) * M + c16 + c]; });
        });
        block_sync<K>();
        static_for<0, CMAX16>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c16 + c < M) {
                Dft<16, INV>::run(x[c]);
                static_for<0, 16>([&](auto ui) { constexpr int u = decltype(ui)::value; tile[(tq16 + 4 * u) * M + c16 + c] = x[c][u]; });
            }
        });
        block_sync<K>();
        return;
    }
    constexpr int RG = S::RG, CMAX = (M + 3) / 4;
    const int tq = lane % RG, cg = lane / RG, c0 = cg * CMAX;
    const cf* rb[4];
    static_for<0, 4>([&](auto ri) { constexpr int r = decltype(ri)::value; rb[r] = tile + LY::slot(tq + RG * r) * M + c0; });
    static_for<FIRST, S::WIDE ? 0 : S::NP4>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int str = pow4(s), len = K / str, ms = len / 4;
        constexpr bool last = (s == S::NP4 - 1) && !S::HAS2;      // the pass that leaves the data in natural order
        const int j = tq & (str - 1), qq = tq / str;
        cf w1, w2, w3;
        if constexpr (ms > 1 && !S::WIDE) { w1 = twd.w[s][0]; w2 = twd.w[s][1]; w3 = twd.w[s][2]; }
        cf* wb[4];
        static_for<0, 4>([&](auto ui) {
            constexpr int u = decltype(ui)::value;
            const int row = j + 4 * str * qq + str * u;
            wb[u] = tile + (last ? row : LY::slot(row)) * M + c0;
        });
        cf x[CMAX][4];
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) { x[c][0] = rb[0][c]; x[c][1] = rb[1][c]; x[c][2] = rb[2][c]; x[c][3] = rb[3][c]; }
        });
        block_sync<K>();                                          // everyone has its inputs in registers
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) {
                Dft<4, INV>::run(x[c]);
                if constexpr (ms > 1) {
                    x[c][1] = cmul_dir<INV>(x[c][1], w1);
                    x[c][2] = cmul_dir<INV>(x[c][2], w2);
                    x[c][3] = cmul_dir<INV>(x[c][3], w3);
                }
                wb[0][c] = x[c][0]; wb[1][c] = x[c][1]; wb[2][c] = x[c][2]; wb[3][c] = x[c][3];
            }
        });
        block_sync<K>();
    });
    if constexpr (S::HAS2) {      // len 2, stride K/2: pairs (tq, tq + K/2), (tq + K/4, tq + 3K/4); reads slots, writes natural rows
        cf y[CMAX][4];
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) { y[c][0] = rb[0][c]; y[c][1] = rb[1][c]; y[c][2] = rb[2][c]; y[c][3] = rb[3][c]; }
        });
        block_sync<K>();
        cf* b = tile + tq * M + c0;
        static_for<0, CMAX>([&](auto ci) {
            constexpr int c = decltype(ci)::value;
            if (c0 + c < M) {
                b[c] = y[c][0] + y[c][2]; b[2 * RG * M + c] = y[c][0] - y[c][2];
                b[RG * M + c] = y[c][1] + y[c][3]; b[3 * RG * M + c] = y[c][1] - y[c][3];
            }
        });
        block_sync<K>();
    }
}

// ---- K = 64: the FIRST radix-4 pass straight from the lanes' registers ------------------------------------------------------
// Before the subcarrier FFT lane q = 16 rr + tq holds row q (all columns); the first pass combines rows tq, tq + 16, tq + 32,
// tq + 48, i.e. the SAME lane of the four 16-lane rows of the wavefront.  A 4 x 4 transpose between four registers and the four
// lane rows (v_permlane32_swap + v_permlane16_swap, two of each per 32-bit component) hands lane (rr, tq) the four inputs of
// the butterfly of column 4 c + rr: the pass needs neither the row store to the tile nor the 12 LDS reads nor the ordering
// point between them, and all 64 lanes work (the LDS form leaves the lanes of the fourth column group idle).
constexpr bool kRegFirstPass = true;

// register i of lane row rr  <-  register rr of lane row i.  v_permlane32_swap(a, b): rows 2,3 of a <-> rows 0,1 of b;
// v_permlane16_swap(a, b): rows 1,3 of a <-> rows 0,2 of b (checked on hardware, scratch/probe/permlane.hip).
__device__ __forceinline__ void lane_row_transpose4(float& f0, float& f1, float& f2, float& f3)
{
    unsigned a0 = __builtin_bit_cast(unsigned, f0), a1 = __builtin_bit_cast(unsigned, f1);
    unsigned a2 = __builtin_bit_cast(unsigned, f2), a3 = __builtin_bit_cast(unsigned, f3);
    const auto r = __builtin_amdgcn_permlane32_swap(a0, a2, false, false);
    const auto s = __builtin_amdgcn_permlane32_swap(a1, a3, false, false);
    const auto t = __builtin_amdgcn_permlane16_swap(r[0], s[0], false, false);
    const auto u = __builtin_amdgcn_permlane16_swap(r[1], s[1], false, false);
    f0 = __builtin_bit_cast(float, (unsigned)t[0]);
    f1 = __builtin_bit_cast(float, (unsigned)t[1]);
    f2 = __builtin_bit_cast(float, (unsigned)u[0]);
    f3 = __builtin_bit_cast(float, (unsigned)u[1]);
}

template <int M, bool INV>
__device__ __forceinline__ void wave_fft_first_pass(cf* tile, int lane, const FftTwiddles<64>& twd, const cf (&row)[M])
{
    using LY = FftLayout<64>;
    const int tq = lane & 15, rr = lane >> 4;
    const cf w1 = twd.w[0][0], w2 = twd.w[0][1], w3 = twd.w[0][2];
    cf* wb[4];
    static_for<0, 4>([&](auto ui) { constexpr int u = decltype(ui)::value; wb[u] = tile + LY::slot(4 * tq + u) * M + rr; });
    static_for<0, (M + 3) / 4>([&](auto gi) {
        constexpr int c4 = 4 * decltype(gi)::value;
        cf a[4];
        static_for<0, 4>([&](auto ii) { constexpr int i = decltype(ii)::value; if constexpr (c4 + i < M) a[i] = row[c4 + i]; else a[i] = mk(0.f, 0.f); });
        lane_row_transpose4(a[0].x, a[1].x, a[2].x, a[3].x);
        lane_row_transpose4(a[0].y, a[1].y, a[2].y, a[3].y);
        Dft<4, INV>::run(a);                                       // rows tq, tq + 16, tq + 32, tq + 48 of column c4 + rr
        a[1] = cmul_dir<INV>(a[1], w1);
        a[2] = cmul_dir<INV>(a[2], w2);
        a[3] = cmul_dir<INV>(a[3], w3);
        if (c4 + rr < M) { wb[0][c4] = a[0]; wb[1][c4] = a[1]; wb[2][c4] = a[2]; wb[3][c4] = a[3]; }
    });
    block_sync<64>();
}

// lane i <- lane (i -+ 1) mod 64: DPP wave rotates (GFX9 dpp_ctrl 0x13C = wave_ror:1, 0x134 = wave_rol:1).
// Spelled as update_dpp with a zero `old` and bound_ctrl: a rotate has a source lane for every lane, so neither changes the value -- but in this form hipcc's DPP
// combine folds a rotate that has ONE use into that use (v_add_f32_dpp / v_fmac_f32_dpp) instead of issuing a v_mov_b32_dpp of its own, which the mov_dpp spelling
// never got: one vector instruction less per neighbour component in every cancellation round of the K = 64 kernels (18 of a round's 171; the filter's
// rotates feed products with taps held in SGPRs, which VOP2-with-DPP cannot encode, and stay).  Same-box A/B, three alternating collections
// (profiles/r06/ic_dpp_fold_ab.csv): MF + 2 IC 134.4 -> 130.9 us per 65 536 blocks, ZF + 2 IC 14.39 -> 14.02 us per 4096; the other two points within the noise.
__device__ __forceinline__ float dpp_wave_ror1(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x13C, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_wave_rol1(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x134, 0xF, 0xF, true));
}

// c + t x for a filter tap t: prototype filters such as RRC / RC have REAL frequency-domain taps (DevicePlan::taps_real, decided once per
// handle), and then the product is two multiply-adds instead of four -- the branch is uniform for the whole launch
template <bool TREAL>
__device__ __forceinline__ cf tap_fma(cf t, cf x, cf c)
{
    if constexpr (TREAL) return mk(fmaf(t.x, x.x, c.x), fmaf(t.x, x.y, c.y)); else return cfma(t, x, c);
}

__device__ __forceinline__ cf decide_point(cf x, const IcParams& ic)
{
    if (ic.decision == 1) {
        const float s = 0.70710678118654752f;
        return mk(x.x > 0.f ? s : -s, x.y > 0.f ? s : -s);
    }
    int idx = 0;
    if (ic.decision == 2) {
        idx = (x.x > 0.f);
    } else {
        float best = __builtin_huge_valf();
        for (int i = 0; i < ic.npoints; ++i) {
            const cf pt = ic.points[i];
            const float dr = x.x - pt.x, di = x.y - pt.y, d = dr * dr + di * di;
            if (d < best) { best = d; idx = i; }
        }
    }
    return ic.points[idx];
}


// ---- the cancellation rounds on the matrix cores (ICK_MFMA) ------------------------------------------------------------------
// One round is  d_new[k][p] = d0[k][p] - sum_r g[(p - r) mod M] (dec[k-1][r] + dec[k+1][r]):  per block a product of the M x M circulant
// of g with an M x 2K matrix of decisions -- the one dense contraction of the receiver.  With QPSK decisions that matrix holds only
// 0, +-s, +-2s: EXACT in f16.  So the product runs as  D = A B + C  on v_mfma_f32_16x16x32_f16 with
//   A (16 x 32, per-handle table p.icA) = the three-term f16 split hi / mid / lo of  -s g[(p - r) mod M] 2^e  (33 significant bits, more than
//     the 24 of the f32 taps themselves: the g[r] largely cancel in the sum, so a two-term split showed 4e-5 relative error on the degenerate
//     all-zero input), hi and mid in one operand, lo in a second,
//   B (32 x 16) = the decisions of 16 subcarriers and one component, sigma[k-1] + sigma[k+1] in units of 2^-e,
//   C = d0, D = d_new in f32 (products of f16 values are exact in the f32 accumulator),
// 16 MFMAs per wavefront (4 groups of 16 subcarriers x re / im x two A operands) and round in place of M(M+1)/2 packed multiply-adds, M(M-1)/2
// packed adds and the neighbour moves of the vector-ALU form.
//
// Round 4: NOTHING of this crosses LDS any more except the two edge rows of a wavefront in blocks of several wavefronts.
//   * Contraction index.  Lane (cr = lane >> 4, cn = lane & 15) of the C / D operands holds timeslots 4 cr .. 4 cr + 3 of subcarrier column cn, and
//     the same lane of B holds contraction entries 8 cr .. 8 cr + 7.  The table maps entry 8 cr + j to timeslot 4 cr + (j & 3), high term for
//     j < 4 and residual term for j >= 4: the B operand of a lane is then its OWN four decisions, twice -- no exchange between lane rows (the
//     round-3 layout ran the contraction index straight through the timeslots and needed an f16 image of the block in LDS, written and read
//     back every round).
//   * Rows.  Lane row g of a wavefront (16 lanes) is group g of the product.  The kernel runs phase D with the rows of a block PERMUTED over the
//     lanes -- lane (g, cn) works on row G cn + g of the block's 16 G rows inside this wavefront (G = 4 for K >= 64; row_of) -- so that the
//     neighbours k - 1 and k + 1 of a group's subcarriers are the SAME lane of groups g - 1 and g + 1; only group 0's lower and group G - 1's
//     upper neighbour sit one lane over (one DPP row shift each, row rotate where the block ends inside the wavefront).
//   * d0.  Lane (g, cn) leaves phase D with all M timeslots of its row; the C operand wants lane (cr, cn) to hold timeslots 4 cr .. 4 cr + 3 of
//     the rows of lanes (0 .. 3, cn): a 4 x 4 transpose between four registers and the four lane rows per timeslot residue and component,
//     32 v_permlane*_swap (lane_row_transpose4), instead of M b64 LDS writes, a barrier and 32 conflicted reads.
//   * Blocks of several wavefronts (K >= 128) pass the decisions of each wavefront's first and last row through a small double-buffered LDS
//     area behind the tiles (128 bytes per wavefront and buffer), one workgroup barrier per round.
// Rows p >= M of the 16 x 16 tile are padding: A is zero there and in the columns of timeslots >= M, d0 is zero there.
enum IcKind { ICK_GENERAL = 0, ICK_REALSYM = 1, ICK_MFMA = 2 };

// How d0 reaches the C / D layout of the matrix-core rounds:
//   true   phase D runs with the block's rows dealt to the lanes in IcMfma's interleaved order and 32 v_permlane*_swap transpose the registers
//          (no LDS, no barrier) -- but the filter then reads rows 4 apart on neighbouring lanes, a 2-way bank conflict on every one of its
//          L M b64 reads that no row stride removes (rows = g mod 4 on a lane row leave 8 of the 16 b64 slots of a 16-lane access);
//   false  phase D keeps the natural order (conflict-free reads); d0 crosses the tile ONCE (M b64 writes, one ordering point, 16 b64 reads in
//          the interleaved order), the rounds themselves stay in registers.
// Measured on one box (profiles/r04/ic_mfma_rounds_in_registers.txt, rocprofv3 means): K=128 M=15 L=4 MF + 2 IC per 8192 blocks 67.4 us (true), 62.5 us
// (false), 63.7 us with round 3's LDS image; per 65 536 blocks 517 / 487 / 498 us.
#ifndef GFDM_IC_REG_TRANSPOSE
#define GFDM_IC_REG_TRANSPOSE 0
#endif
constexpr bool kIcRegTranspose = GFDM_IC_REG_TRANSPOSE != 0;

typedef _Float16 ic_h8 __attribute__((ext_vector_type(8)));
typedef float ic_f4 __attribute__((ext_vector_type(4)));

template <int K, int M> struct IcMfma {
    static constexpr int KW = K < 64 ? K : 64;                // rows of one block inside a wavefront
    static constexpr int G = KW / 16;                         // 16-row groups (lane rows) per block and wavefront: 1, 2, 4
    static constexpr int W = K > 64 ? K / 64 : 1;             // wavefronts per block
    static constexpr bool MULTI = !rowgeom::wave_local(K);    // K >= 128: the wavefronts' edge rows cross LDS, one barrier per round
    static constexpr int EDGE = (int)(rowgeom::ic_mfma_edge_bytes(K) - rowgeom::ic_mfma_pad_bytes(K)) / 2;   // one of the two edge buffers, bytes
    // The tile from the rounds on: [row][M] with ONE element of padding behind every G rows.  A lane row of the C / D operands holds rows G cn + g, i.e. its 16
    // lanes step by G M elements -- an even number, so without the padding the 64 lanes of an access fall on 8 of the 32 b64 slots; with it the step is odd
    // (same-box A/B, profiles/r04/ic_tile_padding_ab.txt: K=128 M=15 L=4 MF + 2 IC 61.9 -> 59.3 us per 8192 blocks, 456 -> 445 us per 65 536).  The padding lies
    // in the 16 spare elements between the tiles of a workgroup (K <= 64) or behind the tile (rowgeom::ic_mfma_pad_bytes).
    static constexpr int PADSH = G == 4 ? 2 : G == 2 ? 1 : 0;
    static constexpr int PADD = ((G * M) % 2 == 0) ? 1 : 0;
    static __device__ __forceinline__ int pa(int r, int m) { return r * M + m + (PADD ? (r >> PADSH) : 0); }
    static __device__ __forceinline__ int pa_linear(int e) { return e + (PADD ? ((e / M) >> PADSH) : 0); }
    // the row of its block that lane q of the block works on from phase D on (kIcRegTranspose)
    static __device__ __forceinline__ int row_of(int q)
    {
        if constexpr (!kIcRegTranspose) return q;
        const int l = q & (KW - 1);
        return (q - l) + G * (l & 15) + (l >> 4);
    }
    // group gi of a wavefront = lane row gi: which block of the wavefront (K < 64) it belongs to
    static constexpr int grp_block(int gi) { return gi / G; }
    struct Pre {
        uint4 a, a2;           // A operands of this lane: [high | 0] and [residual | second residual] of its four timeslots
        float sig[4];          // per group: the decision magnitude 2^-e, 0 on an inactive subcarrier
    };

    // requested with the other tables at the start of the kernel
    static __device__ __forceinline__ void preload(Pre& pre, const DevicePlan& p, const IcParams& ic)
    {
        const int lane = threadIdx.x & 63, cn = lane & 15;
        const int wbase = (K > 64) ? ((threadIdx.x & ~63) & (K - 1)) : 0;      // first row of this wavefront inside its block
        pre.a = reinterpret_cast<const uint4*>(p.icA)[lane];
        pre.a2 = reinterpret_cast<const uint4*>(p.icA)[64 + lane];
        const float sig = (float)__builtin_bit_cast(_Float16, (unsigned short)p.ic_sig);      // 2^-e, carried as its f16 bit pattern
        static_for<0, 4>([&](auto gi) {
            constexpr int g4 = decltype(gi)::value;
            pre.sig[g4] = ic.active[wbase + G * cn + (g4 % G)] ? sig : 0.f;
        });
    }

    // constellation_qpsk::decision_maker (sign tests, zero -> negative point; adv:109-123) on two values, as an f16 pair of +-sig:
    // x * inf is +-inf, or NaN for x = +-0; v_med3_f32 returns the MINIMUM of its operands when one of them is NaN, so
    // med3(x inf, -sig, sig) = sig for x > 0 and -sig for everything else -- two full-rate instructions per value and no condition code
    static __device__ __forceinline__ unsigned decide2(float x0, float x1, float sig)
    {
        const float inf = __builtin_inff();
        const float y0 = __builtin_amdgcn_fmed3f(x0 * inf, -sig, sig), y1 = __builtin_amdgcn_fmed3f(x1 * inf, -sig, sig);
        return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(y0, y1));
    }

    // lane cn <- lane cn - 1 / cn + 1 of its 16-lane row (DPP row_shr:1 = 0x111, row_shl:1 = 0x101, row_ror:n = 0x120 + n).  A block that ends
    // inside the wavefront wraps around inside the row (rotate); otherwise the lane at the end of the row takes `fill` (the neighbouring
    // wavefront's edge row).
    static __device__ __forceinline__ unsigned from_below(unsigned v, unsigned fill)
    {
        if constexpr (MULTI) return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x111, 0xF, 0xF, false);
        else return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x121, 0xF, 0xF, false);
    }
    static __device__ __forceinline__ unsigned from_above(unsigned v, unsigned fill)
    {
        if constexpr (MULTI) return (unsigned)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x101, 0xF, 0xF, false);
        else return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x12F, 0xF, 0xF, false);
    }
    static __device__ __forceinline__ unsigned add_h2(unsigned a, unsigned b)
    {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, __builtin_bit_cast(h2, a) + __builtin_bit_cast(h2, b));
    }

    // d: row row_of(q) of the block after matched filter + inverse DFT.  Leaves the block after ic_iter rounds in its tile, [k][M], synchronised.
    // `edge`: the workgroup's edge-row area (MULTI only).  TS: tile stride of the workgroup's blocks in complex elements.
    template <int TS>
    static __device__ __forceinline__ void rounds(unsigned char* smem, unsigned char* edge, cf* X, int q, const cf (&d)[M], const Pre& pre, int ic_iter)
    {
        const int lane = threadIdx.x & 63, cn = lane & 15, cr = lane >> 4;
        const int wrow = threadIdx.x & ~63;                   // first lane of this wavefront in the workgroup's lane space
        const int wbase = (K > 64) ? (wrow & (K - 1)) : 0;    // first row of this wavefront inside its block
        unsigned char* blk0 = smem + (size_t)(wrow / K) * TS * sizeof(cf);     // tile of the wavefront's first block
        const ic_h8 afrag = __builtin_bit_cast(ic_h8, pre.a), afrag2 = __builtin_bit_cast(ic_h8, pre.a2);
        ic_f4 c0[4][2], cur[4][2];
        if constexpr (!kIcRegTranspose) {
            // d0 through the block's tile (padded layout pa): rows in natural order in, timeslots 4 cr .. 4 cr + 3 of rows wbase + G cn + g out
            static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; X[pa(q, m)] = d[m]; });
            block_sync<K>();
            static_for<0, 4>([&](auto gi) {
                constexpr int g4 = decltype(gi)::value;
                const cf* row = reinterpret_cast<const cf*>(blk0) + grp_block(g4) * TS + pa(wbase + G * cn + (g4 % G), 4 * cr);
                static_for<0, 4>([&](auto ii) {
                    constexpr int i = decltype(ii)::value;
                    cf v = mk(0.f, 0.f);
                    if (12 + i < M || 4 * cr + i < M) v = row[i];
                    c0[g4][0][i] = v.x;
                    c0[g4][1][i] = v.y;
                });
            });
            block_sync<K>();                                  // everyone holds its part of d0: the tile is free for the result
        } else
        // d0 in the C / D layout: register (4 j + i) of lane row g  ->  register i of group g on lane row j, for both components
        static_for<0, 4>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            static_for<0, 2>([&](auto ci) {
                constexpr int c = decltype(ci)::value;
                float t[4];
                static_for<0, 4>([&](auto ji) {
                    constexpr int j = decltype(ji)::value;
                    if constexpr (4 * j + i < M) t[j] = c ? d[4 * j + i].y : d[4 * j + i].x; else t[j] = 0.f;
                });
                lane_row_transpose4(t[0], t[1], t[2], t[3]);
                static_for<0, 4>([&](auto gi) { constexpr int g4 = decltype(gi)::value; c0[g4][c][i] = t[g4]; });
            });
        });
        static_for<0, 4>([&](auto gi) { constexpr int g4 = decltype(gi)::value; cur[g4][0] = c0[g4][0]; cur[g4][1] = c0[g4][1]; });
        const int wave = wrow / 64 % W;                       // this wavefront's place in its block (MULTI)
        for (int it = 0; it < ic_iter; ++it) {                                                           // adv:56-76
            uint2 w[4][2];
            static_for<0, 4>([&](auto gi) {
                constexpr int g4 = decltype(gi)::value;
                const float sig = pre.sig[g4];
                static_for<0, 2>([&](auto ci) {
                    constexpr int c = decltype(ci)::value;
                    w[g4][c] = make_uint2(decide2(cur[g4][c][0], cur[g4][c][1], sig), decide2(cur[g4][c][2], cur[g4][c][3], sig));
                });
            });
            uint2 fill_lo[2] = { make_uint2(0u, 0u), make_uint2(0u, 0u) }, fill_hi[2] = { make_uint2(0u, 0u), make_uint2(0u, 0u) };
            if constexpr (MULTI) {
                // this wavefront's first row is (group 0, cn 0), its last (group 3, cn 15): [buffer][wavefront][first | last][component][cr]
                uint2* eb = reinterpret_cast<uint2*>(edge + (it & 1) * EDGE);
                if (cn == 0) { eb[(wave * 2 + 0) * 8 + cr] = w[0][0]; eb[(wave * 2 + 0) * 8 + 4 + cr] = w[0][1]; }
                if (cn == 15) { eb[(wave * 2 + 1) * 8 + cr] = w[3][0]; eb[(wave * 2 + 1) * 8 + 4 + cr] = w[3][1]; }
                block_sync<K>();
                const int below = (wave + W - 1) % W, above = (wave + 1) % W;                            // wrap mod K: rx:274-299
                fill_lo[0] = eb[(below * 2 + 1) * 8 + cr]; fill_lo[1] = eb[(below * 2 + 1) * 8 + 4 + cr];
                fill_hi[0] = eb[(above * 2 + 0) * 8 + cr]; fill_hi[1] = eb[(above * 2 + 0) * 8 + 4 + cr];
            }
            static_for<0, 4>([&](auto gi) {
                constexpr int g4 = decltype(gi)::value;
                constexpr int b0 = (g4 / G) * G, j = g4 % G;          // first group of this group's block, place inside it
                static_for<0, 2>([&](auto ci) {
                    constexpr int c = decltype(ci)::value;
                    // neighbours k - 1 and k + 1                                                               rx:274-299
                    uint2 lo, hi;
                    if constexpr (j > 0) lo = w[g4 - 1][c];
                    else lo = make_uint2(from_below(w[b0 + G - 1][c].x, fill_lo[c].x), from_below(w[b0 + G - 1][c].y, fill_lo[c].y));
                    if constexpr (j < G - 1) hi = w[g4 + 1][c];
                    else hi = make_uint2(from_above(w[b0][c].x, fill_hi[c].x), from_above(w[b0][c].y, fill_hi[c].y));
                    const unsigned n0 = add_h2(lo.x, hi.x), n1 = add_h2(lo.y, hi.y);
                    const ic_h8 nb = __builtin_bit_cast(ic_h8, make_uint4(n0, n1, n0, n1));
                    // the high terms first: where they cancel (small d0) the residual terms are then added to a small sum and keep their bits
                    cur[g4][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag, nb, c0[g4][c], 0, 0, 0);
                    cur[g4][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag2, nb, cur[g4][c], 0, 0, 0);
                });
            });
        }
        // result -> the tile, [k][M] in the padded layout (the kernel's linear read for the coalesced store uses pa_linear): group g4 of lane (cr, cn) is row wbase + G cn + g4 % G of block grp_block(g4)
        static_for<0, 4>([&](auto gi) {
            constexpr int g4 = decltype(gi)::value;
            cf* row = reinterpret_cast<cf*>(blk0) + grp_block(g4) * TS + pa(wbase + G * cn + (g4 % G), 4 * cr);
            static_for<0, 4>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                if (12 + i < M || 4 * cr + i < M) row[i] = mk(cur[g4][0][i], cur[g4][1][i]);     // (first form: known at compile time for every lane)
            });
        });
        block_sync<K>();
    }
};

// =====================================================================================================================
// EQ: EqSource.  EQ_PREAMBLE runs the preamble channel estimator (gfdm_est.h) in front, on the block's own lanes: the two K-point
// FFTs of the preamble halves reuse the subcarrier FFT on a [K][2] view of the tile, the smoothed estimate (<= K bins) stays in
// LDS, and phase C interpolates it per bin -- the N-bin equaliser vector never exists in HBM (16 K bytes read instead of 8 N).
// (ICK_MFMA: at least two waves per SIMD, i.e. at most 256 registers -- below that bound hipcc keeps the MFMA accumulators in ordinary
// VGPRs, which the dead registers of the FFT phases provide; with the default bound it takes 64 AGPRs ON TOP: 108 + 64 registers = 2 waves
// per SIMD instead of 120 = 4)
// GFDM_IC_WAVES_PER_SIMD: since round 4 the rounds need no LDS beyond the tile, so at K=128 M=15 the LDS would let a CU hold ten blocks = five waves per SIMD
// where the kernel's 98-104 registers held it at eight.  Asking for five (96 registers, 6-11 of them spilled) measured SLOWER on one box
// (profiles/r04/ic_waves_per_simd_ab.txt: MF + 2 IC 65.4 against 63.2 us per 8192 blocks, 467 against 457 us per 65 536; ZF + 2 IC 84 against 75 us): the default stays 2.
// (With the padded tile the MF + 2 IC kernel of that shape compiles to 96 registers by itself; ZF + 2 IC keeps 105, and its five-wave form with the equaliser vector
// requested late is slower as well: profiles/r04/ic_zf_late_equaliser_ab.txt.)
#ifndef GFDM_IC_WAVES_PER_SIMD
#define GFDM_IC_WAVES_PER_SIMD 2
#endif
template <int K, int M, int EQ> constexpr int ic_mfma_waves_per_simd()
{
    constexpr size_t lds = ((EQ == EQ_PREAMBLE) ? rowgeom::lds_bytes(K, M + 2) + rowgeom::est_bytes(K) : rowgeom::lds_bytes(K, M)) + rowgeom::ic_mfma_edge_bytes(K);
    constexpr size_t per_cu = (160 * 1024) / ((lds + 511) / 512 * 512) * (size_t)(rowgeom::wg(K) / 64);      // waves the LDS lets a CU hold
    return per_cu >= 4 * (size_t)GFDM_IC_WAVES_PER_SIMD ? GFDM_IC_WAVES_PER_SIMD : 2;
}
// GFDM_VALU_IC_WAVES_PER_SIMD (round 6, one experiment): the one-wavefront blocks' cancellation kernels (DPP rounds) compile to 77-79 registers = six waves per SIMD
// where their MF / ZF siblings (53-69) get seven or eight.  Asking hipcc for seven (72 registers, 3-5 spilled) or eight (64, 8-12 spilled) measured SLOWER on one box,
// three alternating collections each (profiles/r06/ic_valu_waves_ab.csv): K=64 M=9 MF + 2 IC 11.9 -> 12.9 / 13.7 us per 4096 blocks, 131 -> 137 / 144 us per 65 536;
// ZF + 2 IC 15.0 -> 15.4 / 16.8 and 163 -> 169 / 178 us.  Residency bought with spills does not pay; fewer instructions do.  Default 1 = no bound.
#ifndef GFDM_VALU_IC_WAVES_PER_SIMD
#define GFDM_VALU_IC_WAVES_PER_SIMD 1
#endif
template <int K, int M, int L, int MODE, int EQ, int ICK>
__global__ __launch_bounds__(RowShape<K>::WG, ((MODE == RX_IC && ICK == ICK_MFMA) ? ic_mfma_waves_per_simd<K, M, EQ>() : (MODE == RX_IC && K <= 64) ? GFDM_VALU_IC_WAVES_PER_SIMD : 1)) void k_row_receive(DevicePlan p, IcParams ic, EstPlan est, const cf* __restrict__ twT,
                                                                cf* __restrict__ out, const cf* __restrict__ in,
                                                                const cf* __restrict__ f_eq, int64_t nblocks)
{
    using S = RowShape<K>;
    constexpr int MS = (EQ == EQ_PREAMBLE) ? M + 2 : M;   // tile row stride: with EQ_PREAMBLE two extra columns carry the preamble halves
    using T = RowTile<K, MS>;
    constexpr int N = K * M;
    constexpr bool ICSYM = (ICK == ICK_REALSYM);
    constexpr bool ICMX = (MODE == RX_IC && ICK == ICK_MFMA);   // cancellation rounds on the matrix cores (IcMfma)
    static_assert(!ICMX || rowgeom::ic_mfma(K, M), "IcMfma: K a power of two >= 16, 4 <= M <= 16");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int g = threadIdx.x / K, q = threadIdx.x - g * K;            // q doubles as row index k in phase D
    const int64_t blk = (int64_t)blockIdx.x * S::BPW + g;
    const bool valid = blk < nblocks;
    const int64_t base = (valid ? blk : 0) * N;
    const int64_t in_base = (valid ? blk : 0) * (int64_t)(ic.io.in_stride ? ic.io.in_stride : N) + ic.io.in_offset;   // frame -> block
    cf* X = reinterpret_cast<cf*>(smem) + g * T::TS;                   // the block's single LDS tile, [row][M]

    GFDM_STAMP(0);
    // ---- phase A: timeslot DFT of row q, twiddle W_N^{q m}
    cf v[M];
    static_for<0, M>([&](auto pi) { constexpr int pp = decltype(pi)::value; v[pp] = ld_stream(in + in_base + K * pp + q); });
    FftTwiddles<K> twd;
    load_fft_twiddles<K>(twd, q, p.wK);
    // Everything else the later phases read from global memory is requested here as well: behind an ordering point a load can
    // only be issued where it is needed, and its round trip (scalar cache for the uniform tables, L1 / L2 for the per-lane
    // ones) lands on the wave's critical path.  Filter and IC taps are uniform (SGPRs) and only preloaded while they fit.
    constexpr bool PRE_TAPS = (L * M <= 24);
    cf tapv[PRE_TAPS ? L * M : 1], icgv[(PRE_TAPS && MODE == RX_IC) ? M : 1];
    if constexpr (PRE_TAPS) {
        static_for<0, L * M>([&](auto ii) { constexpr int i = decltype(ii)::value; tapv[i] = p.taps[i]; });
        if constexpr (MODE == RX_IC) static_for<0, M>([&](auto ii) { constexpr int i = decltype(ii)::value; icgv[i] = p.icg[i]; });
    }
    auto tap = [&](auto ii) { constexpr int i = decltype(ii)::value; if constexpr (PRE_TAPS) return tapv[i]; else return p.taps[i]; };
    const bool treal = p.taps_real != 0;
    auto icg = [&](auto ii) { constexpr int i = decltype(ii)::value; if constexpr (PRE_TAPS && MODE == RX_IC) return icgv[i]; else return p.icg[i]; };
    const int wgt = (MODE == RX_IC && !ICMX) ? ic.active[q] : 0;      // multiplicity of subcarrier k in subcarrier_map (0 = inactive)
    typename IcMfma<ICMX ? K : 16, ICMX ? M : 4>::Pre icpre;
    if constexpr (ICMX) IcMfma<K, M>::preload(icpre, p, ic);
    const int rank_q = (MODE != RX_FD && ic.io.demap) ? ic.io.rank[q] : -1;
    // ... and the scalar settings the later phases branch on (kernel arguments, i.e. scalar loads at the point of use otherwise)
    const int io_demap = ic.io.demap, io_nout = ic.io.nout, io_per_timeslot = ic.io.per_timeslot, io_A = ic.io.A;
    const int ic_iter = ic.ic_iter, ic_decision = ic.decision, ic_pc = ic.do_phase_compensation;
    GFDM_STAMP(1);
    // EQ_PREAMBLE: the received preamble's two halves ride through the subcarrier FFT as columns M and M + 1 of the tile
    cf pre0, pre1, inv0, inv1;
    if constexpr (EQ == EQ_PREAMBLE) {
        const cf* pre = f_eq + (valid ? blk : 0) * (int64_t)(est.pre_stride ? est.pre_stride : 2 * K);
        pre0 = ld_stream(pre + q);
        pre1 = ld_stream(pre + K + q);
        inv0 = est.inv0[q];
        inv1 = est.inv1[q];
    }
    cf tw[M];
    static_for<1, M>([&](auto mi) { constexpr int m = decltype(mi)::value; tw[m] = twT[m * K + q]; });
    dft_inplace<M, false>(v);
    constexpr bool REGPASS = (K == 64) && kRegFirstPass;
    if constexpr (REGPASS) {
        cf rowv[MS];
        rowv[0] = v[0];
        static_for<1, M>([&](auto mi) { constexpr int m = decltype(mi)::value; rowv[m] = cmul(v[m], tw[m]); });
        if constexpr (EQ == EQ_PREAMBLE) { rowv[M] = pre0; rowv[M + 1] = pre1; }
        wave_fft_first_pass<MS, false>(X, q, twd, rowv);
    } else {
        cf* xa = X + FftLayout<K>::slot(q) * MS;                   // row q goes to its FFT slot
        xa[0] = v[0];
        static_for<1, M>([&](auto mi) { constexpr int m = decltype(mi)::value; xa[m] = cmul(v[m], tw[m]); });
        if constexpr (EQ == EQ_PREAMBLE) { xa[M] = pre0; xa[M + 1] = pre1; }
        block_sync<K>();
    }
    // The equaliser vector is needed only after the subcarrier FFT: request it now, behind every wave's sample loads
    // (HBM serves requests roughly in issue order, so the samples of all waves arrive first and the transforms start
    // earlier; f_eq streams in while phases A/B run).
    constexpr bool ROWREG = (K == 64 && L == 2 && EQ == EQ_PREAMBLE && !(ICMX && kIcRegTranspose));   // equalised row stays in registers between phases C and D
    cf xrow[ROWREG ? M : 1];
    cf heq[EQ == EQ_VECTOR ? M : 1];
    if constexpr (EQ == EQ_VECTOR) {
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, M>([&](auto ii) { constexpr int i = decltype(ii)::value; heq[i] = ld_stream(f_eq + base + q + K * i); });
    }

    // ---- phase B: subcarrier FFT, in place
    lds_subcarrier_fft<K, MS, false, REGPASS ? 1 : 0>(X, q, twd);

    GFDM_STAMP(2);
    // ---- phase C: X[f] / f_eq[f] in linear order (a conj(b) / |b|^2, reciprocal by v_rcp_f32)              rx:315-316
    if constexpr (EQ == EQ_VECTOR) {
        static_for<0, M>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const cf a = X[q + K * i], b = heq[i];
            const float inv = __builtin_amdgcn_rcpf(b.x * b.x + b.y * b.y);
            X[q + K * i] = mk((a.x * b.x + a.y * b.y) * inv, (a.y * b.x - a.x * b.y) * inv);
        });
        block_sync<K>();
    } else if constexpr (EQ == EQ_PREAMBLE) {
        // lane q holds estimate bin q (est:118-145): scatter it to its place in the fftshift-ordered, edge-replicated array the
        // smoothing filter runs over                                                                       est:147-175
        cf* inter = reinterpret_cast<cf*>(smem + row_lds_bytes<K, MS>()) + g * EstTile<K>::FS;
        cf* F = X + M;                                     // smoothed bin i -> column M of row i (dead once its lane has read it)
        const cf eq = cfma(X[q * MS + M], inv0, cmul(X[q * MS + M + 1], inv1));
        const int pos = est_active_pos(q, est), n_est = est.n_est;
        if (pos >= 0) {
            inter[4 + pos] = eq;
            if (pos == 0) { inter[0] = eq; inter[1] = eq; inter[2] = eq; inter[3] = eq; }
            if (pos == n_est - 1) { inter[n_est + 4] = eq; inter[n_est + 5] = eq; inter[n_est + 6] = eq; inter[n_est + 7] = eq; }
        }
        block_sync<K>();
        if (est.dc_free) {                                                         // DC bin: mean of its neighbours
            if (q == 0) {
                const cf lo = inter[4 + est.A / 2 - 1], hi = inter[4 + est.A / 2 + 1];
                inter[4 + est.A / 2] = mk(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y));
            }
            block_sync<K>();
        }
        if (q < n_est) {                                                           // 9-tap Gaussian         est:176-187
            cf acc = mk(0.f, 0.f);
            static_for<0, 9>([&](auto ti) {
                constexpr int t = decltype(ti)::value;
                const cf x = inter[q + t];
                acc.x += x.x * est.gauss[t];
                acc.y += x.y * est.gauss[t];
            });
            F[q * MS] = acc;
        }
        block_sync<K>();
        // row q = bins M q .. M q + M - 1 lies inside one interpolation segment of the smoothed estimate   est:238-273
        cf lo, hi;
        est_row_segment(F, MS, q, est, lo, hi);
        const cf dlt = mk(hi.x - lo.x, hi.y - lo.y);
        constexpr float step = 1.0f / (float)M;
        static_for<0, M>([&](auto mi) {
            constexpr int m = decltype(mi)::value;
            const float t = (float)m * step;
            const cf a = X[q * MS + m], b = mk(lo.x + dlt.x * t, lo.y + dlt.y * t);
            const float inv = __builtin_amdgcn_rcpf(b.x * b.x + b.y * b.y);
            const cf e = mk((a.x * b.x + a.y * b.y) * inv, (a.y * b.x - a.x * b.y) * inv);
            if constexpr (ROWREG) xrow[m] = e; else X[q * MS + m] = e;      // ROWREG: phase D takes the row from registers
        });
        if constexpr (!ROWREG) block_sync<K>();
    }

    // ---- phase D: S[k][m] = sum_i taps[((i + L/2) % L) M + m] X[(k + i - L/2) mod K][m]                      rx:165-192
    // (the matrix-core rounds want the rows of a block dealt to the lanes in IcMfma's order: kq instead of q from here to the rounds)
    int kq = q;
    if constexpr (ICMX) kq = IcMfma<K, M>::row_of(q);
    cf s[M];
    static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; s[m] = mk(0.f, 0.f); });
    auto filter = [&](auto real_tag) {                     // one copy per kind of taps, chosen by a uniform branch
        constexpr bool TREAL = decltype(real_tag)::value;
        if constexpr (K == 64 && L == 2 && !(ICMX && kIcRegTranspose)) {
            // the block IS the wavefront and the only foreign row is k - 1 = lane k - 1: a DPP wave rotate of the own row replaces the
            // second LDS row read
            const cf* rb = X + q * MS;
            static_for<0, M>([&](auto mi) {
                constexpr int m = decltype(mi)::value;
                cf own;
                if constexpr (ROWREG) own = xrow[m]; else own = rb[m];
                const cf below = mk(dpp_wave_ror1(own.x), dpp_wave_ror1(own.y));
                s[m] = tap_fma<TREAL>(tap(std::integral_constant<int, M + m>{}), below, s[m]);       // i = 0: row k - 1
                s[m] = tap_fma<TREAL>(tap(std::integral_constant<int, m>{}), own, s[m]);             // i = 1: row k
            });
        } else {
            static_for<0, L>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                const cf* rb = X + wrap_k<K>(kq + i - L / 2 + K) * MS;
                static_for<0, M>([&](auto mi) {
                    constexpr int m = decltype(mi)::value;
                    s[m] = tap_fma<TREAL>(tap(std::integral_constant<int, ((i + L / 2) % L) * M + m>{}), rb[m], s[m]);
                });
            });
        }
    };
    if (treal) filter(std::true_type{}); else filter(std::false_type{});
    constexpr float invM = 1.0f / (float)M;
    cf d[M];
    if constexpr (MODE != RX_FD) {
        // from here on S only feeds inverse DFTs that are scaled by 1/M: fold the scale into S (and into the IC taps)
        static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; s[m] = scale(s[m], invM); d[m] = s[m]; });
        dft_inplace<M, true>(d);                                                                         // rx:211-225
    }
    block_sync<K>();                                      // every lane has read its neighbour rows: the tile is free
    GFDM_STAMP(3);

    if constexpr (ICMX) {
        constexpr size_t edge_off = (EQ == EQ_PREAMBLE) ? row_lds_bytes<K, MS>() + EstTile<K>::bytes : row_lds_bytes<K, M>();
        IcMfma<K, M>::template rounds<T::TS>(smem, smem + edge_off + rowgeom::ic_mfma_pad_bytes(K), X, q, d, icpre, ic_iter);
    } else if constexpr (MODE == RX_IC) {
        // One cancellation round of the reference is  d_new = IDFT_M(S - ic (.) DFT_M(nb)) / M  with nb = dec_{k-1} + dec_{k+1}.
        // Both transforms are linear, so  d_new = d0 - g (*) nb  with d0 = IDFT_M(S)/M (already in d) and the M-tap circular
        // convolution kernel g = IDFT_M(ic)/M (host table p.icg).  For the usual real, even prototype filters ic is real and
        // symmetric, hence g is too (ICSYM): M(M+1)/2 packed multiply-adds per row and round instead of two M-point DFTs.
        float* red = reinterpret_cast<float*>(reinterpret_cast<cf*>(smem) + S::BPW * T::TS);
        cf d0[M];
        static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; d0[m] = d[m]; });
        for (int it = 0; it < ic_iter; ++it) {                                                           // adv:56-76
            const bool pc = (ic_pc > 0) && (it == 0);
            float acc = 0.f;
            cf dec[M];
            if (ic_decision == 1 && !pc) {
                // QPSK hot path (constellation_qpsk::decision_maker: sign tests, zero -> negative point); the per-lane
                // amplitudes are 0 on inactive subcarriers, so one compare + one select per component           adv:109-123
                const float sp = (wgt > 0) ? 0.70710678118654752f : 0.f, sn = -sp;
                static_for<0, M>([&](auto mi) {
                    constexpr int m = decltype(mi)::value;
                    dec[m] = mk(d[m].x > 0.f ? sp : sn, d[m].y > 0.f ? sp : sn);
                });
            } else {
                static_for<0, M>([&](auto mi) {                                                          // adv:109-123
                    constexpr int m = decltype(mi)::value;
                    dec[m] = (wgt > 0) ? decide_point(d[m], ic) : mk(0.f, 0.f);
                    if (pc && wgt > 0) acc += (float)wgt * (atan2f(dec[m].y, dec[m].x) - atan2f(d[m].y, d[m].x));
                });
            }
            if (pc) {                                                                                    // adv:59-71, 78-91
                if constexpr (S::LDS_REDUCE) {
                    // the block's lanes are not aligned to wavefronts: sum through the (free) tile
                    float* part = reinterpret_cast<float*>(X);
                    part[q] = acc;
                    block_sync<K>();
                    acc = 0.f;
                    for (int i = 0; i < K; ++i) acc += part[i];
                    block_sync<K>();
                } else {
                    for (int off = 1; off < 64 && off < K; off <<= 1) acc += __shfl_xor(acc, off, 64);
                    if constexpr (K > 64) {
                        if ((q & 63) == 0) red[q >> 6] = acc;
                        block_sync<K>();
                        acc = 0.f;
                        static_for<0, K / 64>([&](auto wi) { acc += red[decltype(wi)::value]; });
                    }
                }
                const float phi = acc / (float)(ic.n_active * M);
                float sn, cs;
                sincosf(phi, &sn, &cs);
                const cf rot = mk(cs, sn);
                static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; d0[m] = cmul(d0[m], rot); });   // rotating S rotates d0
            }
            // neighbours k-1 and k+1 (wrap mod K)                                                           rx:274-299
            cf nb[M];
            if constexpr (K == 64) {
                // the block IS the wavefront: subcarrier k +- 1 is lane +- 1 with wrap-around, i.e. a DPP wave rotate --
                // no LDS traffic and no ordering point in the whole cancellation round
                static_for<0, M>([&](auto mi) {
                    constexpr int m = decltype(mi)::value;
                    nb[m] = mk(dpp_wave_ror1(dec[m].x) + dpp_wave_rol1(dec[m].x), dpp_wave_ror1(dec[m].y) + dpp_wave_rol1(dec[m].y));
                });
            } else {
                static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; X[q * M + m] = dec[m]; });
                block_sync<K>();
                const cf* below = X + wrap_k<K>(q - 1 + K) * M;
                const cf* above = X + wrap_k<K>(q + 1) * M;
                static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; nb[m] = below[m] + above[m]; });
            }
            if constexpr (ICSYM) {
                // real symmetric kernel: both components of a term share the real factor -> one packed v_pk_fma_f32 per term
                // (and one v_pk_add_f32 per symmetric pair)
                typedef float v2f __attribute__((ext_vector_type(2)));
                constexpr int H = (M - 1) / 2;
                v2f nv[M];
                static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; nv[m] = v2f{ nb[m].x, nb[m].y }; });
                static_for<0, M>([&](auto pi) {
                    constexpr int pp = decltype(pi)::value;
                    const float g0 = -icg(std::integral_constant<int, 0>{}).x;
                    v2f acc = __builtin_elementwise_fma(nv[pp], v2f{ g0, g0 }, v2f{ d0[pp].x, d0[pp].y });
                    static_for<1, H + 1>([&](auto ri) {
                        constexpr int r = decltype(ri)::value;
                        const float gr = -icg(std::integral_constant<int, r>{}).x;
                        acc = __builtin_elementwise_fma(nv[(pp - r + M) % M] + nv[(pp + r) % M], v2f{ gr, gr }, acc);
                    });
                    if constexpr (M % 2 == 0) {
                        const float gm = -icg(std::integral_constant<int, M / 2>{}).x;
                        acc = __builtin_elementwise_fma(nv[(pp + M / 2) % M], v2f{ gm, gm }, acc);
                    }
                    d[pp] = mk(acc.x, acc.y);
                });
            } else {
                static_for<0, M>([&](auto pi) {
                    constexpr int pp = decltype(pi)::value;
                    cf acc = d0[pp];
                    static_for<0, M>([&](auto ri) {
                        constexpr int r = decltype(ri)::value;
                        const cf g = icg(std::integral_constant<int, r>{}), x = nb[(pp - r + M) % M];
                        acc = mk(fmaf(-g.x, x.x, fmaf(g.y, x.y, acc.x)), fmaf(-g.x, x.y, fmaf(-g.y, x.x, acc.y)));
                    });
                    d[pp] = acc;
                });
            }
            if constexpr (K != 64) block_sync<K>();       // all neighbour reads done before the tile is rewritten
        }
    }

    GFDM_STAMP(4);
    if (MODE != RX_FD && io_demap) {
        // resource demapper fused into the store: only active subcarriers, in mapper order; for per-timeslot order the lanes of
        // one timeslot write consecutive output symbols, so no LDS staging is needed                     mapper:91-106,136-163
        const int a = rank_q;
        if constexpr (ICMX) static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; d[m] = X[IcMfma<ICMX ? K : 16, ICMX ? M : 4>::pa(q, m)]; });
        if (valid && a >= 0) {
            cf* o = out + blk * (int64_t)io_nout;
            static_for<0, M>([&](auto mi) {
                constexpr int m = decltype(mi)::value;
                const int idx = io_per_timeslot ? (m * io_A + a) : (a * M + m);
                if (idx < io_nout) st_stream(o, idx, d[m]);
            });
        }
    } else {
        // ---- output: row -> tile, linear read, coalesced store
        if constexpr (!ICMX) {                              // (IcMfma leaves its result in the tile)
            static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; X[q * M + m] = (MODE == RX_FD) ? s[m] : d[m]; });
            block_sync<K>();
        }
        if (valid) {
            if constexpr (ICMX) static_for<0, M>([&](auto ii) { constexpr int i = decltype(ii)::value; st_stream(out, base + q + K * i, X[IcMfma<ICMX ? K : 16, ICMX ? M : 4>::pa_linear(q + K * i)]); });
            else static_for<0, M>([&](auto ii) { constexpr int i = decltype(ii)::value; st_stream(out, base + q + K * i, X[q + K * i]); });
        }
    }
    GFDM_STAMP(5);
}

// =====================================================================================================================
// TXMODE 0: plain modulator.  1: input through the resource mapper (transmitter_kernel::modulate).
// 2: mapper in front AND cyclic prefix / suffix + ramp + preamble behind, all ports (transmitter_kernel::generic_work for every
// cyclic shift of the reference's transmitter_cc_impl::general_work, lib/transmitter_cc_impl.cc:165-177) -- see gfdm_tx.h.
template <int K, int M, int L, int TXMODE>
__global__ __launch_bounds__(RowShape<K>::WG) void k_row_modulate(DevicePlan p, TxParams tx, const cf* __restrict__ twT,
                                                                 cf* __restrict__ out, const cf* __restrict__ in, int64_t nblocks)
{
    using S = RowShape<K>;
    using T = RowTile<K, M>;
    constexpr int N = K * M;
    constexpr int PART = (M * L / 2 < M) ? (M * L / 2) : M;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int g = threadIdx.x / K, q = threadIdx.x - g * K;
    const int64_t blk = (int64_t)blockIdx.x * S::BPW + g;
    const bool valid = blk < nblocks;
    const int64_t base = (valid ? blk : 0) * N;
    cf* X = reinterpret_cast<cf*>(smem) + g * T::TS;

    FftTwiddles<K> twd;
    load_fft_twiddles<K>(twd, q, p.wK);
    // the twiddles of the last stage and the (uniform) filter taps are requested up front as well, see k_row_receive
    constexpr bool PRE_TW = (M <= 16), PRE_TAPS = (L * M <= 24);
    cf twv[PRE_TW ? M : 1], tapv[PRE_TAPS ? L * M : 1];
    if constexpr (PRE_TW) static_for<1, M>([&](auto mi) { constexpr int m = decltype(mi)::value; twv[m] = twT[m * K + q]; });
    if constexpr (PRE_TAPS) static_for<0, L * M>([&](auto ii) { constexpr int i = decltype(ii)::value; tapv[i] = p.taps[i]; });
    auto tap = [&](auto ii) { constexpr int i = decltype(ii)::value; if constexpr (PRE_TAPS) return tapv[i]; else return p.taps[i]; };
    const bool treal = p.taps_real != 0;
    cf v[M];
    if constexpr (TXMODE == 0) {
        // symbols [k][p], copied linearly (coalesced) into the tile; lane k then owns row k
        static_for<0, M>([&](auto ii) { constexpr int i = decltype(ii)::value; X[q + K * i] = ld_stream(in + base + q + K * i); });
        block_sync<K>();
        static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; v[m] = X[q * M + m]; });
    } else {
        // resource mapper fused into the load: lane k gathers its own subcarrier's symbols (for per-timeslot order the lanes
        // of one timeslot read consecutive input symbols)
        const cf* sym = in + (valid ? blk : 0) * (int64_t)tx.nin;
        static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; v[m] = tx_symbol(tx, sym, M, q, m); });
    }
    dft_inplace<M, false>(v);                                                                          // mod:109-110
    // gather form of filter + overlap-add: Y[j][m] = sum_i D[(j - i + L/2) mod K][m] taps[((i + L/2) % L) M + m]   mod:116-132
    constexpr float invN = 1.0f / (float)N;
    if constexpr (!(K == 64 && L == 2)) {
        static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; X[q * M + m] = v[m]; });    // own row: no hazard
        block_sync<K>();
    }
    auto filter = [&](auto real_tag) {                     // one copy per kind of taps, chosen by a uniform branch
        constexpr bool TREAL = decltype(real_tag)::value;
        if constexpr (K == 64 && L == 2) {
            // the block is the wavefront and the only foreign row is j + 1: fetch it with a DPP wave rotate, no LDS round trip
            static_for<0, M>([&](auto mi) {
                constexpr int m = decltype(mi)::value;
                const cf up = mk(dpp_wave_rol1(v[m].x), dpp_wave_rol1(v[m].y));          // D[(j + 1) mod K][m]   (i = 0)
                cf acc = mk(0.f, 0.f);
                if constexpr (m < PART) {
                    acc = tap_fma<TREAL>(tap(std::integral_constant<int, M + m>{}), up, acc);
                    acc = tap_fma<TREAL>(tap(std::integral_constant<int, m>{}), v[m], acc);                                     // D[j][m]              (i = 1)
                }
                v[m] = acc;
            });
        } else {
            static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; v[m] = mk(0.f, 0.f); });
            static_for<0, L>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                const cf* rb = X + wrap_k<K>(q - i + L / 2 + K) * M;
                static_for<0, PART>([&](auto mi) {
                    constexpr int m = decltype(mi)::value;
                    v[m] = tap_fma<TREAL>(tap(std::integral_constant<int, ((i + L / 2) % L) * M + m>{}), rb[m], v[m]);
                });
            });
        }
    };
    if (treal) filter(std::true_type{}); else filter(std::false_type{});
    block_sync<K>();                                      // neighbour rows read by everyone before they are overwritten
    constexpr bool REGPASS = (K == 64) && kRegFirstPass;
    static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; v[m] = scale(v[m], invN); });
    if constexpr (REGPASS) {
        wave_fft_first_pass<M, true>(X, q, twd, v);       // first pass of the inverse FFT from the registers
    } else {
        cf* xa = X + FftLayout<K>::slot(q) * M;
        static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; xa[m] = v[m]; });
        block_sync<K>();
    }
    lds_subcarrier_fft<K, M, true, REGPASS ? 1 : 0>(X, q, twd);                                         // inverse over j
    v[0] = X[q * M];
    static_for<1, M>([&](auto mi) {
        constexpr int m = decltype(mi)::value;
        cf w;
        if constexpr (PRE_TW) w = twv[m]; else w = twT[m * K + q];
        v[m] = cmulc(X[q * M + m], w);
    });
    dft_inplace<M, true>(v);                                                                           // mod:137-140
    if (valid) {
        if constexpr (TXMODE == 2) {
            static_for<0, M>([&](auto pi) { constexpr int pp = decltype(pi)::value; tx_store_sample(tx, blk, N, K * pp + q, v[pp]); });
            tx_store_preamble(tx, blk, q, K);
        } else {
            static_for<0, M>([&](auto pi) { constexpr int pp = decltype(pi)::value; st_stream(out, base + K * pp + q, v[pp]); });
        }
    }
}

#ifndef __HIPCC_RTC__
// PART selects which receive kernels a translation unit instantiates (compile time is dominated by the largest shape):
//   0  frequency-domain output and plain demodulation, equaliser none / vector
//   1  interference cancellation on the vector ALU (any constellation, phase compensation, complex IC kernel), equaliser none / vector
//   2  every mode with the equaliser estimated from the preamble inside the kernel (EQ_PREAMBLE), IC rounds on the vector ALU
//   4  interference cancellation with the rounds on the matrix cores (IcMfma: QPSK, real even IC kernel, no phase compensation),
//      equaliser none / vector / preamble  (3 = the modulators)
template <int K, int M, int L, int PART>
hipError_t launch_rx(const DevicePlan& p, const IcParams& ic, const EstPlan* est, const cf* twT, int mode, cf* out, const cf* in, const cf* f_eq,
                     int64_t nblocks, hipStream_t st)
{
    static_assert(row_lds_bytes<K, M>() <= 64 * 1024, "row-lane tile exceeds the default dynamic LDS limit");
    const dim3 grid((unsigned)((nblocks + RowShape<K>::BPW - 1) / RowShape<K>::BPW)), block(RowShape<K>::WG);
    static const EstPlan kNoEst = {};
    const EstPlan& e = est ? *est : kNoEst;
    const bool pre = (PART == 2) || (PART == 4 && est);                          // EQ_PREAMBLE: two more tile columns + the estimate behind the tiles
    size_t lds = pre ? row_lds_bytes<K, M + 2>() + EstTile<K>::bytes : row_lds_bytes<K, M>();
    if (PART == 4) lds += rowgeom::ic_mfma_edge_bytes(K);       // behind everything else: the wavefronts' edge rows of the IcMfma rounds
#define GFDM_RX(MODE_, EQ_, ICK_)                                                                                                       \
    do {                                                                                                                            \
        if (lds > 64 * 1024) {      /* only the largest shape with the estimate behind its tile */                                 \
            hipError_t err_ = hipFuncSetAttribute(reinterpret_cast<const void*>(k_row_receive<K, M, L, MODE_, EQ_, ICK_>),          \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
            if (err_ != hipSuccess) return err_;                                                                                    \
        }                                                                                                                           \
        hipLaunchKernelGGL((k_row_receive<K, M, L, MODE_, EQ_, ICK_>), grid, block, lds, st, p, ic, e, twT, out, in, f_eq, nblocks); \
    } while (0)
    const bool ic_rounds = (mode == RX_IC && ic.ic_iter > 0);
    if constexpr (PART == 4) {
        if constexpr (rowgeom::ic_mfma(K, M)) {
            if (!ic_rounds || !ic_mfma_applies(p, ic)) return hipErrorInvalidValue;
            if (est) GFDM_RX(RX_IC, EQ_PREAMBLE, ICK_MFMA);
            else if (f_eq) GFDM_RX(RX_IC, EQ_VECTOR, ICK_MFMA);
            else GFDM_RX(RX_IC, EQ_NONE, ICK_MFMA);
        } else {
            return hipErrorInvalidValue;
        }
    } else if constexpr (PART == 2) {
        if (!est) return hipErrorInvalidValue;
        if (mode == RX_FD) GFDM_RX(RX_FD, EQ_PREAMBLE, ICK_GENERAL);
        else if (!ic_rounds) GFDM_RX(RX_DEMOD, EQ_PREAMBLE, ICK_GENERAL);
        else if (p.ic_real_sym) GFDM_RX(RX_IC, EQ_PREAMBLE, ICK_REALSYM);
        else GFDM_RX(RX_IC, EQ_PREAMBLE, ICK_GENERAL);
    } else if constexpr (PART == 1) {
        if (est || !ic_rounds) return hipErrorInvalidValue;
        if (p.ic_real_sym) { if (f_eq) GFDM_RX(RX_IC, EQ_VECTOR, ICK_REALSYM); else GFDM_RX(RX_IC, EQ_NONE, ICK_REALSYM); }
        else { if (f_eq) GFDM_RX(RX_IC, EQ_VECTOR, ICK_GENERAL); else GFDM_RX(RX_IC, EQ_NONE, ICK_GENERAL); }
    } else {
        if (est || ic_rounds) return hipErrorInvalidValue;
        if (mode == RX_FD) { if (f_eq) GFDM_RX(RX_FD, EQ_VECTOR, ICK_GENERAL); else GFDM_RX(RX_FD, EQ_NONE, ICK_GENERAL); }
        else { if (f_eq) GFDM_RX(RX_DEMOD, EQ_VECTOR, ICK_GENERAL); else GFDM_RX(RX_DEMOD, EQ_NONE, ICK_GENERAL); }
    }
#undef GFDM_RX
    return hipGetLastError();
}

template <int K, int M, int L>
hipError_t launch_mod(const DevicePlan& p, const TxParams& tx, const cf* twT, cf* out, const cf* in, int64_t nblocks, hipStream_t st)
{
    constexpr size_t lds = row_lds_bytes<K, M>();
    const dim3 grid((unsigned)((nblocks + RowShape<K>::BPW - 1) / RowShape<K>::BPW)), block(RowShape<K>::WG);
    if (tx.mapped && tx.framed) hipLaunchKernelGGL((k_row_modulate<K, M, L, 2>), grid, block, lds, st, p, tx, twT, out, in, nblocks);
    else if (tx.mapped) hipLaunchKernelGGL((k_row_modulate<K, M, L, 1>), grid, block, lds, st, p, tx, twT, out, in, nblocks);
    else hipLaunchKernelGGL((k_row_modulate<K, M, L, 0>), grid, block, lds, st, p, tx, twT, out, in, nblocks);
    return hipGetLastError();
}

#endif  // !__HIPCC_RTC__

// =====================================================================================================================
// estimate_frame of preamble_channel_estimator_cc (lib/preamble_channel_estimator_cc.cc:284-295) in the row-lane layout: K lanes per
// received preamble.  Same pieces as the EQ_PREAMBLE path of k_row_receive; the frame estimate of row q (bins M q .. M q + M - 1)
// is produced in registers, staged through the tile and stored in linear order (coalesced).
template <int K, int M>
__global__ __launch_bounds__(RowShape<K>::WG) void k_row_estimate(EstPlan est, cf* __restrict__ out, const cf* __restrict__ in,
                                                                 int64_t nframes)
{
    using S = RowShape<K>;
    using T = RowTile<K, M>;
    constexpr int N = K * M;
    static_assert(N >= 3 * K, "the tile must hold the [K][2] preamble view plus the smoothed estimate");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int g = threadIdx.x / K, q = threadIdx.x - g * K;
    const int64_t blk = (int64_t)blockIdx.x * S::BPW + g;
    const bool valid = blk < nframes;
    cf* X = reinterpret_cast<cf*>(smem) + g * T::TS;
    cf* inter = reinterpret_cast<cf*>(smem + row_lds_bytes<K, M>()) + g * EstTile<K>::FS;
    cf* F = X + 2 * K;                                                             // smoothed estimate behind the [K][2] view

    const cf* pre = in + (valid ? blk : 0) * (int64_t)(2 * K);
    const cf p0 = ld_stream(pre + q), p1 = ld_stream(pre + K + q);
    const cf inv0 = est.inv0[q], inv1 = est.inv1[q];
    FftTwiddles<K> twd;
    load_fft_twiddles<K>(twd, q, est.wK);
    constexpr bool REGPASS = (K == 64) && kRegFirstPass;
    if constexpr (REGPASS) {
        const cf halves[2] = { p0, p1 };
        wave_fft_first_pass<2, false>(X, q, twd, halves);
    } else {
        cf* xa = X + FftLayout<K>::slot(q) * 2;
        xa[0] = p0;
        xa[1] = p1;
        block_sync<K>();
    }
    lds_subcarrier_fft<K, 2, false, REGPASS ? 1 : 0>(X, q, twd);                   // both halves           est:118-145
    const cf eq = cfma(X[2 * q], inv0, cmul(X[2 * q + 1], inv1));
    const int pos = est_active_pos(q, est), n_est = est.n_est;
    if (pos >= 0) {                                                                //                       est:147-175
        inter[4 + pos] = eq;
        if (pos == 0) { inter[0] = eq; inter[1] = eq; inter[2] = eq; inter[3] = eq; }
        if (pos == n_est - 1) { inter[n_est + 4] = eq; inter[n_est + 5] = eq; inter[n_est + 6] = eq; inter[n_est + 7] = eq; }
    }
    block_sync<K>();
    if (est.dc_free) {
        if (q == 0) {
            const cf lo = inter[4 + est.A / 2 - 1], hi = inter[4 + est.A / 2 + 1];
            inter[4 + est.A / 2] = mk(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y));
        }
        block_sync<K>();
    }
    if (q < n_est) {                                                               // 9-tap Gaussian         est:176-187
        cf acc = mk(0.f, 0.f);
        static_for<0, 9>([&](auto ti) {
            constexpr int t = decltype(ti)::value;
            const cf x = inter[q + t];
            acc.x += x.x * est.gauss[t];
            acc.y += x.y * est.gauss[t];
        });
        F[q] = acc;
    }
    block_sync<K>();
    cf lo, hi;                                                                     //                       est:238-273
    est_row_segment(F, 1, q, est, lo, hi);
    block_sync<K>();                                                               // every lane has its end points: the tile is free
    const cf dlt = mk(hi.x - lo.x, hi.y - lo.y);
    constexpr float step = 1.0f / (float)M;
    static_for<0, M>([&](auto mi) {
        constexpr int m = decltype(mi)::value;
        const float t = (float)m * step;
        X[q * M + m] = mk(lo.x + dlt.x * t, lo.y + dlt.y * t);
    });
    block_sync<K>();
    if (valid) {
        static_for<0, M>([&](auto ii) { constexpr int i = decltype(ii)::value; st_stream(out, blk * N + q + K * i, X[q + K * i]); });
    }
}

#ifndef __HIPCC_RTC__
template <int K, int M>
hipError_t launch_est(const EstPlan& e, cf* out, const cf* in, int64_t nframes, hipStream_t st)
{
    constexpr size_t lds = row_lds_bytes<K, M>() + EstTile<K>::bytes;
    const dim3 grid((unsigned)((nframes + RowShape<K>::BPW - 1) / RowShape<K>::BPW)), block(RowShape<K>::WG);
    if (lds > 64 * 1024) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(k_row_estimate<K, M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
    }
    hipLaunchKernelGGL((k_row_estimate<K, M>), grid, block, lds, st, e, out, in, nframes);
    return hipGetLastError();
}

#endif  // !__HIPCC_RTC__

}  // namespace GFDM_ROWLANE_NS (anonymous in the library build)
}  // namespace gfdm

// gfdm_rowlane_shape.hip is compiled once per (shape, part) -- see the Makefile -- so that the instantiations build in parallel.
#define GFDM_ROWLANE_RX_PART_I(K_, M_, L_, PART_)                                                                                   \
    namespace gfdm {                                                                                                                \
    hipError_t rowlane_rx##PART_##_##K_##_##M_##_##L_(const DevicePlan& p, const IcParams& ic, const EstPlan* est, const cf* twT,  \
                                                      int mode, cf* out, const cf* in, const cf* f_eq, int64_t nblocks,             \
                                                      hipStream_t s)                                                                \
    {                                                                                                                               \
        return launch_rx<K_, M_, L_, PART_>(p, ic, est, twT, mode, out, in, f_eq, nblocks, s);                                      \
    }                                                                                                                               \
    }
#define GFDM_ROWLANE_MOD_I(K_, M_, L_)                                                                                              \
    namespace gfdm {                                                                                                                \
    hipError_t rowlane_mod_##K_##_##M_##_##L_(const DevicePlan& p, const TxParams& tx, const cf* twT, cf* out, const cf* in,        \
                                              int64_t nblocks, hipStream_t s)                                                       \
    {                                                                                                                               \
        return launch_mod<K_, M_, L_>(p, tx, twT, out, in, nblocks, s);                                                             \
    }                                                                                                                               \
    hipError_t rowlane_est_##K_##_##M_##_##L_(const EstPlan& e, cf* out, const cf* in, int64_t nframes, hipStream_t s)              \
    {                                                                                                                               \
        return launch_est<K_, M_>(e, out, in, nframes, s);                                                                          \
    }                                                                                                                               \
    }
// argument macros (-DGFDM_SHAPE_K=..) must be expanded before they are pasted into the function names
#define GFDM_ROWLANE_RX_PART(K_, M_, L_, PART_) GFDM_ROWLANE_RX_PART_I(K_, M_, L_, PART_)
#define GFDM_ROWLANE_MOD(K_, M_, L_) GFDM_ROWLANE_MOD_I(K_, M_, L_)
