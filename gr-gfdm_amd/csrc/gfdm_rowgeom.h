// Launch geometry of the row-lane kernels as plain constexpr functions of the shape: ONE definition for the kernels
// (gfdm_rowlane_impl.h, compile-time shapes) and for the host code that launches run-time instantiated kernels (gfdm_jit.hip).
#pragma once
#ifndef __HIPCC_RTC__
#include <stddef.h>
#endif

#ifndef GFDM_ROW_WG
#define GFDM_ROW_WG 256
#endif

namespace gfdm {
namespace rowgeom {

// Subcarrier counts the family covers: powers of two 4 .. 1024 (radix-4 / wide passes, gfdm_rowlane_impl.h), and every other K <= 1024
// that splits into two or three factors the butterfly codelets hold in registers -- Dft<R> for R <= 16 where such a plan exists
// (96 = 6 x 16, 12, 48, 240; 200 = 2 x 10 x 10, 384, 600, 1000), else R <= 32 (34 = 2 x 17, 62 = 2 x 31, 51, 93, 589 = 19 x 31, a prime
// K <= 31 itself).  Two passes are preferred to three; two-pass plans with factors <= 16 stay below K = 256.
//   the LAST pass takes the largest divisor <= lim, the pass(es) in front the same rule on the quotient; R0 = 1: a single pass
constexpr bool pow2(int K) { return K > 0 && (K & (K - 1)) == 0; }
constexpr int ldiv(int K, int lim)                                           // the largest divisor of K that is <= lim
{
    int r = 1;
    for (int d = 2; d <= lim && d <= K; ++d)
        if (K % d == 0) r = d;
    return r;
}
constexpr bool plan2(int K, int lim) { return K / ldiv(K, lim) <= lim; }
constexpr bool plan3(int K, int lim) { return ldiv(K, lim) > 1 && ldiv(K / ldiv(K, lim), lim) > 1 && plan2(K / ldiv(K, lim), lim); }
// radix limit of K's plan: 16 if a two- or three-pass plan with factors <= 16 exists, else 32
constexpr int plan_lim(int K) { return ((K <= 256 && plan2(K, 16)) || plan3(K, 16)) ? 16 : 32; }
constexpr bool mixed(int K)                                                   // two passes
{
    return !pow2(K) && K >= 3 && K <= 1024 && ((K <= 256 && plan2(K, 16)) || (!plan3(K, 16) && plan2(K, 32)));
}
constexpr int mixed_r1(int K) { return ldiv(K, plan_lim(K)); }
constexpr int mixed_r0(int K) { return K / mixed_r1(K); }
constexpr bool mixed3(int K)                                                  // three passes
{
    return !pow2(K) && !mixed(K) && K >= 3 && K <= 1024 && (plan3(K, 16) || plan3(K, 32));
}
constexpr int mixed3_r2(int K) { return ldiv(K, plan_lim(K)); }
constexpr int mixed3_r1(int K) { return ldiv(K / mixed3_r2(K), plan_lim(K)); }
constexpr int mixed3_r0(int K) { return K / mixed3_r2(K) / mixed3_r1(K); }
constexpr bool supported(int K) { return (pow2(K) && K >= 4 && K <= 1024) || mixed(K) || mixed3(K); }

// threads per workgroup: whole blocks only (a block never straddles two workgroups)
constexpr int wg(int K) { return K >= 128 ? K : pow2(K) ? GFDM_ROW_WG : (GFDM_ROW_WG / K) * K; }
// blocks of K lanes that sit inside ONE wavefront are ordered by the wave's program order; others need the workgroup barrier
constexpr bool wave_local(int K) { return K <= 64 && 64 % K == 0; }
constexpr int bpw(int K) { return wg(K) / K; }                               // blocks per workgroup
constexpr int tile_stride(int K, int M) { return K * M + (bpw(K) > 1 ? 16 : 0); }   // complex elements between the tiles of a workgroup
constexpr size_t lds_bytes(int K, int M) { return (size_t)bpw(K) * (size_t)tile_stride(K, M) * 8 + 64; }
// Matrix-core form of the interference-cancellation rounds (IcMfma in gfdm_rowlane_impl.h): 64-row aligned wavefronts of whole 16-row
// groups and a timeslot row within the 16 x 16 tile of v_mfma_f32_16x16x32_f16.  Since round 4 the rounds keep everything in registers
// (lane-row transposes, DPP row shifts); only blocks that span several wavefronts pass each wavefront's two edge rows through LDS:
// [2 buffers][wavefront][first | last row][component][lane row] x 8 bytes behind the tiles.
constexpr bool ic_mfma(int K, int M) { return pow2(K) && K >= 16 && M >= 4 && M <= 16; }
// ... and where it is the default (measured, profiles/README.md)
constexpr bool ic_mfma_preferred(int K, int M) { return ic_mfma(K, M) && K >= 128; }
constexpr size_t ic_mfma_pad_bytes(int K) { return wave_local(K) ? 0 : (size_t)(K / 4) * 8; }      // padding of the rounds' tile layout (one element per 4 rows)
constexpr size_t ic_mfma_edge_bytes(int K) { return wave_local(K) ? 0 : (size_t)2 * (K / 64) * 2 * 2 * 4 * 8 + ic_mfma_pad_bytes(K); }
constexpr int est_stride(int K) { return K + 10; }                           // EQ_PREAMBLE: edge-extended estimate bins per block
constexpr size_t est_bytes(int K) { return (size_t)bpw(K) * (size_t)est_stride(K) * 8; }

}  // namespace rowgeom
}  // namespace gfdm
