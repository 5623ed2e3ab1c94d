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

// Subcarrier counts the family covers: powers of two 4 .. 1024 (radix-4 / wide passes, gfdm_rowlane_impl.h), any other
// K = R0 * R1 <= 256 with both factors <= 16 (two Stockham passes with the codelets Dft<R0>, Dft<R1>; R0 = 1: a single pass) ...
constexpr bool pow2(int K) { return K > 0 && (K & (K - 1)) == 0; }
constexpr int mixed_r1(int K)                                                // radix of the LAST pass: the largest divisor <= 16
{
    int r = 1;
    for (int d = 2; d <= 16 && d <= K; ++d)
        if (K % d == 0) r = d;
    return r;
}
constexpr int mixed_r0(int K) { return K / mixed_r1(K); }
constexpr bool mixed(int K) { return !pow2(K) && K >= 3 && K <= 256 && mixed_r0(K) <= 16; }
// ... and, where no two-factor plan exists, K = R0 * R1 * R2 <= 1024 with all three <= 16 (200 = 2 x 10 x 10, 320, 384 = 2 x 12 x 16, 600, 960,
// 1000 ...): R2 = the largest divisor <= 16 of K, then the two-factor plan of K / R2.  Three passes (lds_subcarrier_fft3).
constexpr bool mixed3(int K)
{
    return !pow2(K) && !mixed(K) && K >= 3 && K <= 1024 && mixed_r1(K) > 1 && mixed_r0(K / mixed_r1(K)) <= 16 && mixed_r1(K / mixed_r1(K)) > 1;
}
constexpr int mixed3_r2(int K) { return mixed_r1(K); }
constexpr int mixed3_r1(int K) { return mixed_r1(K / mixed_r1(K)); }
constexpr int mixed3_r0(int K) { return K / mixed_r1(K) / mixed3_r1(K); }
constexpr bool supported(int K) { return (pow2(K) && K >= 4 && K <= 1024) || mixed(K) || mixed3(K); }

// threads per workgroup: whole blocks only (a block never straddles two workgroups)
constexpr int wg(int K) { return K >= 128 ? K : pow2(K) ? GFDM_ROW_WG : (GFDM_ROW_WG / K) * K; }
// blocks of K lanes that sit inside ONE wavefront are ordered by the wave's program order; others need the workgroup barrier
constexpr bool wave_local(int K) { return K <= 64 && 64 % K == 0; }
constexpr int bpw(int K) { return wg(K) / K; }                               // blocks per workgroup
constexpr int tile_stride(int K, int M) { return K * M + (bpw(K) > 1 ? 16 : 0); }   // complex elements between the tiles of a workgroup
constexpr size_t lds_bytes(int K, int M) { return (size_t)bpw(K) * (size_t)tile_stride(K, M) * 8 + 64; }
constexpr int est_stride(int K) { return K + 10; }                           // EQ_PREAMBLE: edge-extended estimate bins per block
constexpr size_t est_bytes(int K) { return (size_t)bpw(K) * (size_t)est_stride(K) * 8; }

}  // namespace rowgeom
}  // namespace gfdm
