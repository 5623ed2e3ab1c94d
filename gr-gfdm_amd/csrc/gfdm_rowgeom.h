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

constexpr int wg(int K) { return K >= 128 ? K : GFDM_ROW_WG; }               // threads per workgroup
constexpr int bpw(int K) { return wg(K) / K; }                               // blocks per workgroup
constexpr int tile_stride(int K, int M) { return K * M + (bpw(K) > 1 ? 16 : 0); }   // complex elements between the tiles of a workgroup
constexpr size_t lds_bytes(int K, int M) { return (size_t)bpw(K) * (size_t)tile_stride(K, M) * 8 + 64; }
constexpr int est_stride(int K) { return K + 10; }                           // EQ_PREAMBLE: edge-extended estimate bins per block
constexpr size_t est_bytes(int K) { return (size_t)bpw(K) * (size_t)est_stride(K) * 8; }

}  // namespace rowgeom
}  // namespace gfdm
