// Row-lane kernels of ONE shape and ONE part (gfdm_rowlane_impl.h): compiled by the Makefile once per entry of ROW_SHAPES and
// per part with -DGFDM_SHAPE_K= -DGFDM_SHAPE_M= -DGFDM_SHAPE_L= -DGFDM_SHAPE_PART= (0, 1, 2, 4: receive kernels, 3: modulators).
#include "gfdm_rowlane_impl.h"

#if !defined(GFDM_SHAPE_K) || !defined(GFDM_SHAPE_M) || !defined(GFDM_SHAPE_L) || !defined(GFDM_SHAPE_PART)
#error "compile with -DGFDM_SHAPE_K=.. -DGFDM_SHAPE_M=.. -DGFDM_SHAPE_L=.. -DGFDM_SHAPE_PART=.."
#endif

#if GFDM_SHAPE_PART == 3
GFDM_ROWLANE_MOD(GFDM_SHAPE_K, GFDM_SHAPE_M, GFDM_SHAPE_L)
#else
GFDM_ROWLANE_RX_PART(GFDM_SHAPE_K, GFDM_SHAPE_M, GFDM_SHAPE_L, GFDM_SHAPE_PART)
#endif

#if defined(GFDM_STAMPS)      /* diagnostic build only (scratch/stamps.py): one setter per translation unit, each has its own g_stamp_buf */
#define GFDM_STAMP_SETTER_I(K_, M_, L_, P_) extern "C" int gfdm_debug_set_stamp_buffer_##K_##_##M_##_##L_##_p##P_(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(gfdm::g_stamp_buf), &p, sizeof(p)); }
#define GFDM_STAMP_SETTER(K_, M_, L_, P_) GFDM_STAMP_SETTER_I(K_, M_, L_, P_)
GFDM_STAMP_SETTER(GFDM_SHAPE_K, GFDM_SHAPE_M, GFDM_SHAPE_L, GFDM_SHAPE_PART)
#endif
