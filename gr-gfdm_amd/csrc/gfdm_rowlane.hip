// Row-lane kernel family: shape table and dispatch.  The kernels live in gfdm_rowlane_impl.h and are instantiated by
// gfdm_rowlane_shape.hip, compiled once per shape and part (Makefile: ROW_SHAPES must list the same shapes as GFDM_ROW_SHAPES;
// the library is linked with --no-undefined, so a shape missing there fails the build).
#include "gfdm_plan.h"
#include "gfdm_tx.h"
#include "gfdm_rowgeom.h"

#define GFDM_ROW_SHAPES(X) \
    X(64, 9, 2)            \
    X(32, 5, 2)            \
    X(32, 9, 2)            \
    X(128, 15, 4)          \
    X(256, 31, 2)          \
    X(64, 5, 2)            \
    X(64, 15, 2)           \
    X(128, 9, 2)           \
    X(128, 15, 2)          \
    X(128, 21, 2)          \
    X(4, 16, 2)            \
    X(4, 8, 2)             \
    X(96, 25, 2)

namespace gfdm {

#define RX_DECL(NAME_)                                                                                                              \
    hipError_t NAME_(const DevicePlan& p, const IcParams& ic, const EstPlan* est, const cf* twT, int mode, cf* out, const cf* in,   \
                     const cf* f_eq, int64_t nblocks, hipStream_t s);
#define X(K_, M_, L_)                                                                                                               \
    RX_DECL(rowlane_rx0_##K_##_##M_##_##L_)                                                                                         \
    RX_DECL(rowlane_rx1_##K_##_##M_##_##L_)                                                                                         \
    RX_DECL(rowlane_rx2_##K_##_##M_##_##L_)                                                                                         \
    RX_DECL(rowlane_rx4_##K_##_##M_##_##L_)                                                                                         \
    hipError_t rowlane_mod_##K_##_##M_##_##L_(const DevicePlan& p, const TxParams& tx, const cf* twT, cf* out, const cf* in,        \
                                              int64_t nblocks, hipStream_t s);                                                      \
    hipError_t rowlane_est_##K_##_##M_##_##L_(const EstPlan& e, cf* out, const cf* in, int64_t nframes, hipStream_t s);
GFDM_ROW_SHAPES(X)
#undef X
#undef RX_DECL

bool rowlane_supports(int M, int K, int L)
{
#define X(K_, M_, L_) if (K == K_ && M == M_ && L == L_) return true;
    GFDM_ROW_SHAPES(X)
#undef X
    return false;
}

// the estimator does not depend on the overlap: the first shape with this (K, M)
bool rowlane_supports_estimate(int M, int K)
{
#define X(K_, M_, L_) if (K == K_ && M == M_) return true;
    GFDM_ROW_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_rowlane_estimate(const EstPlan& e, cf* out, const cf* in, int64_t nframes, hipStream_t s)
{
    if (nframes <= 0) return hipSuccess;
#define X(K_, M_, L_) if (e.K == K_ && e.M == M_) return rowlane_est_##K_##_##M_##_##L_(e, out, in, nframes, s);
    GFDM_ROW_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_rowlane_modulate(const DevicePlan& p, const TxParams& tx, const cf* twT, cf* out, const cf* in, int64_t nblocks,
                                   hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
#define X(K_, M_, L_) if (p.K == K_ && p.M == M_ && p.L == L_) return rowlane_mod_##K_##_##M_##_##L_(p, tx, twT, out, in, nblocks, s);
    GFDM_ROW_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_rowlane_receive(const DevicePlan& p, const IcParams& ic, const EstPlan* est, const cf* twT, int mode, cf* out, const cf* in,
                                  const cf* f_eq, int64_t nblocks, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    const bool ic_rounds = (mode == RX_IC && ic.ic_iter > 0);
    // which translation unit holds the kernel (launch_rx); 4: cancellation rounds on the matrix cores
    const int part = (ic_rounds && ic_mfma_applies(p, ic) && rowgeom::ic_mfma(p.K, p.M)) ? 4 : est ? 2 : ic_rounds ? 1 : 0;
#define X(K_, M_, L_)                                                                                              \
    if (p.K == K_ && p.M == M_ && p.L == L_) {                                                                    \
        if (part == 4) return rowlane_rx4_##K_##_##M_##_##L_(p, ic, est, twT, mode, out, in, f_eq, nblocks, s);   \
        if (part == 2) return rowlane_rx2_##K_##_##M_##_##L_(p, ic, est, twT, mode, out, in, f_eq, nblocks, s);   \
        if (part == 1) return rowlane_rx1_##K_##_##M_##_##L_(p, ic, est, twT, mode, out, in, f_eq, nblocks, s);   \
        return rowlane_rx0_##K_##_##M_##_##L_(p, ic, est, twT, mode, out, in, f_eq, nblocks, s);                  \
    }
    GFDM_ROW_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace gfdm
