// TEST-ONLY: loop-back stand-ins for the GFDM kernel launchers of gr-gfdm_amd/csrc/gfdm_plan.h (the real ones are gfx950 device code in
// gfdm_rowlane*.hip / gfdm_generic.hip / gfdm_rader.hip / gfdm_stages.hip).  A "launch" enqueues a host function on the stream that copies in -> out
// through loopback_transform.h; it reads the handle's device tables on the way, so a table freed under a running launch is a sanitizer report.
// No GFDM arithmetic here: the point is the host plumbing around the launches.
#include <hip/hip_runtime.h>

#include "gfdm_plan.h"
#include "gfdm_tx.h"
#include "loopback_transform.h"

#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

namespace {

using gfdm::cf;
using loopback::c2;

inline c2 ld(const cf* p) { return c2{ p->x, p->y }; }
inline void st(cf* p, c2 v) { p->x = v.x; p->y = v.y; }

volatile float g_sink;        // keeps the table reads alive
std::mutex* sink_mutex()
{
    static std::mutex* m = new std::mutex();
    return m;
}
void sink(float v)
{
    std::lock_guard<std::mutex> lk(*sink_mutex());
    g_sink = g_sink + v;
}

float touch_plan(const gfdm::DevicePlan& p)
{
    float acc = 0.f;
    for (int i = 0; i < p.L * p.M; ++i) acc += p.taps[i].x;
    for (int i = 0; i < p.M; ++i) acc += p.ictaps[i].x + p.icg[i].x + p.wM[i].x;
    acc += p.wK[p.K - 1].x + p.wN[p.N - 1].y;
    return acc;
}

struct RxCall {
    gfdm::DevicePlan p;
    gfdm::IcParams ic;
    bool has_est;
    gfdm::EstPlan est;
    int mode;
    cf* out;
    const cf* in;
    const cf* f_eq;
    int64_t nb;
    float tag;
};

void rx_body(const RxCall& a)
{
    const int N = a.p.N, K = a.p.K;
    const int64_t in_stride = a.ic.io.in_stride ? a.ic.io.in_stride : N, in_off = a.ic.io.in_offset;
    const int nout = a.ic.io.nout > 0 ? a.ic.io.nout : N;
    float acc = touch_plan(a.p);
    for (int i = 0; i < a.ic.npoints; ++i) acc += a.ic.points[i].x;
    if (a.ic.active) for (int k = 0; k < K; ++k) acc += (float)a.ic.active[k];
    for (int i = 0; i < a.ic.n_active; ++i) acc += (float)a.ic.smap[i];
    if (a.ic.io.demap) for (int k = 0; k < K; ++k) acc += (float)a.ic.io.rank[k];
    if (a.has_est) acc += a.est.inv0[K - 1].x + a.est.inv1[K - 1].x + a.est.wK[K - 1].x + a.est.w2K[2 * K - 1].x;
    sink(acc);
    const int64_t pre_stride = a.has_est ? (a.est.pre_stride ? a.est.pre_stride : 2 * K) : 0;
    const int rounds = a.mode == gfdm::RX_IC ? a.ic.ic_iter : 0;
    for (int64_t b = 0; b < a.nb; ++b)
        for (int i = 0; i < nout; ++i) {
            const c2 s = ld(a.in + b * in_stride + in_off + (i % N));
            c2 e{ 0.f, 0.f };
            if (a.f_eq) e = a.has_est ? ld(a.f_eq + b * pre_stride + (i % (2 * K))) : ld(a.f_eq + b * (int64_t)N + (i % N));
            st(a.out + b * (int64_t)nout + i, loopback::rx_value(s, e, a.mode, rounds, a.tag));
        }
}

hipError_t rx_enqueue(const gfdm::DevicePlan& p, const gfdm::IcParams& ic, const gfdm::EstPlan* est, int mode, cf* out, const cf* in, const cf* f_eq,
                      int64_t nblocks, hipStream_t s, float tag)
{
    if (nblocks <= 0) return hipSuccess;
    RxCall c{ p, ic, est != nullptr, est ? *est : gfdm::EstPlan{}, mode, out, in, f_eq, nblocks, tag };
    return loopback::enqueue(s, [c] { rx_body(c); });
}

struct ModCall {
    gfdm::DevicePlan p;
    gfdm::TxParams tx;
    cf* out;
    const cf* in;
    int64_t nb;
    float tag;
};

void mod_body(const ModCall& a)
{
    const int N = a.p.N;
    float acc = touch_plan(a.p);
    if (a.tx.mapped) for (int k = 0; k < a.p.K; ++k) acc += (float)a.tx.rank[k];
    if (a.tx.framed) {
        for (int i = 0; i < a.tx.ramp; ++i) acc += a.tx.front[i].x + a.tx.back[i].x;
        for (int i = 0; i < a.tx.plen * a.tx.nports; ++i) acc += a.tx.preambles[i].x;
    }
    sink(acc);
    if (!a.tx.mapped && !a.tx.framed) {
        for (int64_t b = 0; b < a.nb; ++b)
            for (int i = 0; i < N; ++i) st(a.out + b * N + i, loopback::mod_value(ld(a.in + b * N + i), a.tag));
        return;
    }
    const int nin = a.tx.mapped ? a.tx.nin : N;
    if (!a.tx.framed) {                              // mapped symbols -> bare block
        for (int64_t b = 0; b < a.nb; ++b)
            for (int i = 0; i < N; ++i) {
                const c2 s = nin > 0 ? ld(a.in + b * nin + (i % nin)) : c2{ 0.f, 0.f };
                st(a.out + b * N + i, loopback::tx_value(s, 0, 0, a.tag));
            }
        return;
    }
    for (int port = 0; port < a.tx.nports; ++port)
        for (int64_t b = 0; b < a.nb; ++b)
            for (int j = 0; j < a.tx.F; ++j) {
                const c2 s = nin > 0 ? ld(a.in + b * nin + (j % nin)) : c2{ 0.f, 0.f };
                st(a.tx.outs[port] + b * (int64_t)a.tx.F + j, loopback::tx_value(s, port, 1, a.tag));
            }
}

hipError_t mod_enqueue(const gfdm::DevicePlan& p, const gfdm::TxParams& tx, cf* out, const cf* in, int64_t nblocks, hipStream_t s, float tag)
{
    if (nblocks <= 0) return hipSuccess;
    ModCall c{ p, tx, out, in, nblocks, tag };
    return loopback::enqueue(s, [c] { mod_body(c); });
}

int est_elems(const gfdm::EstPlan& e, int stage)
{
    switch (stage) {
    case gfdm::EST_RX_PREAMBLE: return 2 * e.K;
    case gfdm::EST_PREAMBLE_CHANNEL: return e.K;
    case gfdm::EST_FILTERED: return e.n_est;
    default: return e.M * e.K;
    }
}

hipError_t est_enqueue(const gfdm::EstPlan& e, int in_stage, int out_stage, cf* out, const cf* in, int64_t nframes, hipStream_t s, float tag)
{
    if (nframes <= 0) return hipSuccess;
    const gfdm::EstPlan ep = e;
    return loopback::enqueue(s, [=] {
        const int nin = est_elems(ep, in_stage), nout = est_elems(ep, out_stage);
        sink(ep.inv0[ep.K - 1].x + ep.w2K[2 * ep.K - 1].x);
        for (int64_t f = 0; f < nframes; ++f)
            for (int i = 0; i < nout; ++i) st(out + f * nout + i, loopback::est_value(ld(in + f * nin + (i % nin)), in_stage, out_stage, tag));
    });
}

// the shapes compiled into the library (ROW_SHAPES of gr-gfdm_amd/Makefile, as K_M_L)
const int kRowShapes[][3] = { { 64, 9, 2 }, { 32, 5, 2 }, { 32, 9, 2 }, { 128, 15, 4 }, { 256, 31, 2 }, { 64, 5, 2 }, { 64, 15, 2 },
                              { 128, 9, 2 }, { 128, 15, 2 }, { 128, 21, 2 }, { 4, 16, 2 }, { 4, 8, 2 }, { 96, 25, 2 } };

}  // namespace

namespace gfdm {

bool rowlane_supports(int M, int K, int L)
{
    for (const auto& s : kRowShapes)
        if (s[0] == K && s[1] == M && s[2] == L) return true;
    return false;
}
bool rowlane_supports_estimate(int M, int K)
{
    for (const auto& s : kRowShapes)
        if (s[0] == K && s[1] == M) return true;
    return false;
}
bool generic_supports(int, int, bool) { return true; }
bool estimator_supports(int K) { return K <= 4096; }
bool rader_supports(int M, int K) { return M == 127 && K == 16; }
void rader_host_table(int M, std::vector<cf>& tab)
{
    tab.clear();
    if (M == 127) tab.assign(126, make_float2(1.f, 0.f));
}

hipError_t launch_rowlane_receive(const DevicePlan& p, const IcParams& ic, const EstPlan* est, const cf* twT, int mode, cf* out, const cf* in, const cf* f_eq,
                                  int64_t nblocks, hipStream_t s)
{
    sink(twT[(size_t)p.M * p.K - 1].x);
    return rx_enqueue(p, ic, est, mode, out, in, f_eq, nblocks, s, loopback::kTagRowlane);
}
hipError_t launch_generic_receive(const DevicePlan& p, const IcParams& ic, const EstPlan* est, int mode, cf* out, const cf* in, const cf* f_eq, int64_t nblocks,
                                  hipStream_t s)
{
    return rx_enqueue(p, ic, est, mode, out, in, f_eq, nblocks, s, loopback::kTagGeneric);
}
hipError_t launch_rowlane_modulate(const DevicePlan& p, const TxParams& tx, const cf* twT, cf* out, const cf* in, int64_t nblocks, hipStream_t s)
{
    sink(twT[(size_t)p.M * p.K - 1].x);
    return mod_enqueue(p, tx, out, in, nblocks, s, loopback::kTagRowlane);
}
hipError_t launch_generic_modulate(const DevicePlan& p, const TxParams& tx, cf* out, const cf* in, int64_t nblocks, hipStream_t s)
{
    return mod_enqueue(p, tx, out, in, nblocks, s, loopback::kTagGeneric);
}
hipError_t launch_add_frame(const DevicePlan& p, const TxParams& tx, const cf* in, int64_t nblocks, hipStream_t s)
{
    return mod_enqueue(p, tx, nullptr, in, nblocks, s, loopback::kTagGeneric);
}
hipError_t launch_generic_to_td(const DevicePlan& p, cf* out, const cf* in, int64_t nblocks, hipStream_t s)
{
    const DevicePlan pp = p;
    return loopback::enqueue(s, [=] {
        sink(touch_plan(pp));
        for (int64_t i = 0; i < nblocks * pp.N; ++i) st(out + i, loopback::td_value(ld(in + i)));
    });
}
hipError_t launch_generic_cancel(const DevicePlan& p, cf* out, const cf* td, const cf* fd, int64_t nblocks, hipStream_t s)
{
    const DevicePlan pp = p;
    return loopback::enqueue(s, [=] {
        sink(touch_plan(pp));
        for (int64_t i = 0; i < nblocks * pp.N; ++i) st(out + i, loopback::cancel_value(ld(td + i), ld(fd + i)));
    });
}
hipError_t launch_estimate(const EstPlan& e, int in_stage, int out_stage, cf* out, const cf* in, int64_t nframes, hipStream_t s)
{
    return est_enqueue(e, in_stage, out_stage, out, in, nframes, s, loopback::kTagGeneric);
}
hipError_t launch_rowlane_estimate(const EstPlan& e, cf* out, const cf* in, int64_t nframes, hipStream_t s)
{
    return est_enqueue(e, EST_RX_PREAMBLE, EST_FRAME, out, in, nframes, s, loopback::kTagRowlane);
}
hipError_t launch_estimate_snr(const EstPlan& e, float* snr, float* cnrs, const cf* in, int64_t nframes, hipStream_t s)
{
    const EstPlan ep = e;
    return loopback::enqueue(s, [=] {
        for (int64_t f = 0; f < nframes; ++f) {
            snr[f] = in[f * 2 * ep.K].x;
            for (int a = 0; a < ep.A; ++a) cnrs[f * ep.A + a] = in[f * 2 * ep.K + (a % (2 * ep.K))].y;
        }
    });
}
hipError_t launch_prepare_for_zf(cf* out, const cf* in, int64_t n, hipStream_t s)
{
    return loopback::enqueue(s, [=] { for (int64_t i = 0; i < n; ++i) st(out + i, loopback::td_value(ld(in + i))); });
}

}  // namespace gfdm

// kernels of a module the loop-back hiprtc "compiled" (gfdm_jit.hip launches them through hipModuleLaunchKernel with an argument array)
namespace loopback {

hipError_t module_launch(const std::string& name, void** args, hipStream_t s)
{
    using namespace gfdm;
    int t[6] = {};
    const size_t lt = name.find('<');
    if (lt == std::string::npos) return hipErrorInvalidValue;
    const int nt = sscanf(name.c_str() + lt, "<%d,%d,%d,%d,%d,%d>", &t[0], &t[1], &t[2], &t[3], &t[4], &t[5]);
    if (name.find("k_row_receive") != std::string::npos && nt == 6) {
        const DevicePlan& p = *static_cast<const DevicePlan*>(args[0]);
        const IcParams& ic = *static_cast<const IcParams*>(args[1]);
        const EstPlan& est = *static_cast<const EstPlan*>(args[2]);
        const cf* twT = *static_cast<const cf* const*>(args[3]);
        cf* out = *static_cast<cf* const*>(args[4]);
        const cf* in = *static_cast<const cf* const*>(args[5]);
        const cf* f_eq = *static_cast<const cf* const*>(args[6]);
        const int64_t nb = *static_cast<const int64_t*>(args[7]);
        if (p.K != t[0] || p.M != t[1] || p.L != t[2]) return hipErrorInvalidValue;             // the handle launched another shape's kernel
        if ((t[4] != EQ_NONE) != (f_eq != nullptr)) return hipErrorInvalidValue;                   // equalised kernel without a vector or the reverse
        (void)twT[(size_t)p.M * p.K - 1].x;
        return rx_enqueue(p, ic, t[4] == EQ_PREAMBLE ? &est : nullptr, t[3], out, in, f_eq, nb, s, kTagJit);
    }
    if (name.find("k_row_modulate") != std::string::npos && nt == 4) {
        const DevicePlan& p = *static_cast<const DevicePlan*>(args[0]);
        const TxParams& tx = *static_cast<const TxParams*>(args[1]);
        cf* out = *static_cast<cf* const*>(args[3]);
        const cf* in = *static_cast<const cf* const*>(args[4]);
        const int64_t nb = *static_cast<const int64_t*>(args[5]);
        if (p.K != t[0] || p.M != t[1] || p.L != t[2]) return hipErrorInvalidValue;
        return mod_enqueue(p, tx, out, in, nb, s, kTagJit);
    }
    if (name.find("k_row_estimate") != std::string::npos && nt == 2) {
        const EstPlan& e = *static_cast<const EstPlan*>(args[0]);
        cf* out = *static_cast<cf* const*>(args[1]);
        const cf* in = *static_cast<const cf* const*>(args[2]);
        const int64_t nf = *static_cast<const int64_t*>(args[3]);
        return est_enqueue(e, EST_RX_PREAMBLE, EST_FRAME, out, in, nf, s, kTagJit);
    }
    return hipErrorNotFound;
}

}  // namespace loopback
