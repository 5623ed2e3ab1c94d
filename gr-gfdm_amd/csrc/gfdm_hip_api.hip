// C-ABI shim (include/gfdm_hip.h): handle management, table construction, host staging and
// dispatch to the HIP kernel families.  No CPU compute path exists here by design: if the HIP
// device or a kernel launch is unavailable the call fails with an error code.
#include "../../include/gfdm_hip.h"
#include "gfdm_plan.h"
#include "gfdm_rowgeom.h"
#include "gfdm_tx.h"
#include "gfdm_hostpipe.h"

#include <cfloat>
#include <cmath>
#include <complex>
#include <memory>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <vector>

using gfdm::cf;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg)
{
    g_last_error = msg;
    return code;
}

int fail_hip(hipError_t e, const char* what)
{
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();          // reported here: do not leave it behind as the thread's sticky error
    return GFDM_HIP_EHIP;
}

#define HIP_TRY(expr)                                            \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return fail_hip(_e, #expr);        \
    } while (0)

}  // namespace

int gfdm::api_fail(int code, const std::string& msg) { return fail(code, msg); }
int gfdm::api_fail_hip(hipError_t e, const char* what) { return fail_hip(e, what); }

namespace {

// test hook (gfdm_hip_force_generic_family_for_testing): handles created while it is set use the generic kernel family
std::atomic<int> g_force_generic{ 0 };
// gfdm_hip_set_jit: run-time instantiation (hiprtc) of the row-lane kernels for shapes outside the compiled list
//   0 off, 1 compile inside the constructor, 2 compile on a background thread (the handle starts on the generic family and switches over),
//   3 (default) = 1 when the code objects are in the disk cache or the shape compiles quickly (timeslots <= 16), else 2
std::atomic<int> g_jit{ 3 };
// gfdm_hip_set_ic_matrix_cores: 0 = handles created meanwhile run every cancellation round on the vector ALU, 1 (default) = matrix cores where
// they are the faster form (gfdm_rowgeom.h ic_mfma_preferred), 2 = matrix cores wherever the form applies
std::atomic<int> g_ic_mfma{ 1 };
std::atomic<int> g_dft_mfma{ 1 };

struct Plan {
    int device = 0;
    gfdm::DevicePlan dp{};
    std::vector<cf> h_taps, h_ictaps;
    cf* d_tables = nullptr;          // one allocation: taps | ictaps | wM | wK | wN
    hipStream_t stream = nullptr;    // private stream for the *_host entry points
    gfdm::HostPipe pipe;             // the host-buffer batch path: pinned staging sets, completion ticket (gfdm_hostpipe.h)
    std::string kernel_name;
    const cf* d_twT = nullptr;       // [M][K] twiddles W_N^{q m}, transposed so that lane q reads them coalesced (row-lane family)
    int family = gfdm::FAMILY_GENERIC;
    gfdm::JitCache jit;              // FAMILY_ROWLANE_JIT: this handle's pointers to the loaded kernel parts (no lock on the launch path)
    // run-time instantiation in the background (gfdm_hip_set_jit modes 2 / 3): 0 compiling, 1 ready, -1 failed.  While it is 0 the handle
    // runs on the generic family; the first call that sees 1 switches the handle over (a handle is used by one thread at a time)
    std::shared_ptr<std::atomic<int>> jit_pending;
    // the preamble-equalised receive kernels of a run-time instantiated shape, built in the background after set_channel_estimator when they
    // are neither cached nor quick to compile: 0 compiling (estimated calls run on the generic family meanwhile), 1 ready, -1 failed (they stay there)
    std::shared_ptr<std::atomic<int>> jit_pre_pending;

    // a *_host call cuts its batch into chunks, one launch each: the family (and the preamble-equalised kernels' availability) it starts
    // with is the one all of its chunks run on, so that a background instantiation finishing meanwhile cannot change the rounding inside
    // one call's output (FamilyPin below); -1 = not pinned
    int pinned_family = -1, pinned_pre = -1;

    // the family to launch with right now
    int current_family()
    {
        if (pinned_family >= 0) return pinned_family;
        if (jit_pending) {
            const int st = jit_pending->load(std::memory_order_acquire);
            if (st == 1) { family = gfdm::FAMILY_ROWLANE_JIT; kernel_name = "rowlane_jit"; }
            if (st != 0) jit_pending.reset();
        }
        return family;
    }

    ~Plan()
    {
        int prev = 0;
        bool restore = (hipGetDevice(&prev) == hipSuccess);
        (void)hipSetDevice(device);
        pipe.release();
        if (d_tables) (void)hipFree(d_tables);
        if (stream) (void)hipStreamDestroy(stream);
        if (restore) (void)hipSetDevice(prev);
    }
};

// RAII: make the handle's device current for the duration of a call.
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

int ilog2_exact(int v)
{
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

void unit_roots(std::vector<cf>& dst, int n)
{
    const double two_pi = 6.283185307179586476925286766559;
    for (int i = 0; i < n; ++i) {
        const double a = -two_pi * (double)i / (double)n;
        dst.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
    }
}

// Build the plan: normalise taps exactly as the reference constructors do, derive IC taps and twiddle tables.
int plan_create(Plan& pl, int M, int K, int L, const float* taps, int ntaps, int device, bool receiver, unsigned jit_parts = 0)
{
    if (M < 1 || K < 1 || L < 1 || taps == nullptr) return fail(GFDM_HIP_EINVAL, "timeslots, subcarriers, overlap must be >= 1 and taps non-NULL");
    if (ntaps != M * L) {
        char buf[256];
        snprintf(buf, sizeof(buf), "number of frequency taps(%d) MUST be equal to n_timeslots(%d) * overlap(%d) = %d!", ntaps, M, L, M * L);
        return fail(GFDM_HIP_EINVAL_TAPS, buf);
    }
    if (receiver && L < 2) return fail(GFDM_HIP_EINVAL_OVERLAP, "overlap MUST be greater or equal 2");
    if ((int64_t)M * K > (1 << 24)) return fail(GFDM_HIP_EUNSUPPORTED, "block too large");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GFDM_HIP_ENODEV, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(GFDM_HIP_ENODEV, "HIP device ordinal out of range");
    const int N = M * K;
    // (whether the block fits the kernels' LDS tiles is checked once the kernel family is known, below)

    pl.device = device;
    // energy |sum t conj(t)|, factor formed in double and cast (lib/modulator_kernel_cc.cc:75-85)
    double energy = 0.0;
    for (int i = 0; i < ntaps; ++i) energy += (double)taps[2 * i] * taps[2 * i] + (double)taps[2 * i + 1] * taps[2 * i + 1];
    if (!(energy > 0.0)) return fail(GFDM_HIP_EINVAL, "filter taps have zero energy");
    const float scale = (float)(1.0 / std::sqrt(std::fabs(energy) / M));
    pl.h_taps.resize(ntaps);
    for (int i = 0; i < ntaps; ++i) pl.h_taps[i] = make_float2(taps[2 * i] * scale, taps[2 * i + 1] * scale);
    // real taps (imaginary parts below anything float32 arithmetic can see next to the real parts: an RRC response computed in double
    // carries ~1e-17): the kernels then take two multiply-adds per tap product instead of four.  filter_taps() keeps returning h_taps.
    float tmax = 0.f, imax = 0.f;
    for (const cf& t : pl.h_taps) { tmax = std::fmax(tmax, std::fmax(std::fabs(t.x), std::fabs(t.y))); imax = std::fmax(imax, std::fabs(t.y)); }
    const bool taps_real = imax <= 1e-12f * tmax;
    pl.h_ictaps.assign(M, make_float2(0.f, 0.f));
    if (L >= 2)                                                    // lib/receiver_kernel_cc.cc:56-63
        for (int m = 0; m < M; ++m) {
            const cf a = pl.h_taps[m], b = pl.h_taps[M * (L - 1) + m];
            pl.h_ictaps[m] = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
        }

    std::vector<cf> tables;
    tables.reserve((size_t)ntaps + 4 * M + K + 2 * (size_t)N);
    for (const cf& t : pl.h_taps) tables.push_back(taps_real ? make_float2(t.x, 0.f) : t);
    tables.insert(tables.end(), pl.h_ictaps.begin(), pl.h_ictaps.end());
    for (int m = 0; m < M; ++m) tables.push_back(make_float2(pl.h_ictaps[m].x / (float)M, pl.h_ictaps[m].y / (float)M));
    // g = IDFT_M(ic)/M in double: one IC round is d_new = d0 - g (*) (dec_{k-1} + dec_{k+1})  (gfdm_rowlane_impl.h)
    bool ic_real_sym = true;
    std::vector<float> g_real(M);
    {
        const double two_pi = 6.283185307179586476925286766559;
        std::vector<double> gr(M), gi(M);
        double gmax = 0.0;
        for (int r = 0; r < M; ++r) {
            double sr = 0.0, si = 0.0;
            for (int m = 0; m < M; ++m) {
                const double a = two_pi * (double)((r * m) % M) / (double)M;
                sr += pl.h_ictaps[m].x * std::cos(a) - pl.h_ictaps[m].y * std::sin(a);
                si += pl.h_ictaps[m].x * std::sin(a) + pl.h_ictaps[m].y * std::cos(a);
            }
            gr[r] = sr / M; gi[r] = si / M;
            gmax = std::fmax(gmax, std::hypot(gr[r], gi[r]));
        }
        for (int r = 0; r < M; ++r) {
            if (std::fabs(gi[r]) > 1e-7 * gmax || std::fabs(gr[r] - gr[(M - r) % M]) > 1e-7 * gmax) ic_real_sym = false;
            tables.push_back(make_float2((float)gr[r], (float)gi[r]));
            g_real[r] = (float)gr[r];
        }
    }
    unit_roots(tables, M);
    unit_roots(tables, K);
    unit_roots(tables, N);
    const size_t twT_off = tables.size();
    {
        const double two_pi = 6.283185307179586476925286766559;
        for (int m = 0; m < M; ++m)
            for (int q = 0; q < K; ++q) {
                const double a = -two_pi * (double)(((int64_t)q * m) % N) / (double)N;
                tables.push_back(make_float2((float)std::cos(a), (float)std::sin(a)));
            }
    }

    // Matrix-core form of the cancellation rounds (IcMfma, gfdm_rowlane_impl.h): the A operand of v_mfma_f32_16x16x32_f16, lane l holds
    // A[row l & 15][contraction entry 8 (l >> 4) + j], j < 8.  Row p = output timeslot.  Entry 8 cr + j stands for input timeslot r = 4 cr + (j & 3):
    // the first operand holds the high f16 term of a[p][r] in j < 4 (zero in j >= 4), the second its residual term in j < 4 and the next residual term in j >= 4 --
    // so that the B operand of lane row cr is made of the decisions of exactly the four timeslots the C / D operand of that lane row holds --
    // with a[p][r] = -s g[(p - r) mod M] 2^e (s = 1/sqrt 2: the QPSK amplitude; e puts the largest entry near 2^8 so that the residual terms
    // stay normal f16 numbers).  The decisions enter as +-2^-e, exact in f16.
    size_t icA_off = 0;                     // (behind every other table: the pointers below are offsets into `tables`)
    unsigned ic_sig = 0;
    const int mx_mode = g_ic_mfma.load();
    if (receiver && ic_real_sym && (mx_mode == 2 ? gfdm::rowgeom::ic_mfma(K, M) : mx_mode == 1 && gfdm::rowgeom::ic_mfma_preferred(K, M))) {
        const double s = (double)0.70710678118654752f;
        double amax = 0.0;
        for (int r = 0; r < M; ++r) amax = std::fmax(amax, std::fabs(s * (double)g_real[r]));
        int e = 13;
        if (amax > 0.0) e = 8 - (int)std::ceil(std::log2(amax));
        e = e < -4 ? -4 : e > 13 ? 13 : e;
        const double c = std::ldexp(1.0, e);
        if (tables.size() & 1) tables.push_back(make_float2(0.f, 0.f));        // 16-byte alignment of the operand table
        icA_off = tables.size();
        for (int op = 0; op < 2; ++op)                             // operand 0: [high | residual], operand 1: [second residual | 0]
            for (int lane = 0; lane < 64; ++lane) {
                _Float16 h[8];
                for (int j = 0; j < 8; ++j) {
                    const int pr = lane & 15, r = 4 * (lane >> 4) + (j & 3);
                    double a = 0.0;
                    if (pr < M && r < M) a = -s * (double)g_real[((pr - r) % M + M) % M] * c;
                    const _Float16 hi = (_Float16)a;
                    const _Float16 mid = (_Float16)(a - (double)hi);
                    const _Float16 lo = (_Float16)(a - (double)hi - (double)mid);
                    h[j] = (op == 0) ? ((j < 4) ? hi : (_Float16)0.0) : ((j < 4) ? mid : lo);
                }
                cf packed[2];
                static_assert(sizeof packed == sizeof h, "8 f16 = 2 complex floats");
                memcpy(packed, h, sizeof packed);
                tables.insert(tables.end(), packed, packed + 2);
            }
        const _Float16 sig = (_Float16)std::ldexp(1.0, -e);
        unsigned short bits;
        memcpy(&bits, &sig, sizeof bits);
        ic_sig = bits;
    }

    // Matrix-core form of the generic family's timeslot transforms (mx_dft, gfdm_generic.hip): the paired DFT
    //   Q1 = C a.x, Q2 = C a.y, Q3 = S b.x, Q4 = S b.y   with a = v_p + v_{M-p}, b = v_p - v_{M-p}, C[m][p] = Re W_M^{m p}, S[m][p] = Im W_M^{m p}
    // as products of the CONSTANT matrices C, S (rows: outputs m <= M/2, columns: p = 0 (C = 1: the sample v_0 itself), the pairs p = 1..(M-1)/2,
    // the middle sample of an even M) with the block's samples.  A operand of v_mfma_f32_16x16x4_f32: lane l holds A[row l & 15][k = l >> 4].
    size_t dftA_off = 0;
    int dft_mt = 0, dft_ks = 0;
    if (g_dft_mfma.load() && M >= gfdm::MX_DFT_MIN_M) {
        const double two_pi = 6.283185307179586476925286766559;
        const int H = M / 2 + 1, HP = (M - 1) / 2, KD = HP + 1 + ((M & 1) == 0 ? 1 : 0);
        dft_mt = (H + 15) / 16;
        dft_ks = (KD + 3) / 4;
        if (tables.size() & 1) tables.push_back(make_float2(0.f, 0.f));
        dftA_off = tables.size();
        std::vector<float> A((size_t)dft_mt * dft_ks * 128, 0.f);
        for (int mt = 0; mt < dft_mt; ++mt)
            for (int ks = 0; ks < dft_ks; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int m = 16 * mt + (lane & 15), kk = 4 * ks + (lane >> 4);
                    if (m >= H || kk >= KD) continue;
                    const int pidx = (kk <= HP) ? kk : M / 2;
                    const double a = -two_pi * (double)(((int64_t)m * pidx) % M) / (double)M;
                    float* dst = A.data() + ((size_t)(mt * dft_ks + ks) * 2) * 64 + lane;
                    dst[0] = (float)std::cos(a);
                    dst[64] = (kk >= 1 && kk <= HP) ? (float)std::sin(a) : 0.f;
                }
        const cf* packed = reinterpret_cast<const cf*>(A.data());
        tables.insert(tables.end(), packed, packed + A.size() / 2);
    }

    // Rader transforms for prime timeslot counts above the register codelets (gfdm_rader.hip): the kernel spectrum, default mode only
    size_t rader_off = 0;
    if (g_dft_mfma.load() == 1 && !g_force_generic.load() && gfdm::rader_supports(M, K)) {
        std::vector<cf> rtab;
        gfdm::rader_host_table(M, rtab);
        if (!rtab.empty()) {
            rader_off = tables.size();
            tables.insert(tables.end(), rtab.begin(), rtab.end());
        }
    }

    DeviceGuard guard(device);
    if (!guard.ok) return fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    HIP_TRY(hipMalloc(&pl.d_tables, tables.size() * sizeof(cf)));
    HIP_TRY(hipMemcpy(pl.d_tables, tables.data(), tables.size() * sizeof(cf), hipMemcpyHostToDevice));
    HIP_TRY(hipStreamCreateWithFlags(&pl.stream, hipStreamNonBlocking));

    gfdm::DevicePlan& dp = pl.dp;
    dp.M = M; dp.K = K; dp.L = L; dp.N = N;
    dp.log2K = ilog2_exact(K);
    dp.part_len = (M * L / 2 < M) ? (M * L / 2) : M;
    dp.taps = pl.d_tables;
    dp.taps_real = taps_real ? 1 : 0;
    dp.ictaps = dp.taps + ntaps;
    dp.ictaps_m = dp.ictaps + M;
    dp.icg = dp.ictaps_m + M;
    dp.ic_real_sym = ic_real_sym ? 1 : 0;
    dp.icA = ic_sig ? static_cast<const void*>(pl.d_tables + icA_off) : nullptr;
    dp.ic_sig = ic_sig;
    dp.wM = dp.icg + M;
    dp.wK = dp.wM + M;
    dp.wN = dp.wK + K;
    pl.d_twT = pl.d_tables + twT_off;
    dp.dftA = dft_mt ? reinterpret_cast<const float*>(pl.d_tables + dftA_off) : nullptr;
    dp.dft_mt = dft_mt;
    dp.dft_ks = dft_ks;
    dp.dft_always = g_dft_mfma.load() == 2 ? 1 : 0;
    dp.raderB = rader_off ? pl.d_tables + rader_off : nullptr;
    // kernel family: row-lane where the shape is instantiated, else the generic LDS family.  Only the explicit test hook
    // gfdm_hip_force_generic_family_for_testing changes that; no environment variable does.
    pl.family = gfdm::FAMILY_GENERIC;
    if (!g_force_generic.load()) {
        if (gfdm::rowlane_supports(M, K, L)) {
            pl.family = gfdm::FAMILY_ROWLANE;
        } else if (g_jit.load() && gfdm::jit_eligible(M, K, L)) {
            // not in the compiled list: instantiate the row-lane kernels for this shape (seconds to a minute per part, once per shape and
            // machine -- the code objects are cached on disk; gfdm_hip_precompile fills that cache ahead of time).  Any failure (no hiprtc,
            // no compiler, ...) leaves the handle on the generic HIP family; the reason stays readable from gfdm_hip_last_error().
            // Only the parts this kind of handle launches are prepared (a modulator never compiles receiver kernels).
            std::string why;
            DeviceGuard guard(device);
            if (jit_parts == 0) jit_parts = receiver ? (1u << gfdm::JIT_PART_RX) : (1u << gfdm::JIT_PART_MOD);
            int mode = g_jit.load();
            if (mode == 3) {
                bool cached = true;
                for (int part = 0; part < gfdm::JIT_NUM_PARTS; ++part)
                    if (((jit_parts >> part) & 1u) && !gfdm::jit_cached(M, K, L, part)) cached = false;
                mode = (cached || M <= 16) ? 1 : 2;
            }
            // in the background only if the generic family can serve the shape meanwhile (two tiles of the block in LDS or global scratch)
            if (mode == 2 && gfdm::generic_supports(M, K, false)) {
                // a receiver may get a channel estimator attached later: its preamble-equalised kernels are built in the same go, so that
                // the switch-over never leaves a compile for the first estimated call
                const unsigned parts = jit_parts | (receiver ? (1u << gfdm::JIT_PART_RX_PREAMBLE) : 0u);
                pl.jit_pending = std::make_shared<std::atomic<int>>(0);
                gfdm::jit_prepare_async(M, K, L, parts, device, pl.jit_pending);
            } else if (gfdm::jit_prepare(M, K, L, jit_parts, why)) {
                pl.family = gfdm::FAMILY_ROWLANE_JIT;
            } else {
                g_last_error = "run-time instantiation of the row-lane kernels failed, using the generic family: " + why;
            }
        }
    }
    pl.kernel_name = pl.family == gfdm::FAMILY_ROWLANE ? "rowlane" : pl.family == gfdm::FAMILY_ROWLANE_JIT ? "rowlane_jit" :
                     dp.raderB ? "generic_rader" : "generic_lds";       // generic_rader: plain blocks on the Rader kernels (gfdm_rader.hip), the rest generic
    // the generic family holds two tiles of the block in LDS; handles on the row-lane families only use it for the stand-alone
    // transform_subcarriers_to_td / cancel_sc_interference entry points (one tile)
    if (!gfdm::generic_supports(M, K, pl.family != gfdm::FAMILY_GENERIC)) return fail(GFDM_HIP_EUNSUPPORTED, "root tables (2 * timeslots + subcarriers values) do not fit LDS");
    return GFDM_HIP_OK;
}

int status_of(hipError_t e) { return e == hipSuccess ? GFDM_HIP_OK : fail_hip(e, "kernel launch"); }

// resolves the handle's kernel family once, for the duration of one *_host call (Plan::pinned_family).  A handle is used by ONE thread at a time
// (include/gfdm_hip.h, "Threading"), so the pin is plain per-handle state; a pin that is already in place -- a *_host entry point reached from
// inside another one's launch callback -- is inherited and left to its owner, not reset (round-5 advisor).
struct FamilyPin {
    Plan& pl;
    bool owner;
    explicit FamilyPin(Plan& p) : pl(p), owner(p.pinned_family < 0)
    {
        if (!owner) return;
        const int f = pl.current_family();
        int pre = 1;
        if (pl.jit_pre_pending) {
            if (pl.jit_pre_pending->load(std::memory_order_acquire) == 1) pl.jit_pre_pending.reset(); else pre = 0;
        }
        pl.pinned_family = f;
        pl.pinned_pre = pre;
    }
    ~FamilyPin() { if (owner) pl.pinned_family = pl.pinned_pre = -1; }
    FamilyPin(const FamilyPin&) = delete;
    FamilyPin& operator=(const FamilyPin&) = delete;
};

// Host-pointer path of the block entry points (gfdm_hostpipe.h): operands the GPU can address are used in place, the others bounce through
// pinned staging sets in chunks, the kernels run across the link.  `launch(out, in0, in1, nb, stream)` enqueues the kernels of nb blocks and
// returns a status.  Sizes are complex samples per block (frames in / demapped symbols out may differ from the block size); in1 may be read at
// a stride (the preambles of the self-estimating receivers).
template <typename Launch>
int run_host_sized(Plan& pl, float* out, size_t out_pb, const float* in0, size_t in0_pb, const float* in1, size_t in1_stride, size_t in1_pb,
                   int64_t nblocks, Launch launch)
{
    if (nblocks < 0 || out == nullptr || in0 == nullptr) return fail(GFDM_HIP_EINVAL, "NULL buffer or negative block count");
    if (nblocks == 0 || out_pb == 0 || in0_pb == 0) return GFDM_HIP_OK;
    DeviceGuard guard(pl.device);
    if (!guard.ok) return fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    const gfdm::HostOperand ops[3] = { { out, out_pb * sizeof(cf), out_pb * sizeof(cf), true },
                                       { const_cast<float*>(in0), in0_pb * sizeof(cf), in0_pb * sizeof(cf), false },
                                       { const_cast<float*>(in1), in1_stride * sizeof(cf), in1_pb * sizeof(cf), false } };
    const int nops = in1 ? 3 : 2;
    auto fn = [&](void* const* d, int64_t nb, hipStream_t s) {
        return launch(static_cast<cf*>(d[0]), static_cast<const cf*>(d[1]), nops == 3 ? static_cast<const cf*>(d[2]) : nullptr, nb, s);
    };
    FamilyPin pin(pl);
    return pl.pipe.run(pl.stream, ops, nops, nblocks, fn);
}

template <typename Launch>
int run_host(Plan& pl, float* out, const float* in0, const float* in1, int64_t nblocks, Launch launch)
{
    const size_t n = (size_t)pl.dp.N;
    return run_host_sized(pl, out, n, in0, n, in1, n, n, nblocks, launch);
}

template <typename Launch>
int run_device(Plan& pl, void* out, const void* in0, int64_t nblocks, Launch launch)
{
    if (nblocks < 0 || out == nullptr || in0 == nullptr) return fail(GFDM_HIP_EINVAL, "NULL buffer or negative block count");
    if (nblocks == 0) return GFDM_HIP_OK;
    if (nblocks > 0x7fffffff) return fail(GFDM_HIP_EINVAL, "more than 2^31 - 1 blocks per call (one workgroup or wavefront per block)");
    DeviceGuard guard(pl.device);
    if (!guard.ok) return fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    hipError_t e = launch();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return GFDM_HIP_OK;
}

// est != nullptr: f_eq points at the received preambles and the kernel derives the equaliser itself (EQ_PREAMBLE)
hipError_t rx_launch(Plan& pl, const gfdm::IcParams& ic, int mode, cf* out, const cf* in, const cf* f_eq, int64_t nblocks,
                            hipStream_t s, const gfdm::EstPlan* est = nullptr)
{
    const int family = pl.current_family();
    if (family == gfdm::FAMILY_ROWLANE) return gfdm::launch_rowlane_receive(pl.dp, ic, est, pl.d_twT, mode, out, in, f_eq, nblocks, s);
    bool tuned = family == gfdm::FAMILY_ROWLANE_JIT;
    if (tuned && est && pl.pinned_pre >= 0) tuned = pl.pinned_pre == 1;
    else if (tuned && est && pl.jit_pre_pending) {
        const int st = pl.jit_pre_pending->load(std::memory_order_acquire);
        if (st == 1) pl.jit_pre_pending.reset(); else tuned = false;       // still compiling (or failed): this estimated call runs on the generic family
    }
    if (tuned) return gfdm::jit_launch_receive(&pl.jit, pl.dp, ic, est, pl.d_twT, mode, out, in, f_eq, nblocks, s);
    return gfdm::launch_generic_receive(pl.dp, ic, est, mode, out, in, f_eq, nblocks, s);
}

hipError_t mod_launch(Plan& pl, const gfdm::TxParams& tx, cf* out, const cf* in, int64_t nblocks, hipStream_t s)
{
    const int family = pl.current_family();
    if (family == gfdm::FAMILY_ROWLANE) return gfdm::launch_rowlane_modulate(pl.dp, tx, pl.d_twT, out, in, nblocks, s);
    if (family == gfdm::FAMILY_ROWLANE_JIT) return gfdm::jit_launch_modulate(&pl.jit, pl.dp, tx, pl.d_twT, out, in, nblocks, s);
    return gfdm::launch_generic_modulate(pl.dp, tx, out, in, nblocks, s);
}

// name of the family the handle launches with NOW (a background instantiation that has finished is picked up here as well)
const char* plan_kernel_name(const Plan& pl)
{
    (void)const_cast<Plan&>(pl).current_family();
    return pl.kernel_name.c_str();
}

const gfdm::TxParams kNoTx = {};

const gfdm::IcParams kNoIc = { 0, 0, 0, 0, nullptr, nullptr, 0, nullptr };

}  // namespace

struct gfdm_hip_modulator { Plan plan; };
// frame input (cyclic prefix removal) + demapped output, shared by receiver and advanced receiver handles
struct FrameIo {
    gfdm::RxIo io{};
    void* d_rank = nullptr;
    int device = 0;
    bool configured = false;
    ~FrameIo()
    {
        if (d_rank) {
            DeviceGuard guard(device);
            (void)hipFree(d_rank);
        }
    }
};

int frame_io_configure(FrameIo& f, const Plan& pl, int frame_len, int cp_len, const int* smap, int n_map, int per_timeslot)
{
    const int N = pl.dp.N, K = pl.dp.K, M = pl.dp.M;
    if (cp_len < 0 || frame_len < cp_len + N) return fail(GFDM_HIP_EINVAL, "frame_len must be at least cp_len + block size");
    if (n_map < 0 || n_map > K || (n_map > 0 && !smap)) return fail(GFDM_HIP_EINVAL, "bad subcarrier_map");
    if (n_map > 32767) return fail(GFDM_HIP_EUNSUPPORTED, "more than 32767 active subcarriers (the rank table holds 16-bit positions)");
    std::vector<int> sorted(smap, smap + n_map);
    std::sort(sorted.begin(), sorted.end());                      // the reference constructor sorts the map (resource_mapper_kernel_cc.cc:55)
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return fail(GFDM_HIP_EINVAL, "All entries in subcarrier_map MUST be unique!");
    if (n_map > 0 && (sorted.front() < 0 || sorted.back() >= K)) return fail(GFDM_HIP_EINVAL, "subcarrier_map entry out of range");
    std::vector<short> rank(K, (short)-1);
    for (int a = 0; a < n_map; ++a) rank[sorted[a]] = (short)a;
    DeviceGuard guard(pl.device);
    if (!f.d_rank) HIP_TRY(hipMalloc(&f.d_rank, (size_t)K * sizeof(short)));
    HIP_TRY(hipMemcpy(f.d_rank, rank.data(), (size_t)K * sizeof(short), hipMemcpyHostToDevice));
    f.device = pl.device;
    f.io.in_stride = frame_len; f.io.in_offset = cp_len;
    f.io.demap = n_map > 0 ? 1 : 0; f.io.per_timeslot = per_timeslot ? 1 : 0; f.io.A = n_map;
    f.io.nout = n_map > 0 ? n_map * M : N;
    f.io.rank = reinterpret_cast<const short*>(f.d_rank);
    f.configured = true;
    return GFDM_HIP_OK;
}

// effective I/O of one frames call; noutput_size <= 0 selects everything
int frame_io_for_call(const FrameIo& f, const Plan& pl, int noutput_size, gfdm::RxIo& io)
{
    if (!f.configured) return fail(GFDM_HIP_EINVAL, "configure_frames has not been called on this handle");
    io = f.io;
    if (io.demap) {
        if (noutput_size > io.A * pl.dp.M) {
            char buf[200];
            snprintf(buf, sizeof(buf), "output vector size(%d) MUST not exceed active_subcarriers * timeslots(%d)!", noutput_size, io.A * pl.dp.M);
            return fail(GFDM_HIP_EINVAL, buf);                     // resource_mapper_kernel_cc.cc:95-99
        }
        if (noutput_size > 0) io.nout = noutput_size;
    } else if (noutput_size > 0 && noutput_size != io.nout) {
        // no demapper in the store stage: the kernel writes the whole [k][m] block, a shorter caller buffer would be overrun
        char buf[200];
        snprintf(buf, sizeof(buf), "noutput_size(%d) needs a subcarrier map: without one every frame yields block_size(%d) symbols", noutput_size, io.nout);
        return fail(GFDM_HIP_EINVAL, buf);
    }
    return GFDM_HIP_OK;
}

struct gfdm_hip_receiver { Plan plan; FrameIo frames; const gfdm_hip_channel_estimator* est = nullptr; };
struct gfdm_hip_advanced_receiver {
    Plan plan;
    FrameIo frames;
    const gfdm_hip_channel_estimator* est = nullptr;
    gfdm::IcParams ic{};
    void* d_ic = nullptr;     // points | smap | active
    ~gfdm_hip_advanced_receiver()
    {
        if (d_ic) {
            DeviceGuard guard(plan.device);
            (void)hipFree(d_ic);
        }
    }
};

// ---------------------------------------------------------------------------------------------

extern "C" {

const char* gfdm_hip_strerror(int status)
{
    switch (status) {
    case GFDM_HIP_OK: return "success";
    case GFDM_HIP_EINVAL_TAPS: return "number of frequency taps MUST be equal to n_timeslots * overlap";
    case GFDM_HIP_EINVAL_OVERLAP: return "overlap MUST be greater or equal 2";
    case GFDM_HIP_EINVAL: return "invalid argument";
    case GFDM_HIP_ENODEV: return "no usable HIP device";
    case GFDM_HIP_EHIP: return "HIP runtime error";
    case GFDM_HIP_ENOMEM: return "out of device memory";
    case GFDM_HIP_EUNSUPPORTED: return "unsupported configuration";
    default: return "unknown gfdm_hip status";
    }
}

const char* gfdm_hip_last_error(void) { return g_last_error.c_str(); }

int gfdm_hip_force_generic_family_for_testing(int enable)
{
    return g_force_generic.exchange(enable ? 1 : 0);
}

int gfdm_hip_set_jit(int mode)
{
    return g_jit.exchange(mode < 0 ? 0 : mode > 3 ? 3 : mode);
}

int gfdm_hip_precompile(int timeslots, int subcarriers, int overlap, unsigned parts)
{
    if (!gfdm::jit_eligible(timeslots, subcarriers, overlap))
        return gfdm::rowlane_supports(timeslots, subcarriers, overlap) ? GFDM_HIP_OK      // compiled into the library: nothing to do
                                                                        : fail(GFDM_HIP_EUNSUPPORTED, "shape is served by the generic kernel family: nothing to instantiate");
    if (parts == 0) parts = (1u << gfdm::JIT_NUM_PARTS) - 1;
    for (int part = 0; part < gfdm::JIT_NUM_PARTS; ++part) {
        if (!((parts >> part) & 1u)) continue;
        std::string why;
        if (!gfdm::jit_build_only(timeslots, subcarriers, part == gfdm::JIT_PART_EST ? 2 : overlap, part, why)) return fail(GFDM_HIP_EHIP, why);
    }
    return GFDM_HIP_OK;
}

void gfdm_hip_quiesce(void)
{
    gfdm::jit_quiesce();
    gfdm::host_copy_pool_quiesce();
}

int gfdm_hip_set_ic_matrix_cores(int mode)
{
    return g_ic_mfma.exchange(mode < 0 ? 0 : mode > 2 ? 2 : mode);
}

int gfdm_hip_set_dft_matrix_cores(int mode)
{
    return g_dft_mfma.exchange(mode < 0 ? 0 : mode > 2 ? 2 : mode);
}

int gfdm_hip_jit_build_for_testing(int timeslots, int subcarriers, int overlap, int part)
{
    std::string why;
    if (gfdm::jit_build_only(timeslots, subcarriers, overlap, part, why)) return GFDM_HIP_OK;
    return fail(GFDM_HIP_EINVAL, why.c_str());
}

int gfdm_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// GFDM_BUILD_ID: hash of csrc/*, include/gfdm_hip.h and the compiler flags, written by the Makefile (obj/gfdm_build_id.inc) -- the profiles
// under profiles/ carry the id of the library they were taken with, and bench.py refuses to quote counters of another build
#include "gfdm_build_id.inc"
const char* gfdm_hip_version(void) { return "gfdm_hip 0.4 (gfx950) build " GFDM_BUILD_ID; }
const char* gfdm_hip_build_id(void) { return GFDM_BUILD_ID; }

int gfdm_hip_register_host(void* ptr, size_t bytes) { return gfdm::host_register(ptr, bytes); }
int gfdm_hip_unregister_host(void* ptr) { return gfdm::host_unregister(ptr); }
int gfdm_hip_set_host_pipeline(int mode, int64_t chunk_bytes, int depth, int copy_threads, int kernel_streams)
{
    return gfdm::host_pipeline_set(mode, chunk_bytes, depth, copy_threads, kernel_streams);
}
int gfdm_hip_get_host_pipeline(int* mode, int64_t* chunk_bytes, int* depth, int* copy_threads, int* kernel_streams)
{
    gfdm::host_pipeline_get(mode, chunk_bytes, depth, copy_threads, kernel_streams);
    return GFDM_HIP_OK;
}
int gfdm_hip_host_call_times(int64_t* ns5)
{
    if (!ns5) return fail(GFDM_HIP_EINVAL, "NULL argument");
    const gfdm::HostCallStats& st = gfdm::host_last_call();
    ns5[0] = st.ns_setup; ns5[1] = st.ns_copy; ns5[2] = st.ns_launch; ns5[3] = st.ns_post; ns5[4] = st.ns_wait;
    return GFDM_HIP_OK;
}
int gfdm_hip_set_host_streaming_copies_for_testing(int enable) { return gfdm::host_streaming_copies(enable); }
int gfdm_hip_host_call_stats(int64_t* chunks, int64_t* chunk_blocks, int64_t* staged_bytes, unsigned* direct_mask, int* mode, int* copy_threads)
{
    const gfdm::HostCallStats& st = gfdm::host_last_call();
    if (chunks) *chunks = st.chunks;
    if (chunk_blocks) *chunk_blocks = st.chunk_blocks;
    if (staged_bytes) *staged_bytes = st.staged_bytes;
    if (direct_mask) *direct_mask = st.direct_mask;
    if (mode) *mode = st.mode;
    if (copy_threads) *copy_threads = st.copy_threads;
    return GFDM_HIP_OK;
}

// ---- modulator ----

int gfdm_hip_modulator_create(gfdm_hip_modulator** out, int timeslots, int subcarriers, int overlap, const float* taps, int ntaps,
                              int device)
{
    if (!out) return fail(GFDM_HIP_EINVAL, "NULL handle pointer");
    *out = nullptr;
    gfdm_hip_modulator* m = new (std::nothrow) gfdm_hip_modulator();
    if (!m) return fail(GFDM_HIP_ENOMEM, "out of host memory");
    int rc = plan_create(m->plan, timeslots, subcarriers, overlap, taps, ntaps, device, false);
    if (rc != GFDM_HIP_OK) { delete m; return rc; }
    *out = m;
    return GFDM_HIP_OK;
}

int gfdm_hip_modulator_destroy(gfdm_hip_modulator* m) { delete m; return GFDM_HIP_OK; }
int gfdm_hip_modulator_block_size(const gfdm_hip_modulator* m) { return m ? m->plan.dp.N : GFDM_HIP_EINVAL; }

int gfdm_hip_modulator_filter_taps(const gfdm_hip_modulator* m, float* out)
{
    if (!m || !out) return fail(GFDM_HIP_EINVAL, "NULL argument");
    memcpy(out, m->plan.h_taps.data(), m->plan.h_taps.size() * sizeof(cf));
    return GFDM_HIP_OK;
}

const char* gfdm_hip_modulator_kernel_name(const gfdm_hip_modulator* m) { return m ? plan_kernel_name(m->plan) : ""; }

int gfdm_hip_modulator_work_device(gfdm_hip_modulator* m, void* out, const void* in, int64_t nblocks, void* stream)
{
    if (!m) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_device(m->plan, out, in, nblocks, [&]() {
        return mod_launch(m->plan, kNoTx, (cf*)out, (const cf*)in, nblocks, (hipStream_t)stream);
    });
}

int gfdm_hip_modulator_work_host(gfdm_hip_modulator* m, float* out, const float* in, int64_t nblocks)
{
    if (!m) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_host(m->plan, out, in, nullptr, nblocks, [&](cf* o, const cf* i, const cf*, int64_t nb, hipStream_t s) {
        return status_of(mod_launch(m->plan, kNoTx, o, i, nb, s));
    });
}

// ---- receiver ----

int gfdm_hip_receiver_create(gfdm_hip_receiver** out, int timeslots, int subcarriers, int overlap, const float* taps, int ntaps,
                             int device)
{
    if (!out) return fail(GFDM_HIP_EINVAL, "NULL handle pointer");
    *out = nullptr;
    gfdm_hip_receiver* r = new (std::nothrow) gfdm_hip_receiver();
    if (!r) return fail(GFDM_HIP_ENOMEM, "out of host memory");
    int rc = plan_create(r->plan, timeslots, subcarriers, overlap, taps, ntaps, device, true);
    if (rc != GFDM_HIP_OK) { delete r; return rc; }
    *out = r;
    return GFDM_HIP_OK;
}

int gfdm_hip_receiver_destroy(gfdm_hip_receiver* r) { delete r; return GFDM_HIP_OK; }
int gfdm_hip_receiver_block_size(const gfdm_hip_receiver* r) { return r ? r->plan.dp.N : GFDM_HIP_EINVAL; }
int gfdm_hip_receiver_timeslots(const gfdm_hip_receiver* r) { return r ? r->plan.dp.M : GFDM_HIP_EINVAL; }
int gfdm_hip_receiver_subcarriers(const gfdm_hip_receiver* r) { return r ? r->plan.dp.K : GFDM_HIP_EINVAL; }
int gfdm_hip_receiver_overlap(const gfdm_hip_receiver* r) { return r ? r->plan.dp.L : GFDM_HIP_EINVAL; }

int gfdm_hip_receiver_filter_taps(const gfdm_hip_receiver* r, float* out)
{
    if (!r || !out) return fail(GFDM_HIP_EINVAL, "NULL argument");
    memcpy(out, r->plan.h_taps.data(), r->plan.h_taps.size() * sizeof(cf));
    return GFDM_HIP_OK;
}

int gfdm_hip_receiver_ic_filter_taps(const gfdm_hip_receiver* r, float* out)
{
    if (!r || !out) return fail(GFDM_HIP_EINVAL, "NULL argument");
    memcpy(out, r->plan.h_ictaps.data(), r->plan.h_ictaps.size() * sizeof(cf));
    return GFDM_HIP_OK;
}

const char* gfdm_hip_receiver_kernel_name(const gfdm_hip_receiver* r) { return r ? plan_kernel_name(r->plan) : ""; }

int gfdm_hip_receiver_demodulate_device(gfdm_hip_receiver* r, void* out, const void* in, const void* f_eq, int64_t nblocks, void* stream)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_device(r->plan, out, in, nblocks, [&]() {
        return rx_launch(r->plan, kNoIc, gfdm::RX_DEMOD, (cf*)out, (const cf*)in, (const cf*)f_eq, nblocks, (hipStream_t)stream);
    });
}

int gfdm_hip_receiver_demodulate_host(gfdm_hip_receiver* r, float* out, const float* in, const float* f_eq, int64_t nblocks)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_host(r->plan, out, in, f_eq, nblocks, [&](cf* o, const cf* i, const cf* e, int64_t nb, hipStream_t s) {
        return status_of(rx_launch(r->plan, kNoIc, gfdm::RX_DEMOD, o, i, e, nb, s));
    });
}

int gfdm_hip_receiver_fft_filter_downsample_device(gfdm_hip_receiver* r, void* out, const void* in, const void* f_eq, int64_t nblocks,
                                                   void* stream)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_device(r->plan, out, in, nblocks, [&]() {
        return rx_launch(r->plan, kNoIc, gfdm::RX_FD, (cf*)out, (const cf*)in, (const cf*)f_eq, nblocks, (hipStream_t)stream);
    });
}

int gfdm_hip_receiver_fft_filter_downsample_host(gfdm_hip_receiver* r, float* out, const float* in, const float* f_eq, int64_t nblocks)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_host(r->plan, out, in, f_eq, nblocks, [&](cf* o, const cf* i, const cf* e, int64_t nb, hipStream_t s) {
        return status_of(rx_launch(r->plan, kNoIc, gfdm::RX_FD, o, i, e, nb, s));
    });
}

int gfdm_hip_receiver_transform_subcarriers_to_td_device(gfdm_hip_receiver* r, void* out, const void* in, int64_t nblocks, void* stream)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_device(r->plan, out, in, nblocks, [&]() {
        return gfdm::launch_generic_to_td(r->plan.dp, (cf*)out, (const cf*)in, nblocks, (hipStream_t)stream);
    });
}

int gfdm_hip_receiver_transform_subcarriers_to_td_host(gfdm_hip_receiver* r, float* out, const float* in, int64_t nblocks)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_host(r->plan, out, in, nullptr, nblocks, [&](cf* o, const cf* i, const cf*, int64_t nb, hipStream_t s) {
        return status_of(gfdm::launch_generic_to_td(r->plan.dp, o, i, nb, s));
    });
}

int gfdm_hip_receiver_cancel_sc_interference_device(gfdm_hip_receiver* r, void* out, const void* td_in, const void* fd_in,
                                                    int64_t nblocks, void* stream)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (!fd_in) return fail(GFDM_HIP_EINVAL, "NULL buffer");
    return run_device(r->plan, out, td_in, nblocks, [&]() {
        return gfdm::launch_generic_cancel(r->plan.dp, (cf*)out, (const cf*)td_in, (const cf*)fd_in, nblocks, (hipStream_t)stream);
    });
}

int gfdm_hip_receiver_cancel_sc_interference_host(gfdm_hip_receiver* r, float* out, const float* td_in, const float* fd_in,
                                                  int64_t nblocks)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (!fd_in) return fail(GFDM_HIP_EINVAL, "NULL buffer");
    return run_host(r->plan, out, td_in, fd_in, nblocks, [&](cf* o, const cf* i, const cf* e, int64_t nb, hipStream_t s) {
        return status_of(gfdm::launch_generic_cancel(r->plan.dp, o, i, e, nb, s));
    });
}

// ---- advanced receiver ----

int gfdm_hip_advanced_receiver_create(gfdm_hip_advanced_receiver** out, int timeslots, int subcarriers, int overlap, const float* taps,
                                      int ntaps, const int* subcarrier_map, int n_subcarrier_map, int ic_iter,
                                      const float* constellation_points, int n_points, int decision, int do_phase_compensation,
                                      int device)
{
    if (!out) return fail(GFDM_HIP_EINVAL, "NULL handle pointer");
    *out = nullptr;
    if (n_subcarrier_map < 0 || (n_subcarrier_map > 0 && !subcarrier_map)) return fail(GFDM_HIP_EINVAL, "bad subcarrier_map");
    if (n_points < 1 || n_points > 4096 || !constellation_points) return fail(GFDM_HIP_EINVAL, "constellation needs 1..4096 points");
    if (decision < GFDM_HIP_DECIDE_AUTO || decision > GFDM_HIP_DECIDE_BPSK) return fail(GFDM_HIP_EINVAL, "bad decision rule");
    for (int i = 0; i < n_subcarrier_map; ++i)
        if (subcarrier_map[i] < 0 || subcarrier_map[i] >= subcarriers) return fail(GFDM_HIP_EINVAL, "subcarrier_map entry out of range");
    gfdm_hip_advanced_receiver* a = new (std::nothrow) gfdm_hip_advanced_receiver();
    if (!a) return fail(GFDM_HIP_ENOMEM, "out of host memory");
    int rc = plan_create(a->plan, timeslots, subcarriers, overlap, taps, ntaps, device, true, (1u << gfdm::JIT_PART_RX) | (1u << gfdm::JIT_PART_RX_IC));
    if (rc != GFDM_HIP_OK) { delete a; return rc; }

    const cf* pts = reinterpret_cast<const cf*>(constellation_points);
    // "is this GNU Radio's unit constellation": every component within 6 * FLT_EPSILON * |component| (5.1e-7 at 1 / sqrt 2 = 8.5 f32 ulps there) of the unit
    // point's.  Points that went through a double -> float conversion or a product with 1 / sqrt 2 in float land within one ulp; gr::digital's
    // constellation_qpsk is built from the LITERAL SQRT_TWO = 0.707107 (gr-digital/lib/constellation.cc), which as a float sits 2.3e-7 = 3.8 ulps
    // away from 1 / sqrt 2 and must still get the sign tests (round-5 advisor; tests/test_parity_gpu.py test_decision_rule_a_handle_runs_is_reported).
    // The sign-test kernels then cancel with 0.70710678f: 3.2e-7 relative to the literal's points, far inside the 1e-5 of the parity statement.
    auto near = [](cf p, float re, float im) {
        const float tr = 6.f * FLT_EPSILON * (std::fabs(re) > 0.f ? std::fabs(re) : 1.f), ti = 6.f * FLT_EPSILON * (std::fabs(im) > 0.f ? std::fabs(im) : 1.f);
        return std::fabs(p.x - re) <= tr && std::fabs(p.y - im) <= ti;
    };
    if (decision == GFDM_HIP_DECIDE_AUTO) {
        const float s = 0.70710678118654752f;
        if (n_points == 4 && near(pts[0], -s, -s) && near(pts[1], s, -s) && near(pts[2], -s, s) && near(pts[3], s, s))
            decision = GFDM_HIP_DECIDE_QPSK;
        else if (n_points == 2 && near(pts[0], -1.f, 0.f) && near(pts[1], 1.f, 0.f))
            decision = GFDM_HIP_DECIDE_BPSK;
        else
            decision = GFDM_HIP_DECIDE_NEAREST;
    }
    if ((decision == GFDM_HIP_DECIDE_QPSK && n_points != 4) || (decision == GFDM_HIP_DECIDE_BPSK && n_points != 2)) {
        delete a;
        return fail(GFDM_HIP_EINVAL, "decision rule does not match the number of constellation points");
    }
    // The sign-test kernels cancel with the unit constellations' own points (+-1/sqrt 2, +-1).  An explicit QPSK / BPSK rule over other
    // points (scaled, rotated) keeps its decision REGIONS only if it is the nearest-point rule of those points; it is then run as that, on
    // the points as given -- every kernel family reads ic.points for it, so a handle gives the same results on each of them.
    // The rule a handle really runs is reported by gfdm_hip_advanced_receiver_decision.
    {
        const float s = 0.70710678118654752f;
        if (decision == GFDM_HIP_DECIDE_QPSK && !(near(pts[0], -s, -s) && near(pts[1], s, -s) && near(pts[2], -s, s) && near(pts[3], s, s)))
            decision = GFDM_HIP_DECIDE_NEAREST;
        if (decision == GFDM_HIP_DECIDE_BPSK && !(near(pts[0], -1.f, 0.f) && near(pts[1], 1.f, 0.f)))
            decision = GFDM_HIP_DECIDE_NEAREST;
    }

    const size_t pts_bytes = (size_t)n_points * sizeof(cf);
    const size_t smap_bytes = (size_t)(n_subcarrier_map > 0 ? n_subcarrier_map : 1) * sizeof(int);
    const size_t act_bytes = (size_t)subcarriers;
    std::vector<unsigned char> blob(pts_bytes + smap_bytes + act_bytes, 0);
    memcpy(blob.data(), pts, pts_bytes);
    if (n_subcarrier_map > 0) memcpy(blob.data() + pts_bytes, subcarrier_map, (size_t)n_subcarrier_map * sizeof(int));
    for (int i = 0; i < n_subcarrier_map; ++i) {            // multiplicity of each subcarrier in the map (saturating)
        unsigned char& c = blob[pts_bytes + smap_bytes + subcarrier_map[i]];
        if (c < 255) ++c;
    }
    {
        DeviceGuard guard(device);
        hipError_t e = hipMalloc(&a->d_ic, blob.size());
        if (e == hipSuccess) e = hipMemcpy(a->d_ic, blob.data(), blob.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { delete a; return fail_hip(e, "constellation upload"); }
    }
    a->ic.ic_iter = ic_iter;
    a->ic.do_phase_compensation = do_phase_compensation;
    a->ic.decision = decision;
    a->ic.npoints = n_points;
    a->ic.points = reinterpret_cast<const cf*>(a->d_ic);
    a->ic.smap = reinterpret_cast<const int*>(reinterpret_cast<unsigned char*>(a->d_ic) + pts_bytes);
    a->ic.active = reinterpret_cast<unsigned char*>(a->d_ic) + pts_bytes + smap_bytes;
    a->ic.n_active = n_subcarrier_map;
    *out = a;
    return GFDM_HIP_OK;
}

int gfdm_hip_advanced_receiver_destroy(gfdm_hip_advanced_receiver* a) { delete a; return GFDM_HIP_OK; }
int gfdm_hip_advanced_receiver_block_size(const gfdm_hip_advanced_receiver* a) { return a ? a->plan.dp.N : GFDM_HIP_EINVAL; }
int gfdm_hip_advanced_receiver_set_ic(gfdm_hip_advanced_receiver* a, int ic_iter)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    a->ic.ic_iter = ic_iter;
    return GFDM_HIP_OK;
}
int gfdm_hip_advanced_receiver_get_ic(const gfdm_hip_advanced_receiver* a) { return a ? a->ic.ic_iter : GFDM_HIP_EINVAL; }
int gfdm_hip_advanced_receiver_set_phase_compensation(gfdm_hip_advanced_receiver* a, int enable)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    a->ic.do_phase_compensation = enable;
    return GFDM_HIP_OK;
}
int gfdm_hip_advanced_receiver_get_phase_compensation(const gfdm_hip_advanced_receiver* a)
{
    return a ? a->ic.do_phase_compensation : GFDM_HIP_EINVAL;
}
const char* gfdm_hip_advanced_receiver_kernel_name(const gfdm_hip_advanced_receiver* a) { return a ? plan_kernel_name(a->plan) : ""; }
int gfdm_hip_advanced_receiver_decision(const gfdm_hip_advanced_receiver* a) { return a ? a->ic.decision : GFDM_HIP_EINVAL; }

int gfdm_hip_advanced_receiver_work_device(gfdm_hip_advanced_receiver* a, void* out, const void* in, const void* f_eq, int64_t nblocks,
                                           void* stream)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_device(a->plan, out, in, nblocks, [&]() {
        return rx_launch(a->plan, a->ic, gfdm::RX_IC, (cf*)out, (const cf*)in, (const cf*)f_eq, nblocks, (hipStream_t)stream);
    });
}

int gfdm_hip_advanced_receiver_work_host(gfdm_hip_advanced_receiver* a, float* out, const float* in, const float* f_eq, int64_t nblocks)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_host(a->plan, out, in, f_eq, nblocks, [&](cf* o, const cf* i, const cf* e, int64_t nb, hipStream_t s) {
        return status_of(rx_launch(a->plan, a->ic, gfdm::RX_IC, o, i, e, nb, s));
    });
}

}  // extern "C"

// ---- receiver / advanced receiver on raw frames with demapped output ----

extern "C" {

int gfdm_hip_receiver_configure_frames(gfdm_hip_receiver* r, int frame_len, int cp_len, const int* subcarrier_map, int n_subcarrier_map,
                                       int per_timeslot)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return frame_io_configure(r->frames, r->plan, frame_len, cp_len, subcarrier_map, n_subcarrier_map, per_timeslot);
}

int gfdm_hip_advanced_receiver_configure_frames(gfdm_hip_advanced_receiver* a, int frame_len, int cp_len, const int* subcarrier_map,
                                                int n_subcarrier_map, int per_timeslot)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return frame_io_configure(a->frames, a->plan, frame_len, cp_len, subcarrier_map, n_subcarrier_map, per_timeslot);
}

int gfdm_hip_receiver_demodulate_frames_device(gfdm_hip_receiver* r, void* out, const void* in, const void* f_eq, int noutput_size,
                                               int64_t nblocks, void* stream)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    gfdm::IcParams ic = kNoIc;
    int rc = frame_io_for_call(r->frames, r->plan, noutput_size, ic.io);
    if (rc != GFDM_HIP_OK) return rc;
    return run_device(r->plan, out, in, nblocks, [&]() {
        return rx_launch(r->plan, ic, gfdm::RX_DEMOD, (cf*)out, (const cf*)in, (const cf*)f_eq, nblocks, (hipStream_t)stream);
    });
}

int gfdm_hip_advanced_receiver_work_frames_device(gfdm_hip_advanced_receiver* a, void* out, const void* in, const void* f_eq,
                                                  int noutput_size, int64_t nblocks, void* stream)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    gfdm::IcParams ic = a->ic;
    int rc = frame_io_for_call(a->frames, a->plan, noutput_size, ic.io);
    if (rc != GFDM_HIP_OK) return rc;
    return run_device(a->plan, out, in, nblocks, [&]() {
        return rx_launch(a->plan, ic, gfdm::RX_IC, (cf*)out, (const cf*)in, (const cf*)f_eq, nblocks, (hipStream_t)stream);
    });
}

int gfdm_hip_receiver_demodulate_frames_host(gfdm_hip_receiver* r, float* out, const float* in, const float* f_eq, int noutput_size,
                                             int64_t nblocks)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (nblocks < 0) return fail(GFDM_HIP_EINVAL, "negative block count");
    gfdm::RxIo io;
    int rc = frame_io_for_call(r->frames, r->plan, noutput_size, io);
    if (rc != GFDM_HIP_OK) return rc;
    const size_t n = (size_t)r->plan.dp.N;
    return run_host_sized(r->plan, out, (size_t)io.nout, in, (size_t)io.in_stride, f_eq, n, n, nblocks, [&](cf* o, const cf* i, const cf* e, int64_t nb, hipStream_t s) {
        return gfdm_hip_receiver_demodulate_frames_device(r, o, i, e, noutput_size, nb, (void*)s);
    });
}

int gfdm_hip_advanced_receiver_work_frames_host(gfdm_hip_advanced_receiver* a, float* out, const float* in, const float* f_eq,
                                                int noutput_size, int64_t nblocks)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (nblocks < 0) return fail(GFDM_HIP_EINVAL, "negative block count");
    gfdm::RxIo io;
    int rc = frame_io_for_call(a->frames, a->plan, noutput_size, io);
    if (rc != GFDM_HIP_OK) return rc;
    const size_t n = (size_t)a->plan.dp.N;
    return run_host_sized(a->plan, out, (size_t)io.nout, in, (size_t)io.in_stride, f_eq, n, n, nblocks, [&](cf* o, const cf* i, const cf* e, int64_t nb, hipStream_t s) {
        return gfdm_hip_advanced_receiver_work_frames_device(a, o, i, e, noutput_size, nb, (void*)s);
    });
}

}  // extern "C"

// ---- composite transmitter (gr-gfdm transmitter_kernel) ----

struct gfdm_hip_transmitter {
    Plan plan;
    gfdm::TxParams tx{};          // template: device tables, lengths, shifts; per call: outs, nin, mapped/framed
    void* d_blob = nullptr;       // rank | front | back | preambles
    int M = 0, K = 0, A = 0;
    ~gfdm_hip_transmitter()
    {
        DeviceGuard guard(plan.device);
        if (d_blob) (void)hipFree(d_blob);
    }
};

extern "C" {

int gfdm_hip_transmitter_create(gfdm_hip_transmitter** out, int timeslots, int subcarriers, int active_subcarriers, int cp_len,
                                int cs_len, int ramp_len, const int* subcarrier_map, int n_subcarrier_map, int per_timeslot, int overlap,
                                const float* taps, int ntaps, const float* window_taps, int n_window_taps, const int* cyclic_shifts,
                                int n_cyclic_shifts, const float* preambles, int preamble_len, int device)
{
    if (!out) return fail(GFDM_HIP_EINVAL, "NULL handle pointer");
    *out = nullptr;
    const int M = timeslots, K = subcarriers, A = active_subcarriers;
    char buf[256];
    // resource_mapper_kernel_cc constructor checks, lib/resource_mapper_kernel_cc.cc:44-69
    if (M < 1 || K < 1 || A < 1) return fail(GFDM_HIP_EINVAL, "timeslots, subcarriers and active_subcarriers must be >= 1");
    if (A > K) {
        snprintf(buf, sizeof(buf), "active_subcarriers(%d) MUST be smaller or equal to subcarriers(%d)!", A, K);
        return fail(GFDM_HIP_EINVAL, buf);
    }
    if (!subcarrier_map || n_subcarrier_map != A) {
        snprintf(buf, sizeof(buf), "number of subcarrier_map entries(%d) MUST be equal to active_subcarriers(%d)!", n_subcarrier_map, A);
        return fail(GFDM_HIP_EINVAL, buf);
    }
    std::vector<int> smap(subcarrier_map, subcarrier_map + A);
    std::sort(smap.begin(), smap.end());
    if (std::adjacent_find(smap.begin(), smap.end()) != smap.end()) return fail(GFDM_HIP_EINVAL, "All entries in subcarrier_map MUST be unique!");
    if (smap.front() < 0) return fail(GFDM_HIP_EINVAL, "All subcarrier indices MUST be greater or equal to ZERO!");
    if (smap.back() >= K) return fail(GFDM_HIP_EINVAL, "All subcarrier indices MUST be smaller than subcarriers!");
    // add_cyclic_prefix_cc constructor check, lib/add_cyclic_prefix_cc.cc:42-50
    const int N = M * K, window_len = N + cp_len + cs_len;
    if (cp_len < 0 || cs_len < 0 || ramp_len < 0 || !window_taps || (n_window_taps != window_len && n_window_taps != 2 * ramp_len)) {
        snprintf(buf, sizeof(buf), "number of window taps(%d) MUST be equal to 2*ramp_len(%d) OR block_len+cp_len (%d)!", n_window_taps,
                 2 * ramp_len, window_len);
        return fail(GFDM_HIP_EINVAL, buf);
    }
    if (2 * ramp_len > window_len) return fail(GFDM_HIP_EINVAL, "ramp_len too large for the frame");
    // transmitter_kernel constructor checks, lib/transmitter_kernel.cc:57-67 (the C-ABI takes preambles as one [n][len] array)
    if (n_cyclic_shifts < 1 || n_cyclic_shifts > gfdm::TX_MAX_PORTS || !cyclic_shifts)
        return fail(GFDM_HIP_EINVAL, "Number of cyclic shifts and number of preambles do not match!");
    if (preamble_len < 0 || (preamble_len > 0 && !preambles)) return fail(GFDM_HIP_EINVAL, "All preambles must have equal size!");
    for (int i = 0; i < n_cyclic_shifts; ++i)
        if (cyclic_shifts[i] < 0 || cyclic_shifts[i] > cs_len || cp_len + cyclic_shifts[i] > N || cs_len - cyclic_shifts[i] > N)
            return fail(GFDM_HIP_EINVAL, "cyclic shift must lie in [0, cs_len], cp_len + shift and cs_len - shift must not exceed the block");
    if (A > 32767) return fail(GFDM_HIP_EUNSUPPORTED, "more than 32767 active subcarriers (the rank table holds 16-bit positions)");

    gfdm_hip_transmitter* t = new (std::nothrow) gfdm_hip_transmitter();
    if (!t) return fail(GFDM_HIP_ENOMEM, "out of host memory");
    int rc = plan_create(t->plan, M, K, overlap, taps, ntaps, device, false);
    if (rc != GFDM_HIP_OK) { delete t; return rc; }
    t->M = M; t->K = K; t->A = A;

    const size_t rank_bytes = ((size_t)K * sizeof(short) + 15) / 16 * 16;
    const size_t ramp_bytes = (size_t)(ramp_len > 0 ? ramp_len : 1) * sizeof(cf);
    const size_t pre_bytes = (size_t)n_cyclic_shifts * (size_t)(preamble_len > 0 ? preamble_len : 1) * sizeof(cf);
    std::vector<unsigned char> blob(rank_bytes + 2 * ramp_bytes + pre_bytes, 0);
    short* rank = reinterpret_cast<short*>(blob.data());
    for (int k = 0; k < K; ++k) rank[k] = -1;
    for (int a = 0; a < A; ++a) rank[smap[a]] = (short)a;
    const cf* w = reinterpret_cast<const cf*>(window_taps);
    memcpy(blob.data() + rank_bytes, w, (size_t)ramp_len * sizeof(cf));                                     // front ramp  :51-53
    memcpy(blob.data() + rank_bytes + ramp_bytes, w + (n_window_taps - ramp_len), (size_t)ramp_len * sizeof(cf));   // back ramp :54-56
    if (preamble_len > 0) memcpy(blob.data() + rank_bytes + 2 * ramp_bytes, preambles, (size_t)n_cyclic_shifts * preamble_len * sizeof(cf));
    {
        DeviceGuard guard(device);
        hipError_t e = hipMalloc(&t->d_blob, blob.size());
        if (e == hipSuccess) e = hipMemcpy(t->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) { delete t; return fail_hip(e, "transmitter table upload"); }
    }
    gfdm::TxParams& tx = t->tx;
    tx.A = A; tx.per_timeslot = per_timeslot ? 1 : 0; tx.nin = A * M;
    tx.cp = cp_len; tx.cs = cs_len; tx.ramp = ramp_len; tx.plen = preamble_len; tx.F = preamble_len + window_len;
    tx.nports = n_cyclic_shifts;
    for (int i = 0; i < n_cyclic_shifts; ++i) tx.shifts[i] = cyclic_shifts[i];
    unsigned char* d = reinterpret_cast<unsigned char*>(t->d_blob);
    tx.rank = reinterpret_cast<const short*>(d);
    tx.front = reinterpret_cast<const cf*>(d + rank_bytes);
    tx.back = reinterpret_cast<const cf*>(d + rank_bytes + ramp_bytes);
    tx.preambles = reinterpret_cast<const cf*>(d + rank_bytes + 2 * ramp_bytes);
    *out = t;
    return GFDM_HIP_OK;
}

int gfdm_hip_transmitter_destroy(gfdm_hip_transmitter* t) { delete t; return GFDM_HIP_OK; }
int gfdm_hip_transmitter_input_vector_size(const gfdm_hip_transmitter* t) { return t ? t->A * t->M : GFDM_HIP_EINVAL; }
int gfdm_hip_transmitter_output_vector_size(const gfdm_hip_transmitter* t) { return t ? t->tx.F : GFDM_HIP_EINVAL; }
int gfdm_hip_transmitter_block_size(const gfdm_hip_transmitter* t) { return t ? t->plan.dp.N : GFDM_HIP_EINVAL; }
int gfdm_hip_transmitter_n_cyclic_shifts(const gfdm_hip_transmitter* t) { return t ? t->tx.nports : GFDM_HIP_EINVAL; }
int gfdm_hip_transmitter_cyclic_shift(const gfdm_hip_transmitter* t, int port)
{
    return (t && port >= 0 && port < t->tx.nports) ? t->tx.shifts[port] : GFDM_HIP_EINVAL;
}
const char* gfdm_hip_transmitter_kernel_name(const gfdm_hip_transmitter* t) { return t ? plan_kernel_name(t->plan) : ""; }

static int tx_check_nin(const gfdm_hip_transmitter* t, int ninput_size)
{
    if (ninput_size < 0 || ninput_size > t->A * t->M) {
        char buf[200];
        snprintf(buf, sizeof(buf), "input vector size(%d) MUST not exceed active_subcarriers * timeslots(%d)!", ninput_size, t->A * t->M);
        return fail(GFDM_HIP_EINVAL, buf);                 // std::invalid_argument in resource_mapper_kernel_cc.cc:78-82
    }
    return GFDM_HIP_OK;
}

int gfdm_hip_transmitter_work_device(gfdm_hip_transmitter* t, void* const* outs, int n_outs, const void* in, int ninput_size,
                                     int64_t nblocks, void* stream)
{
    if (!t || !outs || !in) return fail(GFDM_HIP_EINVAL, "NULL argument");
    if (n_outs < 1 || n_outs > t->tx.nports) return fail(GFDM_HIP_EINVAL, "n_outs must be between 1 and the number of cyclic shifts");
    int rc = tx_check_nin(t, ninput_size);
    if (rc != GFDM_HIP_OK) return rc;
    gfdm::TxParams tx = t->tx;
    tx.mapped = 1; tx.framed = 1; tx.nin = ninput_size; tx.nports = n_outs;
    for (int i = 0; i < n_outs; ++i) {
        if (!outs[i]) return fail(GFDM_HIP_EINVAL, "NULL output port");
        tx.outs[i] = (cf*)outs[i];
    }
    return run_device(t->plan, outs[0], in, nblocks, [&]() { return mod_launch(t->plan, tx, nullptr, (const cf*)in, nblocks, (hipStream_t)stream); });
}

int gfdm_hip_transmitter_modulate_device(gfdm_hip_transmitter* t, void* out, const void* in, int ninput_size, int64_t nblocks, void* stream)
{
    if (!t) return fail(GFDM_HIP_EINVAL, "NULL handle");
    int rc = tx_check_nin(t, ninput_size);
    if (rc != GFDM_HIP_OK) return rc;
    gfdm::TxParams tx = t->tx;
    tx.mapped = 1; tx.framed = 0; tx.nin = ninput_size;
    return run_device(t->plan, out, in, nblocks, [&]() { return mod_launch(t->plan, tx, (cf*)out, (const cf*)in, nblocks, (hipStream_t)stream); });
}

int gfdm_hip_transmitter_add_frame_device(gfdm_hip_transmitter* t, void* out, const void* in, int cyclic_shift, int64_t nblocks, void* stream)
{
    if (!t) return fail(GFDM_HIP_EINVAL, "NULL handle");
    int port = -1;
    for (int i = 0; i < t->tx.nports; ++i) if (t->tx.shifts[i] == cyclic_shift) { port = i; break; }
    if (port < 0) return fail(GFDM_HIP_EINVAL, "no preamble was registered for this cyclic shift");     // d_preambles lookup, transmitter_kernel.cc:86-90
    gfdm::TxParams tx = t->tx;
    tx.mapped = 0; tx.framed = 1; tx.nports = 1; tx.shifts[0] = cyclic_shift; tx.outs[0] = (cf*)out;
    tx.preambles = t->tx.preambles + (int64_t)port * t->tx.plen;
    return run_device(t->plan, out, in, nblocks, [&]() { return gfdm::launch_add_frame(t->plan.dp, tx, (const cf*)in, nblocks, (hipStream_t)stream); });
}

// host-pointer variants (gfdm_hostpipe.h): operands = the output ports, then the symbols
static int tx_host(gfdm_hip_transmitter* t, float* const* outs, int n_outs, size_t out_elems_per_block, const float* in,
                   size_t in_elems_per_block, int64_t nblocks, const std::function<int(void* const*, const void*, int64_t, hipStream_t)>& enqueue)
{
    if (nblocks < 0 || !outs || !in) return fail(GFDM_HIP_EINVAL, "NULL buffer or negative block count");
    if (nblocks == 0) return GFDM_HIP_OK;
    Plan& pl = t->plan;
    DeviceGuard guard(pl.device);
    if (!guard.ok) return fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    gfdm::HostOperand ops[gfdm::TX_MAX_PORTS + 1];
    for (int i = 0; i < n_outs; ++i) {
        if (!outs[i]) return fail(GFDM_HIP_EINVAL, "NULL output port");
        ops[i] = gfdm::HostOperand{ outs[i], out_elems_per_block * sizeof(cf), out_elems_per_block * sizeof(cf), true };
    }
    ops[n_outs] = gfdm::HostOperand{ const_cast<float*>(in), in_elems_per_block * sizeof(cf), in_elems_per_block * sizeof(cf), false };
    auto fn = [&](void* const* d, int64_t nb, hipStream_t s) { return enqueue(d, d[n_outs], nb, s); };
    FamilyPin pin(pl);
    return pl.pipe.run(pl.stream, ops, n_outs + 1, nblocks, fn);
}

int gfdm_hip_transmitter_work_host(gfdm_hip_transmitter* t, float* const* outs, int n_outs, const float* in, int ninput_size, int64_t nblocks)
{
    if (!t) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (n_outs < 1 || n_outs > t->tx.nports) return fail(GFDM_HIP_EINVAL, "n_outs must be between 1 and the number of cyclic shifts");
    int rc = tx_check_nin(t, ninput_size);
    if (rc != GFDM_HIP_OK) return rc;
    return tx_host(t, outs, n_outs, (size_t)t->tx.F, in, (size_t)ninput_size, nblocks, [&](void* const* d_outs, const void* d_in, int64_t nb, hipStream_t s) {
        return gfdm_hip_transmitter_work_device(t, d_outs, n_outs, d_in, ninput_size, nb, (void*)s);
    });
}

int gfdm_hip_transmitter_modulate_host(gfdm_hip_transmitter* t, float* out, const float* in, int ninput_size, int64_t nblocks)
{
    if (!t) return fail(GFDM_HIP_EINVAL, "NULL handle");
    int rc = tx_check_nin(t, ninput_size);
    if (rc != GFDM_HIP_OK) return rc;
    float* outs[1] = { out };
    return tx_host(t, outs, 1, (size_t)t->plan.dp.N, in, (size_t)ninput_size, nblocks, [&](void* const* d_outs, const void* d_in, int64_t nb, hipStream_t s) {
        return gfdm_hip_transmitter_modulate_device(t, d_outs[0], d_in, ninput_size, nb, (void*)s);
    });
}

int gfdm_hip_transmitter_add_frame_host(gfdm_hip_transmitter* t, float* out, const float* in, int cyclic_shift, int64_t nblocks)
{
    if (!t) return fail(GFDM_HIP_EINVAL, "NULL handle");
    float* outs[1] = { out };
    return tx_host(t, outs, 1, (size_t)t->tx.F, in, (size_t)t->plan.dp.N, nblocks, [&](void* const* d_outs, const void* d_in, int64_t nb, hipStream_t s) {
        return gfdm_hip_transmitter_add_frame_device(t, d_outs[0], d_in, cyclic_shift, nb, (void*)s);
    });
}

}  // extern "C"

// ---- preamble channel estimator (lib/preamble_channel_estimator_cc.cc) ----------------------------------------------

struct gfdm_hip_channel_estimator {
    Plan plan;                       // device, private stream, staging buffers, table allocation (plan.dp is unused)
    gfdm::EstPlan ep{};
    int which_estimator = 0;
};

namespace {

// plain O(K^2) DFT in double: constructor-time only (the reference runs FFTW here, :99-107)
void host_dft(std::vector<std::complex<double>>& out, const float* in, int K)
{
    const double two_pi = 6.283185307179586476925286766559;
    out.assign(K, std::complex<double>(0.0, 0.0));
    std::vector<std::complex<double>> w(K);
    for (int i = 0; i < K; ++i) w[i] = std::complex<double>(std::cos(two_pi * i / K), -std::sin(two_pi * i / K));
    for (int j = 0; j < K; ++j) {
        std::complex<double> acc(0.0, 0.0);
        int e = 0;
        for (int q = 0; q < K; ++q) {
            acc += std::complex<double>(in[2 * q], in[2 * q + 1]) * w[e];
            e += j;
            if (e >= K) e -= K;
        }
        out[j] = acc;
    }
}

int est_stage_elems(const gfdm::EstPlan& e, int stage)
{
    switch (stage) {
    case gfdm::EST_RX_PREAMBLE: return 2 * e.K;
    case gfdm::EST_PREAMBLE_CHANNEL: return e.K;
    case gfdm::EST_FILTERED: return e.n_est;
    default: return e.M * e.K;
    }
}

int est_run_device(gfdm_hip_channel_estimator* c, int in_stage, int out_stage, void* out, const void* in, int64_t nframes, void* stream)
{
    if (!c) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_device(c->plan, out, in, nframes, [&]() {
        if (c->plan.current_family() == gfdm::FAMILY_ROWLANE && in_stage == gfdm::EST_RX_PREAMBLE && out_stage == gfdm::EST_FRAME)
            return gfdm::launch_rowlane_estimate(c->ep, static_cast<cf*>(out), static_cast<const cf*>(in), nframes, static_cast<hipStream_t>(stream));
        if (c->plan.family == gfdm::FAMILY_ROWLANE_JIT && in_stage == gfdm::EST_RX_PREAMBLE && out_stage == gfdm::EST_FRAME)
            return gfdm::jit_launch_estimate(&c->plan.jit, c->ep, static_cast<cf*>(out), static_cast<const cf*>(in), nframes, static_cast<hipStream_t>(stream));
        return gfdm::launch_estimate(c->ep, in_stage, out_stage, static_cast<cf*>(out), static_cast<const cf*>(in), nframes,
                                     static_cast<hipStream_t>(stream));
    });
}

int est_run_host(gfdm_hip_channel_estimator* c, int in_stage, int out_stage, float* out, const float* in, int64_t nframes)
{
    if (!c) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (nframes < 0) return fail(GFDM_HIP_EINVAL, "negative frame count");
    const size_t nout = est_stage_elems(c->ep, out_stage), nin = est_stage_elems(c->ep, in_stage);
    return run_host_sized(c->plan, out, nout, in, nin, nullptr, 0, 0, nframes, [&](cf* o, const cf* i, const cf*, int64_t nf, hipStream_t s) {
        if (c->plan.current_family() == gfdm::FAMILY_ROWLANE && in_stage == gfdm::EST_RX_PREAMBLE && out_stage == gfdm::EST_FRAME)
            return status_of(gfdm::launch_rowlane_estimate(c->ep, o, i, nf, s));
        if (c->plan.family == gfdm::FAMILY_ROWLANE_JIT && in_stage == gfdm::EST_RX_PREAMBLE && out_stage == gfdm::EST_FRAME)
            return status_of(gfdm::jit_launch_estimate(&c->plan.jit, c->ep, o, i, nf, s));
        return status_of(gfdm::launch_estimate(c->ep, in_stage, out_stage, o, i, nf, s));
    });
}

}  // namespace

extern "C" {

int gfdm_hip_channel_estimator_create(gfdm_hip_channel_estimator** out, int timeslots, int fft_len, int active_subcarriers, int is_dc_free,
                                      int which_estimator, const float* preamble, int n_preamble, int device)
{
    if (!out) return fail(GFDM_HIP_EINVAL, "NULL output handle");
    *out = nullptr;
    if (timeslots < 1 || fft_len < 2 || active_subcarriers < 2 || active_subcarriers > fft_len - (is_dc_free ? 1 : 0) || (active_subcarriers & 1) ||
        preamble == nullptr)
        return fail(GFDM_HIP_EINVAL, "timeslots >= 1, fft_len >= 2, even 2 <= active_subcarriers <= fft_len (fft_len - 1 when dc-free), preamble non-NULL");
    if (n_preamble < 2 * fft_len) {
        char buf[160];
        snprintf(buf, sizeof buf, "preamble holds %d samples, MUST hold 2 * fft_len = %d", n_preamble, 2 * fft_len);
        return fail(GFDM_HIP_EINVAL, buf);
    }
    if (!gfdm::estimator_supports(fft_len)) return fail(GFDM_HIP_EUNSUPPORTED, "fft_len too large for the LDS-resident estimator");
    std::unique_ptr<gfdm_hip_channel_estimator> c(new gfdm_hip_channel_estimator);
    const int K = fft_len;
    c->plan.device = device;
    c->which_estimator = which_estimator;
    std::vector<cf> tables;                                            // inv0 | inv1 | wK | w2K
    tables.reserve(5 * (size_t)K);
    std::vector<std::complex<double>> fd;
    for (int h = 0; h < 2; ++h) {
        host_dft(fd, preamble + 2 * (size_t)h * K, K);
        for (int j = 0; j < K; ++j) {                                  // initialize_inv_freq_preamble :108-116
            const std::complex<double> v = std::complex<double>(0.5, 0.0) / fd[j];
            tables.push_back(make_float2((float)v.real(), (float)v.imag()));
        }
    }
    unit_roots(tables, K);
    unit_roots(tables, 2 * K);
    gfdm::EstPlan& e = c->ep;
    e.M = timeslots; e.K = K; e.A = active_subcarriers; e.dc_free = is_dc_free ? 1 : 0;
    e.n_est = active_subcarriers + e.dc_free;
    e.log2K = ilog2_exact(K);
    e.log2K2 = ilog2_exact(2 * K);
    float sum = 0.0f;                                                  // initialize_gaussian_filter :84-97 (float arithmetic)
    for (int i = 0; i < 9; ++i) { e.gauss[i] = std::exp(-0.5f * (float)((i - 4) * (i - 4))); sum += e.gauss[i]; }
    for (int i = 0; i < 9; ++i) e.gauss[i] /= sum;
    DeviceGuard guard(device);
    if (!guard.ok) return fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    HIP_TRY(hipMalloc(&c->plan.d_tables, tables.size() * sizeof(cf)));
    HIP_TRY(hipMemcpy(c->plan.d_tables, tables.data(), tables.size() * sizeof(cf), hipMemcpyHostToDevice));
    HIP_TRY(hipStreamCreateWithFlags(&c->plan.stream, hipStreamNonBlocking));
    e.inv0 = c->plan.d_tables;
    e.inv1 = e.inv0 + K;
    e.wK = e.inv1 + K;
    e.w2K = e.wK + K;
    // estimate_frame runs in the row-lane layout where a shape with this (fft_len, timeslots) is instantiated; the single stages,
    // prepare_for_zf and estimate_snr always use the generic kernels.
    c->plan.family = gfdm::FAMILY_GENERIC;
    if (!g_force_generic.load()) {
        std::string why;
        if (gfdm::rowlane_supports_estimate(timeslots, K)) c->plan.family = gfdm::FAMILY_ROWLANE;
        else if (g_jit.load() && gfdm::jit_eligible(timeslots, K, 2)) {
            // the same policy as plan_create: a compile that takes more than seconds runs in the background, the generic kernels serve meanwhile
            int mode = g_jit.load();
            if (mode == 3) mode = (timeslots <= 16 || gfdm::jit_cached(timeslots, K, 2, gfdm::JIT_PART_EST)) ? 1 : 2;
            if (mode == 2) {
                c->plan.jit_pending = std::make_shared<std::atomic<int>>(0);
                gfdm::jit_prepare_async(timeslots, K, 2, 1u << gfdm::JIT_PART_EST, device, c->plan.jit_pending);
            } else if (gfdm::jit_prepare_estimate(timeslots, K, why)) {
                c->plan.family = gfdm::FAMILY_ROWLANE_JIT;
            }
        }
    }
    c->plan.kernel_name = c->plan.family == gfdm::FAMILY_ROWLANE ? "rowlane" : c->plan.family == gfdm::FAMILY_ROWLANE_JIT ? "rowlane_jit" : "generic_lds";
    *out = c.release();
    return GFDM_HIP_OK;
}

int gfdm_hip_channel_estimator_destroy(gfdm_hip_channel_estimator* c) { delete c; return GFDM_HIP_OK; }
int gfdm_hip_channel_estimator_timeslots(const gfdm_hip_channel_estimator* c) { return c ? c->ep.M : GFDM_HIP_EINVAL; }
int gfdm_hip_channel_estimator_fft_len(const gfdm_hip_channel_estimator* c) { return c ? c->ep.K : GFDM_HIP_EINVAL; }
int gfdm_hip_channel_estimator_active_subcarriers(const gfdm_hip_channel_estimator* c) { return c ? c->ep.A : GFDM_HIP_EINVAL; }
int gfdm_hip_channel_estimator_frame_len(const gfdm_hip_channel_estimator* c) { return c ? c->ep.M * c->ep.K : GFDM_HIP_EINVAL; }
int gfdm_hip_channel_estimator_is_dc_free(const gfdm_hip_channel_estimator* c) { return c ? c->ep.dc_free : GFDM_HIP_EINVAL; }
int gfdm_hip_channel_estimator_filtered_len(const gfdm_hip_channel_estimator* c) { return c ? c->ep.n_est : GFDM_HIP_EINVAL; }
const char* gfdm_hip_channel_estimator_kernel_name(const gfdm_hip_channel_estimator* c) { return c ? plan_kernel_name(c->plan) : ""; }

int gfdm_hip_channel_estimator_preamble_filter_taps(const gfdm_hip_channel_estimator* c, float* out)
{
    if (!c || !out) return fail(GFDM_HIP_EINVAL, "NULL argument");
    memcpy(out, c->ep.gauss, sizeof c->ep.gauss);
    return 9;
}

int gfdm_hip_channel_estimator_estimate_frame_device(gfdm_hip_channel_estimator* c, void* frame_estimate, const void* rx_preamble,
                                                     int64_t nframes, void* stream)
{
    return est_run_device(c, gfdm::EST_RX_PREAMBLE, gfdm::EST_FRAME, frame_estimate, rx_preamble, nframes, stream);
}

int gfdm_hip_channel_estimator_estimate_frame_host(gfdm_hip_channel_estimator* c, float* frame_estimate, const float* rx_preamble, int64_t nframes)
{
    return est_run_host(c, gfdm::EST_RX_PREAMBLE, gfdm::EST_FRAME, frame_estimate, rx_preamble, nframes);
}

int gfdm_hip_channel_estimator_estimate_preamble_channel_device(gfdm_hip_channel_estimator* c, void* fd_preamble_channel, const void* rx_preamble,
                                                                int64_t nframes, void* stream)
{
    return est_run_device(c, gfdm::EST_RX_PREAMBLE, gfdm::EST_PREAMBLE_CHANNEL, fd_preamble_channel, rx_preamble, nframes, stream);
}

int gfdm_hip_channel_estimator_estimate_preamble_channel_host(gfdm_hip_channel_estimator* c, float* fd_preamble_channel, const float* rx_preamble,
                                                              int64_t nframes)
{
    return est_run_host(c, gfdm::EST_RX_PREAMBLE, gfdm::EST_PREAMBLE_CHANNEL, fd_preamble_channel, rx_preamble, nframes);
}

int gfdm_hip_channel_estimator_filter_preamble_estimate_device(gfdm_hip_channel_estimator* c, void* filtered, const void* estimate, int64_t nframes,
                                                               void* stream)
{
    return est_run_device(c, gfdm::EST_PREAMBLE_CHANNEL, gfdm::EST_FILTERED, filtered, estimate, nframes, stream);
}

int gfdm_hip_channel_estimator_filter_preamble_estimate_host(gfdm_hip_channel_estimator* c, float* filtered, const float* estimate, int64_t nframes)
{
    return est_run_host(c, gfdm::EST_PREAMBLE_CHANNEL, gfdm::EST_FILTERED, filtered, estimate, nframes);
}

int gfdm_hip_channel_estimator_interpolate_frame_device(gfdm_hip_channel_estimator* c, void* frame_estimate, const void* filtered, int64_t nframes,
                                                        void* stream)
{
    return est_run_device(c, gfdm::EST_FILTERED, gfdm::EST_FRAME, frame_estimate, filtered, nframes, stream);
}

int gfdm_hip_channel_estimator_interpolate_frame_host(gfdm_hip_channel_estimator* c, float* frame_estimate, const float* filtered, int64_t nframes)
{
    return est_run_host(c, gfdm::EST_FILTERED, gfdm::EST_FRAME, frame_estimate, filtered, nframes);
}

int gfdm_hip_channel_estimator_prepare_for_zf_device(gfdm_hip_channel_estimator* c, void* transformed_frame, const void* frame_estimate,
                                                     int64_t nframes, void* stream)
{
    if (!c) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_device(c->plan, transformed_frame, frame_estimate, nframes, [&]() {
        return gfdm::launch_prepare_for_zf(static_cast<cf*>(transformed_frame), static_cast<const cf*>(frame_estimate),
                                           nframes * c->ep.M * c->ep.K, static_cast<hipStream_t>(stream));
    });
}

int gfdm_hip_channel_estimator_prepare_for_zf_host(gfdm_hip_channel_estimator* c, float* transformed_frame, const float* frame_estimate,
                                                   int64_t nframes)
{
    if (!c) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (nframes < 0) return fail(GFDM_HIP_EINVAL, "negative frame count");
    const size_t n = (size_t)c->ep.M * c->ep.K;
    return run_host_sized(c->plan, transformed_frame, n, frame_estimate, n, nullptr, 0, 0, nframes, [&](cf* o, const cf* i, const cf*, int64_t nf, hipStream_t s) {
        return status_of(gfdm::launch_prepare_for_zf(o, i, nf * (int64_t)n, s));
    });
}

int gfdm_hip_channel_estimator_estimate_snr_device(gfdm_hip_channel_estimator* c, float* snr_lin, float* cnrs, const void* rx_preamble,
                                                   int64_t nframes, void* stream)
{
    if (!c) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (!cnrs) return fail(GFDM_HIP_EINVAL, "NULL buffer");
    return run_device(c->plan, snr_lin, rx_preamble, nframes, [&]() {
        return gfdm::launch_estimate_snr(c->ep, snr_lin, cnrs, static_cast<const cf*>(rx_preamble), nframes, static_cast<hipStream_t>(stream));
    });
}

int gfdm_hip_channel_estimator_estimate_snr_host(gfdm_hip_channel_estimator* c, float* snr_lin, float* cnrs, const float* rx_preamble, int64_t nframes)
{
    if (!c) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (nframes < 0 || !snr_lin || !cnrs || !rx_preamble) return fail(GFDM_HIP_EINVAL, "NULL buffer or negative frame count");
    if (nframes == 0) return GFDM_HIP_OK;
    Plan& pl = c->plan;
    DeviceGuard guard(pl.device);
    if (!guard.ok) return fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    const size_t A = (size_t)c->ep.A, npre = 2 * (size_t)c->ep.K * sizeof(cf);
    const gfdm::HostOperand ops[3] = { { snr_lin, sizeof(float), sizeof(float), true }, { cnrs, A * sizeof(float), A * sizeof(float), true },
                                       { const_cast<float*>(rx_preamble), npre, npre, false } };
    auto fn = [&](void* const* d, int64_t nf, hipStream_t s) {
        return status_of(gfdm::launch_estimate_snr(c->ep, static_cast<float*>(d[0]), static_cast<float*>(d[1]), static_cast<const cf*>(d[2]), nf, s));
    };
    return pl.pipe.run(pl.stream, ops, 3, nframes, fn);
}

}  // extern "C"

// ---- receivers that estimate the channel themselves: preamble_channel_estimator_cc fused in front of the receiver kernel ----

namespace {

int est_attach(Plan& pl, const gfdm_hip_channel_estimator*& slot, const gfdm_hip_channel_estimator* c)
{
    if (c && (c->ep.M != pl.dp.M || c->ep.K != pl.dp.K)) {
        char buf[200];
        snprintf(buf, sizeof buf, "estimator is for timeslots %d x fft_len %d, the receiver for timeslots %d x subcarriers %d", c->ep.M, c->ep.K,
                 pl.dp.M, pl.dp.K);
        return fail(GFDM_HIP_EINVAL, buf);
    }
    if (c && c->plan.device != pl.device) return fail(GFDM_HIP_EINVAL, "estimator and receiver live on different devices");
    if (c && pl.current_family() == gfdm::FAMILY_ROWLANE_JIT) {
        // the preamble-equalised receive kernels of a run-time instantiated shape: load them now rather than in the first call -- in the
        // foreground when that is quick (cached, few timeslots, or gfdm_hip_set_jit(1)), else on the background pool, the estimated calls running
        // on the generic family until they are there (a handle whose other parts came from gfdm_hip_precompile must not block for a compile here)
        const int M = pl.dp.M, K = pl.dp.K, L = pl.dp.L;
        const int mode = g_jit.load();
        if (mode == 1 || M <= 16 || gfdm::jit_cached(M, K, L, gfdm::JIT_PART_RX_PREAMBLE) || !gfdm::generic_supports(M, K, false)) {
            std::string why;
            DeviceGuard guard(pl.device);
            if (!gfdm::jit_prepare(M, K, L, 1u << gfdm::JIT_PART_RX_PREAMBLE, why))
                return fail(GFDM_HIP_EHIP, "run-time instantiation of the preamble-equalised receive kernels failed: " + why);
            pl.jit_pre_pending.reset();
        } else if (!pl.jit_pre_pending) {
            pl.jit_pre_pending = std::make_shared<std::atomic<int>>(0);
            gfdm::jit_prepare_async(M, K, L, 1u << gfdm::JIT_PART_RX_PREAMBLE, pl.device, pl.jit_pre_pending);
        }
    }
    slot = c;
    return GFDM_HIP_OK;
}

// I/O of one estimated call: the frame configuration if there is one, plain blocks otherwise
int est_call_io(const FrameIo& f, const Plan& pl, const gfdm_hip_channel_estimator* c, int preamble_stride, int noutput_size, gfdm::RxIo& io,
                gfdm::EstPlan& ep)
{
    if (!c) return fail(GFDM_HIP_EINVAL, "set_channel_estimator has not been called on this handle");
    if (preamble_stride != 0 && preamble_stride < 2 * pl.dp.K) return fail(GFDM_HIP_EINVAL, "preamble_stride must be 0 (packed) or at least 2 * fft_len");
    if (f.configured) {
        int rc = frame_io_for_call(f, pl, noutput_size, io);
        if (rc != GFDM_HIP_OK) return rc;
    } else {
        if (noutput_size > 0 && noutput_size != pl.dp.N) return fail(GFDM_HIP_EINVAL, "noutput_size needs configure_frames with a subcarrier map");
        io.in_stride = pl.dp.N;
        io.nout = pl.dp.N;
    }
    ep = c->ep;
    ep.pre_stride = preamble_stride ? preamble_stride : 2 * pl.dp.K;
    return GFDM_HIP_OK;
}

template <typename DeviceCall>
int est_call_host(Plan& pl, const gfdm::RxIo& io, const gfdm::EstPlan& ep, float* out, const float* in, const float* rx_preamble, int64_t nblocks,
                  DeviceCall call)
{
    if (nblocks < 0) return fail(GFDM_HIP_EINVAL, "negative block count");
    if (!rx_preamble) return fail(GFDM_HIP_EINVAL, "NULL buffer");
    return run_host_sized(pl, out, (size_t)io.nout, in, (size_t)io.in_stride, rx_preamble, (size_t)ep.pre_stride, 2 * (size_t)ep.K, nblocks,
                          [&](cf* o, const cf* i, const cf* e, int64_t nb, hipStream_t s) { return call(o, i, e, nb, (void*)s); });
}

}  // namespace

extern "C" {

namespace {
int io_layout(const FrameIo& f, const Plan& pl, const gfdm_hip_channel_estimator* c, int estimated, int noutput_size, int* n_in, int* n_out,
              int* est_fft_len)
{
    gfdm::RxIo io{};
    int rc;
    if (estimated) {
        gfdm::EstPlan ep;
        rc = est_call_io(f, pl, c, 0, noutput_size, io, ep);
    } else {
        rc = frame_io_for_call(f, pl, noutput_size, io);
    }
    if (rc != GFDM_HIP_OK) return rc;
    if (n_in) *n_in = io.in_stride;
    if (n_out) *n_out = io.nout;
    if (est_fft_len) *est_fft_len = c ? c->ep.K : 0;
    return GFDM_HIP_OK;
}
}  // namespace

int gfdm_hip_receiver_io_layout(const gfdm_hip_receiver* r, int estimated, int noutput_size, int* n_in, int* n_out, int* est_fft_len)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return io_layout(r->frames, r->plan, r->est, estimated, noutput_size, n_in, n_out, est_fft_len);
}

int gfdm_hip_advanced_receiver_io_layout(const gfdm_hip_advanced_receiver* a, int estimated, int noutput_size, int* n_in, int* n_out,
                                         int* est_fft_len)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return io_layout(a->frames, a->plan, a->est, estimated, noutput_size, n_in, n_out, est_fft_len);
}

int gfdm_hip_receiver_set_channel_estimator(gfdm_hip_receiver* r, const gfdm_hip_channel_estimator* c)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return est_attach(r->plan, r->est, c);
}

int gfdm_hip_advanced_receiver_set_channel_estimator(gfdm_hip_advanced_receiver* a, const gfdm_hip_channel_estimator* c)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return est_attach(a->plan, a->est, c);
}

int gfdm_hip_receiver_demodulate_estimated_device(gfdm_hip_receiver* r, void* out, const void* in, const void* rx_preamble, int preamble_stride,
                                                  int noutput_size, int64_t nblocks, void* stream)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (!rx_preamble) return fail(GFDM_HIP_EINVAL, "NULL buffer");
    gfdm::IcParams ic = kNoIc;
    gfdm::EstPlan ep;
    int rc = est_call_io(r->frames, r->plan, r->est, preamble_stride, noutput_size, ic.io, ep);
    if (rc != GFDM_HIP_OK) return rc;
    return run_device(r->plan, out, in, nblocks, [&]() {
        return rx_launch(r->plan, ic, gfdm::RX_DEMOD, (cf*)out, (const cf*)in, (const cf*)rx_preamble, nblocks, (hipStream_t)stream, &ep);
    });
}

int gfdm_hip_advanced_receiver_work_estimated_device(gfdm_hip_advanced_receiver* a, void* out, const void* in, const void* rx_preamble,
                                                     int preamble_stride, int noutput_size, int64_t nblocks, void* stream)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    if (!rx_preamble) return fail(GFDM_HIP_EINVAL, "NULL buffer");
    gfdm::IcParams ic = a->ic;
    gfdm::EstPlan ep;
    int rc = est_call_io(a->frames, a->plan, a->est, preamble_stride, noutput_size, ic.io, ep);
    if (rc != GFDM_HIP_OK) return rc;
    return run_device(a->plan, out, in, nblocks, [&]() {
        return rx_launch(a->plan, ic, gfdm::RX_IC, (cf*)out, (const cf*)in, (const cf*)rx_preamble, nblocks, (hipStream_t)stream, &ep);
    });
}

int gfdm_hip_receiver_demodulate_estimated_host(gfdm_hip_receiver* r, float* out, const float* in, const float* rx_preamble, int preamble_stride,
                                                int noutput_size, int64_t nblocks)
{
    if (!r) return fail(GFDM_HIP_EINVAL, "NULL handle");
    gfdm::RxIo io{};
    gfdm::EstPlan ep;
    int rc = est_call_io(r->frames, r->plan, r->est, preamble_stride, noutput_size, io, ep);
    if (rc != GFDM_HIP_OK) return rc;
    return est_call_host(r->plan, io, ep, out, in, rx_preamble, nblocks, [&](cf* o, const cf* i, const cf* e, int64_t nb, void* s) {
        return gfdm_hip_receiver_demodulate_estimated_device(r, o, i, e, preamble_stride, noutput_size, nb, s);
    });
}

int gfdm_hip_advanced_receiver_work_estimated_host(gfdm_hip_advanced_receiver* a, float* out, const float* in, const float* rx_preamble,
                                                   int preamble_stride, int noutput_size, int64_t nblocks)
{
    if (!a) return fail(GFDM_HIP_EINVAL, "NULL handle");
    gfdm::RxIo io{};
    gfdm::EstPlan ep;
    int rc = est_call_io(a->frames, a->plan, a->est, preamble_stride, noutput_size, io, ep);
    if (rc != GFDM_HIP_OK) return rc;
    return est_call_host(a->plan, io, ep, out, in, rx_preamble, nblocks, [&](cf* o, const cf* i, const cf* e, int64_t nb, void* s) {
        return gfdm_hip_advanced_receiver_work_estimated_device(a, o, i, e, preamble_stride, noutput_size, nb, s);
    });
}

}  // extern "C"
