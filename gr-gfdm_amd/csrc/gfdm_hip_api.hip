// C-ABI shim (include/gfdm_hip.h): handle management, table construction, host staging and
// dispatch to the HIP kernel families.  No CPU compute path exists here by design: if the HIP
// device or a kernel launch is unavailable the call fails with an error code.
#include "../../include/gfdm_hip.h"
#include "gfdm_plan.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
    return GFDM_HIP_EHIP;
}

#define HIP_TRY(expr)                                            \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return fail_hip(_e, #expr);        \
    } while (0)

struct Plan {
    int device = 0;
    gfdm::DevicePlan dp{};
    std::vector<cf> h_taps, h_ictaps;
    cf* d_tables = nullptr;          // one allocation: taps | ictaps | wM | wK | wN
    hipStream_t stream = nullptr;    // private stream for the *_host entry points
    cf* stage[3] = { nullptr, nullptr, nullptr };
    size_t stage_elems[3] = { 0, 0, 0 };
    std::string kernel_name;
    const cf* d_twT = nullptr;       // [M][K] twiddles of the fast family
    int family = gfdm::FAMILY_GENERIC;

    ~Plan()
    {
        int prev = 0;
        bool restore = (hipGetDevice(&prev) == hipSuccess);
        (void)hipSetDevice(device);
        for (auto& s : stage) if (s) (void)hipFree(s);
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
int plan_create(Plan& pl, int M, int K, int L, const float* taps, int ntaps, int device, bool receiver)
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
    if (!gfdm::generic_supports(N, false)) return fail(GFDM_HIP_EUNSUPPORTED, "block (timeslots*subcarriers) does not fit LDS");

    pl.device = device;
    // energy |sum t conj(t)|, factor formed in double and cast (lib/modulator_kernel_cc.cc:75-85)
    double energy = 0.0;
    for (int i = 0; i < ntaps; ++i) energy += (double)taps[2 * i] * taps[2 * i] + (double)taps[2 * i + 1] * taps[2 * i + 1];
    if (!(energy > 0.0)) return fail(GFDM_HIP_EINVAL, "filter taps have zero energy");
    const float scale = (float)(1.0 / std::sqrt(std::fabs(energy) / M));
    pl.h_taps.resize(ntaps);
    for (int i = 0; i < ntaps; ++i) pl.h_taps[i] = make_float2(taps[2 * i] * scale, taps[2 * i + 1] * scale);
    pl.h_ictaps.assign(M, make_float2(0.f, 0.f));
    if (L >= 2)                                                    // lib/receiver_kernel_cc.cc:56-63
        for (int m = 0; m < M; ++m) {
            const cf a = pl.h_taps[m], b = pl.h_taps[M * (L - 1) + m];
            pl.h_ictaps[m] = make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
        }

    std::vector<cf> tables;
    tables.reserve((size_t)ntaps + 4 * M + K + 2 * (size_t)N);
    tables.insert(tables.end(), pl.h_taps.begin(), pl.h_taps.end());
    tables.insert(tables.end(), pl.h_ictaps.begin(), pl.h_ictaps.end());
    for (int m = 0; m < M; ++m) tables.push_back(make_float2(pl.h_ictaps[m].x / (float)M, pl.h_ictaps[m].y / (float)M));
    // g = IDFT_M(ic)/M in double: one IC round is d_new = d0 - g (*) (dec_{k-1} + dec_{k+1})  (gfdm_rowlane.hip)
    bool ic_real_sym = true;
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
    dp.ictaps = dp.taps + ntaps;
    dp.ictaps_m = dp.ictaps + M;
    dp.icg = dp.ictaps_m + M;
    dp.ic_real_sym = ic_real_sym ? 1 : 0;
    dp.wM = dp.icg + M;
    dp.wK = dp.wM + M;
    dp.wN = dp.wK + K;
    pl.d_twT = pl.d_tables + twT_off;
    // kernel family: row-lane where instantiated, then the 4-rows-per-lane family, else the generic LDS family.
    // GFDM_HIP_FAMILY=generic|fast|rowlane overrides the choice (experiments, A/B timing).
    pl.family = gfdm::rowlane_supports(M, K, L) ? gfdm::FAMILY_ROWLANE : gfdm::fast_supports(M, K, L) ? gfdm::FAMILY_FAST : gfdm::FAMILY_GENERIC;
    if (const char* want = getenv("GFDM_HIP_FAMILY")) {
        if (!strcmp(want, "generic")) pl.family = gfdm::FAMILY_GENERIC;
        else if (!strcmp(want, "fast") && gfdm::fast_supports(M, K, L)) pl.family = gfdm::FAMILY_FAST;
        else if (!strcmp(want, "rowlane") && gfdm::rowlane_supports(M, K, L)) pl.family = gfdm::FAMILY_ROWLANE;
    }
    pl.kernel_name = pl.family == gfdm::FAMILY_ROWLANE ? "rowlane" : pl.family == gfdm::FAMILY_FAST ? "fast_wave_tile" : "generic_lds";
    return GFDM_HIP_OK;
}

int ensure_stage(Plan& pl, int slot, size_t elems)
{
    if (pl.stage_elems[slot] >= elems) return GFDM_HIP_OK;
    if (pl.stage[slot]) { (void)hipFree(pl.stage[slot]); pl.stage[slot] = nullptr; pl.stage_elems[slot] = 0; }
    hipError_t e = hipMalloc(&pl.stage[slot], elems * sizeof(cf));
    if (e != hipSuccess) { g_last_error = "device staging buffer allocation failed"; return GFDM_HIP_ENOMEM; }
    pl.stage_elems[slot] = elems;
    return GFDM_HIP_OK;
}

// Host-pointer convenience path: H2D, launch, D2H, wait.  `launch(out, in0, in1, stream)` enqueues the kernels.
template <typename Launch>
int run_host(Plan& pl, float* out, const float* in0, const float* in1, int64_t nblocks, Launch launch)
{
    if (nblocks < 0 || out == nullptr || in0 == nullptr) return fail(GFDM_HIP_EINVAL, "NULL buffer or negative block count");
    if (nblocks == 0) return GFDM_HIP_OK;
    DeviceGuard guard(pl.device);
    if (!guard.ok) return fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    const size_t elems = (size_t)nblocks * (size_t)pl.dp.N;
    const size_t bytes = elems * sizeof(cf);
    int rc;
    if ((rc = ensure_stage(pl, 0, elems)) != GFDM_HIP_OK) return rc;
    if ((rc = ensure_stage(pl, 1, elems)) != GFDM_HIP_OK) return rc;
    if (in1 && (rc = ensure_stage(pl, 2, elems)) != GFDM_HIP_OK) return rc;
    HIP_TRY(hipMemcpyAsync(pl.stage[1], in0, bytes, hipMemcpyHostToDevice, pl.stream));
    if (in1) HIP_TRY(hipMemcpyAsync(pl.stage[2], in1, bytes, hipMemcpyHostToDevice, pl.stream));
    hipError_t e = launch(pl.stage[0], pl.stage[1], in1 ? pl.stage[2] : nullptr, pl.stream);
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    HIP_TRY(hipMemcpyAsync(out, pl.stage[0], bytes, hipMemcpyDeviceToHost, pl.stream));
    HIP_TRY(hipStreamSynchronize(pl.stream));
    return GFDM_HIP_OK;
}

template <typename Launch>
int run_device(Plan& pl, void* out, const void* in0, int64_t nblocks, Launch launch)
{
    if (nblocks < 0 || out == nullptr || in0 == nullptr) return fail(GFDM_HIP_EINVAL, "NULL buffer or negative block count");
    if (nblocks == 0) return GFDM_HIP_OK;
    DeviceGuard guard(pl.device);
    if (!guard.ok) return fail(GFDM_HIP_ENODEV, "hipSetDevice failed");
    hipError_t e = launch();
    if (e != hipSuccess) return fail_hip(e, "kernel launch");
    return GFDM_HIP_OK;
}

hipError_t rx_launch(Plan& pl, const gfdm::IcParams& ic, int mode, cf* out, const cf* in, const cf* f_eq, int64_t nblocks,
                            hipStream_t s)
{
    if (pl.family == gfdm::FAMILY_ROWLANE) return gfdm::launch_rowlane_receive(pl.dp, ic, pl.d_twT, mode, out, in, f_eq, nblocks, s);
    if (pl.family == gfdm::FAMILY_FAST) return gfdm::launch_fast_receive(pl.dp, ic, pl.d_twT, mode, out, in, f_eq, nblocks, s);
    return gfdm::launch_generic_receive(pl.dp, ic, mode, out, in, f_eq, nblocks, s);
}

hipError_t mod_launch(Plan& pl, cf* out, const cf* in, int64_t nblocks, hipStream_t s)
{
    if (pl.family == gfdm::FAMILY_ROWLANE) return gfdm::launch_rowlane_modulate(pl.dp, pl.d_twT, out, in, nblocks, s);
    if (pl.family == gfdm::FAMILY_FAST) return gfdm::launch_fast_modulate(pl.dp, pl.d_twT, out, in, nblocks, s);
    return gfdm::launch_generic_modulate(pl.dp, out, in, nblocks, s);
}

const gfdm::IcParams kNoIc = { 0, 0, 0, 0, nullptr, nullptr, 0, nullptr };

}  // namespace

struct gfdm_hip_modulator { Plan plan; };
struct gfdm_hip_receiver { Plan plan; };
struct gfdm_hip_advanced_receiver {
    Plan plan;
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

int gfdm_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* gfdm_hip_version(void) { return "gfdm_hip 0.1 (gfx950)"; }

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

const char* gfdm_hip_modulator_kernel_name(const gfdm_hip_modulator* m) { return m ? m->plan.kernel_name.c_str() : ""; }

int gfdm_hip_modulator_work_device(gfdm_hip_modulator* m, void* out, const void* in, int64_t nblocks, void* stream)
{
    if (!m) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_device(m->plan, out, in, nblocks, [&]() {
        return mod_launch(m->plan, (cf*)out, (const cf*)in, nblocks, (hipStream_t)stream);
    });
}

int gfdm_hip_modulator_work_host(gfdm_hip_modulator* m, float* out, const float* in, int64_t nblocks)
{
    if (!m) return fail(GFDM_HIP_EINVAL, "NULL handle");
    return run_host(m->plan, out, in, nullptr, nblocks, [&](cf* o, const cf* i, const cf*, hipStream_t s) {
        return mod_launch(m->plan, o, i, nblocks, s);
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

const char* gfdm_hip_receiver_kernel_name(const gfdm_hip_receiver* r) { return r ? r->plan.kernel_name.c_str() : ""; }

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
    return run_host(r->plan, out, in, f_eq, nblocks, [&](cf* o, const cf* i, const cf* e, hipStream_t s) {
        return rx_launch(r->plan, kNoIc, gfdm::RX_DEMOD, o, i, e, nblocks, s);
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
    return run_host(r->plan, out, in, f_eq, nblocks, [&](cf* o, const cf* i, const cf* e, hipStream_t s) {
        return rx_launch(r->plan, kNoIc, gfdm::RX_FD, o, i, e, nblocks, s);
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
    return run_host(r->plan, out, in, nullptr, nblocks, [&](cf* o, const cf* i, const cf*, hipStream_t s) {
        return gfdm::launch_generic_to_td(r->plan.dp, o, i, nblocks, s);
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
    return run_host(r->plan, out, td_in, fd_in, nblocks, [&](cf* o, const cf* i, const cf* e, hipStream_t s) {
        return gfdm::launch_generic_cancel(r->plan.dp, o, i, e, nblocks, s);
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
    int rc = plan_create(a->plan, timeslots, subcarriers, overlap, taps, ntaps, device, true);
    if (rc != GFDM_HIP_OK) { delete a; return rc; }

    const cf* pts = reinterpret_cast<const cf*>(constellation_points);
    if (decision == GFDM_HIP_DECIDE_AUTO) {
        const float s = 0.70710678118654752f, tol = 1e-6f;
        auto near = [&](cf p, float re, float im) { return std::fabs(p.x - re) < tol && std::fabs(p.y - im) < tol; };
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
const char* gfdm_hip_advanced_receiver_kernel_name(const gfdm_hip_advanced_receiver* a) { return a ? a->plan.kernel_name.c_str() : ""; }

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
    return run_host(a->plan, out, in, f_eq, nblocks, [&](cf* o, const cf* i, const cf* e, hipStream_t s) {
        return rx_launch(a->plan, a->ic, gfdm::RX_IC, o, i, e, nblocks, s);
    });
}

}  // extern "C"
