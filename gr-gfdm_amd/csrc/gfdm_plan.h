// Internal: device-side plan shared by the HIP kernel families and the C-ABI shim.
#pragma once
#ifndef __HIPCC_RTC__          // hiprtc (run-time instantiation of the row-lane kernels, gfdm_jit.hip) brings its own runtime header
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
typedef signed long long int64_t;
#endif

namespace gfdm {

typedef float2 cf;

// Everything a kernel needs to know about one (timeslots, subcarriers, overlap, taps) configuration.
// Tables live in device memory, are built once per handle (double precision on the host, rounded to f32).
struct DevicePlan {
    int M;          // timeslots
    int K;          // subcarriers
    int L;          // overlap
    int N;          // M * K
    int log2K;      // >= 0 when K is a power of two, else -1
    int part_len;   // min(M*L/2, M): bins of each tap part the modulator accumulates (modulator_kernel_cc.cc:101)
    const cf* taps;     // [L*M] normalised filter taps
    int taps_real;      // 1 when every tap is real (RRC / RC and every other real, even prototype filter): two multiply-adds per tap product
    const cf* ictaps;   // [M]   ic[m] = t[m] * t[(L-1)M + m]
    const cf* ictaps_m; // [M]   ic[m] / M (IC taps with the inverse-DFT scale folded in)
    const cf* icg;      // [M]   g = IDFT_M(ic) / M: circular-convolution form of one cancellation round
    int ic_real_sym;    // 1 when g is real and even (real, even prototype filter): only g[0..M/2].x is used
    // matrix-core form of a cancellation round (gfdm_rowlane_impl.h, ICK_MFMA; QPSK decisions, g real, M <= 16):
    const void* icA;    // [2][64 lanes][8 f16] A operands of v_mfma_f32_16x16x32_f16: the circulant -s g[(p - r) mod M] 2^e as three f16 terms
    unsigned ic_sig;    // f16 bits of 2^-e: magnitude of one QPSK decision in the B operand (0: no table, vector-ALU rounds only)
    const cf* wM;       // [M]   exp(-2 pi j p / M)
    const cf* wK;       // [K]   exp(-2 pi j q / K)
    const cf* wN;       // [N]   exp(-2 pi j r / N)
    // matrix-core form of the generic family's timeslot transforms (gfdm_generic.hip, mx_dft; M >= 32): the cosine / sine matrices of the paired
    // DFT as A operands of v_mfma_f32_16x16x4_f32, [dft_mt output tiles][dft_ks k-steps][2: cos, sin][64 lanes]; nullptr = vector-ALU form
    const float* dftA;
    int dft_mt, dft_ks;
    int dft_always;     // 1: wherever the operand scratch fits LDS (gfdm_hip_set_dft_matrix_cores(2)), 0: only where that form is the faster one
    // prime timeslot counts served by Rader transforms (gfdm_rader.hip): FFT_{M-1}(b) / (M - 1) of the convolution kernel b[p] = W_M^(g^-p), in the
    // order the kernels read it; nullptr = dense transforms (generic kernels)
    const cf* raderB;
};

// Where the receiver finds its samples and how it emits its symbols: the cyclic-prefix removal
// of add_cyclic_prefix_cc::remove_cyclic_prefix (lib/add_cyclic_prefix_cc.cc:100-104) is an input offset + stride, the
// resource demapper resource_mapper_kernel_cc::demap_from_resources (lib/resource_mapper_kernel_cc.cc:91-106,136-163)
// a gather in the store stage.  All zero = plain blocks in, plain [k][m] blocks out.
struct RxIo {
    int in_stride;             // samples between consecutive frames in the input (0 = block size N)
    int in_offset;             // samples to skip at the start of every frame (cyclic prefix length)
    int demap;                 // 1: emit only the active subcarriers, in mapper order
    int per_timeslot;          // demapper symbol order
    int A;                     // active subcarriers
    int nout;                  // symbols emitted per block (<= A * M) = output stride
    const short* rank;         // [K] position of subcarrier k in the sorted subcarrier map, -1 when inactive
};

// Interference-cancellation settings of advanced_receiver_kernel_cc (+ the I/O layout above).
struct IcParams {
    int ic_iter;
    int do_phase_compensation;
    int decision;              // gfdm_hip_decision (never AUTO on the device)
    int npoints;
    const cf* points;          // [npoints]
    const unsigned char* active;   // [K] how often the subcarrier occurs in subcarrier_map (0 = inactive)
    int n_active;              // subcarrier_map.size() (duplicates counted, as the reference does)
    const int* smap;           // [n_active] the subcarrier_map itself (order matters for the phase sum)
    RxIo io;
};

// the cancellation rounds run on the matrix cores (IcMfma, gfdm_rowlane_impl.h) when the handle carries the operand table (real even IC
// kernel, shape covered: gfdm_rowgeom.h ic_mfma) and the decisions are QPSK sign tests without phase compensation
inline bool ic_mfma_applies(const DevicePlan& p, const IcParams& ic)
{
    return p.ic_sig != 0 && p.icA != nullptr && ic.decision == 1 && ic.do_phase_compensation <= 0;
}

// from this many timeslots on the generic family's timeslot transforms run on the matrix cores (DevicePlan::dftA)
constexpr int MX_DFT_MIN_M = 32;

enum RxMode {
    RX_FD = 0,        // fft_[equalize_]filter_downsample: out = S
    RX_DEMOD = 1,     // generic_work[_equalize]:         out = IDFT_M(S)/M
    RX_IC = 2         // advanced receiver:               out after ic_iter cancellation rounds
};

#ifndef __HIPCC_RTC__
// ---- generic family (any M, K, L; one workgroup per block, everything staged in LDS) ----
size_t generic_lds_bytes(int N, int ntiles);
struct TxParams;   // gfdm_tx.h: resource mapper in front of / cyclic prefix + preamble behind the modulator
hipError_t launch_generic_modulate(const DevicePlan& p, const TxParams& tx, cf* out, const cf* in, int64_t nblocks, hipStream_t s);
// transmitter_kernel::add_frame: preamble + cyclic prefix/suffix + ramp of already modulated blocks, port 0 of tx
hipError_t launch_add_frame(const DevicePlan& p, const TxParams& tx, const cf* in, int64_t nblocks, hipStream_t s);
struct EstPlan;
hipError_t launch_generic_receive(const DevicePlan& p, const IcParams& ic, const EstPlan* est, int mode, cf* out, const cf* in, const cf* f_eq,
                                  int64_t nblocks, hipStream_t s);
hipError_t launch_generic_to_td(const DevicePlan& p, cf* out, const cf* in, int64_t nblocks, hipStream_t s);
hipError_t launch_generic_cancel(const DevicePlan& p, cf* out, const cf* td, const cf* fd, int64_t nblocks, hipStream_t s);
bool generic_supports(int M, int K, bool one_tile);
// ---- Rader transforms for prime timeslot counts above the register codelets (gfdm_rader.hip; plain blocks only, the generic launchers fall through to them) ----
bool rader_supports(int M, int K);
bool rader_applies_modulate(const DevicePlan& p, const TxParams& tx);
bool rader_applies_receive(const DevicePlan& p, const IcParams& ic, const EstPlan* est, int mode);
hipError_t launch_rader_modulate(const DevicePlan& p, cf* out, const cf* in, int64_t nblocks, hipStream_t s);
hipError_t launch_rader_receive(const DevicePlan& p, const IcParams& ic, int mode, cf* out, const cf* in, const cf* f_eq, int64_t nblocks, hipStream_t s);

// ---- row-lane family (gfdm_rowlane_impl.h, dispatch in gfdm_rowlane.hip): one lane per subcarrier row, in-place radix-4 passes ----
bool rowlane_supports(int M, int K, int L);
hipError_t launch_rowlane_modulate(const DevicePlan& p, const TxParams& tx, const cf* twT, cf* out, const cf* in, int64_t nblocks,
                                   hipStream_t s);
hipError_t launch_rowlane_receive(const DevicePlan& p, const IcParams& ic, const EstPlan* est, const cf* twT, int mode, cf* out, const cf* in,
                                  const cf* f_eq, int64_t nblocks, hipStream_t s);
// ... and the same kernels instantiated at run time (gfdm_jit.hip); jit_prepare* compile / load on the CURRENT device.
// JitCache: a handle's own pointers to the loaded parts, resolved at the first launch of each part -- after that a launch takes no lock
// and no map lookup (one acquire load).
}  // namespace gfdm
#include <atomic>
#include <memory>
#include <string>
#include <vector>
namespace gfdm {
void rader_host_table(int M, std::vector<cf>& tab);          // empty when the timeslot count has no Rader kernels
struct JitCache {
    std::atomic<const void*> part[5];
    JitCache() { for (auto& p : part) p.store(nullptr); }
};
hipError_t jit_launch_modulate(JitCache* cache, const DevicePlan& p, const TxParams& tx, const cf* twT, cf* out, const cf* in, int64_t nblocks, hipStream_t s);
hipError_t jit_launch_receive(JitCache* cache, const DevicePlan& p, const IcParams& ic, const EstPlan* est, const cf* twT, int mode, cf* out, const cf* in,
                              const cf* f_eq, int64_t nblocks, hipStream_t s);
hipError_t jit_launch_estimate(JitCache* cache, const EstPlan& e, cf* out, const cf* in, int64_t nframes, hipStream_t s);
bool jit_cached(int M, int K, int L, int part);               // the part's code object is in the disk cache
// compile / load on a background thread; *state: 0 running, 1 ready, -1 failed
void jit_prepare_async(int M, int K, int L, unsigned parts, int device, std::shared_ptr<std::atomic<int>> state);
// drop the queued background builds, let the ones in flight finish WITHOUT touching the GPU, wait for them (exit path; gfdm_hip_quiesce)
void jit_quiesce();
// error reporting shared by the translation units of the C-ABI (thread-local message behind gfdm_hip_last_error)
int api_fail(int code, const std::string& msg);
int api_fail_hip(hipError_t e, const char* what);
bool jit_eligible(int M, int K, int L);
// parts of a shape (one hiprtc program each): a handle prepares the ones its kind launches, the others load at first use
enum { JIT_PART_RX = 0, JIT_PART_RX_IC = 1, JIT_PART_RX_PREAMBLE = 2, JIT_PART_MOD = 3, JIT_PART_EST = 4, JIT_NUM_PARTS = 5 };
bool jit_prepare(int M, int K, int L, unsigned parts, std::string& err);      // parts: bit p = part p
bool jit_prepare_estimate(int M, int K, std::string& err);
bool jit_build_only(int M, int K, int L, int part, std::string& err);

#endif  // !__HIPCC_RTC__

// Preamble channel estimator (lib/preamble_channel_estimator_cc.cc): tables of one estimator handle.
struct EstPlan {
    int M, K, A;          // timeslots, fft_len (subcarriers), active subcarriers
    int dc_free;
    int n_est;            // A + dc_free: bins of the smoothed estimate
    int log2K, log2K2;    // log2 of K and 2K, or -1: direct DFT
    const cf* inv0;       // [K]  0.5 / FFT_K(preamble first half)
    const cf* inv1;       // [K]  0.5 / FFT_K(preamble second half)
    const cf* wK;         // [K]  exp(-2 pi i j / K)
    const cf* w2K;        // [2K] exp(-2 pi i j / 2K)
    float gauss[9];       // normalised Gaussian smoothing taps (sigma^2 = 1)
    int pre_stride;       // receivers with EQ_PREAMBLE: samples between the preambles of successive blocks (0 = packed, 2K)
};

// how a receiver launch gets its one-tap equaliser
enum EqSource {
    EQ_NONE = 0,
    EQ_VECTOR = 1,        // f_eq: N bins per block (generic_work_equalize)
    EQ_PREAMBLE = 2       // f_eq points at the received preambles; the kernel runs the channel estimator itself
};

#ifndef __HIPCC_RTC__
// stages of the estimator chain; a launch runs in_stage -> out_stage inside one kernel
enum EstStage { EST_RX_PREAMBLE = 0, EST_PREAMBLE_CHANNEL = 1, EST_FILTERED = 2, EST_FRAME = 3 };
size_t estimator_lds_bytes(int K);
bool estimator_supports(int K);
hipError_t launch_estimate(const EstPlan& e, int in_stage, int out_stage, cf* out, const cf* in, int64_t nframes, hipStream_t s);
hipError_t launch_estimate_snr(const EstPlan& e, float* snr, float* cnrs, const cf* in, int64_t nframes, hipStream_t s);
hipError_t launch_prepare_for_zf(cf* out, const cf* in, int64_t n, hipStream_t s);
bool rowlane_supports_estimate(int M, int K);
hipError_t launch_rowlane_estimate(const EstPlan& e, cf* out, const cf* in, int64_t nframes, hipStream_t s);   // rx preamble -> frame

#endif  // !__HIPCC_RTC__

// FAMILY_ROWLANE_JIT: the row-lane kernels instantiated at run time through hiprtc for a shape outside the compiled list (gfdm_jit.hip)
enum KernelFamily { FAMILY_GENERIC = 0, FAMILY_ROWLANE = 2, FAMILY_ROWLANE_JIT = 3 };

}  // namespace gfdm
