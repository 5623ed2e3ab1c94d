// TEST-ONLY driver of the sanitizer builds (tests/sanitize/Makefile, tests/test_sanitizers.py): the call mix of scratch/fuzz_host_path.py -- ragged block counts,
// every route / chunk size / depth / copy-thread / stream setting, pageable, registered, partly registered, adjacent-registration, pinned and device operands,
// in-place calls -- on the product's host-side code (gfdm_hip_api.hip, gfdm_hostpipe.hip, gfdm_jit.hip, the C++ classes, sharded_batch.h, batched_work.h),
// compiled unchanged against the loop-back HIP layer of tests/sanitize/loopback, with
//   * several threads driving handles of their own at once, another one calling gfdm_hip_quiesce() while their calls are in flight,
//   * run-time instantiated shapes building in the background (loop-back hiprtc) while their handles are used and destroyed,
//   * a sharded batch (one host thread per "device") and the GNU Radio wrappers' batched work() bodies,
//   * every runtime call failing once (injected) with the call after it required to work,
//   * the process ending with background builds in flight.
// Every result is compared bit for bit with what the loop-back kernels must have produced (loopback_transform.h).
//   host_fuzz <seconds> [seed] [threads]
#include <gfdm_hip.h>
#include <gfdm/advanced_receiver_kernel_cc.h>
#include <gfdm/batched_work.h>
#include <gfdm/host_memory.h>
#include <gfdm/modulator_kernel_cc.h>
#include <gfdm/receiver_kernel_cc.h>
#include <gfdm/sharded_batch.h>

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include "loopback_transform.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <unistd.h>

using loopback::c2;
typedef std::complex<float> cfl;

namespace {

std::atomic<long> g_calls{ 0 }, g_failures{ 0 }, g_direct_ops{ 0 }, g_chunked{ 0 }, g_jit_tagged{ 0 }, g_generic_tagged{ 0 };
std::atomic<bool> g_stop{ false };

#define CHECK(cond, ...)                                                                      \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            fprintf(stderr, "host_fuzz FAILED %s:%d: %s -- ", __FILE__, __LINE__, #cond);     \
            fprintf(stderr, __VA_ARGS__);                                                     \
            fprintf(stderr, " [last error: %s]\n", gfdm_hip_last_error());                    \
            g_failures.fetch_add(1);                                                          \
            abort();                                                                          \
        }                                                                                     \
    } while (0)

const size_t kPage = 4096;

// injected runtime failures (failure_sweep) hit the PRODUCT's calls only, not the driver's own allocations
template <class F> int product(F f) { loopback::arm(true); const int rc = f(); loopback::arm(false); return rc; }

// ---- operand memory of every kind the host path distinguishes -------------------------------------------------------------------------------
enum Kind { PAGEABLE, REGISTERED, REG_INSIDE, REG_SPAN2, REG_PARTIAL, PINNED, DEVICE, NUM_KINDS };

struct Buf {
    Kind kind = PAGEABLE;
    float* p = nullptr;              // what the call gets
    size_t bytes = 0;
    void* base = nullptr;            // what is freed
    size_t base_bytes = 0;
    std::vector<std::pair<void*, size_t>> regs;

    Buf() = default;
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;

    void alloc(Kind k, size_t nbytes, std::mt19937& rng)
    {
        kind = k;
        bytes = nbytes ? nbytes : 8;
        const size_t pages = (bytes + kPage - 1) / kPage;
        auto map = [&](size_t n) {
            void* m = mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            CHECK(m != MAP_FAILED, "mmap of %zu bytes", n);
            return m;
        };
        auto reg = [&](void* q, size_t n) {
            CHECK(gfdm_hip_register_host(q, n) == GFDM_HIP_OK, "register %p + %zu", q, n);
            regs.emplace_back(q, n);
        };
        switch (k) {
        case PAGEABLE:                                     // malloc'ed to the byte: an overrun by the host path is an ASan report
            base = malloc(bytes); base_bytes = bytes; p = static_cast<float*>(base);
            break;
        case REGISTERED:
            base_bytes = pages * kPage; base = map(base_bytes); p = static_cast<float*>(base);
            reg(base, base_bytes);
            break;
        case REG_INSIDE: {                                 // a block somewhere inside a larger registered range (a scheduler's circular buffer)
            base_bytes = (pages + 3) * kPage; base = map(base_bytes);
            reg(base, base_bytes);
            const size_t off = (size_t)(rng() % (3 * kPage / 8)) * 8;
            p = reinterpret_cast<float*>(static_cast<char*>(base) + off);
            break;
        }
        case REG_SPAN2: {                                  // two registrations side by side, the operand across both: not ONE registration -> bounced
            base_bytes = (pages + 1) * kPage; base = map(base_bytes);
            const size_t first = ((pages + 1) / 2) * kPage;
            reg(base, first);
            reg(static_cast<char*>(base) + first, base_bytes - first);
            p = reinterpret_cast<float*>(static_cast<char*>(base) + (kPage - 64 < first ? kPage - 64 : 0));
            if (reinterpret_cast<char*>(p) + bytes > static_cast<char*>(base) + base_bytes) p = static_cast<float*>(base);
            break;
        }
        case REG_PARTIAL: {                                // begins in registered pages, ends in pages nobody registered
            base_bytes = (pages + 2) * kPage; base = map(base_bytes);
            reg(base, kPage);
            p = reinterpret_cast<float*>(static_cast<char*>(base) + kPage - 128);
            break;
        }
        case PINNED:
            CHECK(hipHostMalloc(&base, bytes, hipHostMallocMapped) == hipSuccess, "hipHostMalloc");
            base_bytes = bytes; p = static_cast<float*>(base);
            break;
        case DEVICE:
            CHECK(hipMalloc(&base, bytes) == hipSuccess, "hipMalloc");
            base_bytes = bytes; p = static_cast<float*>(base);
            break;
        default: abort();
        }
    }
    ~Buf()
    {
        for (auto& r : regs) CHECK(gfdm_hip_unregister_host(r.first) == GFDM_HIP_OK, "unregister %p", r.first);
        switch (kind) {
        case PAGEABLE: free(base); break;
        case PINNED: if (base) (void)hipHostFree(base); break;
        case DEVICE: if (base) (void)hipFree(base); break;
        default: if (base) munmap(base, base_bytes);
        }
    }
    bool in_place_capable() const { return kind == REGISTERED || kind == REG_INSIDE || kind == PINNED || kind == DEVICE; }
};

inline c2 value_at(unsigned seed, int64_t i) { return c2{ (float)((int)((seed * 2654435761u + (unsigned)i * 40503u) >> 16 & 0x3ff) - 512) * 0.125f, (float)((int)(i % 977) - 488) * 0.25f }; }
void fill(Buf& b, unsigned seed, int64_t ncomplex)
{
    for (int64_t i = 0; i < ncomplex; ++i) { const c2 v = value_at(seed, i); b.p[2 * i] = v.x; b.p[2 * i + 1] = v.y; }
}
void poison(Buf& b, int64_t ncomplex)
{
    for (int64_t i = 0; i < 2 * ncomplex; ++i) b.p[i] = NAN;
}
inline c2 at(const Buf& b, int64_t i) { return c2{ b.p[2 * i], b.p[2 * i + 1] }; }
inline bool same(c2 a, c2 b) { return memcmp(&a, &b, sizeof a) == 0; }

// ---- shapes -------------------------------------------------------------------------------------------------------------------------------------
struct Shape { int M, K, L; int kind; };      // kind 0 compiled row-lane, 1 run-time instantiated (quick: constructor), 2 run-time instantiated (background), 3 generic
const Shape kShapes[] = { { 9, 64, 2, 0 }, { 5, 32, 2, 0 }, { 15, 128, 4, 0 }, { 7, 12, 2, 1 }, { 6, 16, 2, 1 }, { 21, 64, 2, 2 }, { 25, 32, 4, 2 },
                          { 33, 16, 2, 2 }, { 21, 37, 2, 3 }, { 127, 16, 2, 3 } };
const int kNumShapes = sizeof kShapes / sizeof kShapes[0];

std::vector<float> taps_for(const Shape& s)
{
    std::vector<float> t((size_t)2 * s.M * s.L);
    for (size_t i = 0; i < t.size() / 2; ++i) { t[2 * i] = 1.f / (1.f + (float)i); t[2 * i + 1] = 0.f; }
    return t;
}

// which tags a handle of this shape may produce right now, and the tag ALL blocks of one call carried
float expect_uniform_tag(const Shape& s, c2 got, c2 want_untagged)
{
    const float d = got.x - want_untagged.x;
    (void)s;
    return d;
}
bool tag_allowed(const Shape& s, float tag)
{
    if (s.kind == 0) return tag == loopback::kTagRowlane;
    if (s.kind == 3) return tag == loopback::kTagGeneric;
    return tag == loopback::kTagJit || tag == loopback::kTagGeneric;      // a background build may or may not have finished; a failed one stays generic
}

void random_pipeline(std::mt19937& rng, int64_t block_bytes)
{
    static const int64_t chunk_choice[] = { 0, 0, 1, -1 /* 8 blocks */, -2 /* 3 blocks + 5 */, 1 << 20, 1 << 30 };
    int64_t chunk = chunk_choice[rng() % 7];
    if (chunk == -1) chunk = 8 * block_bytes;
    if (chunk == -2) chunk = 3 * block_bytes + 5;
    CHECK(gfdm_hip_set_host_pipeline((int)(rng() % 4), chunk, 1 + (int)(rng() % 4), (int)(rng() % 4), 1 + (int)(rng() % 2)) == GFDM_HIP_OK, "set_host_pipeline");
    (void)gfdm_hip_set_host_streaming_copies_for_testing((int)(rng() % 2));
}

Kind pick_kind(std::mt19937& rng) { return (Kind)(rng() % NUM_KINDS); }

void note_stats()
{
    int64_t chunks = 0, cb = 0, staged = 0; unsigned mask = 0; int mode = 0, ct = 0;
    gfdm_hip_host_call_stats(&chunks, &cb, &staged, &mask, &mode, &ct);
    g_direct_ops.fetch_add(__builtin_popcount(mask));
    if (chunks > 1) g_chunked.fetch_add(1);
}

// ---- one worker: handles of its own, random calls ---------------------------------------------------------------------------------------------
struct Handles {
    Shape s{};
    gfdm_hip_modulator* mod = nullptr;
    gfdm_hip_receiver* rx = nullptr;
    gfdm_hip_advanced_receiver* adv = nullptr;
    gfdm_hip_transmitter* tx = nullptr;
    gfdm_hip_channel_estimator* est = nullptr;
    int device = 0, cp = 0, A = 0, ic_iter = 2, F = 0, nports = 0;
    bool frames = false, estimated = false;
    ~Handles()
    {
        if (mod) gfdm_hip_modulator_destroy(mod);
        if (rx) gfdm_hip_receiver_destroy(rx);
        if (adv) gfdm_hip_advanced_receiver_destroy(adv);
        if (tx) gfdm_hip_transmitter_destroy(tx);
        if (est) gfdm_hip_channel_estimator_destroy(est);
    }
};

std::unique_ptr<Handles> make_handles(std::mt19937& rng, bool allow_failure)
{
    auto h = std::make_unique<Handles>();
    h->s = kShapes[rng() % kNumShapes];
    const Shape& s = h->s;
    h->device = (int)(rng() % 2);
    const std::vector<float> taps = taps_for(s);
    const int nt = s.M * s.L;
    int rc = product([&] { return gfdm_hip_modulator_create(&h->mod, s.M, s.K, s.L, taps.data(), nt, h->device); });
    if (rc != GFDM_HIP_OK) { CHECK(allow_failure, "modulator_create %d", rc); return nullptr; }
    rc = product([&] { return gfdm_hip_receiver_create(&h->rx, s.M, s.K, s.L, taps.data(), nt, h->device); });
    if (rc != GFDM_HIP_OK) { CHECK(allow_failure, "receiver_create %d", rc); return nullptr; }
    std::vector<int> smap;
    h->A = s.K - 2 * (s.K / 8);
    for (int k = 0; k < s.K && (int)smap.size() < h->A; ++k) smap.push_back((k * 5 + 1) % s.K == 0 ? 0 : k);
    smap.clear();
    for (int k = 1; k <= h->A; ++k) smap.push_back(k % s.K);
    const float qpsk[8] = { -0.70710678f, -0.70710678f, 0.70710678f, -0.70710678f, -0.70710678f, 0.70710678f, 0.70710678f, 0.70710678f };
    h->ic_iter = (int)(rng() % 3);
    rc = product([&] { return gfdm_hip_advanced_receiver_create(&h->adv, s.M, s.K, s.L, taps.data(), nt, smap.data(), (int)smap.size(), h->ic_iter, qpsk, 4, GFDM_HIP_DECIDE_AUTO,
                                           (int)(rng() % 2), h->device); });
    if (rc != GFDM_HIP_OK) { CHECK(allow_failure, "advanced_receiver_create %d", rc); return nullptr; }
    CHECK(allow_failure || gfdm_hip_advanced_receiver_decision(h->adv) == GFDM_HIP_DECIDE_QPSK, "QPSK points recognised");
    if (rng() % 2) {                                       // raw frames in (cyclic prefix skipped), demapped symbols out
        h->cp = (int)(rng() % 9);
        const int N = s.M * s.K;
        h->frames = true;
        h->F = N + h->cp + 3;
        std::vector<int> sorted(smap);
        rc = product([&] { return gfdm_hip_receiver_configure_frames(h->rx, h->F, h->cp, sorted.data(), h->A, (int)(rng() % 2)); });
        if (rc == GFDM_HIP_OK) rc = product([&] { return gfdm_hip_advanced_receiver_configure_frames(h->adv, h->F, h->cp, sorted.data(), h->A, (int)(rng() % 2)); });
        if (rc != GFDM_HIP_OK) { CHECK(allow_failure, "configure_frames %d", rc); return nullptr; }
    }
    if (rng() % 2) {                                       // composite transmitter, up to three ports
        h->nports = 1 + (int)(rng() % 3);
        const int N = s.M * s.K, cp = 4, cs = 2, ramp = 2, plen = 2 * s.K;
        std::vector<float> window((size_t)2 * (N + cp + cs), 1.f), pre((size_t)2 * h->nports * plen, 0.5f);
        std::vector<int> shifts;
        for (int i = 0; i < h->nports; ++i) shifts.push_back(i % (cs + 1));
        rc = product([&] { return gfdm_hip_transmitter_create(&h->tx, s.M, s.K, h->A, cp, cs, ramp, smap.data(), h->A, 0, s.L, taps.data(), nt, window.data(), N + cp + cs, shifts.data(),
                                         h->nports, pre.data(), plen, h->device); });
        if (rc != GFDM_HIP_OK) { CHECK(allow_failure, "transmitter_create %d", rc); return nullptr; }
    }
    if (rng() % 3 == 0) {
        std::vector<float> pre((size_t)4 * s.K);
        for (size_t i = 0; i < pre.size(); ++i) pre[i] = 1.f + (float)(i % 7);
        const int A = (h->A & ~1) >= 2 ? (h->A & ~1) : 2;
        rc = product([&] { return gfdm_hip_channel_estimator_create(&h->est, s.M, s.K, A > s.K - 1 ? ((s.K - 1) & ~1) : A, 1, 1, pre.data(), 2 * s.K, h->device); });
        if (rc != GFDM_HIP_OK) { CHECK(allow_failure, "channel_estimator_create %d: %s", rc, gfdm_hip_last_error()); return nullptr; }
    }
    if (h->est && rng() % 2) {                             // receivers that estimate the channel themselves (run-time instantiated shapes: their preamble-equalised
        rc = product([&] { return gfdm_hip_receiver_set_channel_estimator(h->rx, h->est); });            // kernels may come from the background pool)
        if (rc == GFDM_HIP_OK) rc = product([&] { return gfdm_hip_advanced_receiver_set_channel_estimator(h->adv, h->est); });
        if (rc != GFDM_HIP_OK) { CHECK(allow_failure, "set_channel_estimator %d", rc); return nullptr; }
        h->estimated = true;
    }
    return h;
}

// one random host call on the handles; returns false when the call reported an error (only legal while failures are being injected)
bool one_call(Handles& h, std::mt19937& rng, bool allow_failure)
{
    const Shape& s = h.s;
    const int N = s.M * s.K;
    static const int nb_small[] = { 1, 1, 2, 3, 5, 8, 13, 33, 64, 100, 257 };
    int64_t nb = nb_small[rng() % (N > 2000 ? 7 : 11)];
    random_pipeline(rng, (int64_t)N * 8);
    CHECK(hipSetDevice(h.device) == hipSuccess, "hipSetDevice");      // device operands must live on the handle's GPU (another GPU's memory is refused: tested below)
    const unsigned seed = (unsigned)rng();
    int which = (int)(rng() % 11);
    if (which == 10 && !h.estimated) which = 4;
    if (which == 7 && !h.tx) which = 0;
    if (which == 8 && !h.est) which = 1;
    if ((which == 5 || which == 6) && !h.frames) which = 2 + (int)(rng() % 2);
    const bool with_eq = rng() % 2;
    Buf out, in0, in1;
    int rc = GFDM_HIP_OK;
    auto check_rx = [&](int mode, int rounds, int64_t in_stride, int in_off, int nout, const Buf* e, int64_t e_stride = -1, int e_mod = 0) {
        if (e_stride < 0) { e_stride = N; e_mod = N; }          // the equaliser vector: N bins per block; a preamble: 2 K samples at its stride
        float tag = 0.f;
        for (int64_t b = 0; b < nb; ++b)
            for (int i = 0; i < nout; ++i) {
                const c2 sv = at(in0, b * in_stride + in_off + (i % N));
                const c2 ev = e ? at(*e, b * e_stride + (i % e_mod)) : c2{ 0.f, 0.f };
                const c2 got = at(out, b * nout + i);
                if (b == 0 && i == 0) {
                    tag = -1.f;
                    for (float t : { loopback::kTagRowlane, loopback::kTagJit, loopback::kTagGeneric })
                        if (same(got, loopback::rx_value(sv, ev, mode, rounds, t))) tag = t;
                    CHECK(tag_allowed(s, tag), "family tag %g of shape M%d K%d L%d kind %d: call %d mode %d rounds %d eq %d nb %ld stride %ld off %d nout %d: got (%g, %g) from (%g, %g) eq (%g, %g)", tag,
                          s.M, s.K, s.L, s.kind, which, mode, rounds, e != nullptr, (long)nb, (long)in_stride, in_off, nout, got.x, got.y, sv.x, sv.y, ev.x, ev.y);
                }
                const c2 want = loopback::rx_value(sv, ev, mode, rounds, tag);
                CHECK(same(got, want), "rx mode %d shape M%d K%d block %ld of %ld element %d: got (%g, %g) want (%g, %g)", mode, s.M, s.K, (long)b, (long)nb, i, got.x, got.y,
                      want.x, want.y);
            }
        (tag == loopback::kTagJit ? g_jit_tagged : tag == loopback::kTagGeneric ? g_generic_tagged : g_calls).fetch_add(tag == loopback::kTagRowlane ? 0 : 1);
    };
    switch (which) {
    case 0: {                                              // modulator; sometimes in place
        in0.alloc(pick_kind(rng), (size_t)nb * N * 8, rng); fill(in0, seed, nb * N);
        const bool inplace = rng() % 4 == 0;
        if (!inplace) { out.alloc(pick_kind(rng), (size_t)nb * N * 8, rng); poison(out, nb * N); }
        std::vector<c2> keep;
        if (inplace) for (int64_t i = 0; i < nb * N; ++i) keep.push_back(at(in0, i));
        rc = product([&] { return gfdm_hip_modulator_work_host(h.mod, inplace ? in0.p : out.p, in0.p, nb); });
        if (inplace && in0.kind == DEVICE && !allow_failure) {      // the one in-place form the host path refuses (it could not bounce the output)
            CHECK(rc == GFDM_HIP_EINVAL, "in-place call on device memory returned %d", rc);
            rc = GFDM_HIP_OK;
            break;
        }
        if (rc != GFDM_HIP_OK) break;
        const Buf& o = inplace ? in0 : out;
        float tag = 0.f;
        for (int64_t i = 0; i < nb * N; ++i) {
            const c2 sv = inplace ? keep[(size_t)i] : at(in0, i);
            if (i == 0) { tag = at(o, 0).x - loopback::mod_value(sv, 0.f).x; CHECK(tag_allowed(s, tag), "modulator family tag %g", tag); }
            CHECK(same(at(o, i), loopback::mod_value(sv, tag)), "modulate%s element %ld", inplace ? " in place" : "", (long)i);
        }
        break;
    }
    case 1: case 2: case 3: case 4: {                      // receiver demodulate / fft_filter_downsample, advanced receiver; with or without the equaliser vector
        in0.alloc(pick_kind(rng), (size_t)nb * N * 8, rng); fill(in0, seed, nb * N);
        if (with_eq) { in1.alloc(pick_kind(rng), (size_t)nb * N * 8, rng); fill(in1, seed ^ 0x5a5a, nb * N); }
        out.alloc(pick_kind(rng), (size_t)nb * N * 8, rng); poison(out, nb * N);
        const float* e = with_eq ? in1.p : nullptr;
        if (which == 1) rc = product([&] { return gfdm_hip_receiver_demodulate_host(h.rx, out.p, in0.p, e, nb); });
        else if (which == 2) rc = product([&] { return gfdm_hip_receiver_fft_filter_downsample_host(h.rx, out.p, in0.p, e, nb); });
        else rc = product([&] { return gfdm_hip_advanced_receiver_work_host(h.adv, out.p, in0.p, e, nb); });
        if (rc != GFDM_HIP_OK) break;
        check_rx(which == 1 ? 1 : which == 2 ? 0 : 2, which >= 3 ? h.ic_iter : 0, N, 0, N, with_eq ? &in1 : nullptr);
        break;
    }
    case 5: case 6: {                                      // raw frames in, demapped symbols out
        const int nout = h.A * s.M;
        in0.alloc(pick_kind(rng), (size_t)nb * h.F * 8, rng); fill(in0, seed, nb * h.F);
        if (with_eq) { in1.alloc(pick_kind(rng), (size_t)nb * N * 8, rng); fill(in1, seed ^ 0x1234, nb * N); }
        out.alloc(pick_kind(rng), (size_t)nb * nout * 8, rng); poison(out, nb * nout);
        const float* e = with_eq ? in1.p : nullptr;
        if (which == 5) rc = product([&] { return gfdm_hip_receiver_demodulate_frames_host(h.rx, out.p, in0.p, e, 0, nb); });
        else rc = product([&] { return gfdm_hip_advanced_receiver_work_frames_host(h.adv, out.p, in0.p, e, 0, nb); });
        if (rc != GFDM_HIP_OK) break;
        check_rx(which == 5 ? 1 : 2, which == 6 ? h.ic_iter : 0, h.F, h.cp, nout, with_eq ? &in1 : nullptr);
        break;
    }
    case 7: {                                              // transmitter: every port an operand of its own
        const int nin = gfdm_hip_transmitter_input_vector_size(h.tx), F = gfdm_hip_transmitter_output_vector_size(h.tx);
        in0.alloc(pick_kind(rng), (size_t)nb * nin * 8, rng); fill(in0, seed, nb * nin);
        std::vector<std::unique_ptr<Buf>> ports;
        std::vector<float*> outs;
        for (int p = 0; p < h.nports; ++p) {
            ports.emplace_back(new Buf());
            ports.back()->alloc(pick_kind(rng), (size_t)nb * F * 8, rng);
            poison(*ports.back(), nb * F);
            outs.push_back(ports.back()->p);
        }
        rc = product([&] { return gfdm_hip_transmitter_work_host(h.tx, outs.data(), h.nports, in0.p, nin, nb); });
        if (rc != GFDM_HIP_OK) break;
        float tag = 0.f;
        for (int p = 0; p < h.nports; ++p)
            for (int64_t b = 0; b < nb; ++b)
                for (int j = 0; j < F; ++j) {
                    const c2 sv = at(in0, b * nin + (j % nin));
                    const c2 got = at(*ports[(size_t)p], b * F + j);
                    if (p == 0 && b == 0 && j == 0) { tag = got.x - loopback::tx_value(sv, 0, 1, 0.f).x; CHECK(tag_allowed(s, tag), "transmitter family tag %g", tag); }
                    CHECK(same(got, loopback::tx_value(sv, p, 1, tag)), "transmitter port %d block %ld sample %d", p, (long)b, j);
                }
        break;
    }
    case 8: {                                              // channel estimator: rx preamble -> frame estimate
        const int nin = 2 * s.K;
        in0.alloc(pick_kind(rng), (size_t)nb * nin * 8, rng); fill(in0, seed, nb * nin);
        out.alloc(pick_kind(rng), (size_t)nb * N * 8, rng); poison(out, nb * N);
        rc = product([&] { return gfdm_hip_channel_estimator_estimate_frame_host(h.est, out.p, in0.p, nb); });
        if (rc != GFDM_HIP_OK) break;
        float tag = 0.f;
        for (int64_t b = 0; b < nb; ++b)
            for (int i = 0; i < N; ++i) {
                const c2 sv = at(in0, b * nin + (i % nin));
                if (b == 0 && i == 0) tag = at(out, 0).x - loopback::est_value(sv, 0, 3, 0.f).x;
                CHECK(same(at(out, b * N + i), loopback::est_value(sv, 0, 3, tag)), "estimate_frame block %ld element %d", (long)b, i);
            }
        break;
    }
    case 10: {                                             // self-estimating receivers: the received preambles as third operand, read at a stride
        const int pre_stride = (rng() % 2) ? 0 : 2 * s.K + 2 * (int)(rng() % 5), ps = pre_stride ? pre_stride : 2 * s.K;
        const int64_t in_stride = h.frames ? h.F : N;
        const int nout = h.frames ? h.A * s.M : N;
        in0.alloc(pick_kind(rng), (size_t)nb * in_stride * 8, rng); fill(in0, seed, nb * in_stride);
        in1.alloc(pick_kind(rng), ((size_t)(nb - 1) * ps + 2 * s.K) * 8, rng); fill(in1, seed ^ 0x4242, (nb - 1) * ps + 2 * s.K);
        out.alloc(pick_kind(rng), (size_t)nb * nout * 8, rng); poison(out, nb * nout);
        const bool adv = rng() % 2;
        if (adv) rc = product([&] { return gfdm_hip_advanced_receiver_work_estimated_host(h.adv, out.p, in0.p, in1.p, pre_stride, 0, nb); });
        else rc = product([&] { return gfdm_hip_receiver_demodulate_estimated_host(h.rx, out.p, in0.p, in1.p, pre_stride, 0, nb); });
        if (rc != GFDM_HIP_OK) break;
        check_rx(adv ? 2 : 1, adv ? h.ic_iter : 0, in_stride, h.frames ? h.cp : 0, nout, &in1, ps, 2 * s.K);
        break;
    }
    default: {                                             // the stand-alone stages (generic kernels): cancel_sc_interference with three operands
        in0.alloc(pick_kind(rng), (size_t)nb * N * 8, rng); fill(in0, seed, nb * N);
        in1.alloc(pick_kind(rng), (size_t)nb * N * 8, rng); fill(in1, seed ^ 0x777, nb * N);
        out.alloc(pick_kind(rng), (size_t)nb * N * 8, rng); poison(out, nb * N);
        rc = product([&] { return gfdm_hip_receiver_cancel_sc_interference_host(h.rx, out.p, in0.p, in1.p, nb); });
        if (rc != GFDM_HIP_OK) break;
        for (int64_t i = 0; i < nb * N; ++i) CHECK(same(at(out, i), loopback::cancel_value(at(in0, i), at(in1, i))), "cancel element %ld", (long)i);
        break;
    }
    }
    if (rc != GFDM_HIP_OK) {
        CHECK(allow_failure, "call %d on shape M%d K%d L%d nb %ld failed with %d", which, s.M, s.K, s.L, (long)nb, rc);
        return false;
    }
    note_stats();
    g_calls.fetch_add(1);
    return true;
}

void worker(unsigned seed, double seconds)
{
    std::mt19937 rng(seed);
    const auto t0 = std::chrono::steady_clock::now();
    while (!g_stop.load() && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        (void)gfdm_hip_set_jit(1 + (int)(rng() % 3));
        auto h = make_handles(rng, false);
        const int ncalls = 1 + (int)(rng() % 12);
        for (int i = 0; i < ncalls; ++i) one_call(*h, rng, false);
    }
}

// the C++ layer: one batch over three handles on two "devices" (one host thread per shard), and a GNU Radio work() body
void cpp_worker(unsigned seed, double seconds)
{
    using namespace gr::gfdm;
    std::mt19937 rng(seed);
    const Shape s = kShapes[0];
    const int N = s.M * s.K;
    std::vector<cfl> taps((size_t)s.M * s.L);
    for (size_t i = 0; i < taps.size(); ++i) taps[i] = cfl(1.f / (1.f + (float)i), 0.f);
    sharded_batch<receiver_kernel_cc> rx(std::vector<int>{ 0, 1, 0 }, s.M, s.K, s.L, taps);
    modulator_kernel_cc mod(s.M, s.K, s.L, taps);
    std::vector<int> smap;
    for (int k = 0; k < s.K; ++k) smap.push_back(k);
    advanced_receiver_kernel_cc adv(s.M, s.K, s.L, taps, smap, 2, constellation::qpsk(), 0);
    const auto t0 = std::chrono::steady_clock::now();
    while (!g_stop.load() && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        const long nb = 1 + (long)(rng() % 40);
        std::vector<cfl> in((size_t)nb * N), eq((size_t)nb * N), out((size_t)nb * N, cfl(NAN, NAN));
        const unsigned sd = (unsigned)rng();
        for (size_t i = 0; i < in.size(); ++i) { const c2 v = value_at(sd, (int64_t)i); in[i] = cfl(v.x, v.y); const c2 w = value_at(sd ^ 99, (int64_t)i); eq[i] = cfl(w.x, w.y); }
        const bool e = rng() % 2;
        rx.generic_work_batch(out.data(), in.data(), e ? eq.data() : nullptr, nb);
        for (size_t i = 0; i < in.size(); ++i) {
            const c2 want = loopback::rx_value(c2{ in[i].real(), in[i].imag() }, e ? c2{ eq[i].real(), eq[i].imag() } : c2{ 0.f, 0.f }, 1, 0, loopback::kTagRowlane);
            CHECK(out[i].real() == want.x && out[i].imag() == want.y, "sharded batch element %zu of %ld blocks", i, nb);
        }
        // simple_modulator_cc_impl::work / advanced_receiver_sb_cc_impl::work through batched_work.h, a ragged noutput_items
        const int items = (int)nb * N + (int)(rng() % N);
        std::vector<cfl> sym((size_t)items), frames((size_t)items, cfl(NAN, NAN)), back((size_t)items, cfl(NAN, NAN));
        for (size_t i = 0; i < sym.size(); ++i) { const c2 v = value_at(sd + 7, (int64_t)i); sym[i] = cfl(v.x, v.y); }
        CHECK(batched::sync_work(mod, items, sym.data(), frames.data()) == items, "sync_work item count");
        CHECK(batched::sync_work_equalize(adv, items, frames.data(), nullptr, back.data()) == (int)nb * N, "sync_work_equalize item count");
        for (size_t i = 0; i < (size_t)nb * N; ++i) {
            const c2 m = loopback::mod_value(c2{ sym[i].real(), sym[i].imag() }, loopback::kTagRowlane);
            const c2 want = loopback::rx_value(m, c2{ 0.f, 0.f }, 2, 2, loopback::kTagRowlane);
            CHECK(back[i].real() == want.x && back[i].imag() == want.y, "batched work element %zu", i);
        }
        g_calls.fetch_add(3);
    }
}

void quiescer(double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    std::mt19937 rng(12345);
    while (!g_stop.load() && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        std::this_thread::sleep_for(std::chrono::milliseconds(5 + rng() % 40));
        gfdm_hip_quiesce();
    }
}

// every runtime call failing once: the call reports an error (or survives), nothing leaks or crashes, and the next call works
void failure_sweep(unsigned seed)
{
    static const char* apis[] = { "hipMalloc", "hipHostMalloc", "hipHostGetDevicePointer", "hipStreamCreateWithFlags", "hipEventCreateWithFlags", "launch", "hipMemcpyAsync",
                                  "hipMemcpy", "hipEventRecord", "hipStreamWaitEvent", "hipPointerGetAttributes", "hipMemGetAddressRange", "hipSetDevice", "hipGetDeviceCount",
                                  "hipStreamQuery", "hipStreamSynchronize", "hipModuleLoadData", "hipModuleGetFunction", "hiprtcCompileProgram", "hiprtcCreateProgram" };
    std::mt19937 rng(seed);
    long injected_errors = 0, survived = 0;
    (void)gfdm_hip_set_jit(1);
    for (const char* api : apis)
        for (int skip = 0; skip < 12; ++skip) {
            std::mt19937 r2(seed + (unsigned)skip * 7919u);           // the same handle / call sequence for every API at this depth
            loopback::fail_next(api, skip, 1, strcmp(api, "hipMalloc") == 0 || strcmp(api, "hipHostMalloc") == 0 ? hipErrorOutOfMemory : hipErrorUnknown);
            bool ok = true;
            {
                auto h = make_handles(r2, true);
                if (!h) ok = false;
                for (int i = 0; h && i < 3; ++i)
                    if (!one_call(*h, r2, true)) ok = false;
            }
            loopback::clear_failures();
            (ok ? survived : injected_errors)++;
            auto h = make_handles(rng, false);                          // and afterwards everything works
            one_call(*h, rng, false);
        }
    {   // a registration the driver refuses to pin: reported, not remembered
        void* m = mmap(nullptr, 2 * kPage, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        loopback::fail_next("hipHostRegister", 0, 1, hipErrorUnknown);
        CHECK(product([&] { return gfdm_hip_register_host(m, 2 * kPage); }) == GFDM_HIP_EHIP, "failed registration");
        loopback::clear_failures();
        CHECK(gfdm_hip_unregister_host(m) == GFDM_HIP_EINVAL, "a failed registration must not be remembered");
        CHECK(gfdm_hip_register_host(static_cast<char*>(m) + 8, kPage) == GFDM_HIP_EINVAL, "partial pages are refused");
        CHECK(gfdm_hip_register_host(m, 2 * kPage) == GFDM_HIP_OK && gfdm_hip_unregister_host(m) == GFDM_HIP_OK, "registration afterwards");
        munmap(m, 2 * kPage);
    }
    printf("failure sweep: %zu runtime calls x 12 positions: %ld calls reported the injected error, %ld runs were not reached by it; every following call correct\n",
           sizeof apis / sizeof apis[0], injected_errors, survived);
}

// refusals the host path owes its callers (a device operand shorter than the call, or running on into another allocation, must be EINVAL -- bouncing
// it would dereference a device address on the CPU), under both answers a runtime gives for memory it does not know
void edge_cases(unsigned seed)
{
    std::mt19937 rng(seed);
    for (int unknown_is_error = 0; unknown_is_error < 2; ++unknown_is_error) {
        loopback::set_unknown_pointer_is_error(unknown_is_error != 0);
        Handles h;
        h.s = kShapes[0];
        const Shape& s = h.s;
        const int N = s.M * s.K;
        const std::vector<float> taps = taps_for(s);
        CHECK(gfdm_hip_receiver_create(&h.rx, s.M, s.K, s.L, taps.data(), s.M * s.L, 1) == GFDM_HIP_OK, "receiver on device 1");
        CHECK(hipSetDevice(1) == hipSuccess, "hipSetDevice");
        const int64_t nb = 7;
        Buf in, out_short, out_other, out_ok;
        in.alloc(PAGEABLE, (size_t)nb * N * 8, rng); fill(in, seed, nb * N);
        out_short.alloc(DEVICE, (size_t)nb * N * 8 - 8, rng);                   // one sample short
        CHECK(gfdm_hip_receiver_demodulate_host(h.rx, out_short.p, in.p, nullptr, nb) == GFDM_HIP_EINVAL, "undersized device output (unknown pointers %s)",
              unknown_is_error ? "are errors" : "are 'unregistered'");
        CHECK(gfdm_hip_receiver_demodulate_host(h.rx, out_short.p, in.p, nullptr, nb - 1) == GFDM_HIP_OK, "the same buffer holds one block less");
        CHECK(hipSetDevice(0) == hipSuccess, "hipSetDevice");
        out_other.alloc(DEVICE, (size_t)nb * N * 8, rng);                       // lives on GPU 0, the handle on GPU 1
        CHECK(gfdm_hip_receiver_demodulate_host(h.rx, out_other.p, in.p, nullptr, nb) == GFDM_HIP_EINVAL, "output in another GPU's memory");
        out_ok.alloc(PAGEABLE, (size_t)nb * N * 8, rng);
        CHECK(gfdm_hip_receiver_demodulate_host(h.rx, out_ok.p, in.p, nullptr, -1) == GFDM_HIP_EINVAL, "negative block count");
        CHECK(gfdm_hip_receiver_demodulate_host(h.rx, nullptr, in.p, nullptr, 1) == GFDM_HIP_EINVAL, "NULL output");
        poison(out_ok, nb * N);
        CHECK(gfdm_hip_receiver_demodulate_host(h.rx, out_ok.p, in.p, nullptr, 0) == GFDM_HIP_OK && std::isnan(out_ok.p[0]), "zero blocks: nothing touched");
        CHECK(gfdm_hip_receiver_demodulate_host(h.rx, out_ok.p, in.p, nullptr, nb) == GFDM_HIP_OK, "pageable call (unknown pointers %d)", unknown_is_error);
        for (int64_t i = 0; i < nb * N; ++i) CHECK(same(at(out_ok, i), loopback::rx_value(at(in, i), c2{ 0.f, 0.f }, 1, 0, loopback::kTagRowlane)), "element %ld", (long)i);
        for (int i = 0; i < 40; ++i) { auto hh = make_handles(rng, false); one_call(*hh, rng, false); }
    }
    loopback::set_unknown_pointer_is_error(false);
    printf("edge cases: undersized / foreign-GPU device operands refused, argument checks, both unknown-pointer conventions of the runtime\n");
}

}  // namespace

int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0;
    const unsigned seed = argc > 2 ? (unsigned)atoi(argv[2]) : 1u;
    const int nthreads = argc > 3 ? atoi(argv[3]) : 4;
    // a cache directory of this run's own: the first handles compile, later ones (and the retry path) read the cache
    std::string cache_s = std::string(getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp") + "/gfdm_sanitize_cache_XXXXXX";
    char* cache = &cache_s[0];
    CHECK(mkdtemp(cache) != nullptr, "mkdtemp %s", cache);
    setenv("GFDM_HIP_CACHE_DIR", cache, 1);
    loopback::set_device_count(2);
    loopback::set_compile_ms(2, 25);
    CHECK(gfdm_hip_device_count() == 2, "two loop-back devices");

    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) th.emplace_back(worker, seed * 1000u + (unsigned)t, seconds);
    th.emplace_back(cpp_worker, seed * 1000u + 77u, seconds);
    th.emplace_back(quiescer, seconds);
    for (auto& t : th) t.join();
    const long calls = g_calls.load();
    printf("host fuzz: %ld calls on %d threads + sharded batch + quiesce thread, %ld operands in place, %ld chunked calls, %ld on run-time instantiated kernels, %ld of those shapes still generic;"
           " loop-back: %ld launches, %ld compiles, %ld modules\n",
           calls, nthreads, g_direct_ops.load(), g_chunked.load(), g_jit_tagged.load(), g_generic_tagged.load(), loopback::stats().launches, loopback::compiles(),
           loopback::stats().modules_loaded);
    CHECK(calls > 0, "no call completed");

    // damaged cache files: truncated code object, garbage names -- the next handle of that shape must compile afresh and work
    {
        std::string cmd = std::string("for f in ") + cache + "/*.hsaco; do head -c 9 \"$f\" > \"$f.t\" && mv \"$f.t\" \"$f\"; done 2>/dev/null";
        (void)!system(cmd.c_str());
        gfdm_hip_quiesce();
        std::mt19937 rng(seed + 5);
        (void)gfdm_hip_set_jit(1);
        for (int i = 0; i < 6; ++i) { auto h = make_handles(rng, false); one_call(*h, rng, false); }
    }
    failure_sweep(seed);
    edge_cases(seed + 3);

    gfdm_hip_quiesce();
    const loopback::Stats st = loopback::stats();
    CHECK(loopback::live_allocations() == 0, "%ld loop-back allocations still live (staging sets, tables)", loopback::live_allocations());
    CHECK(st.streams_created == st.streams_destroyed, "streams created %ld destroyed %ld", st.streams_created, st.streams_destroyed);
    CHECK(st.registers == st.unregisters, "registrations %ld unregistrations %ld", st.registers, st.unregisters);

    // the process ends with background builds queued and in flight (static destructors: jit workers, copy pool)
    {
        loopback::set_compile_ms(150, 300);
        (void)gfdm_hip_set_jit(2);
        std::string cmd = std::string("rm -rf ") + cache + "/*";
        (void)!system(cmd.c_str());
        std::mt19937 rng(seed + 9);
        std::vector<std::unique_ptr<Handles>> hs;
        for (int i = 0; i < 8; ++i) hs.push_back(make_handles(rng, false));
        for (auto& h : hs) one_call(*h, rng, false);
        hs.clear();
    }
    // the cache directory goes while those builds are still running (they make it again, or find their file's directory gone: both must be harmless)
    (void)!system((std::string("rm -rf ") + cache).c_str());
    printf("host fuzz OK\n");
    fflush(stdout);
    return 0;
}
