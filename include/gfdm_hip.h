/* gfdm_hip.h -- C-ABI of the MI355X (gfx950) GFDM modulator / receiver / IC-receiver kernels.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Each entry point names the gr-gfdm interface it replaces (paths relative to
 * the gr-gfdm checkout).  The GR-free C++ classes of gr-gfdm_amd/cpp/ and the
 * pybind11 module `gfdm_python` are thin wrappers over these functions; see
 * INTEGRATION.md for the reference-side binding.
 *
 * Conventions
 *   - samples are interleaved float32 (re, im) == std::complex<float> == gr_complex
 *     (include/gfdm/gfdm_kernel_utils.h:40).  One GFDM block = timeslots*subcarriers
 *     samples; batched calls take `nblocks` blocks back to back (block stride = block_size),
 *     exactly the stream layout the GNU Radio wrappers feed one block at a time
 *     (lib/simple_receiver_cc_impl.cc:70-74, lib/advanced_receiver_sb_cc_impl.cc:98-113).
 *   - `*_host` calls take host pointers and return when the result is in `out` (the reference's
 *     synchronous generic_work contract); see "the host-buffer batch path" below for how the bytes move.
 *   - `*_device` calls take device pointers on the handle's device and only ENQUEUE work on
 *     `stream` (a hipStream_t passed as void*, NULL = default stream).  No synchronisation,
 *     no allocation: they are hipGraph-capturable.  Buffers must not alias.
 *   - every function returns GFDM_HIP_OK (0) or a negative gfdm_hip_status; the text of the
 *     last failure on the calling thread is available from gfdm_hip_last_error().
 *   - a handle is not thread-safe (neither are the reference kernels, which own scratch
 *     buffers: include/gfdm/receiver_kernel_cc.h:103-118); use one handle per thread/stream.
 *   - there is NO CPU fallback: without a usable gfx950 device creation fails with
 *     GFDM_HIP_ENODEV.
 */
#ifndef GFDM_HIP_H
#define GFDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gfdm_hip_status {
    GFDM_HIP_OK = 0,
    GFDM_HIP_EINVAL_TAPS = -1,     /* std::invalid_argument: taps.size() != timeslots*overlap
                                      (lib/modulator_kernel_cc.cc:39-46, lib/receiver_kernel_cc.cc:40-47) */
    GFDM_HIP_EINVAL_OVERLAP = -2,  /* std::invalid_argument: receiver overlap < 2 (lib/receiver_kernel_cc.cc:48-52) */
    GFDM_HIP_EINVAL = -3,          /* any other bad argument (NULL pointer, negative size, bad subcarrier index ...) */
    GFDM_HIP_ENODEV = -4,          /* no usable HIP device */
    GFDM_HIP_EHIP = -5,            /* a HIP runtime call failed */
    GFDM_HIP_ENOMEM = -6,
    GFDM_HIP_EUNSUPPORTED = -7     /* beyond the device: timeslots * subcarriers > 2^24, or 2 * timeslots + subcarriers above ~20 000 */
} gfdm_hip_status;

/* how hard decisions are taken in the IC loop (gr::digital::constellation::decision_maker,
 * called at lib/advanced_receiver_kernel_cc.cc:119) */
typedef enum gfdm_hip_decision {
    GFDM_HIP_DECIDE_AUTO = -1,     /* QPSK/BPSK sign tests when the points match those constellations, else NEAREST */
    GFDM_HIP_DECIDE_NEAREST = 0,   /* minimum Euclidean distance, first minimum wins */
    GFDM_HIP_DECIDE_QPSK = 1,      /* index = 2*(im > 0) + (re > 0)   (constellation_qpsk) */
    GFDM_HIP_DECIDE_BPSK = 2       /* index = (re > 0)                (constellation_bpsk) */
    /* QPSK / BPSK name the sign tests of GNU Radio's unit constellations, points (+-1 +-j)/sqrt 2 in the order --, +-, -+, ++ and -1, +1.
     * Given with other points (scaled, rotated) a handle decides by NEAREST over the points as given -- for a scaled QPSK / BPSK the same
     * regions and the same tie rule (zero -> the first, all-negative point). */
} gfdm_hip_decision;

typedef struct gfdm_hip_modulator gfdm_hip_modulator;
typedef struct gfdm_hip_receiver gfdm_hip_receiver;
typedef struct gfdm_hip_advanced_receiver gfdm_hip_advanced_receiver;

const char* gfdm_hip_strerror(int status);
const char* gfdm_hip_last_error(void);
int gfdm_hip_device_count(void);
/* TEST HOOK, not part of the drop-in surface: handles created while `enable` is non-zero use the generic kernel family even
 * for shapes the tuned row-lane family serves, so that the tests can run both families on the same shape.  Returns the previous
 * setting.  Nothing else (no environment variable) changes the kernel family of a handle. */
int gfdm_hip_force_generic_family_for_testing(int enable);
/* Run-time instantiation of the tuned (row-lane) kernels for shapes outside the library's compiled list: a handle for a shape with
 * a power-of-two number of subcarriers (4 .. 1024) or one up to 1024 that is a product of two or three factors <= 32 (12, 20, 34, 48, 96, 100, 240, 384, 600, 1000 ...),
 * 3 .. 48 timeslots and overlap 2 .. 8 -- as far as the block fits the CU's 160 KB LDS (every shape up to 512 subcarriers does; 1024 x 17) -- gets the kernels compiled for exactly that
 * shape through hiprtc (1-70 s per part, growing with the number of timeslots; a handle compiles only the parts of its kind -- a modulator
 * the modulator kernels, a receiver the receive kernels, ... -- the rest at first use; the code objects are cached under
 * $GFDM_HIP_CACHE_DIR, else $XDG_CACHE_HOME/gfdm_hip, else ~/.cache/gfdm_hip -- a directory of this user that nobody else can write, or no
 * cache at all -- so this happens once per shape and machine) and reports kernel_name "rowlane_jit".
 * gfdm_hip_set_jit(mode) says WHEN the compile happens for handles created afterwards (returns the previous mode):
 *   0  never: such shapes use the generic kernel family (no compile step, several times slower kernels);
 *   1  inside the constructor, which then takes as long as the compile;
 *   2  on a background thread: the constructor returns at once, the handle runs on the generic family (same results, kernel_name
 *      "generic_lds") and switches to the tuned kernels -- kernel_name "rowlane_jit" -- at the first call after the build has finished;
 *   3  (default) like 1 when the code objects are already in the disk cache or the shape compiles within seconds (timeslots <= 16),
 *      like 2 otherwise: no constructor blocks for longer than a few seconds.
 * If hiprtc is unavailable the generic family is used.  gfdm_hip_precompile fills the disk cache ahead of time. */
int gfdm_hip_set_jit(int mode);
/* Deployment step: compile the tuned kernels of a shape into the disk cache WITHOUT creating a handle (no GPU needed), so that the first
 * flowgraph using the shape starts with them.  parts: bit 0 receive, 1 receive + IC, 2 preamble-equalised receive, 3 modulate,
 * 4 estimator (0 = all).  Returns GFDM_HIP_OK also for shapes that are compiled into the library; GFDM_HIP_EUNSUPPORTED for shapes only
 * the generic family serves.  Command line: python -m gfdm_amd.precompile <timeslots> <subcarriers> <overlap> [...]. */
int gfdm_hip_precompile(int timeslots, int subcarriers, int overlap, unsigned parts);
/* Stops the background builds of gfdm_hip_set_jit modes 2 / 3: what is still queued is dropped (those handles stay on the generic kernels), the
 * builds in flight finish their compile (the code object still reaches the disk cache) without touching the GPU, and the call returns when the
 * two pool threads are idle -- at most one compile (10-70 s) later.  The library does this by itself when it is unloaded; an application that
 * tears the HIP runtime down before that (an interpreter at exit: the Python package registers it with atexit) calls it first.  Later requests
 * start the pool again. */
void gfdm_hip_quiesce(void);
/* The interference-cancellation rounds of the advanced receiver (lib/advanced_receiver_kernel_cc.cc:56-76, lib/receiver_kernel_cc.cc:274-299)
 * can run on the matrix cores (v_mfma_f32_16x16x32_f16; the QPSK decisions are exact in f16, the IC taps enter as a three-term f16 split with
 * 33 significant bits, sums in f32) where that form applies: QPSK sign decisions, a real even IC kernel (any real, even prototype filter),
 * no phase compensation, 4 .. 16 timeslots and a power-of-two number of subcarriers >= 16 on the row-lane kernels.  Mode of the handles
 * created afterwards (returns the previous mode): 0 = vector ALU only (f32 throughout), 1 (default) = matrix cores where they are the
 * faster form (blocks of two or more wavefronts: subcarriers >= 128), 2 = matrix cores wherever the form applies. */
int gfdm_hip_set_ic_matrix_cores(int mode);
/* The timeslot-axis transforms of the generic kernel family (shapes outside the tuned row-lane families: more than 48 timeslots, a number of
 * subcarriers with a prime factor above 31 or above 1024; lib/modulator_kernel_cc.cc:109-110,137-140, lib/receiver_kernel_cc.cc:211-225,
 * 274-299, 304-305) are direct sums; from 32 timeslots on they can run on the matrix cores as products of the constant cosine / sine matrices
 * with the block's samples (v_mfma_f32_16x16x4_f32: f32 operands, f32 sums -- no reduced precision).  Mode of the handles created afterwards
 * (returns the previous mode): 0 = vector ALU only, 1 (default) = matrix cores where they are the faster form (the 16 x 16 x 4 operand tiles
 * well filled, at least four of them per block, and the operand scratch does not cost the CU its second workgroup), 2 = wherever the form fits.
 * In mode 1 the reference's own QA shape -- 127 timeslots, 16 subcarriers (python/qa_simple_receiver_cc.py:58-83) -- does not use dense
 * transforms at all for plain blocks (modulate, fft_[equalize_]filter_downsample, generic_work[_equalize], the advanced receiver's cancellation
 * rounds): its 127-point transforms are Rader transforms, cyclic convolutions of length 126 = 9 x 14 through prime-factor FFTs
 * (csrc/gfdm_rader.hip; kernel_name "generic_rader"); frames / demapper and the self-estimating receivers of that shape stay on the generic
 * kernels.  Modes 0 and 2 keep the dense forms for that shape too (A/B, tests). */
int gfdm_hip_set_dft_matrix_cores(int mode);
/* TEST HOOK: compile (or find in the disk cache) part 0..4 (receive, receive + IC, preamble-equalised receive, modulate, estimator) of
 * the row-lane kernels for a shape through hiprtc WITHOUT loading it --
 * needs no GPU, so the CPU-side tests can check that the embedded kernel sources build. */
int gfdm_hip_jit_build_for_testing(int timeslots, int subcarriers, int overlap, int part);
const char* gfdm_hip_version(void);
/* 16 hex digits: hash of the kernel / C-ABI sources and compiler flags this library was built from.  The rocprofv3 summaries under
 * profiles/ name the build they were measured with; bench.py only quotes counters whose build id equals the loaded library's. */
const char* gfdm_hip_build_id(void);

/* ---- the host-buffer batch path (*_host entry points) -------------------------------------------------------------------------
 * The reference's callers hand HOST pointers and advance them block by block (lib/simple_receiver_cc_impl.cc:61-77,
 * lib/advanced_receiver_sb_cc_impl.cc:86-123 -- in, f_eq and out --, lib/transmitter_cc_impl.cc:165-177); a *_host call takes the
 * whole run of a scheduler call.  How its bytes cross the PCIe link:
 *   - buffers the GPU can address -- registered with gfdm_hip_register_host, allocated with hipHostMalloc, or device memory -- are
 *     used IN PLACE: the kernels run on the caller's memory, bound by the link (~57 GB/s each way on the MI355X boxes);
 *   - ordinary pageable buffers are bounced through pinned staging sets in chunks: the kernel of chunk c works across the link
 *     while the calling thread and a small pool of copy threads move chunk c + 1 in and chunk c - 1 out.  A call that fits one
 *     chunk (<= 512 KiB, e.g. the one block per call of an unchanged GNU Radio wrapper) is copy in, one launch, completion ticket, copy out.
 *   Measured (MI355X box, K=64 M=9, profiles/r04/bench_default.json): pageable 6.4 M blocks/s matched filter, 4.3 M ZF + 2 IC at 65 536 blocks
 *   per call; registered 9.6 M / 6.0 M (88 / 82 GB/s over the link, both directions together); one block per call 13-16 us either way.
 * Results do not depend on the route (same kernels, same blocks).  The call returns when `out` is complete.
 *
 * gfdm_hip_register_host: pin a long-lived buffer (a GNU Radio circular buffer, an application's frame store) and map it for every GPU,
 * ~16 us per MiB once; later *_host calls on any part of it skip the bounce.  The range must be WHOLE PAGES the caller owns -- start and size
 * multiples of the page size (mmap'ed buffers, aligned_alloc / posix_memalign with a page-multiple size), GFDM_HIP_EINVAL otherwise: pinning is
 * per page, and a page shared with other objects must not be pinned and unpinned under them.  Unregister before the memory is freed. */
int gfdm_hip_register_host(void* ptr, size_t bytes);
int gfdm_hip_unregister_host(void* ptr);
/* Process-wide tuning of the bounce (negative = leave unchanged).  mode = how the bytes cross the link: 0 (default) under the kernels' own
 * accesses to the pinned staging memory, 1 copy engines (H2D / kernel / D2H on three streams, device staging), 2 inputs under the kernel's
 * reads and outputs by copy engine, 3 the reverse -- 1 .. 3 are kept for A/B measurements (profiles/r04/host_path_sweep.txt: mode 0 wins
 * or ties everywhere but the largest matched-filter calls); chunk_bytes = staged bytes per chunk over all operands, 0 = automatic (one chunk
 * up to 512 KiB, else about sqrt(1.3 x MiB staged) chunks, at least two, of at most 16 MiB: profiles/r04/host_chunk_sweep.txt); depth = staging
 * sets (1 .. 4, default 3); copy_threads = pool threads that help the calling thread with the bounce copies of chunks >= 1 MiB (0 .. 16,
 * default 3); kernel_streams = 2 (default): the chunks alternate
 * between two streams (a kernel reads its blocks, then writes them, i.e. uses one direction of the link at a time; with two streams the
 * reads of one chunk can run under the writes of another), 1 = a single stream. */
int gfdm_hip_set_host_pipeline(int mode, int64_t chunk_bytes, int depth, int copy_threads, int kernel_streams);
int gfdm_hip_get_host_pipeline(int* mode, int64_t* chunk_bytes, int* depth, int* copy_threads, int* kernel_streams);
/* What the last *_host call of the calling thread did (any pointer may be NULL): kernel launches, blocks per chunk, bytes bounced,
 * bit i of direct_mask = operand i was used in place (operands in signature order: outputs first, then inputs), route, pool threads used. */
int gfdm_hip_host_call_stats(int64_t* chunks, int64_t* chunk_blocks, int64_t* staged_bytes, unsigned* direct_mask, int* mode, int* copy_threads);
/* ... and where the calling thread spent it, nanoseconds: ns5[0] sorting the operands and sizing the staging, [1] bounce copies, [2] enqueueing
 * the kernels, [3] posting completion tickets, [4] waiting for them.  (A one-block call on the MI355X box: 0.3 / 0.4 / 3.8 / 3.1 / 5.1 us.) */
int gfdm_hip_host_call_times(int64_t* ns5);
/* TEST / MEASUREMENT HOOK: the bounce copies of calls that stage 2 MiB or more use non-temporal (streaming) stores; 0 = plain memcpy everywhere
 * (A/B).  Returns the previous setting. */
int gfdm_hip_set_host_streaming_copies_for_testing(int enable);

/* ---- modulator_kernel_cc (include/gfdm/modulator_kernel_cc.h:41-51) -------------------- */

/* ctor, lib/modulator_kernel_cc.cc:30-66.  taps: ntaps interleaved complex; device: HIP ordinal. */
int gfdm_hip_modulator_create(gfdm_hip_modulator** out, int timeslots, int subcarriers, int overlap,
                              const float* taps, int ntaps, int device);
int gfdm_hip_modulator_destroy(gfdm_hip_modulator* m);
int gfdm_hip_modulator_block_size(const gfdm_hip_modulator* m);                 /* block_size(), .h:50 */
int gfdm_hip_modulator_filter_taps(const gfdm_hip_modulator* m, float* out);    /* filter_taps(), .cc:92-95; overlap*timeslots complex */
const char* gfdm_hip_modulator_kernel_name(const gfdm_hip_modulator* m);        /* which HIP kernel family serves this shape */
/* generic_work(out, in), lib/modulator_kernel_cc.cc:98-141, over nblocks blocks */
int gfdm_hip_modulator_work_host(gfdm_hip_modulator* m, float* out, const float* in, int64_t nblocks);
int gfdm_hip_modulator_work_device(gfdm_hip_modulator* m, void* out, const void* in, int64_t nblocks, void* stream);

/* ---- receiver_kernel_cc (include/gfdm/receiver_kernel_cc.h:52-89) ---------------------- */

/* ctor, lib/receiver_kernel_cc.cc:31-88 */
int gfdm_hip_receiver_create(gfdm_hip_receiver** out, int timeslots, int subcarriers, int overlap,
                             const float* taps, int ntaps, int device);
int gfdm_hip_receiver_destroy(gfdm_hip_receiver* r);
int gfdm_hip_receiver_block_size(const gfdm_hip_receiver* r);
int gfdm_hip_receiver_timeslots(const gfdm_hip_receiver* r);
int gfdm_hip_receiver_subcarriers(const gfdm_hip_receiver* r);
int gfdm_hip_receiver_overlap(const gfdm_hip_receiver* r);
int gfdm_hip_receiver_filter_taps(const gfdm_hip_receiver* r, float* out);      /* .cc:120-123 */
int gfdm_hip_receiver_ic_filter_taps(const gfdm_hip_receiver* r, float* out);   /* .cc:125-128; timeslots complex */
const char* gfdm_hip_receiver_kernel_name(const gfdm_hip_receiver* r);

/* generic_work (f_eq == NULL, .cc:322-326) / generic_work_equalize (f_eq != NULL, .cc:328-334);
 * f_eq holds one block_size vector PER BLOCK (lib/advanced_receiver_sb_cc_impl.cc:98-104) */
int gfdm_hip_receiver_demodulate_host(gfdm_hip_receiver* r, float* out, const float* in, const float* f_eq, int64_t nblocks);
int gfdm_hip_receiver_demodulate_device(gfdm_hip_receiver* r, void* out, const void* in, const void* f_eq, int64_t nblocks, void* stream);
/* fft_filter_downsample (.cc:301-307) / fft_equalize_filter_downsample (.cc:309-320) */
int gfdm_hip_receiver_fft_filter_downsample_host(gfdm_hip_receiver* r, float* out, const float* in, const float* f_eq, int64_t nblocks);
int gfdm_hip_receiver_fft_filter_downsample_device(gfdm_hip_receiver* r, void* out, const void* in, const void* f_eq, int64_t nblocks, void* stream);
/* transform_subcarriers_to_td (.cc:211-225) */
int gfdm_hip_receiver_transform_subcarriers_to_td_host(gfdm_hip_receiver* r, float* out, const float* in, int64_t nblocks);
int gfdm_hip_receiver_transform_subcarriers_to_td_device(gfdm_hip_receiver* r, void* out, const void* in, int64_t nblocks, void* stream);
/* cancel_sc_interference(out, td_in, fd_in) (.cc:274-299) */
int gfdm_hip_receiver_cancel_sc_interference_host(gfdm_hip_receiver* r, float* out, const float* td_in, const float* fd_in, int64_t nblocks);
int gfdm_hip_receiver_cancel_sc_interference_device(gfdm_hip_receiver* r, void* out, const void* td_in, const void* fd_in, int64_t nblocks, void* stream);

/* ---- advanced_receiver_kernel_cc (include/gfdm/advanced_receiver_kernel_cc.h:37-78) ---- */

/* ctor, lib/advanced_receiver_kernel_cc.cc:32-52.  The gr::digital::constellation_sptr argument of the
 * reference is passed as its points() array plus the decision rule. */
int gfdm_hip_advanced_receiver_create(gfdm_hip_advanced_receiver** out, int timeslots, int subcarriers, int overlap,
                                      const float* taps, int ntaps,
                                      const int* subcarrier_map, int n_subcarrier_map,
                                      int ic_iter,
                                      const float* constellation_points, int n_points, int decision,
                                      int do_phase_compensation, int device);
int gfdm_hip_advanced_receiver_destroy(gfdm_hip_advanced_receiver* a);
int gfdm_hip_advanced_receiver_block_size(const gfdm_hip_advanced_receiver* a);
int gfdm_hip_advanced_receiver_set_ic(gfdm_hip_advanced_receiver* a, int ic_iter);                 /* .h:57 */
int gfdm_hip_advanced_receiver_get_ic(const gfdm_hip_advanced_receiver* a);                        /* .h:58 */
int gfdm_hip_advanced_receiver_set_phase_compensation(gfdm_hip_advanced_receiver* a, int enable);  /* .h:60-63 */
int gfdm_hip_advanced_receiver_get_phase_compensation(const gfdm_hip_advanced_receiver* a);        /* .h:64 */
const char* gfdm_hip_advanced_receiver_kernel_name(const gfdm_hip_advanced_receiver* a);
/* the decision rule the handle RUNS (a gfdm_hip_decision, never AUTO): QPSK / BPSK -- the sign tests, and with a real even IC kernel the
 * matrix-core rounds -- only for GNU Radio's unit constellations (every component within 6 * FLT_EPSILON relative of (+-1 +-j)/sqrt 2 resp. -1, +1 -- this admits gr::digital's 0.707107 literal -- in
 * that order); any other points, also when created with an explicit QPSK / BPSK, are decided by NEAREST over the points as given */
int gfdm_hip_advanced_receiver_decision(const gfdm_hip_advanced_receiver* a);
/* generic_work (f_eq == NULL, .cc:93-98) / generic_work_equalize (f_eq != NULL, .cc:100-107) */
int gfdm_hip_advanced_receiver_work_host(gfdm_hip_advanced_receiver* a, float* out, const float* in, const float* f_eq, int64_t nblocks);
int gfdm_hip_advanced_receiver_work_device(gfdm_hip_advanced_receiver* a, void* out, const void* in, const void* f_eq, int64_t nblocks, void* stream);

/* ---- receivers on raw frames with demapped output -----------------------------------
 * In the reference flowgraph the receiver sits between remove_prefix (add_cyclic_prefix_cc::remove_cyclic_prefix,
 * lib/add_cyclic_prefix_cc.cc:100-104) and the resource demapper (resource_mapper_kernel_cc::demap_from_resources,
 * lib/resource_mapper_kernel_cc.cc:91-106,136-163).  Here both are the receiver kernel's load / store stage:
 * configure_frames declares the frame layout once, the *_frames_* calls then read frames of frame_len samples (the block
 * starts cp_len samples in) and write only the active subcarriers' symbols in mapper order, noutput_size per frame
 * (<= 0: all of them).  n_subcarrier_map == 0 keeps the plain [k][m] block output.  f_eq stays one block_size vector per frame.
 * The demapper walks the SORTED map, as the reference constructor does. */
int gfdm_hip_receiver_configure_frames(gfdm_hip_receiver* r, int frame_len, int cp_len, const int* subcarrier_map, int n_subcarrier_map,
                                       int per_timeslot);
int gfdm_hip_receiver_demodulate_frames_host(gfdm_hip_receiver* r, float* out, const float* in, const float* f_eq, int noutput_size,
                                             int64_t nblocks);
int gfdm_hip_receiver_demodulate_frames_device(gfdm_hip_receiver* r, void* out, const void* in, const void* f_eq, int noutput_size,
                                               int64_t nblocks, void* stream);
int gfdm_hip_advanced_receiver_configure_frames(gfdm_hip_advanced_receiver* a, int frame_len, int cp_len, const int* subcarrier_map,
                                                int n_subcarrier_map, int per_timeslot);
int gfdm_hip_advanced_receiver_work_frames_host(gfdm_hip_advanced_receiver* a, float* out, const float* in, const float* f_eq,
                                                int noutput_size, int64_t nblocks);
int gfdm_hip_advanced_receiver_work_frames_device(gfdm_hip_advanced_receiver* a, void* out, const void* in, const void* f_eq,
                                                  int noutput_size, int64_t nblocks, void* stream);

/* ---- transmitter_kernel (include/gfdm/transmitter_kernel.h:43-85) -------------
 * resource mapper -> modulator -> cyclic prefix / suffix with cyclic shift + window ramp -> preamble, FUSED into one HIP kernel:
 * the mapper is the modulator's load stage, prefix/suffix/ramp/preamble its store stage, for all cyclic shifts ("ports") at once. */
typedef struct gfdm_hip_transmitter gfdm_hip_transmitter;

/* ctor, lib/transmitter_kernel.cc:33-73 (which constructs resource_mapper_kernel_cc, modulator_kernel_cc, add_cyclic_prefix_cc).
 * window_taps: n_window_taps complex (the whole window block+cp+cs, or 2*ramp_len); preambles: [n_cyclic_shifts][preamble_len] complex.
 * Argument errors of the three reference constructors are reported as GFDM_HIP_EINVAL / GFDM_HIP_EINVAL_TAPS with their messages. */
int gfdm_hip_transmitter_create(gfdm_hip_transmitter** out, int timeslots, int subcarriers, int active_subcarriers, int cp_len,
                                int cs_len, int ramp_len, const int* subcarrier_map, int n_subcarrier_map, int per_timeslot, int overlap,
                                const float* taps, int ntaps, const float* window_taps, int n_window_taps, const int* cyclic_shifts,
                                int n_cyclic_shifts, const float* preambles, int preamble_len, int device);
int gfdm_hip_transmitter_destroy(gfdm_hip_transmitter* t);
int gfdm_hip_transmitter_input_vector_size(const gfdm_hip_transmitter* t);    /* .h:63, active_subcarriers * timeslots */
int gfdm_hip_transmitter_output_vector_size(const gfdm_hip_transmitter* t);   /* .h:64, preamble + cp + block + cs */
int gfdm_hip_transmitter_block_size(const gfdm_hip_transmitter* t);
int gfdm_hip_transmitter_n_cyclic_shifts(const gfdm_hip_transmitter* t);
int gfdm_hip_transmitter_cyclic_shift(const gfdm_hip_transmitter* t, int port); /* cyclic_shifts()[port], .h:69 */
const char* gfdm_hip_transmitter_kernel_name(const gfdm_hip_transmitter* t);
/* generic_work (lib/transmitter_kernel.cc:100-106) generalised to the first n_outs cyclic shifts, i.e. what
 * transmitter_cc_impl::general_work does per frame (lib/transmitter_cc_impl.cc:165-177).  in: nblocks x ninput_size symbols,
 * outs[i]: nblocks x output_vector_size samples for cyclic_shifts[i]. */
int gfdm_hip_transmitter_work_host(gfdm_hip_transmitter* t, float* const* outs, int n_outs, const float* in, int ninput_size, int64_t nblocks);
int gfdm_hip_transmitter_work_device(gfdm_hip_transmitter* t, void* const* outs, int n_outs, const void* in, int ninput_size,
                                     int64_t nblocks, void* stream);
/* modulate (lib/transmitter_kernel.cc:78-84): mapper + modulator, out = bare blocks */
int gfdm_hip_transmitter_modulate_host(gfdm_hip_transmitter* t, float* out, const float* in, int ninput_size, int64_t nblocks);
int gfdm_hip_transmitter_modulate_device(gfdm_hip_transmitter* t, void* out, const void* in, int ninput_size, int64_t nblocks, void* stream);
/* add_frame (lib/transmitter_kernel.cc:92-98): preamble of `cyclic_shift` + cyclic prefix/suffix + ramp of modulated blocks */
int gfdm_hip_transmitter_add_frame_host(gfdm_hip_transmitter* t, float* out, const float* in, int cyclic_shift, int64_t nblocks);
int gfdm_hip_transmitter_add_frame_device(gfdm_hip_transmitter* t, void* out, const void* in, int cyclic_shift, int64_t nblocks, void* stream);

/* ---- preamble_channel_estimator_cc (include/gfdm/preamble_channel_estimator_cc.h:45-78) ----
 * Received preamble (2 * fft_len samples) -> frequency-domain channel estimate of a whole frame (timeslots * fft_len bins), the
 * f_eq input of the receivers above.  One HIP kernel per call, one workgroup per preamble, every stage LDS-resident;
 * `nframes` preambles / estimates back to back. */
typedef struct gfdm_hip_channel_estimator gfdm_hip_channel_estimator;

/* ctor, lib/preamble_channel_estimator_cc.cc:32-76.  preamble: n_preamble >= 2 * fft_len complex (the known core preamble). */
int gfdm_hip_channel_estimator_create(gfdm_hip_channel_estimator** out, int timeslots, int fft_len, int active_subcarriers, int is_dc_free,
                                      int which_estimator, const float* preamble, int n_preamble, int device);
int gfdm_hip_channel_estimator_destroy(gfdm_hip_channel_estimator* c);
int gfdm_hip_channel_estimator_timeslots(const gfdm_hip_channel_estimator* c);            /* .h:58 */
int gfdm_hip_channel_estimator_fft_len(const gfdm_hip_channel_estimator* c);              /* .h:57 */
int gfdm_hip_channel_estimator_active_subcarriers(const gfdm_hip_channel_estimator* c);   /* .h:60 */
int gfdm_hip_channel_estimator_frame_len(const gfdm_hip_channel_estimator* c);            /* .h:59, timeslots * fft_len */
int gfdm_hip_channel_estimator_is_dc_free(const gfdm_hip_channel_estimator* c);           /* .h:61 */
int gfdm_hip_channel_estimator_filtered_len(const gfdm_hip_channel_estimator* c);         /* active_subcarriers + is_dc_free */
const char* gfdm_hip_channel_estimator_kernel_name(const gfdm_hip_channel_estimator* c);  /* kernel family of estimate_frame */
int gfdm_hip_channel_estimator_preamble_filter_taps(const gfdm_hip_channel_estimator* c, float* out);   /* .h:62, 9 floats; returns 9 */
/* estimate_frame (lib/preamble_channel_estimator_cc.cc:284-295): the three stages below fused.  Bins the reference leaves
 * unwritten when not dc-free ([(A/2 - 1) M, (A/2) M)) carry the value of the constant region next to them. */
int gfdm_hip_channel_estimator_estimate_frame_host(gfdm_hip_channel_estimator* c, float* frame_estimate, const float* rx_preamble, int64_t nframes);
int gfdm_hip_channel_estimator_estimate_frame_device(gfdm_hip_channel_estimator* c, void* frame_estimate, const void* rx_preamble,
                                                     int64_t nframes, void* stream);
/* estimate_preamble_channel (:118-145): rx preamble -> fft_len bins */
int gfdm_hip_channel_estimator_estimate_preamble_channel_host(gfdm_hip_channel_estimator* c, float* fd_preamble_channel, const float* rx_preamble,
                                                              int64_t nframes);
int gfdm_hip_channel_estimator_estimate_preamble_channel_device(gfdm_hip_channel_estimator* c, void* fd_preamble_channel, const void* rx_preamble,
                                                                int64_t nframes, void* stream);
/* filter_preamble_estimate (:147-187): fft_len bins -> filtered_len smoothed bins (fftshift order) */
int gfdm_hip_channel_estimator_filter_preamble_estimate_host(gfdm_hip_channel_estimator* c, float* filtered, const float* estimate, int64_t nframes);
int gfdm_hip_channel_estimator_filter_preamble_estimate_device(gfdm_hip_channel_estimator* c, void* filtered, const void* estimate, int64_t nframes,
                                                               void* stream);
/* interpolate_frame (:238-273): filtered_len bins -> frame_len bins */
int gfdm_hip_channel_estimator_interpolate_frame_host(gfdm_hip_channel_estimator* c, float* frame_estimate, const float* filtered, int64_t nframes);
int gfdm_hip_channel_estimator_interpolate_frame_device(gfdm_hip_channel_estimator* c, void* frame_estimate, const void* filtered, int64_t nframes,
                                                        void* stream);
/* prepare_for_zf (:275-281): conj(1 / frame_estimate) */
int gfdm_hip_channel_estimator_prepare_for_zf_host(gfdm_hip_channel_estimator* c, float* transformed_frame, const float* frame_estimate,
                                                   int64_t nframes);
int gfdm_hip_channel_estimator_prepare_for_zf_device(gfdm_hip_channel_estimator* c, void* transformed_frame, const void* frame_estimate,
                                                     int64_t nframes, void* stream);
/* estimate_snr (:189-236): snr_lin[nframes] (the return value), cnrs[nframes][active_subcarriers] (the std::vector argument) */
int gfdm_hip_channel_estimator_estimate_snr_host(gfdm_hip_channel_estimator* c, float* snr_lin, float* cnrs, const float* rx_preamble, int64_t nframes);
int gfdm_hip_channel_estimator_estimate_snr_device(gfdm_hip_channel_estimator* c, float* snr_lin, float* cnrs, const void* rx_preamble,
                                                   int64_t nframes, void* stream);

/* ---- receivers that estimate the channel themselves (second half) ----
 * The chain  channel_estimator_cc -> (f_eq input of) simple_receiver_cc / advanced_receiver_sb_cc  of
 * examples/hier_gfdm_receiver.grc in ONE kernel: the receiver kernel runs estimate_frame on its block's received preamble
 * and applies the result as its one-tap equaliser; the N-bin estimate never exists in HBM (the preamble's 2 * fft_len samples
 * are read instead of N equaliser bins).  Results equal estimate_frame followed by generic_work_equalize.
 * set_channel_estimator: the estimator handle must match (timeslots, subcarriers, device) and outlive its use; NULL detaches.  (A receiver on
 *   run-time instantiated kernels loads its preamble-equalised kernels here: at once when they are cached or quick to build, otherwise on the
 *   background pool -- the estimated calls run on the generic kernel family, same results, until they are there.)
 * rx_preamble: preamble of block b at rx_preamble + b * preamble_stride complex (0 = packed, 2 * fft_len) -- a burst buffer
 *   holding preamble and frame back to back is passed as `in` and, offset to the core preamble, as `rx_preamble`.
 * The block I/O follows configure_frames when that was called (frames in, demapped symbols out, noutput_size as there),
 * else plain blocks in and out (noutput_size must then be <= 0 or block_size). */
/* Buffer sizes of ONE *_frames_* (estimated == 0) or *_estimated_* (estimated != 0) call with this noutput_size, in complex samples
 * per frame / block: n_in read from `in`, n_out written to `out`, and the attached estimator's fft_len (0 = none).  The library is
 * the authority on these sizes (a caller that derives them itself can under-allocate `out`): EINVAL where the call itself would
 * fail -- no configure_frames / estimator, noutput_size above active_subcarriers * timeslots
 * (lib/resource_mapper_kernel_cc.cc:95-99), or noutput_size given without a subcarrier map (the block is written whole). */
int gfdm_hip_receiver_io_layout(const gfdm_hip_receiver* r, int estimated, int noutput_size, int* n_in, int* n_out, int* est_fft_len);
int gfdm_hip_advanced_receiver_io_layout(const gfdm_hip_advanced_receiver* a, int estimated, int noutput_size, int* n_in, int* n_out,
                                         int* est_fft_len);
int gfdm_hip_receiver_set_channel_estimator(gfdm_hip_receiver* r, const gfdm_hip_channel_estimator* c);
int gfdm_hip_advanced_receiver_set_channel_estimator(gfdm_hip_advanced_receiver* a, const gfdm_hip_channel_estimator* c);
int gfdm_hip_receiver_demodulate_estimated_host(gfdm_hip_receiver* r, float* out, const float* in, const float* rx_preamble, int preamble_stride,
                                                int noutput_size, int64_t nblocks);
int gfdm_hip_receiver_demodulate_estimated_device(gfdm_hip_receiver* r, void* out, const void* in, const void* rx_preamble, int preamble_stride,
                                                  int noutput_size, int64_t nblocks, void* stream);
int gfdm_hip_advanced_receiver_work_estimated_host(gfdm_hip_advanced_receiver* a, float* out, const float* in, const float* rx_preamble,
                                                   int preamble_stride, int noutput_size, int64_t nblocks);
int gfdm_hip_advanced_receiver_work_estimated_device(gfdm_hip_advanced_receiver* a, void* out, const void* in, const void* rx_preamble,
                                                     int preamble_stride, int noutput_size, int64_t nblocks, void* stream);

/* ---- resource_mapper_kernel_cc (include/gfdm/resource_mapper_kernel_cc.h:38-60, lib/resource_mapper_kernel_cc.cc) -----------------
 * The stand-alone form of the mapper / demapper that the transmitter and the *_frames_* receivers above have fused into their
 * kernels: data symbols <-> the [subcarriers][timeslots] grid the modulator reads and the receivers
 * write.  Blocks back to back: map reads ninput_size symbols per block and writes frame_size; demap reads frame_size and writes
 * noutput_size.  Errors as the reference constructor / methods throw them (:44-69, :78-82, :95-99), as GFDM_HIP_EINVAL with the
 * message in gfdm_hip_last_error().  Two deliberate differences: a subcarrier index EQUAL to `subcarriers` is refused (the reference
 * tests '>' at :65 and then writes past the grid), and the per-subcarrier demapper writes exactly noutput_size symbols (the reference
 * loop :150-161 writes one more when noutput_size < block_size). */
typedef struct gfdm_hip_resource_mapper gfdm_hip_resource_mapper;
int gfdm_hip_resource_mapper_create(gfdm_hip_resource_mapper** out, int timeslots, int subcarriers, int active_subcarriers,
                                    const int* subcarrier_map, int n_subcarrier_map, int per_timeslot, int device);
int gfdm_hip_resource_mapper_destroy(gfdm_hip_resource_mapper* m);
int gfdm_hip_resource_mapper_block_size(const gfdm_hip_resource_mapper* m);      /* timeslots * active_subcarriers   (.h:47) */
int gfdm_hip_resource_mapper_frame_size(const gfdm_hip_resource_mapper* m);      /* timeslots * subcarriers          (.h:46) */
/* map_to_resources (.h:50-52, .cc:74-89): grid zeroed, the first ninput_size symbols placed, the rest of the grid stays zero */
int gfdm_hip_resource_mapper_map_host(gfdm_hip_resource_mapper* m, float* out, const float* in, int ninput_size, int64_t nblocks);
int gfdm_hip_resource_mapper_map_device(gfdm_hip_resource_mapper* m, void* out, const void* in, int ninput_size, int64_t nblocks, void* stream);
/* demap_from_resources (.h:53-55, .cc:91-106) */
int gfdm_hip_resource_mapper_demap_host(gfdm_hip_resource_mapper* m, float* out, const float* in, int noutput_size, int64_t nblocks);
int gfdm_hip_resource_mapper_demap_device(gfdm_hip_resource_mapper* m, void* out, const void* in, int noutput_size, int64_t nblocks, void* stream);

/* ---- add_cyclic_prefix_cc (include/gfdm/add_cyclic_prefix_cc.h:40-60, lib/add_cyclic_prefix_cc.cc) -------------------------------
 * Cyclic prefix + suffix with cyclic shift and the block-pinching window ramps, and prefix removal; the stand-alone form of the
 * transmitter's store stage / the frame receivers' load offset.  window_taps: either the whole window (block_len + cp_len + cs_len
 * taps) or only its 2 * ramp_len ramp taps (.cc:42-56).  add: block_len samples in, frame_size = block_len + cp_len + cs_len out;
 * remove: the reverse (frame[cp_len : cp_len + block_len], .cc:100-104).  cyclic_shift must lie in [0, cs_len] with cp_len + shift
 * <= block_len (the reference reads outside its input otherwise). */
typedef struct gfdm_hip_cyclic_prefixer gfdm_hip_cyclic_prefixer;
int gfdm_hip_cyclic_prefixer_create(gfdm_hip_cyclic_prefixer** out, int block_len, int cp_len, int cs_len, int ramp_len, const float* window_taps,
                                    int n_window_taps, int cyclic_shift, int device);
int gfdm_hip_cyclic_prefixer_destroy(gfdm_hip_cyclic_prefixer* c);
int gfdm_hip_cyclic_prefixer_block_size(const gfdm_hip_cyclic_prefixer* c);
int gfdm_hip_cyclic_prefixer_frame_size(const gfdm_hip_cyclic_prefixer* c);
int gfdm_hip_cyclic_prefixer_cyclic_shift(const gfdm_hip_cyclic_prefixer* c);
/* add_cyclic_prefix(out, in, cyclic_shift) (.cc:66-98); generic_work = the constructor's shift */
int gfdm_hip_cyclic_prefixer_add_host(gfdm_hip_cyclic_prefixer* c, float* out, const float* in, int cyclic_shift, int64_t nblocks);
int gfdm_hip_cyclic_prefixer_add_device(gfdm_hip_cyclic_prefixer* c, void* out, const void* in, int cyclic_shift, int64_t nblocks, void* stream);
int gfdm_hip_cyclic_prefixer_remove_host(gfdm_hip_cyclic_prefixer* c, float* out, const float* in, int64_t nblocks);
int gfdm_hip_cyclic_prefixer_remove_device(gfdm_hip_cyclic_prefixer* c, void* out, const void* in, int64_t nblocks, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GFDM_HIP_H */
