/* Multi-threaded timing driver of the plain-C CPU oracle (bench.py's cpu_baseline leg).
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never linked or called by the product path (see gfdm_oracle.c).
 *
 * T pthreads, pinned one per allowed CPU, each with ITS OWN kernel object and private buffers, run the reference's loop
 * structure -- one kernel object processing its blocks one after the other, exactly how the GNU Radio wrappers drive the
 * reference kernels (lib/simple_receiver_cc_impl.cc:61-77, lib/simple_modulator_cc_impl.cc:62-80,
 * lib/advanced_receiver_sb_cc_impl.cc:95-113) -- until a common deadline.  No Python, no GIL, no shared state in the timed
 * loop: the figure is what the host's cores deliver on this algorithm, not what a Python thread pool lets through. */
#define _GNU_SOURCE
#include "gfdm_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

enum { BENCH_MOD_DEMOD = 0, BENCH_DEMOD = 1, BENCH_DEMOD_IC = 2, BENCH_MOD = 3 };

/* start gate: the workers report in once their set-up is done and wait for `go`; `abort` releases them when a worker could not be started
 * (a pthread barrier cannot be opened with fewer threads than it was initialised for, and waiting on one cannot be cancelled) */
typedef struct {
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int arrived, go, abort;
} gate_t;

typedef struct {
    int M, K, L, ntaps, mode, use_eq, ic_iter, chunk, cpu;
    const float* taps;
    double seconds;        /* length of the timed loop */
    gate_t* gate;          /* every worker waits here once its kernel object and buffers exist */
    double* t_start;       /* CLOCK_MONOTONIC seconds at which the driver saw the barrier open: start of the timed region */
    long blocks;           /* out */
    int failed;            /* out */
} worker_t;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void* worker(void* arg)
{
    worker_t* w = (worker_t*)arg;
    if (w->cpu >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(w->cpu, &set);
        (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    }
    gfdm_oracle* o = gfdm_oracle_create(w->M, w->K, w->L, w->taps, w->ntaps);
    const size_t N = (size_t)w->M * (size_t)w->K, n = N * (size_t)w->chunk;
    float* sym = (float*)malloc(sizeof(float) * 2 * n);
    float* frames = (float*)malloc(sizeof(float) * 2 * n);
    float* out = (float*)malloc(sizeof(float) * 2 * n);
    float* eq = w->use_eq ? (float*)malloc(sizeof(float) * 2 * n) : NULL;
    int* smap = (int*)malloc(sizeof(int) * (size_t)w->K);
    if (!o || !sym || !frames || !out || !smap || (w->use_eq && !eq)) w->failed = 1;
    uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(w->cpu + 2);
    const float a = 0.70710678f;
    for (size_t i = 0; i < 2 * n && !w->failed; ++i) { /* QPSK symbols */
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        sym[i] = (s >> 40) & 1 ? a : -a;
    }
    for (size_t i = 0; i < n && eq && !w->failed; ++i) {   /* a smooth, nowhere-small equaliser vector */
        const float ph = 6.2831853f * (float)(i % N) / (float)N;
        eq[2 * i] = 1.0f + 0.4f * cosf(ph);
        eq[2 * i + 1] = 0.3f * sinf(ph);
    }
    for (int k = 0; k < w->K && smap; ++k) smap[k] = k;
    const float pts[8] = { -a, -a, a, -a, -a, a, a, a };
    if (!w->failed) gfdm_oracle_modulate(o, frames, sym, w->chunk);     /* receiver input: modulated frames (decisions well conditioned) */
    /* set-up (kernel object, buffers, input generation) is NOT timed: the clock starts when every worker has reached this point */
    pthread_mutex_lock(&w->gate->mu);
    ++w->gate->arrived;
    pthread_cond_broadcast(&w->gate->cv);
    while (!w->gate->go && !w->gate->abort) pthread_cond_wait(&w->gate->cv, &w->gate->mu);
    const int aborted = w->gate->abort;
    pthread_mutex_unlock(&w->gate->mu);
    const double deadline = now_s() + w->seconds;
    long done = 0;
    if (!w->failed && !aborted) do {
        if (w->mode == BENCH_MOD) {
            gfdm_oracle_modulate(o, frames, sym, w->chunk);
        } else if (w->mode == BENCH_MOD_DEMOD) {
            gfdm_oracle_modulate(o, frames, sym, w->chunk);
            gfdm_oracle_demodulate(o, out, frames, eq, w->chunk);
        } else if (w->mode == BENCH_DEMOD) {
            gfdm_oracle_demodulate(o, out, frames, eq, w->chunk);
        } else {
            gfdm_oracle_advanced_receive(o, out, frames, eq, w->chunk, smap, w->K, pts, 4, GFDM_ORACLE_DECIDE_QPSK, w->ic_iter, 0);
        }
        done += w->chunk;
    } while (now_s() < deadline);
    w->blocks = done;
    free(sym); free(frames); free(out); free(eq); free(smap);
    if (o) gfdm_oracle_destroy(o);
    return NULL;
}

/* Runs `nthreads` workers for about `seconds`; thread t is pinned to cpus[t] (cpus == NULL: not pinned).
 * mode: 0 modulate + demodulate, 1 demodulate, 2 demodulate + ic_iter IC rounds (QPSK, all subcarriers active), 3 modulate;
 * use_eq: with the per-block equaliser vector (generic_work_equalize).  chunk: blocks a worker processes between two looks at
 * the clock.  Returns the blocks processed by all workers (or -1), *elapsed_s = wall time from the moment every worker had finished
 * its set-up (the start gate) to the last join. */
long gfdm_oracle_bench(int timeslots, int subcarriers, int overlap, const float* taps, int ntaps, int mode, int use_eq, int ic_iter,
                       int nthreads, const int* cpus, double seconds, int chunk, double* elapsed_s)
{
    if (nthreads < 1 || chunk < 1) return -1;
    worker_t* w = (worker_t*)calloc((size_t)nthreads, sizeof(worker_t));
    pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
    if (!w || !th) { free(w); free(th); return -1; }
    gate_t gate = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, 0 };
    int started = 0;
    double t0 = 0.0;
    for (int t = 0; t < nthreads; ++t) {
        w[t] = (worker_t){ timeslots, subcarriers, overlap, ntaps, mode, use_eq, ic_iter, chunk, cpus ? cpus[t] : -1, taps,
                           seconds, &gate, &t0, 0, 0 };
    }
    for (int t = 0; t < nthreads; ++t) {
        if (pthread_create(&th[t], NULL, worker, &w[t]) != 0) break;
        ++started;
    }
    pthread_mutex_lock(&gate.mu);
    if (started < nthreads) {          /* not every worker exists: release the ones that do and give up cleanly */
        gate.abort = 1;
    } else {
        while (gate.arrived < nthreads) pthread_cond_wait(&gate.cv, &gate.mu);
        gate.go = 1;
    }
    pthread_cond_broadcast(&gate.cv);
    pthread_mutex_unlock(&gate.mu);
    t0 = now_s();
    if (started < nthreads) {
        for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
        free(w); free(th);
        return -1;
    }
    long total = 0;
    int failed = 0;
    for (int t = 0; t < nthreads; ++t) {
        pthread_join(th[t], NULL);
        total += w[t].blocks;
        failed |= w[t].failed;
    }
    if (elapsed_s) *elapsed_s = now_s() - t0;
    free(w); free(th);
    return failed ? -1 : total;
}
