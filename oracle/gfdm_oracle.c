/* CPU oracle (plain C, float32) for the gr-gfdm sparse-frequency-domain kernels.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/, __graft_entry__.smoke() and
 * the timed CPU baseline ("port") of bench.py.  The product path never links it.
 *
 * Restates, block by block and in the reference's own loop structure (one
 * kernel object, one block per call, K tiny FFTs bracketed by copies, L
 * multiply/accumulate passes of length M per subcarrier), the algorithm of
 *
 *   modulator_kernel_cc::generic_work                lib/modulator_kernel_cc.cc:98-141
 *   tap normalisation                                lib/modulator_kernel_cc.cc:70-85, lib/receiver_kernel_cc.cc:99-113
 *   IC taps                                          lib/receiver_kernel_cc.cc:56-63
 *   filter_subcarriers_and_downsample_fd             lib/receiver_kernel_cc.cc:165-192
 *   transform_subcarriers_to_td                      lib/receiver_kernel_cc.cc:211-225
 *   cancel_sc_interference                           lib/receiver_kernel_cc.cc:274-299
 *   fft_[equalize_]filter_downsample, generic_work*  lib/receiver_kernel_cc.cc:301-334
 *   perform_ic_iterations, phase offset, decisions   lib/advanced_receiver_kernel_cc.cc:56-123
 *
 * The reference delegates its transforms to FFTW3f and its vector arithmetic
 * to VOLK (neither vendored, neither installed here).  Their published
 * contracts are restated: an unnormalised out-of-place complex DFT, forward =
 * exp(-j...), backward = exp(+j...); element-wise complex multiply / divide /
 * add / subtract / scale in float32.  The FFT below is an own mixed-radix
 * Stockham implementation (radix 4/2/3/5 butterflies, direct O(p^2) butterfly
 * for other primes).
 *
 * Pinning: checked against the pygfdm golden vectors in tests/golden/ --
 * modulator, receiver (overlap 2), IC taps, cancel_sc_interference and the IC
 * loop (make_golden.py, make_golden_ic.py) -- and the numpy restatement
 * oracle/gfdm_ref.py (tests/test_oracle.py).  The C++ reference could not be
 * built here (FFTW3f/VOLK/GNU Radio absent), so there is no oracle/_ref.
 * gfdm_oracle_bench.c is the multi-threaded timing driver of bench.py's
 * cpu_baseline leg.
 */
#define _GNU_SOURCE
#include "gfdm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float re, im; } cf;

static inline cf cf_mul(cf a, cf b) { cf r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return r; }
static inline cf cf_add(cf a, cf b) { cf r = { a.re + b.re, a.im + b.im }; return r; }
static inline cf cf_sub(cf a, cf b) { cf r = { a.re - b.re, a.im - b.im }; return r; }

/* ------------------------------------------------------------------ FFT -- */

#define OFFT_MAX_STAGES 32

typedef struct {
    int n;
    int sign;                 /* -1 forward, +1 backward */
    int nstages;
    int radix[OFFT_MAX_STAGES];
    cf* stage_tw[OFFT_MAX_STAGES];   /* per stage: (len/radix) * (radix-1) twiddles W_len^{p*u} */
    cf* prime_tw[OFFT_MAX_STAGES];   /* per stage with generic radix r: W_r^{t} table, r entries */
    cf* work;                 /* n scratch */
    cf* work2;                /* n scratch (keeps caller buffers untouched) */
} offt_plan;

static cf unit_root(long num, long den, int sign)
{
    double a = 2.0 * M_PI * (double)(num % den) / (double)den;
    cf r = { (float)cos(a), (float)(sign * sin(a)) };
    return r;
}

static void offt_destroy(offt_plan* p)
{
    if (!p) return;
    for (int s = 0; s < p->nstages; ++s) { free(p->stage_tw[s]); free(p->prime_tw[s]); }
    free(p->work); free(p->work2); free(p);
}

static offt_plan* offt_create(int n, int forward)
{
    offt_plan* p = (offt_plan*)calloc(1, sizeof(offt_plan));
    p->n = n; p->sign = forward ? -1 : +1;
    int rem = n;
    while (rem % 4 == 0) { p->radix[p->nstages++] = 4; rem /= 4; }
    while (rem % 2 == 0) { p->radix[p->nstages++] = 2; rem /= 2; }
    for (int f = 3; rem > 1; f += 2)
        while (rem % f == 0) { p->radix[p->nstages++] = f; rem /= f; }
    int len = n;
    for (int s = 0; s < p->nstages; ++s) {
        int r = p->radix[s], m = len / r;
        p->stage_tw[s] = (cf*)malloc(sizeof(cf) * (size_t)m * (r - 1));
        for (int q = 0; q < m; ++q)
            for (int u = 1; u < r; ++u)
                p->stage_tw[s][q * (r - 1) + (u - 1)] = unit_root((long)q * u, len, p->sign);
        if (r != 2 && r != 4) {
            p->prime_tw[s] = (cf*)malloc(sizeof(cf) * r);
            for (int t = 0; t < r; ++t) p->prime_tw[s][t] = unit_root(t, r, p->sign);
        }
        len = m;
    }
    p->work = (cf*)malloc(sizeof(cf) * n);
    p->work2 = (cf*)malloc(sizeof(cf) * n);
    return p;
}

/* One decimation-in-frequency Stockham pass: sequence length len, stride str. */
static void offt_pass(const offt_plan* p, int s, int len, int str, const cf* x, cf* y)
{
    const int r = p->radix[s], m = len / r;
    const cf* tw = p->stage_tw[s];
    const float sg = (float)p->sign;
    if (r == 2) {
        for (int q = 0; q < m; ++q) {
            const cf w = tw[q];
            for (int j = 0; j < str; ++j) {
                cf a = x[j + str * q], b = x[j + str * (q + m)];
                y[j + str * (2 * q)] = cf_add(a, b);
                y[j + str * (2 * q + 1)] = cf_mul(cf_sub(a, b), w);
            }
        }
    } else if (r == 4) {
        for (int q = 0; q < m; ++q) {
            const cf w1 = tw[3 * q], w2 = tw[3 * q + 1], w3 = tw[3 * q + 2];
            for (int j = 0; j < str; ++j) {
                cf a = x[j + str * q], b = x[j + str * (q + m)], c = x[j + str * (q + 2 * m)], d = x[j + str * (q + 3 * m)];
                cf apc = cf_add(a, c), amc = cf_sub(a, c), bpd = cf_add(b, d), bmd = cf_sub(b, d);
                cf jbmd = { -sg * bmd.im, sg * bmd.re };          /* (sign j) * (b - d) */
                y[j + str * (4 * q)] = cf_add(apc, bpd);
                y[j + str * (4 * q + 1)] = cf_mul(cf_add(amc, jbmd), w1);
                y[j + str * (4 * q + 2)] = cf_mul(cf_sub(apc, bpd), w2);
                y[j + str * (4 * q + 3)] = cf_mul(cf_sub(amc, jbmd), w3);
            }
        }
    } else if (r == 3) {
        const cf e = p->prime_tw[s][1];                          /* W_3 */
        for (int q = 0; q < m; ++q) {
            const cf w1 = tw[2 * q], w2 = tw[2 * q + 1];
            for (int j = 0; j < str; ++j) {
                cf a = x[j + str * q], b = x[j + str * (q + m)], c = x[j + str * (q + 2 * m)];
                cf bpc = cf_add(b, c), bmc = cf_sub(b, c);
                cf t = { a.re + e.re * bpc.re, a.im + e.re * bpc.im };
                cf u = { -e.im * bmc.im, e.im * bmc.re };          /* j*Im(W3)*(b-c) */
                y[j + str * (3 * q)] = cf_add(a, bpc);
                y[j + str * (3 * q + 1)] = cf_mul(cf_add(t, u), w1);
                y[j + str * (3 * q + 2)] = cf_mul(cf_sub(t, u), w2);
            }
        }
    } else {
        const cf* wr = p->prime_tw[s];
        cf a[64];
        cf* big = NULL;
        cf* v = a;
        if (r > 64) { big = (cf*)malloc(sizeof(cf) * r); v = big; }
        for (int q = 0; q < m; ++q) {
            for (int j = 0; j < str; ++j) {
                for (int t = 0; t < r; ++t) v[t] = x[j + str * (q + t * m)];
                for (int u = 0; u < r; ++u) {
                    cf acc = v[0];
                    int idx = 0;
                    for (int t = 1; t < r; ++t) {
                        idx += u; if (idx >= r) idx -= r;
                        acc = cf_add(acc, cf_mul(v[t], wr[idx]));
                    }
                    y[j + str * (r * q + u)] = (u == 0) ? acc : cf_mul(acc, tw[q * (r - 1) + (u - 1)]);
                }
            }
        }
        free(big);
    }
}

/* out-of-place, unnormalised; in and out must not alias */
static void offt_execute(const offt_plan* p, const cf* in, cf* out)
{
    if (p->nstages == 0) { out[0] = in[0]; return; }
    const cf* src = in;
    int len = p->n, str = 1;
    for (int s = 0; s < p->nstages; ++s) {
        cf* dst = (s == p->nstages - 1) ? out : ((s & 1) ? p->work2 : p->work);
        offt_pass(p, s, len, str, src, dst);
        len /= p->radix[s]; str *= p->radix[s];
        src = dst;
    }
}

/* --------------------------------------------------------------- kernel -- */

struct gfdm_oracle {
    int M, K, L, N;
    cf* taps;       /* L*M, normalised */
    cf* ictaps;     /* M */
    offt_plan *fft_m, *ifft_m, *fft_n, *ifft_n;
    cf *sub_in, *sub_out, *filtered;      /* M */
    cf *big_in, *big_out, *equalized;     /* N */
    cf *sc_filtered, *freq_block, *ic_time, *ic_freq;   /* N */
};

gfdm_oracle* gfdm_oracle_create(int M, int K, int L, const float* taps, int ntaps)
{
    if (ntaps != M * L || M < 1 || K < 1 || L < 1) return NULL;
    gfdm_oracle* o = (gfdm_oracle*)calloc(1, sizeof(gfdm_oracle));
    o->M = M; o->K = K; o->L = L; o->N = M * K;
    o->taps = (cf*)malloc(sizeof(cf) * M * L);
    o->ictaps = (cf*)calloc(M, sizeof(cf));
    /* energy = |sum t*conj(t)| accumulated in float32 as the dot-product primitive does;
     * factor formed in double then cast (modulator_kernel_cc.cc:80-81) */
    cf acc = { 0.f, 0.f };
    const cf* t = (const cf*)taps;
    for (int i = 0; i < M * L; ++i) { acc.re += t[i].re * t[i].re + t[i].im * t[i].im; }
    const float scale = (float)(1.0 / sqrt(fabs((double)acc.re) / M));
    for (int i = 0; i < M * L; ++i) { o->taps[i].re = t[i].re * scale; o->taps[i].im = t[i].im * scale; }
    if (L >= 2)
        for (int m = 0; m < M; ++m) o->ictaps[m] = cf_mul(o->taps[m], o->taps[M * (L - 1) + m]);
    o->fft_m = offt_create(M, 1); o->ifft_m = offt_create(M, 0);
    o->fft_n = offt_create(o->N, 1); o->ifft_n = offt_create(o->N, 0);
    o->sub_in = (cf*)malloc(sizeof(cf) * M); o->sub_out = (cf*)malloc(sizeof(cf) * M); o->filtered = (cf*)malloc(sizeof(cf) * M);
    o->big_in = (cf*)malloc(sizeof(cf) * o->N); o->big_out = (cf*)malloc(sizeof(cf) * o->N); o->equalized = (cf*)malloc(sizeof(cf) * o->N);
    o->sc_filtered = (cf*)malloc(sizeof(cf) * o->N); o->freq_block = (cf*)malloc(sizeof(cf) * o->N);
    o->ic_time = (cf*)malloc(sizeof(cf) * o->N); o->ic_freq = (cf*)malloc(sizeof(cf) * o->N);
    return o;
}

void gfdm_oracle_destroy(gfdm_oracle* o)
{
    if (!o) return;
    offt_destroy(o->fft_m); offt_destroy(o->ifft_m); offt_destroy(o->fft_n); offt_destroy(o->ifft_n);
    free(o->taps); free(o->ictaps); free(o->sub_in); free(o->sub_out); free(o->filtered);
    free(o->big_in); free(o->big_out); free(o->equalized); free(o->sc_filtered); free(o->freq_block);
    free(o->ic_time); free(o->ic_freq); free(o);
}

int gfdm_oracle_block_size(const gfdm_oracle* o) { return o->N; }
void gfdm_oracle_filter_taps(const gfdm_oracle* o, float* out) { memcpy(out, o->taps, sizeof(cf) * o->M * o->L); }
void gfdm_oracle_ic_filter_taps(const gfdm_oracle* o, float* out) { memcpy(out, o->ictaps, sizeof(cf) * o->M); }

/* modulator_kernel_cc::generic_work, one block */
static void modulate_block(gfdm_oracle* o, cf* out, const cf* in)
{
    const int M = o->M, K = o->K, L = o->L, N = o->N;
    const int part_len = (M * L / 2 < M) ? M * L / 2 : M;
    memset(o->big_in, 0, sizeof(cf) * N);
    for (int k = 0; k < K; ++k) {
        memcpy(o->sub_in, in + (size_t)k * M, sizeof(cf) * M);
        offt_execute(o->fft_m, o->sub_in, o->sub_out);
        for (int i = 0; i < L; ++i) {
            const cf* tp = o->taps + ((i + L / 2) % L) * M;
            cf* dst = o->big_in + ((k + i + K - L / 2) % K) * M;
            for (int m = 0; m < M; ++m) o->filtered[m] = cf_mul(o->sub_out[m], tp[m]);
            for (int m = 0; m < part_len; ++m) dst[m] = cf_add(dst[m], o->filtered[m]);
        }
    }
    offt_execute(o->ifft_n, o->big_in, o->big_out);
    const float s = (float)(1.0 / N);
    for (int n = 0; n < N; ++n) { out[n].re = o->big_out[n].re * s; out[n].im = o->big_out[n].im * s; }
}

/* filter_subcarriers_and_downsample_fd */
static void filter_downsample_fd(gfdm_oracle* o, cf* out, const cf* X)
{
    const int M = o->M, K = o->K, L = o->L;
    memset(out, 0, sizeof(cf) * o->N);
    for (int k = 0; k < K; ++k) {
        cf* dst = out + (size_t)k * M;
        for (int i = 0; i < L; ++i) {
            const cf* src = X + ((k + i + K - L / 2) % K) * M;
            const cf* tp = o->taps + ((i + L / 2) % L) * M;
            for (int m = 0; m < M; ++m) o->filtered[m] = cf_mul(tp[m], src[m]);
            for (int m = 0; m < M; ++m) dst[m] = cf_add(dst[m], o->filtered[m]);
        }
    }
}

static void fft_filter_downsample_block(gfdm_oracle* o, cf* out, const cf* in, const cf* f_eq)
{
    memcpy(o->big_in, in, sizeof(cf) * o->N);
    offt_execute(o->fft_n, o->big_in, o->big_out);
    if (f_eq) {
        for (int n = 0; n < o->N; ++n) {            /* full complex divide, a / b = a conj(b) / |b|^2 */
            const cf a = o->big_out[n], b = f_eq[n];
            const float d = b.re * b.re + b.im * b.im;
            o->equalized[n].re = (a.re * b.re + a.im * b.im) / d;
            o->equalized[n].im = (a.im * b.re - a.re * b.im) / d;
        }
        filter_downsample_fd(o, out, o->equalized);
    } else {
        filter_downsample_fd(o, out, o->big_out);
    }
}

static void to_td_block(gfdm_oracle* o, cf* out, const cf* in)
{
    const int M = o->M, K = o->K;
    const float s = (float)(1.0 / M);
    for (int k = 0; k < K; ++k) {
        memcpy(o->sub_in, in + (size_t)k * M, sizeof(cf) * M);
        offt_execute(o->ifft_m, o->sub_in, o->sub_out);
        for (int m = 0; m < M; ++m) { out[(size_t)k * M + m].re = o->sub_out[m].re * s; out[(size_t)k * M + m].im = o->sub_out[m].im * s; }
    }
}

static void cancel_block(gfdm_oracle* o, cf* out, const cf* td, const cf* fd)
{
    const int M = o->M, K = o->K;
    for (int k = 0; k < K; ++k) {
        const cf* prev = td + (size_t)((k - 1 + K) % K) * M;
        const cf* next = td + (size_t)((k + 1 + K) % K) * M;
        for (int m = 0; m < M; ++m) o->sub_in[m] = cf_add(prev[m], next[m]);
        offt_execute(o->fft_m, o->sub_in, o->sub_out);
        for (int m = 0; m < M; ++m) o->filtered[m] = cf_mul(o->ictaps[m], o->sub_out[m]);
        for (int m = 0; m < M; ++m) out[(size_t)k * M + m] = cf_sub(fd[(size_t)k * M + m], o->filtered[m]);
    }
}

void gfdm_oracle_modulate(gfdm_oracle* o, float* out, const float* in, long nblocks)
{
    for (long b = 0; b < nblocks; ++b) modulate_block(o, (cf*)out + b * o->N, (const cf*)in + b * o->N);
}

void gfdm_oracle_fft_filter_downsample(gfdm_oracle* o, float* out, const float* in, const float* f_eq, long nblocks)
{
    for (long b = 0; b < nblocks; ++b)
        fft_filter_downsample_block(o, (cf*)out + b * o->N, (const cf*)in + b * o->N, f_eq ? (const cf*)f_eq + b * o->N : NULL);
}

void gfdm_oracle_transform_subcarriers_to_td(gfdm_oracle* o, float* out, const float* in, long nblocks)
{
    for (long b = 0; b < nblocks; ++b) to_td_block(o, (cf*)out + b * o->N, (const cf*)in + b * o->N);
}

void gfdm_oracle_cancel_sc_interference(gfdm_oracle* o, float* out, const float* td, const float* fd, long nblocks)
{
    for (long b = 0; b < nblocks; ++b)
        cancel_block(o, (cf*)out + b * o->N, (const cf*)td + b * o->N, (const cf*)fd + b * o->N);
}

void gfdm_oracle_demodulate(gfdm_oracle* o, float* out, const float* in, const float* f_eq, long nblocks)
{
    for (long b = 0; b < nblocks; ++b) {
        fft_filter_downsample_block(o, o->sc_filtered, (const cf*)in + b * o->N, f_eq ? (const cf*)f_eq + b * o->N : NULL);
        to_td_block(o, (cf*)out + b * o->N, o->sc_filtered);
    }
}

static int decide(const cf x, const cf* pts, int npts, int kind)
{
    if (kind == GFDM_ORACLE_DECIDE_QPSK) return 2 * (x.im > 0.f) + (x.re > 0.f);
    if (kind == GFDM_ORACLE_DECIDE_BPSK) return (x.re > 0.f);
    int best = 0; float bd = INFINITY;
    for (int i = 0; i < npts; ++i) {
        const float dr = x.re - pts[i].re, di = x.im - pts[i].im, d = dr * dr + di * di;
        if (d < bd) { bd = d; best = i; }
    }
    return best;
}

void gfdm_oracle_advanced_receive(gfdm_oracle* o, float* out_f, const float* in_f, const float* f_eq_f, long nblocks,
                                  const int* smap, int nsmap, const float* points_f, int npoints,
                                  int kind, int ic_iter, int do_pc)
{
    const int M = o->M, N = o->N;
    const cf* pts = (const cf*)points_f;
    for (long b = 0; b < nblocks; ++b) {
        cf* out = (cf*)out_f + b * N;
        fft_filter_downsample_block(o, o->freq_block, (const cf*)in_f + b * N, f_eq_f ? (const cf*)f_eq_f + b * N : NULL);
        to_td_block(o, out, o->freq_block);
        for (int j = 0; j < ic_iter; ++j) {
            memset(o->ic_time, 0, sizeof(cf) * N);
            for (int a = 0; a < nsmap; ++a)
                for (int m = 0; m < M; ++m) {
                    const int pos = smap[a] * M + m;
                    o->ic_time[pos] = pts[decide(out[pos], pts, npoints, kind)];
                }
            if (do_pc > 0 && j == 0) {
                float acc = 0.f;
                for (int a = 0; a < nsmap; ++a)
                    for (int m = 0; m < M; ++m) {
                        const int pos = smap[a] * M + m;
                        acc += atan2f(o->ic_time[pos].im, o->ic_time[pos].re) - atan2f(out[pos].im, out[pos].re);
                    }
                const float phi = acc / (float)(nsmap * M);
                const cf rot = { cosf(phi), sinf(phi) };
                for (int n = 0; n < N; ++n) o->freq_block[n] = cf_mul(o->freq_block[n], rot);
            }
            cancel_block(o, o->ic_freq, o->ic_time, o->freq_block);
            to_td_block(o, out, o->ic_freq);
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------------
 * Composite transmitter: restates lib/resource_mapper_kernel_cc.cc:74-89,108-134 (map_to_resources),
 * lib/add_cyclic_prefix_cc.cc:66-98 (cyclic extension with shift, ramp) and lib/transmitter_kernel.cc:78-107. */
struct gfdm_oracle_tx {
    gfdm_oracle* mod;
    int M, K, A, cp, cs, ramp, per_timeslot, nshifts, plen;
    int* smap;            /* sorted, as the reference constructor leaves it */
    int* shifts;
    cf *front, *back;     /* ramp_len taps each */
    cf* preambles;        /* nshifts x plen */
    cf *mapped, *frame;   /* N each */
};

static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

gfdm_oracle_tx* gfdm_oracle_tx_create(int M, int K, int A, int cp, int cs, int ramp, const int* smap, int per_timeslot, int L,
                                      const float* taps, int ntaps, const float* window, int n_window, const int* shifts, int nshifts,
                                      const float* preambles, int plen)
{
    const int N = M * K;
    if (A > K || A < 1 || nshifts < 1 || plen < 0) return NULL;
    if (n_window != N + cp + cs && n_window != 2 * ramp) return NULL;
    gfdm_oracle* mod = gfdm_oracle_create(M, K, L, taps, ntaps);
    if (!mod) return NULL;
    gfdm_oracle_tx* t = (gfdm_oracle_tx*)calloc(1, sizeof(*t));
    t->mod = mod; t->M = M; t->K = K; t->A = A; t->cp = cp; t->cs = cs; t->ramp = ramp; t->per_timeslot = per_timeslot;
    t->nshifts = nshifts; t->plen = plen;
    t->smap = (int*)malloc(sizeof(int) * A); memcpy(t->smap, smap, sizeof(int) * A); qsort(t->smap, A, sizeof(int), cmp_int);
    t->shifts = (int*)malloc(sizeof(int) * nshifts); memcpy(t->shifts, shifts, sizeof(int) * nshifts);
    t->front = (cf*)malloc(sizeof(cf) * (ramp > 0 ? ramp : 1)); t->back = (cf*)malloc(sizeof(cf) * (ramp > 0 ? ramp : 1));
    memcpy(t->front, window, sizeof(cf) * ramp);
    memcpy(t->back, (const cf*)window + (n_window - ramp), sizeof(cf) * ramp);
    t->preambles = (cf*)malloc(sizeof(cf) * (size_t)nshifts * (plen > 0 ? plen : 1));
    memcpy(t->preambles, preambles, sizeof(cf) * (size_t)nshifts * plen);
    t->mapped = (cf*)malloc(sizeof(cf) * N); t->frame = (cf*)malloc(sizeof(cf) * N);
    return t;
}

void gfdm_oracle_tx_destroy(gfdm_oracle_tx* t)
{
    if (!t) return;
    gfdm_oracle_destroy(t->mod);
    free(t->smap); free(t->shifts); free(t->front); free(t->back); free(t->preambles); free(t->mapped); free(t->frame); free(t);
}

int gfdm_oracle_tx_input_vector_size(const gfdm_oracle_tx* t) { return t->A * t->M; }
int gfdm_oracle_tx_output_vector_size(const gfdm_oracle_tx* t) { return t->plen + t->cp + t->M * t->K + t->cs; }

void gfdm_oracle_tx_work(gfdm_oracle_tx* t, float* out_f, const float* in_f, int nin, long nblocks, int port)
{
    const int M = t->M, K = t->K, A = t->A, N = M * K, F = gfdm_oracle_tx_output_vector_size(t);
    const int shift = t->shifts[port];
    for (long b = 0; b < nblocks; ++b) {
        const cf* in = (const cf*)in_f + b * nin;
        cf* out = (cf*)out_f + b * F;
        memset(t->mapped, 0, sizeof(cf) * N);                                   /* map_to_resources */
        int ctr = 0;
        if (t->per_timeslot) {
            for (int ti = 0; ti < M; ++ti)
                for (int a = 0; a < A; ++a, ++ctr)
                    if (ctr < nin) t->mapped[M * t->smap[a] + ti] = in[ctr];
        } else {
            for (int a = 0; a < A; ++a)
                for (int ti = 0; ti < M; ++ti, ++ctr)
                    if (ctr < nin) t->mapped[M * t->smap[a] + ti] = in[ctr];
        }
        modulate_block(t->mod, t->frame, t->mapped);
        memcpy(out, t->preambles + (size_t)port * t->plen, sizeof(cf) * t->plen);   /* insert_preamble */
        cf* body = out + t->plen;                                                /* add_cyclic_extension */
        const int scp = t->cp + shift, scs = t->cs - shift;
        memcpy(body, t->frame + (N - scp), sizeof(cf) * scp);
        memcpy(body + scp, t->frame, sizeof(cf) * N);
        memcpy(body + scp + N, t->frame, sizeof(cf) * scs);
        const int tail = N + t->cp + t->cs - t->ramp;                             /* apply_ramp */
        for (int i = 0; i < t->ramp; ++i) { body[i] = cf_mul(body[i], t->front[i]); body[tail + i] = cf_mul(body[tail + i], t->back[i]); }
    }
}
