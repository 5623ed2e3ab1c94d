/* CPU oracle (plain C, float32) for the gr-gfdm sparse-frequency-domain kernels.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg -- never by the product path.
 * See gfdm_oracle.c for the reference lines each function follows.
 *
 * All sample buffers are interleaved (re, im) float32, i.e. std::complex<float>.
 */
#ifndef GFDM_ORACLE_H
#define GFDM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gfdm_oracle gfdm_oracle;

enum { GFDM_ORACLE_DECIDE_NEAREST = 0, GFDM_ORACLE_DECIDE_QPSK = 1, GFDM_ORACLE_DECIDE_BPSK = 2 };

/* returns NULL when ntaps != timeslots*overlap (the reference's invalid_argument) */
gfdm_oracle* gfdm_oracle_create(int timeslots, int subcarriers, int overlap, const float* taps, int ntaps);
void gfdm_oracle_destroy(gfdm_oracle* o);
int gfdm_oracle_block_size(const gfdm_oracle* o);
void gfdm_oracle_filter_taps(const gfdm_oracle* o, float* out);    /* overlap*timeslots complex */
void gfdm_oracle_ic_filter_taps(const gfdm_oracle* o, float* out); /* timeslots complex */

void gfdm_oracle_modulate(gfdm_oracle* o, float* out, const float* in, long nblocks);
void gfdm_oracle_fft_filter_downsample(gfdm_oracle* o, float* out, const float* in, const float* f_eq, long nblocks);
void gfdm_oracle_transform_subcarriers_to_td(gfdm_oracle* o, float* out, const float* in, long nblocks);
void gfdm_oracle_cancel_sc_interference(gfdm_oracle* o, float* out, const float* td, const float* fd, long nblocks);
void gfdm_oracle_demodulate(gfdm_oracle* o, float* out, const float* in, const float* f_eq, long nblocks);
void gfdm_oracle_advanced_receive(gfdm_oracle* o, float* out, const float* in, const float* f_eq, long nblocks,
                                  const int* subcarrier_map, int nsubcarrier_map, const float* points, int npoints,
                                  int decision_kind, int ic_iter, int do_phase_compensation);

/* ---- composite transmitter: mapper -> modulator -> cyclic prefix/suffix + ramp -> preamble ---- */
typedef struct gfdm_oracle_tx gfdm_oracle_tx;
/* window: n_window complex taps (whole window or 2*ramp_len); preambles: n_shifts rows of preamble_len complex.
 * Returns NULL on the argument errors the reference constructors throw for. */
gfdm_oracle_tx* gfdm_oracle_tx_create(int timeslots, int subcarriers, int active_subcarriers, int cp_len, int cs_len, int ramp_len,
                                      const int* subcarrier_map, int per_timeslot, int overlap, const float* taps, int ntaps,
                                      const float* window, int n_window, const int* cyclic_shifts, int n_shifts,
                                      const float* preambles, int preamble_len);
void gfdm_oracle_tx_destroy(gfdm_oracle_tx* t);
int gfdm_oracle_tx_input_vector_size(const gfdm_oracle_tx* t);
int gfdm_oracle_tx_output_vector_size(const gfdm_oracle_tx* t);
/* transmitter_kernel::modulate + add_frame for cyclic_shifts[port]; in: nblocks x ninput_size symbols, out: nblocks x output_vector_size */
void gfdm_oracle_tx_work(gfdm_oracle_tx* t, float* out, const float* in, int ninput_size, long nblocks, int port);

/* ---- timing driver (gfdm_oracle_bench.c): nthreads pinned pthreads, one kernel object each, until a common deadline ---- */
long gfdm_oracle_bench(int timeslots, int subcarriers, int overlap, const float* taps, int ntaps, int mode, int use_eq, int ic_iter,
                       int nthreads, const int* cpus, double seconds, int chunk, double* elapsed_s);

#ifdef __cplusplus
}
#endif
#endif
