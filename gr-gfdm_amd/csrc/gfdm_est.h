// Preamble channel estimator, device-side pieces shared by the stand-alone estimator kernel (gfdm_generic.hip) and by the
// receivers that derive their one-tap equaliser from the received preamble inside the kernel (EQ_PREAMBLE).
// Restated from gr-gfdm lib/preamble_channel_estimator_cc.cc:
//   filter_preamble_estimate :147-187   active bins in fftshift order, DC interpolated when dc-free, 9-tap Gaussian
//   interpolate_frame        :238-273   linear interpolation K bins -> M*K bins, written here as a gather per output bin
#pragma once
#include "gfdm_plan.h"

namespace gfdm {

// one tap input of the smoothing filter: position u of the edge-replicated, fftshift-ordered active-bin array
__device__ __forceinline__ cf est_filter_src(const cf* est, int u, const EstPlan& e)
{
    int v = u - 4;                                   // 4 = taps / 2 replicated edge bins on either side
    v = v < 0 ? 0 : (v > e.n_est - 1 ? e.n_est - 1 : v);
    const int half = e.A >> 1;
    if (v < half) return est[e.K - half + v];
    if (e.dc_free && v == half) {
        const cf lo = est[e.K - 1], hi = est[1];
        return make_float2(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y));
    }
    return est[v - half];                            // dc-free: bins 1.., else bins 0..
}

// position of FFT bin b in the fftshift-ordered active-bin array (0 .. n_est - 1), -1 when the smoothing filter never reads the
// bin; the inverse of est_filter_src's mapping (the interpolated DC slot of the dc-free layout has no source bin)
__device__ __forceinline__ int est_active_pos(int b, const EstPlan& e)
{
    const int half = e.A >> 1;
    if (b >= e.K - half) return b - (e.K - half);
    if (b >= e.dc_free && b < e.dc_free + half) return half + b;
    return -1;
}

// row j of the [K][M] frame estimate (bins M j .. M j + M - 1) lies in ONE interpolation segment: its end points.
// filt[i * stride] = smoothed estimate bin i.
__device__ __forceinline__ void est_row_segment(const cf* filt, int stride, int j, const EstPlan& e, cf& lo, cf& hi)
{
    const int n_est = e.n_est;
    const int upper = n_est - 1 - n_est / 2, low_start = e.K / 2 + (e.K - e.A) / 2;
    int a, b;
    if (j < upper) { a = n_est / 2 + j; b = a + 1; }
    else if (j < e.K / 2) { a = b = n_est - 1; }
    else if (j < low_start) { a = b = 0; }
    else { a = j - low_start; b = a + 1; }
    lo = filt[a * stride];
    hi = filt[b * stride];
}

// smoothed estimate, bin i of n_est
__device__ __forceinline__ cf est_filter_bin(const cf* est, int i, const EstPlan& e)
{
    cf acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const cf v = est_filter_src(est, i + t, e);
        acc.x += v.x * e.gauss[t];
        acc.y += v.y * e.gauss[t];
    }
    return acc;
}

// frame estimate at FFT bin n of M*K from the smoothed estimate (MC > 0: timeslots known at compile time)
template <int MC>
__device__ __forceinline__ cf est_frame_bin(const cf* filt, int n, const EstPlan& e)
{
    const int M = MC > 0 ? MC : e.M;
    const int n_est = e.n_est, center = (M * e.K) / 2;
    const int upper_end = M * (n_est - 1 - n_est / 2);     // bins [0, upper_end): positive frequencies, interpolated
    const int low_start = center + M * (e.K - e.A) / 2;    // first interpolated negative-frequency bin
    int seg, j;
    if (n < upper_end) { seg = n / M; j = n - seg * M; seg += n_est / 2; }
    else if (n < center) return filt[n_est - 1];
    else if (n < low_start) return filt[0];
    else { seg = (n - low_start) / M; j = (n - low_start) - seg * M; }
    const cf lo = filt[seg], hi = filt[seg + 1];
    const float t = (float)j * (1.0f / (float)M);
    return make_float2(lo.x + (hi.x - lo.x) * t, lo.y + (hi.y - lo.y) * t);
}

}  // namespace gfdm
