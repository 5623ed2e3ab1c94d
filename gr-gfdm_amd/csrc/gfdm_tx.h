// Device-side helpers of the composite transmitter (gr-gfdm transmitter_kernel = resource mapper -> modulator ->
// cyclic prefix/suffix with cyclic shift + window ramp -> preamble), fused into the modulator kernels' load and store
// stages: the mapped symbol grid and the bare modulated block never exist in HBM.
//
// Restated from gr-gfdm: lib/resource_mapper_kernel_cc.cc:74-89,108-134 (map_to_resources),
// lib/add_cyclic_prefix_cc.cc:66-98 (add_cyclic_extension, apply_ramp), lib/transmitter_kernel.cc:78-107.
#pragma once
#include "gfdm_plan.h"
#include "gfdm_dft.h"

namespace gfdm {

constexpr int TX_MAX_PORTS = 8;

struct TxParams {
    int mapped;                 // 1: input is `nin` data symbols per block, placed by the resource mapper; 0: full K x M grid
    int framed;                 // 1: output is preamble + cyclic prefix + block + cyclic suffix per port; 0: bare block
    int A;                      // active subcarriers
    int per_timeslot;           // symbol order of the mapper
    int nin;                    // symbols per block in the input (<= A * M, the rest of the grid is zero)
    int cp, cs, ramp;           // cyclic prefix / suffix / ramp lengths
    int plen;                   // preamble length
    int F;                      // output samples per block and port = plen + cp + N + cs
    int nports;
    int shifts[TX_MAX_PORTS];   // cyclic shift of every port
    cf* outs[TX_MAX_PORTS];     // output base pointer of every port
    const short* rank;          // [K] position of subcarrier k in the SORTED subcarrier map, -1 when inactive
    const cf* front;            // [ramp] window taps of the rising ramp
    const cf* back;             // [ramp] window taps of the falling ramp
    const cf* preambles;        // [nports][plen]
};

// D[k][t] of the mapped grid (zero for inactive subcarriers and beyond the supplied symbols)
__device__ __forceinline__ cf tx_symbol(const TxParams& t, const cf* __restrict__ in_block, int M, int k, int ti)
{
    const int a = t.rank[k];
    if (a < 0) return make_float2(0.f, 0.f);
    const int idx = t.per_timeslot ? (ti * t.A + a) : (a * M + ti);
    return (idx < t.nin) ? dft::ld_stream(in_block + idx) : make_float2(0.f, 0.f);
}

__device__ __forceinline__ cf tx_window(const TxParams& t, int f, int N, cf x)
{
    const int FL = t.cp + N + t.cs;
    if (f < t.ramp) {
        const cf w = t.front[f];
        return make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x);
    }
    if (f >= FL - t.ramp) {
        const cf w = t.back[f - (FL - t.ramp)];
        return make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x);
    }
    return x;
}

// time sample n of block blk: body position plus, where it applies, its copy in the cyclic prefix / suffix, every port
__device__ __forceinline__ void tx_store_sample(const TxParams& t, int64_t blk, int N, int n, cf x)
{
    // one port: non-temporal stores like the other kernels (15.2 vs 17.4 us per 4096 frames); several ports: plain stores -- the
    // frames of the ports are not line-aligned, a wave writes many partial lines, and streaming them was slower (4 ports: 40 vs 35 us
    // per 4096 frames, 482 vs 468 us per 65 536)
    if (t.nports == 1) {
        cf* o = t.outs[0] + blk * (int64_t)t.F + t.plen;
        const int s = t.shifts[0];
        const int scp = t.cp + s, scs = t.cs - s;
        dft::st_stream(o, scp + n, tx_window(t, scp + n, N, x));
        if (n >= N - scp) dft::st_stream(o, n - (N - scp), tx_window(t, n - (N - scp), N, x));
        if (n < scs) dft::st_stream(o, scp + N + n, tx_window(t, scp + N + n, N, x));
        return;
    }
    for (int port = 0; port < t.nports; ++port) {
        cf* o = t.outs[port] + blk * (int64_t)t.F + t.plen;
        const int s = t.shifts[port];
        const int scp = t.cp + s, scs = t.cs - s;
        o[scp + n] = tx_window(t, scp + n, N, x);
        if (n >= N - scp) o[n - (N - scp)] = tx_window(t, n - (N - scp), N, x);
        if (n < scs) o[scp + N + n] = tx_window(t, scp + N + n, N, x);
    }
}

// preamble samples j = first, first + step, ... of every port
__device__ __forceinline__ void tx_store_preamble(const TxParams& t, int64_t blk, int first, int step)
{
    for (int port = 0; port < t.nports; ++port) {
        cf* o = t.outs[port] + blk * (int64_t)t.F;
        const cf* pre = t.preambles + (int64_t)port * t.plen;
        if (t.nports == 1) { for (int j = first; j < t.plen; j += step) dft::st_stream(o, j, pre[j]); }
        else { for (int j = first; j < t.plen; j += step) o[j] = pre[j]; }
    }
}

}  // namespace gfdm
