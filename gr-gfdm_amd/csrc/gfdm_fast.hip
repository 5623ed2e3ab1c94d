// Tuned HIP kernel family for gfx950: register-resident K x M block tiles, one wavefront per workgroup.
//
// Decomposition (n = K p + q, f = M j + m):
//     X[M j + m] = sum_q W_K^{q j} * W_N^{q m} * ( sum_p x[K p + q] W_M^{p m} )
//
// Thread layout.  A GFDM block is owned by TPB = K/4 lanes; lane t keeps the FOUR subcarrier rows
// q = t + (K/4) r, r = 0..3, each with all M timeslot columns, in VGPRs (8 M registers).  A 64-lane
// wavefront therefore carries G = 64 / TPB independent blocks (4 blocks at K = 64) and all lanes do
// identical work in every phase:
//   phase A   M-point DFT per row (compile-time codelet, gfdm_dft.h) + twiddle W_N^{q m}
//   phase B   K-point FFT over the subcarrier axis as radix-4 Stockham passes.  With this row
//             assignment the inputs of EVERY pass are exactly the four rows a lane already holds
//             (j + str (q' + m_s r) = t + (K/4) r), so each pass is four in-register butterflies per
//             column; only the pass OUTPUTS move, through an LDS tile (ds_write_b64 / ds_read_b64).
//   phase C   one-tap equaliser on the LDS tile in linear order (coalesced f_eq loads)
//   phase D   L-tap filter + fold over neighbouring subcarriers (rows 4t-L/2 .. 4t+3+L-1-L/2 from the
//             tile), M-point inverse DFT per row, optional interference-cancellation rounds with the
//             decided symbols exchanged through the tile, output staged through LDS so that the
//             global store is linear 16 B per lane.
// HBM traffic is exactly: read x (+ f_eq), write out.  The modulator is the transposed flow.
//
// Algorithm restated from gr-gfdm: lib/modulator_kernel_cc.cc:98-141, lib/receiver_kernel_cc.cc:165-334,
// lib/advanced_receiver_kernel_cc.cc:56-123.
#include "gfdm_dft.h"
#include "gfdm_plan.h"

namespace gfdm {
namespace {

using namespace dft;

constexpr int WAVE = 64;

template <int K> struct FftShape {
    static constexpr int log2K = (K == 4) ? 2 : (K == 8) ? 3 : (K == 16) ? 4 : (K == 32) ? 5 : (K == 64) ? 6 : (K == 128) ? 7 : (K == 256) ? 8 : -1;
    static_assert(log2K > 0, "fast family supports K = 4 .. 256, power of two");
    static constexpr int NP4 = log2K / 2;          // radix-4 passes
    static constexpr bool HAS2 = (log2K & 1) != 0; // one trailing radix-2 pass
    static constexpr int TPB = K / 4;              // lanes per block
    static constexpr int G = WAVE / TPB;           // blocks per wavefront
};

constexpr int ipow4(int s) { return 1 << (2 * s); }

template <int K, int M> struct Tile {
    static constexpr int N = K * M;
    static constexpr int PAD = (FftShape<K>::G > 1) ? 16 : 0;     // complex; makes the tile stride == 32 (mod 64) dwords
    static constexpr int STRIDE = N + PAD;                         // complex elements per block tile
};

// LDS access with compile-time element offsets (become the instruction's immediate offset field)
__device__ __forceinline__ void lds_store(cf* base, int off, cf v) { base[off] = v; }
__device__ __forceinline__ cf lds_load(const cf* base, int off) { return base[off]; }

// One wavefront == one workgroup: the barrier is only a wave-level ordering point for LDS.
__device__ __forceinline__ void wave_sync() { __syncthreads(); }

// ---- phase B: in-register radix-4 passes over the subcarrier axis -------------------------------------------------
// v[r][m]: row t + (K/4) r.  tw: per-lane forward roots for the exchanging passes, tw[s][u-1] = W_K^{q u 4^s}.
template <int K, int M, bool INV>
__device__ __forceinline__ void subcarrier_fft(cf (&v)[4][M], cf* tile, int t, const cf (&tw)[FftShape<K>::NP4][3])
{
    using S = FftShape<K>;
    static_for<0, S::NP4>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int str = ipow4(s);
        constexpr int len = K / str;
        constexpr int ms = len / 4;                       // q range of this pass
        static_for<0, M>([&](auto mi) {
            constexpr int m = decltype(mi)::value;
            cf x[4] = { v[0][m], v[1][m], v[2][m], v[3][m] };
            Dft<4, INV>::run(x);
            if constexpr (ms > 1) {                       // W_len^{q u}
                x[1] = cmul_dir<INV>(x[1], tw[s][0]);
                x[2] = cmul_dir<INV>(x[2], tw[s][1]);
                x[3] = cmul_dir<INV>(x[3], tw[s][2]);
            }
            v[0][m] = x[0]; v[1][m] = x[1]; v[2][m] = x[2]; v[3][m] = x[3];
        });
        if constexpr (ms > 1) {
            // outputs go to rows j + str (4 q + u); the next pass wants rows t + (K/4) r
            const int j = t & (str - 1), q = t / str;
            cf* wbase = tile + (j + 4 * str * q) * M;
            wave_sync();                                   // previous readers of the tile are done
            static_for<0, 4>([&](auto ui) {
                constexpr int u = decltype(ui)::value;
                static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; lds_store(wbase, str * u * M + m, v[u][m]); });
            });
            wave_sync();
            const cf* rbase = tile + t * M;
            static_for<0, 4>([&](auto ri) {
                constexpr int r = decltype(ri)::value;
                static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; v[r][m] = lds_load(rbase, (K / 4) * r * M + m); });
            });
        }
    });
    if constexpr (S::HAS2) {                              // len 2, stride K/2: rows (t, t+K/2) and (t+K/4, t+3K/4)
        static_for<0, M>([&](auto mi) {
            constexpr int m = decltype(mi)::value;
            const cf a = v[0][m], b = v[2][m], c = v[1][m], d = v[3][m];
            v[0][m] = a + b; v[2][m] = a - b; v[1][m] = c + d; v[3][m] = c - d;
        });
    }
}
// NOTE on HAS2: with a trailing radix-2 pass the LAST radix-4 pass has ms == 2 and exchanges through LDS like the others.

template <int K>
__device__ __forceinline__ void load_pass_twiddles(cf (&tw)[FftShape<K>::NP4][3], const cf* __restrict__ wK, int t)
{
    static_for<0, FftShape<K>::NP4>([&](auto si) {
        constexpr int s = decltype(si)::value;
        constexpr int str = ipow4(s);
        const int q = t / str;
        static_for<0, 3>([&](auto ui) { constexpr int u = decltype(ui)::value + 1; tw[s][u - 1] = wK[(q * u * str) & (K - 1)]; });
    });
}

__device__ __forceinline__ cf decide_pt(cf x, const IcParams& ic)
{
    if (ic.decision == 1) {
        const float s = 0.70710678118654752f;
        return mk(x.x > 0.f ? s : -s, x.y > 0.f ? s : -s);
    }
    int idx = 0;
    if (ic.decision == 2) {
        idx = (x.x > 0.f);
    } else {
        float best = INFINITY;
        for (int i = 0; i < ic.npoints; ++i) {
            const cf pt = ic.points[i];
            const float dr = x.x - pt.x, di = x.y - pt.y, d = dr * dr + di * di;
            if (d < best) { best = d; idx = i; }
        }
    }
    return ic.points[idx];
}

// =====================================================================================================================
// receiver: MODE = RX_FD (out = S), RX_DEMOD (out = IDFT_M(S)/M), RX_IC (interference cancellation rounds)
template <int K, int M, int L, int MODE, bool EQ>
__global__ __launch_bounds__(WAVE, 2) void k_fast_receive(DevicePlan p, IcParams ic, const cf* __restrict__ twT, cf* __restrict__ out,
                                                          const cf* __restrict__ in, const cf* __restrict__ f_eq, int64_t nblocks)
{
    using S = FftShape<K>;
    using T = Tile<K, M>;
    constexpr int N = K * M, TPB = S::TPB, G = S::G;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int g = lane / TPB, t = lane - g * TPB;
    const int64_t blk = (int64_t)blockIdx.x * G + g;
    const bool valid = blk < nblocks;
    cf* tile = reinterpret_cast<cf*>(smem) + g * T::STRIDE;
    const cf* x = in + (valid ? blk : 0) * N;

    // ---- phase A: load rows q = t + (K/4) r (coalesced along q), M-point DFT, twiddle W_N^{q m}
    cf v[4][M];
    static_for<0, 4>([&](auto ri) {
        constexpr int r = decltype(ri)::value;
        static_for<0, M>([&](auto pi) { constexpr int pp = decltype(pi)::value; v[r][pp] = x[K * pp + t + (K / 4) * r]; });
    });
    cf tw[S::NP4][3];
    load_pass_twiddles<K>(tw, p.wK, t);
    static_for<0, 4>([&](auto ri) {
        constexpr int r = decltype(ri)::value;
        dft_inplace<M, false>(v[r]);
        static_for<1, M>([&](auto mi) {
            constexpr int m = decltype(mi)::value;
            v[r][m] = cmul(v[r][m], twT[m * K + t + (K / 4) * r]);
        });
    });

    // ---- phase B: K-point FFT over q for every column m; afterwards v[u][m] = X[(t + (K/4) u) M + m]
    subcarrier_fft<K, M, false>(v, tile, t, tw);

    // ---- X -> tile in natural [j][m] order
    wave_sync();
    {
        cf* wbase = tile + t * M;
        static_for<0, 4>([&](auto ui) {
            constexpr int u = decltype(ui)::value;
            static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; lds_store(wbase, (K / 4) * u * M + m, v[u][m]); });
        });
    }
    wave_sync();

    // ---- phase C: X[f] /= f_eq[f], linear over the block (element e = t + TPB i), coalesced global reads
    if constexpr (EQ) {
        const cf* eq = f_eq + (valid ? blk : 0) * N;
        constexpr int PER = N / TPB;            // = 4 M
        cf h[PER];
        static_for<0, PER>([&](auto ii) { constexpr int i = decltype(ii)::value; h[i] = eq[t + TPB * i]; });
        static_for<0, PER>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const cf a = tile[t + TPB * i], b = h[i];
            const float inv = 1.f / (b.x * b.x + b.y * b.y);
            tile[t + TPB * i] = mk((a.x * b.x + a.y * b.y) * inv, (a.y * b.x - a.x * b.y) * inv);
        });
        wave_sync();
    }

    // ---- phase D: rows k = 4 t + u.  S[u][m] = sum_i taps[((i + L/2) % L) M + m] * X[(k + i - L/2) mod K][m]
    constexpr int NR = 4 + L - 1;                // rows 4t - L/2 .. 4t + 3 + (L-1) - L/2
    cf s[4][M];
    {
        cf xr[NR][M];
        static_for<0, NR>([&](auto ji) {
            constexpr int jj = decltype(ji)::value;
            const int row = (4 * t + jj - L / 2 + K) & (K - 1);
            const cf* rb = tile + row * M;
            static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; xr[jj][m] = lds_load(rb, m); });
        });
        static_for<0, 4>([&](auto ui) {
            constexpr int u = decltype(ui)::value;
            static_for<0, M>([&](auto mi) {
                constexpr int m = decltype(mi)::value;
                cf acc = mk(0.f, 0.f);
                static_for<0, L>([&](auto ii) {
                    constexpr int i = decltype(ii)::value;
                    acc = cfma(p.taps[((i + L / 2) % L) * M + m], xr[u + i][m], acc);
                });
                s[u][m] = acc;
            });
        });
    }

    cf d[4][M];
    constexpr float invM = 1.0f / (float)M;
    if constexpr (MODE != RX_FD) {
        static_for<0, 4>([&](auto ui) {
            constexpr int u = decltype(ui)::value;
            static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; d[u][m] = s[u][m]; });
            dft_inplace<M, true>(d[u]);
            static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; d[u][m] = scale(d[u][m], invM); });
        });
    }

    if constexpr (MODE == RX_IC) {
        // multiplicity of this lane's four subcarriers in subcarrier_map (0 = inactive; the map may repeat an entry,
        // which the reference's phase-offset loop counts twice: adv:78-91)
        int wgt[4] = { 0, 0, 0, 0 };
        for (int a = 0; a < ic.n_active; ++a) {
            const int k = ic.smap[a] - 4 * t;
            static_for<0, 4>([&](auto ui) { constexpr int u = decltype(ui)::value; wgt[u] += (k == u); });
        }
        for (int it = 0; it < ic.ic_iter; ++it) {
            const bool pc = (ic.do_phase_compensation > 0) && (it == 0);
            float acc = 0.f;
            wave_sync();                                      // earlier readers of the tile are done
            // hard decisions on active subcarriers, zero elsewhere, straight into the tile     adv:109-123
            {
                cf* wb = tile + 4 * t * M;
                static_for<0, 4>([&](auto ui) {
                    constexpr int u = decltype(ui)::value;
                    static_for<0, M>([&](auto mi) {
                        constexpr int m = decltype(mi)::value;
                        const cf dec = (wgt[u] > 0) ? decide_pt(d[u][m], ic) : mk(0.f, 0.f);
                        if (pc && wgt[u] > 0) acc += (float)wgt[u] * (atan2f(dec.y, dec.x) - atan2f(d[u][m].y, d[u][m].x));
                        lds_store(wb, u * M + m, dec);
                    });
                });
            }
            if (pc) {                                                                        // adv:59-71
                for (int off = 1; off < TPB; off <<= 1) acc += __shfl_xor(acc, off, WAVE);
                const float phi = acc / (float)(ic.n_active * M);
                float sn, cs;
                sincosf(phi, &sn, &cs);
                const cf rot = mk(cs, sn);
                static_for<0, 4>([&](auto ui) {
                    constexpr int u = decltype(ui)::value;
                    static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; s[u][m] = cmul(s[u][m], rot); });
                });
            }
            wave_sync();
            // S'_k = S_k - ic * FFT_M(dec_{k-1} + dec_{k+1}), neighbours wrap mod K              rx:274-299
            static_for<0, 4>([&](auto ui) {
                constexpr int u = decltype(ui)::value;
                const cf* below = tile + ((4 * t + u - 1 + K) & (K - 1)) * M;
                const cf* above = tile + ((4 * t + u + 1) & (K - 1)) * M;
                cf nb[M];
                static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; nb[m] = lds_load(below, m) + lds_load(above, m); });
                dft_inplace<M, false>(nb);
                static_for<0, M>([&](auto mi) {
                    constexpr int m = decltype(mi)::value;
                    nb[m] = s[u][m] - cmul(p.ictaps[m], nb[m]);
                });
                dft_inplace<M, true>(nb);
                static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; d[u][m] = scale(nb[m], invM); });
            });
        }
    }

    // ---- output: rows 4t .. 4t+3 are 4M consecutive samples of the block; stage through LDS, store linearly
    wave_sync();
    {
        cf* wb = tile + 4 * t * M;
        static_for<0, 4>([&](auto ui) {
            constexpr int u = decltype(ui)::value;
            static_for<0, M>([&](auto mi) {
                constexpr int m = decltype(mi)::value;
                lds_store(wb, u * M + m, (MODE == RX_FD) ? s[u][m] : d[u][m]);
            });
        });
    }
    wave_sync();
    if (valid) {
        cf* o = out + blk * N;
        constexpr int PER = N / TPB;
        static_for<0, PER>([&](auto ii) { constexpr int i = decltype(ii)::value; o[t + TPB * i] = tile[t + TPB * i]; });
    }
}

// =====================================================================================================================
// modulator: transposed flow.  D_k = FFT_M(d_k); Y[j][m] = sum_i D[(j - i + L/2) mod K][m] taps[((i+L/2)%L) M + m];
// z[q][m] = sum_j Y[j][m] conj(W_K^{q j}); x[K p + q] = (1/N) sum_m z[q][m] conj(W_N^{q m}) conj(W_M^{p m})
template <int K, int M, int L>
__global__ __launch_bounds__(WAVE, 2) void k_fast_modulate(DevicePlan p, const cf* __restrict__ twT, cf* __restrict__ out,
                                                           const cf* __restrict__ in, int64_t nblocks)
{
    using S = FftShape<K>;
    using T = Tile<K, M>;
    constexpr int N = K * M, TPB = S::TPB, G = S::G;
    constexpr int PART = (M * L / 2 < M) ? (M * L / 2) : M;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x;
    const int g = lane / TPB, t = lane - g * TPB;
    const int64_t blk = (int64_t)blockIdx.x * G + g;
    const bool valid = blk < nblocks;
    cf* tile = reinterpret_cast<cf*>(smem) + g * T::STRIDE;
    const cf* x = in + (valid ? blk : 0) * N;

    // symbols arrive subcarrier-major [k][p]: copy linearly (coalesced) into the tile, then pick rows k = 4t + u
    constexpr int PER = N / TPB;
    {
        cf tmp[PER];
        static_for<0, PER>([&](auto ii) { constexpr int i = decltype(ii)::value; tmp[i] = x[t + TPB * i]; });
        static_for<0, PER>([&](auto ii) { constexpr int i = decltype(ii)::value; tile[t + TPB * i] = tmp[i]; });
    }
    cf tw[S::NP4][3];
    load_pass_twiddles<K>(tw, p.wK, t);
    wave_sync();
    cf v[4][M];
    {
        const cf* rb = tile + 4 * t * M;
        static_for<0, 4>([&](auto ui) {
            constexpr int u = decltype(ui)::value;
            static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; v[u][m] = lds_load(rb, u * M + m); });
            dft_inplace<M, false>(v[u]);                                             // D_k                         :109-110
        });
    }
    wave_sync();
    {
        cf* wb = tile + 4 * t * M;
        static_for<0, 4>([&](auto ui) {
            constexpr int u = decltype(ui)::value;
            static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; lds_store(wb, u * M + m, v[u][m]); });
        });
    }
    wave_sync();
    // gather form of filter + overlap-add, directly into the row assignment of the subcarrier FFT: j = t + (K/4) r  :116-132
    static_for<0, 4>([&](auto ri) {
        constexpr int r = decltype(ri)::value;
        static_for<0, M>([&](auto mi) { constexpr int m = decltype(mi)::value; v[r][m] = mk(0.f, 0.f); });
        static_for<0, L>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            const int row = (t + (K / 4) * r - i + L / 2 + K) & (K - 1);
            const cf* rb = tile + row * M;
            static_for<0, PART>([&](auto mi) {
                constexpr int m = decltype(mi)::value;
                v[r][m] = cfma(lds_load(rb, m), p.taps[((i + L / 2) % L) * M + m], v[r][m]);
            });
        });
    });
    subcarrier_fft<K, M, true>(v, tile, t, tw);                                      // z[q = t + (K/4) u][m]
    constexpr float invN = 1.0f / (float)N;
    cf* o = out + (valid ? blk : 0) * N;
    static_for<0, 4>([&](auto ui) {
        constexpr int u = decltype(ui)::value;
        static_for<1, M>([&](auto mi) {
            constexpr int m = decltype(mi)::value;
            v[u][m] = cmulc(v[u][m], twT[m * K + t + (K / 4) * u]);
        });
        dft_inplace<M, true>(v[u]);                                                  // over m -> timeslot p        :137-140
        if (valid) {
            static_for<0, M>([&](auto pi) {
                constexpr int pp = decltype(pi)::value;
                o[K * pp + t + (K / 4) * u] = scale(v[u][pp], invN);
            });
        }
    });
}

template <int K, int M> constexpr size_t fast_lds_bytes() { return (size_t)FftShape<K>::G * Tile<K, M>::STRIDE * sizeof(cf); }

template <int K, int M, int L>
hipError_t launch_rx(const DevicePlan& p, const IcParams& ic, const cf* twT, int mode, cf* out, const cf* in, const cf* f_eq,
                     int64_t nblocks, hipStream_t st)
{
    constexpr int G = FftShape<K>::G;
    const dim3 grid((unsigned)((nblocks + G - 1) / G)), block(WAVE);
    constexpr size_t lds = fast_lds_bytes<K, M>();
#define GFDM_RX(MODE_, EQ_) hipLaunchKernelGGL((k_fast_receive<K, M, L, MODE_, EQ_>), grid, block, lds, st, p, ic, twT, out, in, f_eq, nblocks)
    if (mode == RX_FD) { if (f_eq) GFDM_RX(RX_FD, true); else GFDM_RX(RX_FD, false); }
    else if (mode == RX_DEMOD || ic.ic_iter <= 0) { if (f_eq) GFDM_RX(RX_DEMOD, true); else GFDM_RX(RX_DEMOD, false); }
    else { if (f_eq) GFDM_RX(RX_IC, true); else GFDM_RX(RX_IC, false); }
#undef GFDM_RX
    return hipGetLastError();
}

template <int K, int M, int L>
hipError_t launch_mod(const DevicePlan& p, const cf* twT, cf* out, const cf* in, int64_t nblocks, hipStream_t st)
{
    constexpr int G = FftShape<K>::G;
    const dim3 grid((unsigned)((nblocks + G - 1) / G)), block(WAVE);
    constexpr size_t lds = fast_lds_bytes<K, M>();
    hipLaunchKernelGGL((k_fast_modulate<K, M, L>), grid, block, lds, st, p, twT, out, in, nblocks);
    return hipGetLastError();
}

}  // namespace

// Shapes served by this family (others fall back to the generic LDS family).
// (kept to the benchmark shape: the row-lane family is the default everywhere, this one is the A/B alternative)
#define GFDM_FAST_SHAPES(X) \
    X(64, 9, 2)

bool fast_supports(int M, int K, int L)
{
#define X(K_, M_, L_) if (K == K_ && M == M_ && L == L_) return true;
    GFDM_FAST_SHAPES(X)
#undef X
    return false;
}

hipError_t launch_fast_modulate(const DevicePlan& p, const cf* twT, cf* out, const cf* in, int64_t nblocks, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
#define X(K_, M_, L_) if (p.K == K_ && p.M == M_ && p.L == L_) return launch_mod<K_, M_, L_>(p, twT, out, in, nblocks, s);
    GFDM_FAST_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

hipError_t launch_fast_receive(const DevicePlan& p, const IcParams& ic, const cf* twT, int mode, cf* out, const cf* in, const cf* f_eq,
                               int64_t nblocks, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
#define X(K_, M_, L_) if (p.K == K_ && p.M == M_ && p.L == L_) return launch_rx<K_, M_, L_>(p, ic, twT, mode, out, in, f_eq, nblocks, s);
    GFDM_FAST_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace gfdm
