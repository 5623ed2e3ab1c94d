// EXPERIMENT (not in the product): the 31-point transform of a wavefront on the matrix cores (WaveDft31 below) against the vector-ALU codelet
// PrimeDft<31> of gfdm_dft.h.  Result on MI355X: bit-identical outputs, but 4248 cycles per transform and wavefront against 2440 -- the f32 MFMA
// rate equals the packed-f32 vector rate, the dense 16 x 16 form has the same multiply-adds as the symmetric codelet (1024 vs 900 per row), and
// the 128 lane swaps + the dependent MFMA chains come on top.  (In the generic family the matrix cores win because they relieve LDS, not flops.)
// Round 4 added the split-precision form the review asked for (WaveDft31Bf16): the same paired 16 x 16 products on v_mfma_f32_16x16x32_bf16 with matrix AND data as
// three-term bf16 splits (24 bits each; six of the nine cross products kept, two per instruction: 3 MFMAs of depth 32 per product instead of 4 of depth 4).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I../../gr-gfdm_amd/csrc -o wave_dft31 wave_dft31.hip && ./wave_dft31
#include "gfdm_dft.h"
#include <cstdio>
#include <vector>
#include <cmath>
namespace gfdm { namespace dft {
// ---------------------------------------------------------------- the 31-point transform of a whole wavefront on the matrix cores
// M = 31 (BASELINE configs[4]) is the one timeslot count whose direct codelet makes a kernel vector-ALU bound: 450 packed multiply-adds per
// transform and lane, at half rate on gfx950.  In the paired form above the transform of a row is  [Q1 | Q2] = C [a.x | a.y],
// [Q3 | Q4] = S [b.x | b.y]  with the CONSTANT 16 x 16 matrices C[k][n] = cos(2 pi k n / 31), S[k][n] = sin(2 pi k n / 31) (k: the 16 output
// pairs k / 31 - k; n = 0: the sample x_0, C = 1, S = 0; n = 1..15: the sample pairs) -- exactly one 16 x 16 tile of v_mfma_f32_16x16x4_f32
// (f32 operands, f32 sums: the same numbers as the fmaf chain), four k-steps.  A wavefront holds 64 rows, one per lane; the B operand of the
// 16 rows of lane row g wants (lane row q, lane n) <- sample pair 4 ks + q of row n: a 4 x 4 transpose between four registers and the four
// lane rows (v_permlane32_swap + v_permlane16_swap), and the same transpose brings the products back to "one row per lane".
// Per transform and lane: 64 MFMAs (2048 cycles of the matrix pipe, beside the vector ALU) + 128 lane swaps + ~150 adds instead of ~3800
// issue cycles.  Every lane of the wavefront must be active and hold a row (garbage rows are harmless: columns do not mix).
// register i of lane row rr  <-  register rr of lane row i.  v_permlane32_swap(a, b): rows 2,3 of a <-> rows 0,1 of b;
// v_permlane16_swap(a, b): rows 1,3 of a <-> rows 0,2 of b (checked on hardware, scratch/probe/permlane.hip).
__device__ __forceinline__ void lane_row_transpose4(float& f0, float& f1, float& f2, float& f3)
{
    unsigned a0 = __builtin_bit_cast(unsigned, f0), a1 = __builtin_bit_cast(unsigned, f1);
    unsigned a2 = __builtin_bit_cast(unsigned, f2), a3 = __builtin_bit_cast(unsigned, f3);
    const auto r = __builtin_amdgcn_permlane32_swap(a0, a2, false, false);
    const auto s = __builtin_amdgcn_permlane32_swap(a1, a3, false, false);
    const auto t = __builtin_amdgcn_permlane16_swap(r[0], s[0], false, false);
    const auto u = __builtin_amdgcn_permlane16_swap(r[1], s[1], false, false);
    f0 = __builtin_bit_cast(float, (unsigned)t[0]);
    f1 = __builtin_bit_cast(float, (unsigned)t[1]);
    f2 = __builtin_bit_cast(float, (unsigned)u[0]);
    f3 = __builtin_bit_cast(float, (unsigned)u[1]);
}

// A operands: lane l holds A[row l & 15][k = l >> 4] of k-step ks, i.e. C / S at (output pair l & 15, sample pair 4 ks + (l >> 4))
struct Dft31Operands {
    float c[4][64], s[4][64];
    constexpr Dft31Operands() : c{}, s{}
    {
        for (int ks = 0; ks < 4; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int k = l & 15, n = 4 * ks + (l >> 4);
                c[ks][l] = (float)cos2pi((long)k * n, 31);
                s[ks][l] = (n == 0) ? 0.f : (float)sin2pi((long)k * n, 31);
            }
    }
};
__device__ const Dft31Operands kDft31Operands = Dft31Operands();

typedef float mfma_f4 __attribute__((ext_vector_type(4)));

struct WaveDft31 {
    float c[4], s[4];
    __device__ __forceinline__ void preload(int lane)              // (early: the table sits in global memory)
    {
        static_for<0, 4>([&](auto ki) { constexpr int ks = decltype(ki)::value; c[ks] = kDft31Operands.c[ks][lane]; s[ks] = kDft31Operands.s[ks][lane]; });
    }
    // products of one constant matrix (operands w) with the planes px, py (16 sample pairs each) of all four lane rows
    __device__ __forceinline__ void products(const float (&w)[4], const float (&px)[16], const float (&py)[16], mfma_f4 (&qx)[4], mfma_f4 (&qy)[4]) const
    {
        static_for<0, 4>([&](auto gi) { constexpr int g = decltype(gi)::value; qx[g] = mfma_f4{ 0.f, 0.f, 0.f, 0.f }; qy[g] = qx[g]; });
        static_for<0, 4>([&](auto ki) {
            constexpr int ks = decltype(ki)::value;
            float t0 = px[4 * ks], t1 = px[4 * ks + 1], t2 = px[4 * ks + 2], t3 = px[4 * ks + 3];
            lane_row_transpose4(t0, t1, t2, t3);                   // t_g: lane row q holds sample pair 4 ks + q of the rows of lane row g
            qx[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks], t0, qx[0], 0, 0, 0);
            qx[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks], t1, qx[1], 0, 0, 0);
            qx[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks], t2, qx[2], 0, 0, 0);
            qx[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks], t3, qx[3], 0, 0, 0);
            float u0 = py[4 * ks], u1 = py[4 * ks + 1], u2 = py[4 * ks + 2], u3 = py[4 * ks + 3];
            lane_row_transpose4(u0, u1, u2, u3);
            qy[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks], u0, qy[0], 0, 0, 0);
            qy[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks], u1, qy[1], 0, 0, 0);
            qy[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks], u2, qy[2], 0, 0, 0);
            qy[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ks], u3, qy[3], 0, 0, 0);
        });
        // q_g[i] of lane row q = output pair 4 q + i of the rows of lane row g  ->  q_q[i] of lane row g: one row per lane again
        static_for<0, 4>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            float a0 = qx[0][i], a1 = qx[1][i], a2 = qx[2][i], a3 = qx[3][i];
            lane_row_transpose4(a0, a1, a2, a3);
            qx[0][i] = a0; qx[1][i] = a1; qx[2][i] = a2; qx[3][i] = a3;
            float b0 = qy[0][i], b1 = qy[1][i], b2 = qy[2][i], b3 = qy[3][i];
            lane_row_transpose4(b0, b1, b2, b3);
            qy[0][i] = b0; qy[1][i] = b1; qy[2][i] = b2; qy[3][i] = b3;
        });
    }
    template <bool INV>
    __device__ __forceinline__ void run(cf (&x)[31]) const
    {
        float ax[16], ay[16], bx[16], by[16];
        ax[0] = x[0].x; ay[0] = x[0].y; bx[0] = 0.f; by[0] = 0.f;
        static_for<1, 16>([&](auto ni) {
            constexpr int n = decltype(ni)::value;
            ax[n] = x[n].x + x[31 - n].x; ay[n] = x[n].y + x[31 - n].y;
            bx[n] = x[n].x - x[31 - n].x; by[n] = x[n].y - x[31 - n].y;
        });
        mfma_f4 rex[4], rey[4], imx[4], imy[4];
        products(c, ax, ay, rex, rey);
        products(s, bx, by, imx, imy);
        x[0] = mk(rex[0][0], rey[0][0]);
        static_for<1, 16>([&](auto ki) {
            constexpr int k = decltype(ki)::value;
            const float rx = rex[k / 4][k % 4], ry = rey[k / 4][k % 4], ix = imx[k / 4][k % 4], iy = imy[k / 4][k % 4];
            // forward: y[k] = re - j im, y[31 - k] = re + j im ; inverse swaps them
            const cf lo = mk(rx + iy, ry - ix), hi = mk(rx - iy, ry + ix);
            x[k] = INV ? hi : lo;
            x[31 - k] = INV ? lo : hi;
        });
    }
};


// ---------------------------------------------------------------- the same transform on bf16 x 3 (round 4)
// Contraction entry 8 cr + j of lane row cr stands for sample pair n = 4 cr + (j & 3) (as in IcMfma: the B operand of a lane is built from the four values the lane itself
// holds after the lane-row transpose).  x = hi + mid + lo (bf16 each), w likewise;  w x ~ hi hi + hi mid + mid hi + mid mid + hi lo + lo hi:
//   A1 = [w_hi | w_hi] . B1 = [x_hi | x_mid],   A2 = [w_mid | w_mid] . B1,   A3 = [w_hi | w_lo] . B2 = [x_lo | x_hi]
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
struct Bf16Split { unsigned short hi, mid, lo; };
constexpr unsigned f32_bits(float f) { return __builtin_bit_cast(unsigned, f); }
constexpr float bits_f32(unsigned u) { return __builtin_bit_cast(float, u); }
constexpr unsigned short bf16_rne(float f) { const unsigned u = f32_bits(f); return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }
constexpr Bf16Split split3(double v)
{
    const float f = (float)v;
    const unsigned short h = bf16_rne(f);
    const float r1 = f - bits_f32((unsigned)h << 16);
    const unsigned short m = bf16_rne(r1);
    const float r2 = r1 - bits_f32((unsigned)m << 16);
    return Bf16Split{ h, m, bf16_rne(r2) };
}
struct Dft31Bf16Operands {
    unsigned a[2][3][64][4];       // [C | S][A1, A2, A3][lane][4 dwords = 8 bf16]
    constexpr Dft31Bf16Operands() : a{}
    {
        for (int mat = 0; mat < 2; ++mat)
            for (int l = 0; l < 64; ++l) {
                const int k = l & 15, cr = l >> 4;
                unsigned short e[3][8] = {};
                for (int j = 0; j < 8; ++j) {
                    const int n = 4 * cr + (j & 3);
                    const double w = mat == 0 ? cos2pi((long)k * n, 31) : (n == 0 ? 0.0 : sin2pi((long)k * n, 31));
                    const Bf16Split sp = split3(w);
                    e[0][j] = sp.hi;
                    e[1][j] = sp.mid;
                    e[2][j] = (j < 4) ? sp.hi : sp.lo;
                }
                for (int op = 0; op < 3; ++op)
                    for (int d = 0; d < 4; ++d) a[mat][op][l][d] = (unsigned)e[op][2 * d] | ((unsigned)e[op][2 * d + 1] << 16);
            }
    }
};
__device__ const Dft31Bf16Operands kDft31Bf16 = Dft31Bf16Operands();

struct WaveDft31Bf16 {
    uint4 a[2][3];
    __device__ __forceinline__ void preload(int lane)
    {
        static_for<0, 2>([&](auto mi) { constexpr int m = decltype(mi)::value;
            static_for<0, 3>([&](auto oi) { constexpr int o = decltype(oi)::value;
                a[m][o] = make_uint4(kDft31Bf16.a[m][o][lane][0], kDft31Bf16.a[m][o][lane][1], kDft31Bf16.a[m][o][lane][2], kDft31Bf16.a[m][o][lane][3]); }); });
    }
    // two f32 values -> their three bf16 terms, packed pairwise (low half = the first value)
    static __device__ __forceinline__ void split2(float x0, float x1, unsigned& hi, unsigned& mid, unsigned& lo)
    {
        const bf16x2 h = __builtin_convertvector(float2_t{ x0, x1 }, bf16x2);
        hi = __builtin_bit_cast(unsigned, h);
        const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xFFFF0000u);
        const bf16x2 m = __builtin_convertvector(float2_t{ r0, r1 }, bf16x2);
        mid = __builtin_bit_cast(unsigned, m);
        const float s0 = r0 - __builtin_bit_cast(float, mid << 16), s1 = r1 - __builtin_bit_cast(float, mid & 0xFFFF0000u);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(float2_t{ s0, s1 }, bf16x2));
    }
    typedef float float2_t __attribute__((ext_vector_type(2)));
    template <int MAT>
    __device__ __forceinline__ void products(const float (&p)[16], mfma_f4 (&q)[4]) const
    {
        float t[4][4];                                        // t[g][i]: sample pair 4 cr + i of the rows of lane row g, on lane row cr
        static_for<0, 4>([&](auto ii) {
            constexpr int i = decltype(ii)::value;
            float r0 = p[i], r1 = p[4 + i], r2 = p[8 + i], r3 = p[12 + i];
            lane_row_transpose4(r0, r1, r2, r3);
            t[0][i] = r0; t[1][i] = r1; t[2][i] = r2; t[3][i] = r3;
        });
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, a[MAT][0]), a2 = __builtin_bit_cast(bf16x8, a[MAT][1]), a3 = __builtin_bit_cast(bf16x8, a[MAT][2]);
        static_for<0, 4>([&](auto gi) {
            constexpr int g = decltype(gi)::value;
            unsigned h0, m0, l0, h1, m1, l1;
            split2(t[g][0], t[g][1], h0, m0, l0);
            split2(t[g][2], t[g][3], h1, m1, l1);
            const bf16x8 b1 = __builtin_bit_cast(bf16x8, make_uint4(h0, h1, m0, m1)), b2 = __builtin_bit_cast(bf16x8, make_uint4(l0, l1, h0, h1));
            mfma_f4 acc = mfma_f4{ 0.f, 0.f, 0.f, 0.f };
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, acc, 0, 0, 0);
            q[g] = acc;
        });
        static_for<0, 4>([&](auto ii) {                      // back to one row per lane: q[j][i] of lane row g = output pair 4 j + i of the lane's own row
            constexpr int i = decltype(ii)::value;
            float r0 = q[0][i], r1 = q[1][i], r2 = q[2][i], r3 = q[3][i];
            lane_row_transpose4(r0, r1, r2, r3);
            q[0][i] = r0; q[1][i] = r1; q[2][i] = r2; q[3][i] = r3;
        });
    }
    template <bool INV>
    __device__ __forceinline__ void run(cf (&x)[31]) const
    {
        float ax[16], ay[16], bx[16], by[16];
        ax[0] = x[0].x; ay[0] = x[0].y; bx[0] = 0.f; by[0] = 0.f;
        static_for<1, 16>([&](auto ni) {
            constexpr int n = decltype(ni)::value;
            ax[n] = x[n].x + x[31 - n].x; ay[n] = x[n].y + x[31 - n].y;
            bx[n] = x[n].x - x[31 - n].x; by[n] = x[n].y - x[31 - n].y;
        });
        mfma_f4 rex[4], rey[4], imx[4], imy[4];
        products<0>(ax, rex); products<0>(ay, rey); products<1>(bx, imx); products<1>(by, imy);
        x[0] = mk(rex[0][0], rey[0][0]);
        static_for<1, 16>([&](auto ki) {
            constexpr int k = decltype(ki)::value;
            const float rx = rex[k / 4][k % 4], ry = rey[k / 4][k % 4], ix = imx[k / 4][k % 4], iy = imy[k / 4][k % 4];
            const cf lo = mk(rx + iy, ry - ix), hi = mk(rx - iy, ry + ix);
            x[k] = INV ? hi : lo;
            x[31 - k] = INV ? lo : hi;
        });
    }
};

} }
using namespace gfdm::dft;

template <bool INV>
__global__ __launch_bounds__(256, 2) void k(const cf* in, cf* out_valu, cf* out_mx, long long* cyc, int reps)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    WaveDft31 w;
    w.preload(threadIdx.x & 63);
    cf x[31], y[31];
    static_for<0, 31>([&](auto i) { constexpr int j = decltype(i)::value; x[j] = in[t * 31 + j]; y[j] = x[j]; });
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) PrimeDft<31, INV>::run(x);
    long long t1 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) w.run<INV>(y);
    long long t2 = __builtin_readcyclecounter();
    static_for<0, 31>([&](auto i) { constexpr int j = decltype(i)::value; out_valu[t * 31 + j] = x[j]; out_mx[t * 31 + j] = y[j]; });
    if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}

template <bool INV>
__global__ __launch_bounds__(256, 2) void kb(const cf* in, cf* out_bf, long long* cyc, int reps)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    WaveDft31Bf16 w;
    w.preload(threadIdx.x & 63);
    cf y[31];
    static_for<0, 31>([&](auto i) { constexpr int j = decltype(i)::value; y[j] = in[t * 31 + j]; });
    long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) w.run<INV>(y);
    long long t1 = __builtin_readcyclecounter();
    static_for<0, 31>([&](auto i) { constexpr int j = decltype(i)::value; out_bf[t * 31 + j] = y[j]; });
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    const int blocks = 512, T = blocks * 256, n = T * 31;
    std::vector<cf> in(n), a(n), b(n);
    unsigned s = 12345;
    for (auto& v : in) { s = s * 1664525u + 1013904223u; v.x = (float)(s >> 8) / 8388608.f - 1.f; s = s * 1664525u + 1013904223u; v.y = (float)(s >> 8) / 8388608.f - 1.f; }
    cf *din, *da, *db; long long* dc; long long hc[2];
    hipMalloc(&din, n * sizeof(cf)); hipMalloc(&da, n * sizeof(cf)); hipMalloc(&db, n * sizeof(cf)); hipMalloc(&dc, 16);
    hipMemcpy(din, in.data(), n * sizeof(cf), hipMemcpyHostToDevice);
    int bad = 0;
    for (int inv = 0; inv < 2; ++inv) {
        if (inv) hipLaunchKernelGGL(k<true>, dim3(blocks), dim3(256), 0, 0, din, da, db, dc, 1);
        else hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, din, da, db, dc, 1);
        hipMemcpy(a.data(), da, n * sizeof(cf), hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), db, n * sizeof(cf), hipMemcpyDeviceToHost);
        double num = 0, den = 0, ref_err = 0;
        for (int i = 0; i < n; ++i) { num += (a[i].x - b[i].x) * (double)(a[i].x - b[i].x) + (a[i].y - b[i].y) * (double)(a[i].y - b[i].y); den += a[i].x * (double)a[i].x + a[i].y * (double)a[i].y; }
        // float64 reference of row 0
        for (int kk = 0; kk < 31; ++kk) {
            double re = 0, im = 0;
            for (int p = 0; p < 31; ++p) { const double ang = (inv ? 2.0 : -2.0) * M_PI * (double)((kk * p) % 31) / 31.0; re += in[p].x * cos(ang) - in[p].y * sin(ang); im += in[p].x * sin(ang) + in[p].y * cos(ang); }
            ref_err = fmax(ref_err, hypot(b[kk].x - re, b[kk].y - im));
        }
        printf("%s: matrix cores vs vector ALU, relative difference %.3e; row 0 vs float64: max abs error %.3e\n", inv ? "inverse" : "forward", sqrt(num / den), ref_err);
        bad += !(sqrt(num / den) < 2e-6) || !(ref_err < 1e-4);
    }
    {   // the bf16 x 3 form: accuracy against the vector-ALU codelet over all rows and against float64 on row 0, then cycles
        hipLaunchKernelGGL(k<false>, dim3(blocks), dim3(256), 0, 0, din, da, db, dc, 1);
        hipLaunchKernelGGL(kb<false>, dim3(blocks), dim3(256), 0, 0, din, db, dc, 1);
        hipMemcpy(a.data(), da, n * sizeof(cf), hipMemcpyDeviceToHost);
        hipMemcpy(b.data(), db, n * sizeof(cf), hipMemcpyDeviceToHost);
        double num = 0, den = 0, worst = 0;
        for (int r = 0; r < T; ++r) {
            double nr = 0, dr = 0;
            for (int j = 0; j < 31; ++j) { const int i = r * 31 + j; nr += (a[i].x - b[i].x) * (double)(a[i].x - b[i].x) + (a[i].y - b[i].y) * (double)(a[i].y - b[i].y); dr += a[i].x * (double)a[i].x + a[i].y * (double)a[i].y; }
            num += nr; den += dr; worst = fmax(worst, sqrt(nr / dr));
        }
        printf("bf16 x 3 matrix cores vs vector ALU: relative difference %.3e over all rows, worst row %.3e\n", sqrt(num / den), worst);
        bad += !(worst < 1e-5);
        for (int blk = 1; blk <= 512; blk *= 512) {
            hipLaunchKernelGGL(kb<false>, dim3(blk), dim3(256), 0, 0, din, db, dc, 200);
            hipLaunchKernelGGL(kb<false>, dim3(blk), dim3(256), 0, 0, din, db, dc, 200);
            hipMemcpy(hc, dc, 8, hipMemcpyDeviceToHost);
            printf("%d workgroup(s) of 4 wavefronts: bf16 x 3 matrix cores %.0f cycles per transform\n", blk, hc[0] / 200.0);
        }
    }
    for (int blk = 1; blk <= 512; blk *= 512) {
        hipLaunchKernelGGL(k<false>, dim3(blk), dim3(256), 0, 0, din, da, db, dc, 200);
        hipLaunchKernelGGL(k<false>, dim3(blk), dim3(256), 0, 0, din, da, db, dc, 200);
        hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
        printf("%d workgroup(s) of 4 wavefronts: vector-ALU codelet %.0f cycles per transform, matrix cores %.0f\n", blk, hc[0] / 200.0, hc[1] / 200.0);
    }
    return bad;
}
