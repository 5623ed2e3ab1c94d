// Probe: issue rate of packed f32 VALU ops (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) against scalar v_fma_f32 on gfx950, and the
// op_sel / neg modifier semantics needed for complex arithmetic on (re, im) register pairs.
//   hipcc -O3 --offload-arch=gfx950 pk_rate.hip -o pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

template <int KIND>
__global__ void rate(float* out, int iters)
{
    v2f a0 = { 1.f + threadIdx.x, 2.f }, a1 = { 3.f, 4.f }, a2 = { 5.f, 6.f }, a3 = { 7.f, 8.f };
    v2f a4 = { 1.5f, 2.5f }, a5 = { 3.5f, 4.5f }, a6 = { 5.5f, 6.5f }, a7 = { 7.5f, 8.5f };
    const v2f w = { 0.999f, 1.0e-3f };
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {         // 16 scalar FMAs per group (8 pairs x 2 components)
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(w.x), "v"(w.y));)
        } else if (KIND == 1) {  // 8 packed FMAs per group = the same 16 flops-pairs in half the instructions
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                              "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));)
        } else if (KIND == 2) {  // packed add with op_sel swap + neg_hi (a + (-j) b)
            REP8(asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                              "v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                              "v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                              "v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));)
        } else if (KIND == 3) {  // 8 scalar adds per group (instruction-count match of KIND 1/2)
            REP8(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                              "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(w.y));)
        } else if (KIND == 4) {  // DPP row rotate movs
            REP8(asm volatile("v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %2 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %4 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %6 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 wave_ror:1 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));)
        } else if (KIND == 5) {  // permlane32 swaps
            REP8(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                              "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                              : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));)
        }
    }
    v2f s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s.x == 12345.678f) out[threadIdx.x] = s.x + s.y;
}

// semantics check: complex helpers on (re, im) pairs
__device__ __forceinline__ v2f pk_add_mj(v2f a, v2f b)    // a + (-j) b = (a.re + b.im, a.im - b.re)
{
    v2f r;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_add_pj(v2f a, v2f b)    // a + (+j) b = (a.re - b.im, a.im + b.re)
{
    v2f r;
    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ v2f pk_cmul(v2f a, v2f w)      // a * w
{
    v2f t, r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                       // (a.re w.re, a.im w.re)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));   // (-a.im w.im + t.lo, a.re w.im + t.hi)
    return r;
}
__device__ __forceinline__ v2f pk_cmul_conj(v2f a, v2f w) // a * conj(w)
{
    v2f t, r;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));   // (a.im w.im + t.lo, -a.re w.im + t.hi)
    return r;
}
__global__ void sem(float* o)
{
    v2f a = { 1.25f + threadIdx.x, -2.5f }, b = { 0.75f, 3.0f - threadIdx.x };
    v2f r0 = pk_add_mj(a, b), r1 = pk_add_pj(a, b), r2 = pk_cmul(a, b), r3 = pk_cmul_conj(a, b);
    float* p = o + threadIdx.x * 8;
    p[0] = r0.x; p[1] = r0.y; p[2] = r1.x; p[3] = r1.y; p[4] = r2.x; p[5] = r2.y; p[6] = r3.x; p[7] = r3.y;
}

template <int KIND> float time_kind(float* d, int iters, int grid)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate<KIND><<<grid, 256>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate<KIND><<<grid, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* d; hipMalloc(&d, 1 << 20);
    const int iters = 200;
    const char* nm[6] = { "v_fma_f32 x128 (scalar, 128 flop-lanes)", "v_pk_fma_f32 x64 (same flops)", "v_pk_add_f32 x64 op_sel+neg",
                          "v_add_f32 x64", "v_mov_b32_dpp wave_ror x64", "v_permlane32/16_swap x64" };
    for (int grid : { 1024, 4096 }) {      // 4 and 16 waves per CU
        float t[6] = { time_kind<0>(d, iters, grid), time_kind<1>(d, iters, grid), time_kind<2>(d, iters, grid), time_kind<3>(d, iters, grid),
                       time_kind<4>(d, iters, grid), time_kind<5>(d, iters, grid) };
        const int ninstr[6] = { 64, 64, 64, 64, 64, 64 };
        for (int k = 0; k < 6; ++k) {
            // per SIMD: waves = grid * 4 / 1024; instr per wave = iters * ninstr
            const double waves_per_simd = grid * 4.0 / 1024.0;
            const double cyc = t[k] * 1e-3 * 2.4e9 / (waves_per_simd * iters * ninstr[k]);
            printf("grid %5d  %-42s %8.3f ms  ~%.2f cycles per wave-instruction at 2.4 GHz\n", grid, nm[k], t[k], cyc);
        }
    }
    sem<<<1, 64>>>(d);
    float h[512]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t) {
        const float ar = 1.25f + t, ai = -2.5f, br = 0.75f, bi = 3.0f - t;
        const float exp[8] = { ar + bi, ai - br, ar - bi, ai + br, ar * br - ai * bi, ar * bi + ai * br, ar * br + ai * bi, -ar * bi + ai * br };
        for (int i = 0; i < 8; ++i) if (fabsf(h[t * 8 + i] - exp[i]) > 1e-4f * (1 + fabsf(exp[i]))) { if (bad < 8) printf("MISMATCH lane %d slot %d: %g vs %g\n", t, i, h[t * 8 + i], exp[i]); ++bad; }
    }
    printf("semantics: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return 0;
}
