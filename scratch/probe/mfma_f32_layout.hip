// Lane maps of v_mfma_f32_16x16x4_f32 checked with exact integer data (the matrix-core timeslot transforms of gfdm_generic.hip rely on them):
//   A: lane l holds A[row l & 15][k = l >> 4], B: lane l holds B[k = l >> 4][col l & 15]
//   C / D: lane l holds D[row 4 (l >> 4) + i][col l & 15], i < 4
// Also times a chain of dependent / independent MFMAs (cycles per instruction).
// hipcc --offload-arch=gfx950 -O2 -o mfma_f32_layout mfma_f32_layout.hip && ./mfma_f32_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* A, const float* B, const float* C, float* D)
{
    const int l = threadIdx.x;
    const float a = A[(l & 15) * 4 + (l >> 4)];
    const float b = B[(l >> 4) * 16 + (l & 15)];
    f4 c;
    for (int i = 0; i < 4; ++i) c[i] = C[(4 * (l >> 4) + i) * 16 + (l & 15)];
    const f4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * (l >> 4) + i) * 16 + (l & 15)] = d[i];
}

__global__ void timing(float* out, long long* cyc, int n)
{
    const int l = threadIdx.x;
    float a = 1.f + l * 1e-3f, b = 1.f - l * 1e-3f;
    f4 q1 = {0, 0, 0, 0}, q2 = q1, q3 = q1, q4 = q1;
    const long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        q1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q1, 0, 0, 0);
        q2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, q2, 0, 0, 0);
        q3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, q3, 0, 0, 0);
        q4 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, q4, 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    out[l] = q1[0] + q2[1] + q3[2] + q4[3];
    if (l == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    std::vector<float> A(16 * 4), B(4 * 16), C(16 * 16), D(16 * 16), R(16 * 16);
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) A[i * 4 + kk] = (float)((i * 7 + kk * 3) % 11 - 5);
    for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) B[kk * 16 + j] = (float)((kk * 5 + j * 2 + (kk * j) % 3) % 9 - 4);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) C[i * 16 + j] = (float)(i * 100 + j);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = C[i * 16 + j];
        for (int kk = 0; kk < 4; ++kk) s += A[i * 4 + kk] * B[kk * 16 + j];
        R[i * 16 + j] = s;
    }
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += (D[i] != R[i]);
    printf("mfma_f32_16x16x4_f32 lane maps: %s (%d of 256 elements differ)\n", bad ? "MISMATCH" : "as assumed", bad);
    long long* dc; float* dout; long long hc[4];
    hipMalloc(&dc, 64); hipMalloc(&dout, 1024 * 4);
    for (int waves = 1; waves <= 4; waves *= 2) {
        hipLaunchKernelGGL(timing, dim3(1), dim3(64 * waves), 0, 0, dout, dc, 1000);
        hipLaunchKernelGGL(timing, dim3(1), dim3(64 * waves), 0, 0, dout, dc, 1000);
        hipMemcpy(hc, dc, 8, hipMemcpyDeviceToHost);
        printf("  %d wave(s) on one CU: %.1f shader-clock cycles per mfma_f32_16x16x4_f32 (4 independent chains)\n", waves, hc[0] / 4000.0);
    }
    return bad != 0;
}
