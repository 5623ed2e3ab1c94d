// Lane maps of v_mfma_f32_16x16x32_f16 checked with exact integer data (the guide documents the bf16 form; the IC rounds of
// gfdm_rowlane_impl.h rely on the f16 form using the same maps):
//   A: lane l holds A[row l & 15][k = 8 (l >> 4) + j], B: lane l holds B[k = 8 (l >> 4) + j][col l & 15], j < 8
//   C / D: lane l holds D[row 4 (l >> 4) + i][col l & 15], i < 4
// hipcc --offload-arch=gfx950 -O2 -o mfma_f16_layout mfma_f16_layout.hip && ./mfma_f16_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k(const float* A, const float* B, const float* C, float* D)
{
    const int l = threadIdx.x;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + j];
        b[j] = (_Float16)B[(8 * (l >> 4) + j) * 16 + (l & 15)];
    }
    f4 c;
    for (int i = 0; i < 4; ++i) c[i] = C[(4 * (l >> 4) + i) * 16 + (l & 15)];
    const f4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * (l >> 4) + i) * 16 + (l & 15)] = d[i];
}

int main()
{
    std::vector<float> A(16 * 32), B(32 * 16), C(16 * 16), D(16 * 16), R(16 * 16);
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 32; ++kk) A[i * 32 + kk] = (float)((i * 7 + kk * 3) % 11 - 5);
    for (int kk = 0; kk < 32; ++kk) for (int j = 0; j < 16; ++j) B[kk * 16 + j] = (float)((kk * 5 + j * 2 + (kk * j) % 3) % 9 - 4);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) C[i * 16 + j] = (float)(i * 100 + j);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = C[i * 16 + j];
        for (int kk = 0; kk < 32; ++kk) s += A[i * 32 + kk] * B[kk * 16 + j];
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
    printf("mfma_f32_16x16x32_f16 lane maps: %s (%d of 256 elements differ)\n", bad ? "MISMATCH" : "as assumed", bad);
    return bad != 0;
}
