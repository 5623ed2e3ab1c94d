// Micro-benchmark: achievable HBM copy bandwidth for the access shapes the GFDM kernels use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each 64-thread WG copies one 4608-byte block (576 float2): 9 x 8-byte accesses per lane
__global__ __launch_bounds__(64) void copy_f2_block(float2* __restrict__ o, const float2* __restrict__ i, long nb)
{
    long b = blockIdx.x; const float2* x = i + b * 576; float2* y = o + b * 576;
    float2 v[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) v[p] = x[64 * p + threadIdx.x];
#pragma unroll
    for (int p = 0; p < 9; ++p) y[64 * p + threadIdx.x] = v[p];
}
// same bytes per WG with 16-byte accesses: 288 float4 = 4.5 per lane
__global__ __launch_bounds__(64) void copy_f4_block(float4* __restrict__ o, const float4* __restrict__ i, long nb)
{
    long b = blockIdx.x; const float4* x = i + b * 288; float4* y = o + b * 288;
    float4 v[5];
#pragma unroll
    for (int p = 0; p < 4; ++p) v[p] = x[64 * p + threadIdx.x];
    if (threadIdx.x < 32) v[4] = x[256 + threadIdx.x];
#pragma unroll
    for (int p = 0; p < 4; ++p) y[64 * p + threadIdx.x] = v[p];
    if (threadIdx.x < 32) y[256 + threadIdx.x] = v[4];
}
// grid-stride float4 copy with 256-thread WGs (the "textbook" streaming copy)
__global__ __launch_bounds__(256) void copy_f4_stream(float4* __restrict__ o, const float4* __restrict__ i, long n)
{
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < n; k += (long)gridDim.x * 256) o[k] = i[k];
}
// read f2 (8 B) gather, write f4 through LDS transpose
__global__ __launch_bounds__(64) void copy_f2in_f4out(float4* __restrict__ o, const float2* __restrict__ i, long nb)
{
    __shared__ float2 t[576];
    long b = blockIdx.x; const float2* x = i + b * 576; float4* y = o + b * 288;
#pragma unroll
    for (int p = 0; p < 9; ++p) t[64 * p + threadIdx.x] = x[64 * p + threadIdx.x];
    __syncthreads();
    const float4* t4 = reinterpret_cast<const float4*>(t);
#pragma unroll
    for (int p = 0; p < 4; ++p) y[64 * p + threadIdx.x] = t4[64 * p + threadIdx.x];
    if (threadIdx.x < 32) y[256 + threadIdx.x] = t4[256 + threadIdx.x];
}

int main()
{
    const long nb_list[] = { 4096, 65536 };
    for (long nb : nb_list) {
        const long ring = (nb == 4096) ? 40 : 6;                     // distinct buffer sets (defeat the 256 MiB Infinity Cache)
        const size_t bytes = (size_t)nb * 4608;
        std::vector<void*> in(ring), out(ring);
        for (long r = 0; r < ring; ++r) { CK(hipMalloc(&in[r], bytes)); CK(hipMalloc(&out[r], bytes)); CK(hipMemset(in[r], 1, bytes)); }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int variant = 0; variant < 4; ++variant) {
            const int reps = 200;
            float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipEventRecord(e0));
                for (int r = 0; r < reps; ++r) {
                    void* i = in[r % ring]; void* o = out[r % ring];
                    if (variant == 0) hipLaunchKernelGGL(copy_f2_block, dim3(nb), dim3(64), 0, 0, (float2*)o, (const float2*)i, nb);
                    if (variant == 1) hipLaunchKernelGGL(copy_f4_block, dim3(nb), dim3(64), 0, 0, (float4*)o, (const float4*)i, nb);
                    if (variant == 2) hipLaunchKernelGGL(copy_f4_stream, dim3(2048), dim3(256), 0, 0, (float4*)o, (const float4*)i, (long)(bytes / 16));
                    if (variant == 3) hipLaunchKernelGGL(copy_f2in_f4out, dim3(nb), dim3(64), 0, 0, (float4*)o, (const float2*)i, nb);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            }
            const char* names[] = { "f2 per-block (9x8B/lane)", "f4 per-block (4.5x16B/lane)", "f4 grid-stride 256thr", "f2 in, f4 out via LDS" };
            printf("nb=%6ld %-30s %8.2f us/launch  %7.1f GB/s (read+write)\n", nb, names[variant], ms * 1000 / reps, 2.0 * bytes * reps / (ms * 1e-3) / 1e9);
        }
        for (long r = 0; r < ring; ++r) { hipFree(in[r]); hipFree(out[r]); }
    }
    return 0;
}
