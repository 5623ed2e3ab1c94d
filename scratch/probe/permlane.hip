// Probe: what do v_permlane32_swap / v_permlane16_swap do on gfx950?   hipcc --offload-arch=gfx950 permlane.hip -o permlane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o)
{
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
int main()
{
    unsigned* d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[4] = { "swap32 first ", "swap32 second", "swap16 first ", "swap16 second" };
    for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int r = 0; r < 4; ++r) printf("  row%d starts %3u", r, h[64 * v + 16 * r]); printf("\n"); }
    return 0;
}
