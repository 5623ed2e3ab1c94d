// Shader clock as the GPU runs it: one wave spins for `ticks` of the 100 MHz constant counter (s_memrealtime) and reports how many
// shader-clock cycles (s_memtime) went by.  Queued on the stream in front of and behind a burst of launches it names the clock state the
// burst ran in -- sysfs (pp_dpm_sclk) is sampled too slowly for a 2 ms burst and reads an idle state most of the time.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o scratch/probe/libsclk_probe.so scratch/probe/sclk_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ void k_sclk_probe(unsigned long long* out, unsigned long long ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long r0 = wall_clock64(), c0 = clock64();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) r1 = wall_clock64();
    out[0] = clock64() - c0;
    out[1] = r1 - r0;
}

extern "C" int sclk_probe(void* stream, void* out2, unsigned long long ticks)
{
    hipLaunchKernelGGL(k_sclk_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out2, ticks);
    return (int)hipGetLastError();
}
