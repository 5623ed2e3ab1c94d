// What does a one-block host call cost on this stack?  launch + wait of an empty kernel with (a) hipStreamSynchronize, (b) a flag in
// pinned host memory that a second tiny kernel on the same stream sets and the host polls, (c) the flag set by the work kernel itself.
// hipcc --offload-arch=gfx950 -O2 -o host_latency host_latency.hip && ./host_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <atomic>

__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_flag(volatile unsigned* flag, unsigned v) { __threadfence_system(); *flag = v; }
__global__ void k_work_flag(const float* in, float* out, volatile unsigned* flag, unsigned v)
{
    out[threadIdx.x] = in[threadIdx.x] * 2.f;
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *flag = v;
}

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned* flag; unsigned* dflag;
    (void)hipHostMalloc((void**)&flag, 64, hipHostMallocMapped);
    (void)hipHostGetDevicePointer((void**)&dflag, flag, 0);
    float *hin, *hout, *din, *dout;
    (void)hipHostMalloc((void**)&hin, 8192, hipHostMallocMapped); (void)hipHostMalloc((void**)&hout, 8192, hipHostMallocMapped);
    (void)hipHostGetDevicePointer((void**)&din, hin, 0); (void)hipHostGetDevicePointer((void**)&dout, hout, 0);
    const int reps = 2000;
    for (int w = 0; w < 200; ++w) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, nullptr); (void)hipStreamSynchronize(s); }
    double t0 = now();
    for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, nullptr); (void)hipStreamSynchronize(s); }
    printf("(a) empty kernel + hipStreamSynchronize:        %.2f us\n", (now() - t0) / reps);
    *flag = 0;
    t0 = now();
    for (int i = 1; i <= reps; ++i) {
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, nullptr);
        hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s, dflag, (unsigned)i);
        while (*(volatile unsigned*)flag != (unsigned)i) { }
    }
    printf("(b) empty kernel + flag kernel, host polls:     %.2f us\n", (now() - t0) / reps);
    (void)hipStreamSynchronize(s);
    *flag = 0;
    t0 = now();
    for (int i = 1; i <= reps; ++i) {
        hipLaunchKernelGGL(k_work_flag, dim3(1), dim3(256), 0, s, din, dout, dflag, (unsigned)i);
        while (*(volatile unsigned*)flag != (unsigned)i) { }
    }
    printf("(c) one kernel that sets the flag, host polls:  %.2f us\n", (now() - t0) / reps);
    (void)hipStreamSynchronize(s);
    t0 = now();
    for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, nullptr); }
    printf("(d) launch enqueue alone (no wait):             %.2f us\n", (now() - t0) / reps);
    (void)hipStreamSynchronize(s);
    hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    t0 = now();
    for (int i = 0; i < reps; ++i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, nullptr); (void)hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) { } }
    printf("(e) empty kernel + event record + query spin:   %.2f us\n", (now() - t0) / reps);
    return 0;
}
