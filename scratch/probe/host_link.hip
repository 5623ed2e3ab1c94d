// Probe for the host-buffer batch path (round 4): what does the PCIe link of the MI355X box give, and through which mechanism?
//   hipcc -O3 --offload-arch=gfx950 -o host_link host_link.hip -lpthread && ./host_link
// Rows: CPU memcpy pageable -> pinned (1, 2, 4 threads), copy-engine H2D / D2H from pinned and from pageable memory, both directions at once,
// a kernel reading / writing pinned host memory across the link (what the small *_host calls already do), hipHostRegister cost, event wait latency.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void k_copy_v(v2f* __restrict__ dst, const v2f* __restrict__ src, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
static void k_copy_launch(int grid, hipStream_t s, float2* dst, const float2* src, size_t n)
{
    hipLaunchKernelGGL(k_copy_v, dim3(grid), dim3(256), 0, s, (v2f*)dst, (const v2f*)src, n);
}

static void par_memcpy(char* d, const char* s, size_t n, int T)
{
    if (T <= 1) { memcpy(d, s, n); return; }
    std::vector<std::thread> th;
    size_t per = (n / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; ++t) {
        size_t o = per * t; if (o >= n) break;
        size_t len = (o + per > n) ? n - o : per;
        th.emplace_back([=] { memcpy(d + o, s + o, len); });
    }
    for (auto& x : th) x.join();
}

int main()
{
    const size_t B = 64u << 20;
    char* pageable_a = (char*)aligned_alloc(4096, B);
    char* pageable_b = (char*)aligned_alloc(4096, B);
    memset(pageable_a, 1, B); memset(pageable_b, 2, B);
    char *pin_a, *pin_b, *dev_a, *dev_b;
    CK(hipHostMalloc((void**)&pin_a, B, hipHostMallocMapped));
    CK(hipHostMalloc((void**)&pin_b, B, hipHostMallocMapped));
    memset(pin_a, 3, B); memset(pin_b, 4, B);
    CK(hipMalloc((void**)&dev_a, B)); CK(hipMalloc((void**)&dev_b, B));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    const int R = 10;

    for (int T : { 1, 2, 4, 8 }) {
        par_memcpy(pin_a, pageable_a, B, T);
        double t0 = now();
        for (int r = 0; r < R; ++r) par_memcpy(pin_a, pageable_a, B, T);
        double dt = (now() - t0) / R;
        printf("memcpy pageable->pinned  %d thread(s): %6.1f GB/s\n", T, B / dt * 1e-9);
    }
    {
        memcpy(pageable_b, pin_b, B);
        double t0 = now();
        for (int r = 0; r < R; ++r) memcpy(pageable_b, pin_b, B);
        printf("memcpy pinned->pageable  1 thread   : %6.1f GB/s\n", B / ((now() - t0) / R) * 1e-9);
    }
    auto timed = [&](const char* name, auto&& fn, double bytes) {
        fn(); CK(hipDeviceSynchronize());
        double t0 = now();
        for (int r = 0; r < R; ++r) fn();
        CK(hipDeviceSynchronize());
        printf("%-52s: %6.1f GB/s\n", name, bytes / ((now() - t0) / R) * 1e-9);
    };
    timed("copy engine H2D from pinned", [&] { CK(hipMemcpyAsync(dev_a, pin_a, B, hipMemcpyHostToDevice, s0)); }, B);
    timed("copy engine D2H to pinned", [&] { CK(hipMemcpyAsync(pin_b, dev_b, B, hipMemcpyDeviceToHost, s1)); }, B);
    timed("copy engine H2D + D2H at once (sum)", [&] { CK(hipMemcpyAsync(dev_a, pin_a, B, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(pin_b, dev_b, B, hipMemcpyDeviceToHost, s1)); }, 2.0 * B);
    timed("copy engine H2D from pageable", [&] { CK(hipMemcpyAsync(dev_a, pageable_a, B, hipMemcpyHostToDevice, s0)); }, B);
    timed("copy engine D2H to pageable", [&] { CK(hipMemcpyAsync(pageable_b, dev_b, B, hipMemcpyDeviceToHost, s1)); }, B);
    for (size_t chunk : { (size_t)256 << 10, (size_t)1 << 20, (size_t)4 << 20 }) {
        char name[96];
        snprintf(name, sizeof name, "copy engine H2D from pinned in %zu KiB pieces", chunk >> 10);
        timed(name, [&] { for (size_t o = 0; o < B; o += chunk) CK(hipMemcpyAsync(dev_a + o, pin_a + o, chunk, hipMemcpyHostToDevice, s0)); }, B);
    }
    float2 *pa_d, *pb_d;
    CK(hipHostGetDevicePointer((void**)&pa_d, pin_a, 0)); CK(hipHostGetDevicePointer((void**)&pb_d, pin_b, 0));
    const size_t n = B / sizeof(float2);
    for (int grid : { 256, 1024, 4096 }) {
        char name[96];
        snprintf(name, sizeof name, "kernel reads pinned host -> HBM, grid %d x 256", grid);
        timed(name, [&] { k_copy_launch(grid, s0, (float2*)dev_a, pa_d, n); }, B);
        snprintf(name, sizeof name, "kernel HBM -> writes pinned host, grid %d x 256", grid);
        timed(name, [&] { k_copy_launch(grid, s0, pb_d, (const float2*)dev_b, n); }, B);
        snprintf(name, sizeof name, "kernel pinned -> pinned (both directions, sum), grid %d", grid);
        timed(name, [&] { k_copy_launch(grid, s0, pb_d, pa_d, n); }, 2.0 * B);
    }
    {
        double t0 = now();
        CK(hipHostRegister(pageable_a, B, hipHostRegisterMapped));
        double t_reg = now() - t0;
        float2* ra_d;
        CK(hipHostGetDevicePointer((void**)&ra_d, pageable_a, 0));
        printf("hipHostRegister of 64 MiB: %.2f ms (%.1f us per MiB)\n", t_reg * 1e3, t_reg * 1e6 / 64);
        timed("copy engine H2D from registered memory", [&] { CK(hipMemcpyAsync(dev_a, pageable_a, B, hipMemcpyHostToDevice, s0)); }, B);
        timed("kernel reads registered host -> HBM, grid 1024", [&] { k_copy_launch(1024, s0, (float2*)dev_a, ra_d, n); }, B);
        hipPointerAttribute_t at;
        t0 = now();
        for (int r = 0; r < 1000; ++r) (void)hipPointerGetAttributes(&at, pageable_a + 4096 * r);
        printf("hipPointerGetAttributes (registered): %.2f us, type %d\n", (now() - t0) * 1e3, (int)at.type);
        t0 = now();
        hipError_t e = hipSuccess;
        for (int r = 0; r < 1000; ++r) { e = hipPointerGetAttributes(&at, pageable_b + 4096 * r); (void)hipGetLastError(); }
        printf("hipPointerGetAttributes (pageable): %.2f us, result '%s' type %d\n", (now() - t0) * 1e3, hipGetErrorString(e), (int)at.type);
        t0 = now();
        CK(hipHostUnregister(pageable_a));
        printf("hipHostUnregister: %.2f ms\n", (now() - t0) * 1e3);
    }
    {
        hipEvent_t ev;
        CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        double t0 = now();
        for (int r = 0; r < 200; ++r) {
            k_copy_launch(1, s0, (float2*)dev_a, (const float2*)dev_b, (size_t)64);
            CK(hipEventRecord(ev, s0));
            CK(hipEventSynchronize(ev));
        }
        printf("tiny kernel + event record + hipEventSynchronize: %.1f us per round\n", (now() - t0) / 200 * 1e6);
        t0 = now();
        for (int r = 0; r < 200; ++r) {
            k_copy_launch(1, s0, (float2*)dev_a, (const float2*)dev_b, (size_t)64);
            CK(hipStreamSynchronize(s0));
        }
        printf("tiny kernel + hipStreamSynchronize: %.1f us per round\n", (now() - t0) / 200 * 1e6);
        t0 = now();
        for (int r = 0; r < 200; ++r) {
            CK(hipMemcpyAsync(dev_a, pin_a, 4608, hipMemcpyHostToDevice, s0));
            k_copy_launch(1, s0, (float2*)dev_b, (const float2*)dev_a, (size_t)64);
            CK(hipMemcpyAsync(pin_b, dev_b, 4608, hipMemcpyDeviceToHost, s0));
            CK(hipStreamSynchronize(s0));
        }
        printf("H2D 4.6 KB + tiny kernel + D2H 4.6 KB + sync: %.1f us per round\n", (now() - t0) / 200 * 1e6);
    }
    printf("hardware threads %u\n", std::thread::hardware_concurrency());
    return 0;
}
