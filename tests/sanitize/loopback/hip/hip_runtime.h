// TEST-ONLY loop-back HIP layer (tests/sanitize): the handful of runtime calls the host side of the product uses, implemented on the CPU so that
// gr-gfdm_amd/csrc/{gfdm_hip_api,gfdm_hostpipe,gfdm_jit}.hip -- compiled UNCHANGED as plain C++ with this directory in front of the include path --
// run in the GPU-less container under ThreadSanitizer / AddressSanitizer / UBSan.  Nothing under gr-gfdm_amd/ includes or links this.
//
//   device memory   = malloc'ed host memory, remembered in a registry (type Device, owning device ordinal)
//   pinned memory   = page-aligned host memory in the same registry (type Host); hipHostRegister adds caller ranges, identity device pointers
//   a stream        = a worker thread executing its queue in order (so completion really is asynchronous and another thread writes the results)
//   an event        = a generation counter signalled from the recording stream's thread
//   a kernel launch = a host function enqueued on the stream (hipLaunchKernelGGL; the GFDM kernels' launchers are loopback_kernels.cc)
// It is NOT a HIP emulator: only what those three files call exists, with the semantics they rely on.
#ifndef GFDM_TEST_LOOPBACK_HIP_RUNTIME_H
#define GFDM_TEST_LOOPBACK_HIP_RUNTIME_H

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>

#define GFDM_LOOPBACK_HIP 1

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_SYMBOL(x) x

typedef enum hipError_t {
    hipSuccess = 0,
    hipErrorInvalidValue = 1,
    hipErrorOutOfMemory = 2,
    hipErrorNotInitialized = 3,
    hipErrorInvalidDevice = 101,
    hipErrorInvalidImage = 200,
    hipErrorNotFound = 500,
    hipErrorNotReady = 600,
    hipErrorLaunchFailure = 719,
    hipErrorHostMemoryAlreadyRegistered = 712,
    hipErrorHostMemoryNotRegistered = 713,
    hipErrorUnknown = 999
} hipError_t;

struct float2 { float x, y; };
static inline float2 make_float2(float x, float y) { float2 r; r.x = x; r.y = y; return r; }
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct float4 { float x, y, z, w; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct loopback_stream;
struct loopback_event;
struct loopback_module;
struct loopback_function;
typedef loopback_stream* hipStream_t;
typedef loopback_event* hipEvent_t;
typedef loopback_module* hipModule_t;
typedef loopback_function* hipFunction_t;
typedef void* hipDeviceptr_t;

enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeManaged = 3 };
struct hipPointerAttribute_t {
    hipMemoryType type;
    int device;
    void* devicePointer;
    void* hostPointer;
    int isManaged;
    unsigned allocationFlags;
};
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

#define hipStreamNonBlocking 0x01
#define hipEventDisableTiming 0x02
#define hipHostMallocMapped 0x02
#define hipHostRegisterPortable 0x01
#define hipHostRegisterMapped 0x02

extern "C" {
const char* hipGetErrorString(hipError_t e);
hipError_t hipGetLastError(void);
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void* p);
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned flags);
hipError_t hipHostRegister(void* p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void* p);
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p);
hipError_t hipMemGetAddressRange(hipDeviceptr_t* base, size_t* size, hipDeviceptr_t p);
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipDeviceSynchronize(void);
hipError_t hipModuleLoadData(hipModule_t* m, const void* image);
hipError_t hipModuleUnload(hipModule_t m);
hipError_t hipModuleGetFunction(hipFunction_t* f, hipModule_t m, const char* name);
hipError_t hipModuleLaunchKernel(hipFunction_t f, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz, unsigned shmem, hipStream_t s,
                                 void** args, void** extra);
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute attr, int value);
}

template <class T> static inline hipError_t hipMalloc(T** p, size_t bytes) { return hipMalloc(reinterpret_cast<void**>(p), bytes); }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t bytes, unsigned flags) { return hipHostMalloc(reinterpret_cast<void**>(p), bytes, flags); }

// ---- what the tests (and loopback_kernels.cc) use on top --------------------------------------------------------------------------------------
namespace loopback {
// enqueue `fn` on the stream (nullptr = the calling thread's current device's default stream); false when an injected launch failure fired
hipError_t enqueue(hipStream_t s, std::function<void()> fn);
// the next `count` calls of the named runtime function (e.g. "hipHostMalloc", "launch", "hipMemcpyAsync", "hipStreamCreateWithFlags",
// "hipModuleLoadData") after skipping `skip` of them fail with `err`
void fail_next(const char* api, int skip, int count, hipError_t err);
void clear_failures();
// injected failures fire only on threads that are inside arm(true) ... arm(false): the driver arms around the product call, so its own allocations are spared
void arm(bool on);
// statistics for the assertions of the driver
struct Stats { long launches, async_copies, streams_created, streams_destroyed, host_allocs, host_frees, dev_allocs, dev_frees, registers, unregisters, modules_loaded; };
Stats stats();
// live allocations of the layer's own (leak check at the end of a run); registrations excluded
long live_allocations();
void set_device_count(int n);
// hipPointerGetAttributes on memory the layer has never seen: true = hipErrorInvalidValue (ROCm <= 5), false (default) = hipSuccess + hipMemoryTypeUnregistered
void set_unknown_pointer_is_error(bool on);
}  // namespace loopback

static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_store(ptr, val, order, scope) __atomic_store_n((ptr), (val), (order))
#define __hip_atomic_load(ptr, order, scope) __atomic_load_n((ptr), (order))

// kernel<<<>>> as the product spells it: the "kernel" is an ordinary function here, run later on the stream's thread with copies of its arguments
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                                         \
    do {                                                                                                    \
        (void)(grid); (void)(block); (void)(shmem);                                                         \
        (void)loopback::enqueue((stream), std::bind((kernel), __VA_ARGS__));                                \
    } while (0)

#endif
