// TEST-ONLY loop-back hiprtc (tests/sanitize, see hip_runtime.h): "compiles" in a few milliseconds into a self-describing blob that the loop-back
// hipModuleLoadData understands, so that gfdm_jit.hip's disk cache, background pool, load / retry and exit paths run without a compiler or a GPU.
#ifndef GFDM_TEST_LOOPBACK_HIPRTC_H
#define GFDM_TEST_LOOPBACK_HIPRTC_H

#include <hip/hip_runtime.h>

typedef enum hiprtcResult { HIPRTC_SUCCESS = 0, HIPRTC_ERROR_OUT_OF_MEMORY = 1, HIPRTC_ERROR_COMPILATION = 6, HIPRTC_ERROR_INVALID_INPUT = 3 } hiprtcResult;
struct loopback_rtc_program;
typedef loopback_rtc_program* hiprtcProgram;

extern "C" {
const char* hiprtcGetErrorString(hiprtcResult r);
hiprtcResult hiprtcVersion(int* major, int* minor);
hiprtcResult hiprtcCreateProgram(hiprtcProgram* prog, const char* src, const char* name, int nheaders, const char* const* headers, const char* const* include_names);
hiprtcResult hiprtcDestroyProgram(hiprtcProgram* prog);
hiprtcResult hiprtcAddNameExpression(hiprtcProgram prog, const char* expr);
hiprtcResult hiprtcCompileProgram(hiprtcProgram prog, int nopts, const char** opts);
hiprtcResult hiprtcGetProgramLogSize(hiprtcProgram prog, size_t* n);
hiprtcResult hiprtcGetProgramLog(hiprtcProgram prog, char* log);
hiprtcResult hiprtcGetLoweredName(hiprtcProgram prog, const char* expr, const char** lowered);
hiprtcResult hiprtcGetCodeSize(hiprtcProgram prog, size_t* n);
hiprtcResult hiprtcGetCode(hiprtcProgram prog, char* code);
}

namespace loopback {
void set_compile_ms(int lo, int hi);      // a "compile" sleeps a random time in [lo, hi] milliseconds (default 2..20)
long compiles();                          // programs compiled so far
}

#endif
