// HOST-ONLY stand-in for <hip/hip_runtime.h>, used by `make asan` and by nothing else: the host glue of libttl_hip (api.hip:
// config validation, weight loading by name, arena set-up, the launch sequences' pointer arithmetic, the error paths) is
// compiled as plain C++ with -fsanitize=address against this header and asan/launch_stubs.cpp, and run on the CPU
// (tests/test_asan_cpu.py).  "Device" memory is host heap memory, so AddressSanitizer sees every byte the glue and the
// host-side stubs of the weight-image kernels touch.  GPU AddressSanitizer does not exist on the target pool; this never
// runs on a GPU and is not part of libttl_hip.so.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

typedef enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorUnknown = 999 } hipError_t;
typedef struct ihipStream_t* hipStream_t;
typedef struct ihipEvent_t* hipEvent_t;
typedef struct ihipGraph* hipGraph_t;
typedef struct ihipGraphExec* hipGraphExec_t;
typedef enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 } hipMemcpyKind;
typedef enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 } hipStreamCaptureMode;
typedef enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 } hipFuncAttribute;
typedef struct hipDeviceProp_t { int multiProcessorCount; } hipDeviceProp_t;
#ifdef __cplusplus
struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
#endif
typedef struct int3 { int x, y, z; } int3;

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorOutOfMemory ? "hipErrorOutOfMemory" : "hipError (asan stub)"; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int dev) { (void)dev; p->multiProcessorCount = 256; return hipSuccess; }
static inline hipError_t hipGetLastError(void) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipMallocAsync(void** p, size_t n, hipStream_t s) { (void)s; return hipMalloc(p, n); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipFreeAsync(void* p, hipStream_t s) { (void)s; free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k) { (void)k; memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st) { (void)k; (void)st; memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t s) { (void)s; memset(d, v, n); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t s) { (void)s; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { (void)s; return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v) { (void)f; (void)a; (void)v; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)malloc(8); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { (void)e; (void)s; return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { (void)a; (void)b; *ms = 0.f; return hipSuccess; }
static inline hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode m) { (void)s; (void)m; return hipSuccess; }
static inline hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g) { (void)s; *g = (hipGraph_t)malloc(8); return hipSuccess; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t* x, hipGraph_t g, void* a, void* b, size_t n) { (void)g; (void)a; (void)b; (void)n; *x = (hipGraphExec_t)malloc(8); return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t x, hipStream_t s) { (void)x; (void)s; return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { free(g); return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t x) { free(x); return hipSuccess; }
