// Probe kernels for tools/coresidency_probe.py: pure-ALU spins that allocate a chosen number of VGPRs / bytes of LDS (not part of the product).
#include <hip/hip_runtime.h>
template <int VG, int THREADS> __global__ __launch_bounds__(THREADS) void spin_kernel(int iters, float* sink) {
    extern __shared__ float lds[];
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    if (VG > 200) asm volatile("v_mov_b32 v223, 0" ::: "v223");
    else if (VG > 64) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    else if (VG > 32) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    else asm volatile("v_mov_b32 v31, 0" ::: "v31");
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    if (a == 123.456f) { *sink = a; lds[threadIdx.x] = a; }
}
// stamps: per block {start, end} of s_memrealtime (100 MHz) and the hardware id (XCC / SE / CU) it ran on
__global__ __launch_bounds__(256) void stamp_spin_kernel(int iters, unsigned long long* stamps, float* sink) {
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long r0 = wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) { a = a * b + 0.5f; b = b * 0.99999f + 1e-6f; }
    if (a == 123.456f) *sink = a;
    unsigned long long r1 = wall_clock64();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = r0; stamps[2 * blockIdx.x + 1] = r1; }
    (void)t0;
}
static float* sink_ptr() { static float* sink = nullptr; if (!sink) (void)hipMalloc(&sink, 4); return sink; }
extern "C" int ttl_probe_spin(int blocks, int vgprs, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    float* sink = sink_ptr();
    if (vgprs > 64) hipLaunchKernelGGL((spin_kernel<96, 256>), dim3(blocks), dim3(256), 0, s, iters, sink);
    else if (vgprs > 32) hipLaunchKernelGGL((spin_kernel<64, 256>), dim3(blocks), dim3(256), 0, s, iters, sink);
    else hipLaunchKernelGGL((spin_kernel<32, 256>), dim3(blocks), dim3(256), 0, s, iters, sink);
    return (int)hipGetLastError();
}
// a stand-in with the big-M GEMM's footprint: 512 threads, `vgprs` (32 or 224) registers, `lds_bytes` of dynamic LDS
extern "C" int ttl_probe_spin_big(int blocks, int vgprs, int lds_bytes, int iters, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    float* sink = sink_ptr();
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute((const void*)spin_kernel<224, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)spin_kernel<32, 512>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    if (vgprs > 200) hipLaunchKernelGGL((spin_kernel<224, 512>), dim3(blocks), dim3(512), lds_bytes, s, iters, sink);
    else hipLaunchKernelGGL((spin_kernel<32, 512>), dim3(blocks), dim3(512), lds_bytes, s, iters, sink);
    return (int)hipGetLastError();
}
extern "C" int ttl_probe_stamp_spin(int blocks, int iters, unsigned long long* stamps, void* stream) {
    hipLaunchKernelGGL(stamp_spin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, iters, stamps, sink_ptr());
    return (int)hipGetLastError();
}
