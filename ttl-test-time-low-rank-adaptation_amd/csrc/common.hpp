// Shared device/host helpers for libttl_hip (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// ---- switches.  The PRODUCT libraries (libttl_hip.so, libttl_hip_fp16.so, libttl_hip_strict.so) read exactly the environment
// variables of this table — ttl_runtime_switches() (include/ttl_hip.h) lists them with default and current value, bench.py prints them
// under protocol.kernel_env and refuses to run with a non-default one unless --variant-env is given.  Every other TTL_* knob the
// sources mention is a CLOSED experiment (measured neutral or slower, profiles/r0*_experiments.txt): TTL_EXPERIMENT(name, default)
// is the constant `default` — no getenv, no string in the binary — unless the library is built with -DTTL_EXPERIMENTS
// (libttl_hip_fp16_exp.so: TEST / tools only, never benched as the product; tools/hip_variant.sh builds its variants that way).
enum TtlSwitch { SW_GEMM_HUGE = 0, SW_GEMM_HUGE_NARROW, SW_GEMM_HUGE_MIN_FILL, SW_BWD_COMPACT, SW_CONCURRENCY, SW_COUNT };
int ttl_switch(TtlSwitch id);          // api.hip: the variable's value in the environment NOW, else its default
static inline int ttl_env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
#ifdef TTL_EXPERIMENTS
#define TTL_EXPERIMENT(name, dflt) ttl_env_int(name, dflt)
#else
#define TTL_EXPERIMENT(name, dflt) (dflt)
#endif

// ---- operand type of every MFMA product (activations, weights, attention probabilities).
// Default build: bf16 (BASELINE.json north_star).  -DTTL_OPERAND_FP16 builds libttl_hip_fp16.so with
// IEEE half operands — the dtype of the reference's own GPU path (torch.cuda.amp.autocast(), ttl.py:79)
// — same MFMA rate, 3 more mantissa bits; its backward carries a fixed 2^10 loss scale like the
// reference's GradScaler(init_scale=1000) (ttl.py:222).  Accumulation is fp32 in both.
#ifdef TTL_OPERAND_FP32
// -DTTL_OPERAND_FP32 builds libttl_hip_strict.so: a TEST-ONLY third build (SURVEY §7.2 "strict kernel instantiation") in which
// every operand buffer holds fp32 and every product runs on v_mfma_f32_32x32x2_f32 / fp32 FMAs (strict_gemm.hip,
// strict_attention.hip and the fp32 branches of lora.hip replace gemm.hip, gemm_big.hip, attention.hip and the MFMA16 kernels).
// The host glue (api.hip: every launch sequence), LayerNorm forward / backward, head, loss, optimizer, casts and im2col are the
// SAME sources as the two product builds.  It exists so that the north_star's 1e-3 tolerance on LoRA gradients / weights can be
// asserted against the reference's fp32 path without the operand-rounding noise of a 16-bit forward; it is never benched.
typedef float op_t;
typedef float op_scalar;
#define TTL_OPERAND_NAME "fp32"
#define TTL_GRAD_SCALE 1.0f
#define TTL_DS_PRESCALE 1.0f
#else
typedef uint16_t op_t;  // storage type; arithmetic is always fp32
#ifdef TTL_OPERAND_FP16
typedef _Float16 op_scalar;
#define TTL_OPERAND_NAME "fp16"
#define TTL_GRAD_SCALE 1024.0f
// Attention backward, dS = P o (dP - delta): the one gradient-side operand that is a product with a probability, so most of
// its elements are orders of magnitude below the rest of the (loss-scaled) backward and fall into fp16's subnormal range
// under the reference's 2^10 scale (K = 1000 with every view selected: one gradient tensor 8e-3 off; with dS kept exact
// 1.5e-3, which is what the 16-bit FORWARD leaves: tools/fp16_grad_points.py, profiles/r03_fp16_grad_points.txt).  dS is
// therefore rounded to fp16 as dS * 2^8 and the products that consume it (dQ = dS K, dK = dS^T Q) are scaled back by 2^-8 in
// fp32: exact powers of two, no change to any other operand.  An overflow of the pre-scaled value is an inf in the
// gradients like any other: found_inf, whole step skipped, loss scale halved (GradScaler contract).
#define TTL_DS_PRESCALE 256.0f
#define MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#define MFMA32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#else
typedef __bf16 op_scalar;
#define TTL_OPERAND_NAME "bf16"
#define TTL_GRAD_SCALE 1.0f
#define TTL_DS_PRESCALE 1.0f      // bf16 has fp32's exponent range
#define MFMA16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#define MFMA32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
#endif
#endif  // TTL_OPERAND_FP32
#ifndef TTL_HOST_STUB   // (`make asan`: the host glue compiled as plain C++ against asan/hip/hip_runtime.h needs the storage type only)
typedef __attribute__((ext_vector_type(8))) op_scalar opx8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// fp32 -> operand, round to nearest even (plain cast: hipcc emits v_cvt_pk_bf16_f32 / v_cvt_f16_f32;
// NaN stays NaN)
#ifdef TTL_OPERAND_FP32
__device__ __forceinline__ op_t f32_to_op(float f) { return f; }
__device__ __forceinline__ float op_to_f32(op_t v) { return v; }
// four / eight consecutive operand elements from fp32 values (16-B aligned destination in the 16-bit builds' terms)
__device__ __forceinline__ void st_op4(op_t* p, float a, float b, float c, float d) { *(float4*)p = make_float4(a, b, c, d); }
__device__ __forceinline__ void st_op8(op_t* p, const float (&v)[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
#else
__device__ __forceinline__ op_t f32_to_op(float f) {
    op_scalar b = (op_scalar)f;
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack_op2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) op_scalar op2;
    op2 v = {(op_scalar)lo, (op_scalar)hi};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void st_op4(op_t* p, float a, float b, float c, float d) { *(u32x2*)p = u32x2{pack_op2(a, b), pack_op2(c, d)}; }
__device__ __forceinline__ void st_op8(op_t* p, const float (&v)[8]) {
    *(u32x4*)p = u32x4{pack_op2(v[0], v[1]), pack_op2(v[2], v[3]), pack_op2(v[4], v[5]), pack_op2(v[6], v[7])};
}
#ifdef TTL_OPERAND_FP16
__device__ __forceinline__ float op_to_f32(op_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ float op_lo(uint32_t w) { return op_to_f32((op_t)(w & 0xFFFFu)); }
__device__ __forceinline__ float op_hi(uint32_t w) { return op_to_f32((op_t)(w >> 16)); }
#else
__device__ __forceinline__ float op_to_f32(op_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ float op_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float op_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
#endif
#endif  // TTL_OPERAND_FP32

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// quick_gelu(u) = u * sigmoid(1.702 u)   (HF ACT2FN["quick_gelu"], modeling_clip.py:346-350)
// sigmoid through v_exp_f32 (base 2) + v_rcp_f32: 4 VALU instead of the ~15 of an IEEE division; both
// are accurate to ~1 ulp, far inside the 16-bit operand the result is rounded to
__device__ __forceinline__ float sigmoid_1702(float u) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * u));
}
__device__ __forceinline__ float quick_gelu_f(float u) { return u * sigmoid_1702(u); }
__device__ __forceinline__ float quick_gelu_grad_f(float u) {
    float s = sigmoid_1702(u);
    return s * (1.0f + 1.702f * u * (1.0f - s));
}

// ---- in-kernel clock stamps (-DTTL_CLOCK_STAMPS: DIAGNOSTIC builds only, tools/r06_clock_stamps.sh; in the product build no stamp
// executes).  MI355X_MICROARCH.md DVFS give-back item 6: clock held = d s_memtime / d s_memrealtime x 100 MHz.  Every workgroup keeps
// its stamps in scalar registers and writes them ONCE, behind its last tile, into a buffer of their own that nothing else reads.
#ifdef TTL_CLOCK_STAMPS
struct TtlClockStamp { unsigned long long t0, r0, t1, r1, k0, kr0, k1, kr1; };     // whole kernel (t, r); K loop of the first tile (k, kr)
#define TTL_STAMP_SLOTS 2048
#define TTL_STAMP_DECL unsigned long long st_t0 = 0, st_r0 = 0, st_k0 = 0, st_kr0 = 0, st_k1 = 0, st_kr1 = 0
#define TTL_STAMP_BEGIN() do { st_t0 = __builtin_amdgcn_s_memtime(); st_r0 = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TTL_STAMP_K0(first) do { if (first) { st_k0 = __builtin_amdgcn_s_memtime(); st_kr0 = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define TTL_STAMP_K1(first) do { if (first) { st_k1 = __builtin_amdgcn_s_memtime(); st_kr1 = __builtin_amdgcn_s_memrealtime(); } } while (0)
#define TTL_STAMP_END(buf) do { if (threadIdx.x == 0 && blockIdx.x < TTL_STAMP_SLOTS) { \
        buf[blockIdx.x] = TtlClockStamp{st_t0, st_r0, __builtin_amdgcn_s_memtime(), __builtin_amdgcn_s_memrealtime(), st_k0, st_kr0, st_k1, st_kr1}; } } while (0)
#else
#define TTL_STAMP_DECL
#define TTL_STAMP_BEGIN() do {} while (0)
#define TTL_STAMP_K0(first) do {} while (0)
#define TTL_STAMP_K1(first) do {} while (0)
#define TTL_STAMP_END(buf) do {} while (0)
#endif

// bijective XCD-aware remap of a 1-D block id: blocks b and b+8 share an XCD (round-robin
// dispatch), so give each XCD a contiguous chunk of the tile order for L2 reuse.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}
#endif  // TTL_HOST_STUB
