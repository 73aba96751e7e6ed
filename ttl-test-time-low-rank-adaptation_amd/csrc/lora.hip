// Rank-r LoRA side of the q/v projections (peft LoRA Linear as configured at
// clip/custom_clip.py:583-590: y = xW^T + b + (alpha/r)·(x A^T) B^T).
//
// Forward and dgrad ride the big MFMA GEMMs through K-extension:
//     [x1 | s·x1A^T] · [W | B]^T            and            [dq dk dv | dU] · [W^T | A^T]^T
// so this file only holds the skinny pieces:
//   lora_refresh : bf16 images of A/B inside the extended weight matrices (after every step)
//   lora_skinny  : [M,D] x [D,2r]  (U = s·x1·A^T on the forward, dU = s·dy·B on the backward)
//   lora_wgrad   : dA = dU^T x1, dB = dy^T (s·U): reductions over all M = N·T tokens, done as
//                  chunked MFMA products (both operands k-strided -> ds_read_b64_tr_b16) plus a
//                  deterministic second-pass sum (no float atomics: bitwise reproducible grads)
#include <atomic>

#include "kernels.hpp"

namespace {

__global__ void refresh_kernel(const float* __restrict__ Aq, const float* __restrict__ Bq, const float* __restrict__ Av,
                               const float* __restrict__ Bv, int D, int r, op_t* __restrict__ wext, int ldw,
                               op_t* __restrict__ wtext, int ldwt, op_t* __restrict__ acat,
                               op_t* __restrict__ btcat) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * r) return;
    {   // B side: i = n*r + j
        int n = i / r, j = i - n * r;
        op_t bq = f32_to_op(Bq[i]), bv = f32_to_op(Bv[i]);
        wext[(size_t)n * ldw + D + j] = bq;
        wext[(size_t)(2 * D + n) * ldw + D + r + j] = bv;
        btcat[(size_t)j * D + n] = bq;
        btcat[(size_t)(r + j) * D + n] = bv;
    }
    {   // A side: i = j*D + d
        int j = i / D, d = i - j * D;
        op_t aq = f32_to_op(Aq[i]), av = f32_to_op(Av[i]);
        acat[(size_t)j * D + d] = aq;
        acat[(size_t)(r + j) * D + d] = av;
        wtext[(size_t)d * ldwt + 3 * D + j] = aq;
        wtext[(size_t)d * ldwt + 3 * D + r + j] = av;
    }
}

// One wave: 16 rows of X times all 2r columns.  MFMA A operand = Wcat rows (output row = column
// c of the result), B operand = X rows (output column = token) -> each lane ends with 4
// consecutive result columns of one token: one 8-byte store.  The kernel is pure latency (a
// few hundred waves, each a serial K loop), so ALL of a wave's X fragments are requested up
// front (KS x 16 B per lane in flight) and Wcat is staged once per workgroup in LDS.
template <int NCG, int KS>   // KS = D / 32 k-steps
__global__ __launch_bounds__(256) void skinny_kernel(const op_t* __restrict__ X, int ldx, int xoff_q, int xoff_v,
                                                     const op_t* __restrict__ W, int D, float scale,
                                                     op_t* __restrict__ out, int ldo, int M) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // Wcat [16*NCG][D] operand type, rows padded by 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int rowb = D * 2 + 16;
    for (int q = tid; q < 16 * NCG * (D / 8); q += 256) {
        int r = q / (D / 8), c = q - r * (D / 8);
        *(u32x4*)(smem + r * rowb + c * 16) = *(const u32x4*)(W + (size_t)r * D + c * 8);
    }
    const int m0 = (blockIdx.x * 4 + (tid >> 6)) * 16;
    const int li = lane & 15, lg = lane >> 4;
    const int row = min(m0 + li, M - 1);
    const bool same = (xoff_q == xoff_v);
    const op_t* xq = X + (size_t)row * ldx + xoff_q + 8 * lg;
    const op_t* xv = X + (size_t)row * ldx + xoff_v + 8 * lg;
    opx8 fq[KS], fv[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) fq[k] = *(const opx8*)(xq + 32 * k);
    if (!same) {
#pragma unroll
        for (int k = 0; k < KS; ++k) fv[k] = *(const opx8*)(xv + 32 * k);
    }
    __syncthreads();
    f32x4 acc[NCG];
#pragma unroll
    for (int c = 0; c < NCG; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < KS; ++k) {
#pragma unroll
        for (int c = 0; c < NCG; ++c) {
            opx8 w = *(const opx8*)(smem + (c * 16 + li) * rowb + (32 * k + 8 * lg) * 2);
            acc[c] = MFMA16(w, (c < NCG / 2 || same) ? fq[k] : fv[k], acc[c], 0, 0, 0);
        }
    }
    if (m0 < M && m0 + li < M) {
#pragma unroll
        for (int c = 0; c < NCG; ++c)
            *(u32x2*)(out + (size_t)(m0 + li) * ldo + c * 16 + 4 * lg) =
                u32x2{pack_op2(acc[c][0] * scale, acc[c][1] * scale), pack_op2(acc[c][2] * scale, acc[c][3] * scale)};
    }
}

// ---- weight gradients: partial[prod][chunk][r][D] = S_chunk^T · G_chunk over 256-token chunks
constexpr int WG_CH = 256;   // tokens per chunk
constexpr int WG_BN = 128;   // result columns per block

__device__ __forceinline__ int wg_u(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <int R>
__global__ __launch_bounds__(256) void wgrad_kernel(const op_t* __restrict__ x1ext, int ldx, const op_t* __restrict__ dqkv,
                                                    int ldd, int M, int D, float* __restrict__ partial, int nch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sG = smem;                       // [256][128] bf16, 256-B rows, 16-B chunk c at c ^ (u(row) << 1)
    char* sS = smem + WG_CH * WG_BN * 2;   // [256][R] bf16, plain
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = blockIdx.x, cb = blockIdx.y, prod = blockIdx.z;
    const int m0 = ch * WG_CH, n0 = cb * WG_BN;
    // operands of this product
    const op_t* S; int lds_; const op_t* G; int ldg;
    if (prod == 0)      { S = x1ext + D;           lds_ = ldx; G = dqkv;          ldg = ldd; }  // dB_q^T = Us_q^T dq
    else if (prod == 1) { S = x1ext + D + R;       lds_ = ldx; G = dqkv + 2 * D;  ldg = ldd; }  // dB_v^T = Us_v^T dv
    else if (prod == 2) { S = dqkv + 3 * D;        lds_ = ldd; G = x1ext;         ldg = ldx; }  // dA_q = dU_q^T x1
    else                { S = dqkv + 3 * D + R;    lds_ = ldd; G = x1ext;         ldg = ldx; }  // dA_v = dU_v^T x1
    // stage G tile: 256 rows x 16 chunks of 16 B
    for (int q = tid; q < WG_CH * 16; q += 256) {
        int r = q >> 4, c = q & 15;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m0 + r < M) v = *(const u32x4*)(G + (size_t)(m0 + r) * ldg + n0 + c * 8);
        *(u32x4*)(sG + r * 256 + ((c ^ (wg_u(r) << 1)) << 4)) = v;
    }
    for (int q = tid; q < WG_CH * (R / 8); q += 256) {
        int r = q / (R / 8), c = q - r * (R / 8);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m0 + r < M) v = *(const u32x4*)(S + (size_t)(m0 + r) * lds_ + c * 8);
        *(u32x4*)(sS + r * (R * 2) + (c << 4)) = v;
    }
    __syncthreads();
    // wave w owns result columns n0 + 32w .. +31 (two 16-col groups), all R rows
    const int li = lane & 15, lg = lane >> 4, tq = li >> 2, tp = li & 3;
    f32x4 acc[R / 16][2];
#pragma unroll
    for (int a = 0; a < R / 16; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = acc[a][0]; }
    typedef __attribute__((ext_vector_type(8))) short s16x8;
#pragma unroll 2
    for (int ks = 0; ks < WG_CH / 32; ++ks) {
        const int rlo = 32 * ks + 8 * lg + tq, rhi = rlo + 4;
        opx8 sf[R / 16], gf[2];
#pragma unroll
        for (int a = 0; a < R / 16; ++a) {
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sS + rlo * (R * 2) + (a * 16 + 4 * tp) * 2));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sS + rhi * (R * 2) + (a * 16 + 4 * tp) * 2));
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            sf[a] = __builtin_bit_cast(opx8, v);
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int slot = (32 * wave + 16 * b) / 4 + tp;  // 8-byte slot inside the 256-B row
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sG + rlo * 256 + ((slot ^ (wg_u(rlo) << 2)) << 3)));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sG + rhi * 256 + ((slot ^ (wg_u(rhi) << 2)) << 3)));
            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            gf[b] = __builtin_bit_cast(opx8, v);
        }
#pragma unroll
        for (int a = 0; a < R / 16; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = MFMA16(sf[a], gf[b], acc[a][b], 0, 0, 0);
    }
    float* po = partial + ((size_t)prod * nch + ch) * R * D;
#pragma unroll
    for (int a = 0; a < R / 16; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                po[(size_t)(a * 16 + 4 * lg + e) * D + n0 + 32 * wave + 16 * b + li] = acc[a][b][e];
}

// 64 outputs x 4 chunk-slices per block: slice q sums chunks q, q+4, ... in order, the four slice sums are added in a
// fixed order -> deterministic; one thread walking all ~50 chunks serially left the 9.8 MB read latency-bound (16 us)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nch, int D, int r,
                                                           float* __restrict__ gAq, float* __restrict__ gBq,
                                                           float* __restrict__ gAv, float* __restrict__ gBv,
                                                           const float* __restrict__ scaler_f, int* __restrict__ scaler_i) {
    __shared__ float part[4][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    const int prod = blockIdx.y;
    float a0 = 0.f, a1 = 0.f;
    if (i < r * D) {
        const float* p = partial + (size_t)prod * nch * r * D + i;
        int c = slice;
        for (; c + 4 < nch; c += 8) { a0 += p[(size_t)c * r * D]; a1 += p[(size_t)(c + 4) * r * D]; }
        for (; c < nch; c += 4) a0 += p[(size_t)c * r * D];
    }
    part[slice][threadIdx.x & 63] = a0 + a1;
    __syncthreads();
    if (slice != 0 || i >= r * D) return;
    float s = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    // scaler.unscale_: undo the backward's loss scale; any inf/nan makes the optimizer skip the WHOLE step (deyo.py:186-188)
    if (scaler_f) s *= scaler_f[1];
    if (scaler_i && !isfinite(s)) atomicOr(scaler_i, 1);
    int j = i / D, d = i - j * D;
    if (prod == 0) gBq[(size_t)d * r + j] = s;
    else if (prod == 1) gBv[(size_t)d * r + j] = s;
    else if (prod == 2) gAq[i] = s;
    else gAv[i] = s;
}

}  // namespace

hipError_t launch_lora_refresh(const float* Aq, const float* Bq, const float* Av, const float* Bv, int D, int r,
                               op_t* wqkv_ext, int ldw, op_t* wqkvT_ext, int ldwt, op_t* a_cat, op_t* bT_cat,
                               hipStream_t s) {
    hipLaunchKernelGGL(refresh_kernel, dim3((D * r + 255) / 256), dim3(256), 0, s, Aq, Bq, Av, Bv, D, r, wqkv_ext, ldw,
                       wqkvT_ext, ldwt, a_cat, bT_cat);
    return hipGetLastError();
}

template <int NCG, int KS>
static hipError_t skinny_launch(const op_t* X, int ldx, int xoff_q, int xoff_v, const op_t* Wcat, int D, float scale, op_t* out,
                                int ldo, int M, hipStream_t s) {
    const int smem = 16 * NCG * (D * 2 + 16);
    static bool done = false;
    if (!done) {
        hipError_t e = hipFuncSetAttribute((const void*)skinny_kernel<NCG, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        done = true;
    }
    hipLaunchKernelGGL((skinny_kernel<NCG, KS>), dim3((M + 63) / 64), dim3(256), smem, s, X, ldx, xoff_q, xoff_v, Wcat, D, scale, out,
                       ldo, M);
    return hipGetLastError();
}

hipError_t launch_lora_skinny(const op_t* X, int ldx, int xoff_q, int xoff_v, const op_t* Wcat, int D, int r,
                              float scale, op_t* out, int ldo, int M, hipStream_t s) {
    const int ncg = 2 * r / 16, ks = D / 32;
#define SK(N_, K_) if (ncg == N_ && ks == K_ && D % 32 == 0) return skinny_launch<N_, K_>(X, ldx, xoff_q, xoff_v, Wcat, D, scale, out, ldo, M, s)
    SK(2, 24); SK(4, 24);      // ViT-B/16, r = 16 / 32
    SK(2, 32); SK(4, 32);      // ViT-L/14
    SK(2, 16); SK(4, 16);      // text tower of ViT-B/16 (D = 512)
    SK(2, 4);  SK(4, 4);       // reduced test geometry (D = 128)
#undef SK
    return hipErrorInvalidValue;
}

int lora_wgrad_chunks(int M) { return (M + WG_CH - 1) / WG_CH; }

hipError_t launch_lora_wgrad(const op_t* x1ext, int ldx, const op_t* dqkv, int ldd, int M, int D, int r, float* partial,
                             float* gAq, float* gBq, float* gAv, float* gBv, hipStream_t s, const float* scaler_f, int* scaler_i) {
    if (D % WG_BN) return hipErrorInvalidValue;
    const int nch = lora_wgrad_chunks(M);
    dim3 grid(nch, D / WG_BN, 4);
    if (r == 16) {
        constexpr int SMEM = WG_CH * WG_BN * 2 + WG_CH * 16 * 2;
        static std::atomic<bool> done{false};
        if (!done.load()) { hipError_t e = hipFuncSetAttribute((const void*)wgrad_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); if (e != hipSuccess) return e; done.store(true); }
        hipLaunchKernelGGL((wgrad_kernel<16>), grid, dim3(256), SMEM, s, x1ext, ldx, dqkv, ldd, M, D, partial, nch);
    } else if (r == 32) {
        constexpr int SMEM = WG_CH * WG_BN * 2 + WG_CH * 32 * 2;
        static std::atomic<bool> done{false};
        if (!done.load()) { hipError_t e = hipFuncSetAttribute((const void*)wgrad_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); if (e != hipSuccess) return e; done.store(true); }
        hipLaunchKernelGGL((wgrad_kernel<32>), grid, dim3(256), SMEM, s, x1ext, ldx, dqkv, ldd, M, D, partial, nch);
    } else return hipErrorInvalidValue;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((r * D + 63) / 64, 4), dim3(256), 0, s, partial, nch, D, r, gAq, gBq, gAv, gBv, scaler_f, scaler_i);
    return hipGetLastError();
}
