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

// bf16 images of one layer's adapters.  q/k/v adapters (slot k = position among the enabled ones): B_t -> columns D + k*r of rows
// t*D.. of wqkv_ext (forward K-extension) and row block k of btcat (= B_t^T); A_t -> row block k of acat and columns 3D + k*r of
// wqkvT_ext (dgrad K-extension).  out_proj adapter: B_o -> columns D.. of wo_ext, A_o^T -> columns D.. of woT_ext, plus
// acat_o = A_o and btcat_o = B_o^T for the skinny products.
__global__ void refresh_kernel(LoraPtrs P, int3 slot /* of q, k, v; -1 = none */, int D, int r, op_t* __restrict__ wext, int ldw,
                               op_t* __restrict__ wtext, int ldwt, op_t* __restrict__ acat, op_t* __restrict__ btcat,
                               op_t* __restrict__ woext, op_t* __restrict__ wotext, int ldwo, op_t* __restrict__ acat_o,
                               op_t* __restrict__ btcat_o) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D * r) return;
    const int nB = i / r, jB = i - nB * r;      // B side: i = n*r + j
    const int jA = i / D, dA = i - jA * D;      // A side: i = j*D + d
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int k = t == 0 ? slot.x : t == 1 ? slot.y : slot.z;
        if (k < 0) continue;
        const op_t b = f32_to_op(P.B[t][i]), a = f32_to_op(P.A[t][i]);
        wext[(size_t)(t * D + nB) * ldw + D + k * r + jB] = b;
        btcat[(size_t)(k * r + jB) * D + nB] = b;
        acat[(size_t)(k * r + jA) * D + dA] = a;
        wtext[(size_t)dA * ldwt + 3 * D + k * r + jA] = a;
    }
    if (P.A[3]) {
        const op_t b = f32_to_op(P.B[3][i]), a = f32_to_op(P.A[3][i]);
        woext[(size_t)nB * ldwo + D + jB] = b;
        btcat_o[(size_t)jB * D + nB] = b;
        acat_o[(size_t)jA * D + dA] = a;
        wotext[(size_t)dA * ldwo + D + jA] = a;
    }
}

#ifdef TTL_OPERAND_FP32
// ---- strict (fp32) build: the same product in plain fp32 FMAs.  One wave per SK32_ROWS rows of X; lane l owns elements
// d = l, l + 64, ... of every row (held in registers), walks the output columns and reduces across the wave.
constexpr int SK32_ROWS = 4, SK32_MAXC = 16;     // D <= 1024
__global__ __launch_bounds__(256) void skinny_f32_kernel(const op_t* __restrict__ X, long long ldx, int3 xoff, int r,
                                                          const op_t* __restrict__ Wcat, int D, float scale,
                                                          op_t* __restrict__ out, long long ldo, int M, const int* __restrict__ rowmap) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.y;
    const int xo = t == 0 ? xoff.x : t == 1 ? xoff.y : xoff.z;
    const int m0 = (blockIdx.x * 4 + wave) * SK32_ROWS;
    if (m0 >= M) return;
    const int nc = D >> 6;
    long long prow[SK32_ROWS];
    float x[SK32_ROWS][SK32_MAXC];
#pragma unroll
    for (int i = 0; i < SK32_ROWS; ++i) {
        const int row = min(m0 + i, M - 1);
        prow[i] = rowmap ? rowmap[row] : row;
#pragma unroll
        for (int c = 0; c < SK32_MAXC; ++c) x[i][c] = c < nc ? X[prow[i] * ldx + xo + lane + 64 * c] : 0.f;
    }
    for (int col = 0; col < r; ++col) {
        const op_t* w = Wcat + (size_t)(t * r + col) * D;
        float acc[SK32_ROWS] = {};
#pragma unroll
        for (int c = 0; c < SK32_MAXC; ++c) {
            const float wv = c < nc ? w[lane + 64 * c] : 0.f;
#pragma unroll
            for (int i = 0; i < SK32_ROWS; ++i) acc[i] = fmaf(x[i][c], wv, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < SK32_ROWS; ++i) {
            const float v = wave_sum(acc[i]);
            if (lane == 0 && m0 + i < M) out[prow[i] * ldo + t * r + col] = v * scale;
        }
    }
}
#else
// One wave: 16 rows of X times the r columns of ONE adapter (blockIdx.y = slot of the adapter).  MFMA A operand = Wcat rows
// (output row = column c of the result), B operand = X rows (output column = token) -> each lane ends with 4 consecutive
// result columns of one token: one 8-byte store.  The kernel is pure latency (a few hundred waves, each a serial K loop), so
// ALL of a wave's X fragments are requested up front (KS x 16 B per lane in flight) and the adapter's rows of Wcat are staged
// once per workgroup in LDS.  rowmap (optional): logical row m lives at physical row rowmap[m] of X and out.
// TTL_SKINNY_WPB: waves (16-row groups) per workgroup.  64 views are 788 groups: 4 per workgroup leave 59 of the 256 CUs without any
// (197 workgroups); fewer per workgroup spread the X reads over every CU at the price of staging Wcat more often (from L2).
#ifndef TTL_SKINNY_WPB
#define TTL_SKINNY_WPB 4
#endif
constexpr int SK_WPB = TTL_SKINNY_WPB;
template <int NCG, int KS>   // NCG = r / 16 column groups, KS = D / 32 k-steps
__global__ __launch_bounds__(64 * SK_WPB) void skinny_kernel(const op_t* __restrict__ X, long long ldx, int3 xoff,
                                                     const op_t* __restrict__ Wcat, int D, float scale,
                                                     op_t* __restrict__ out, long long ldo, int M, const int* __restrict__ rowmap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // W [16*NCG][D] operand type, rows padded by 16 B
    const int tid = threadIdx.x, lane = tid & 63;
    const int t = blockIdx.y;
    const int xo = t == 0 ? xoff.x : t == 1 ? xoff.y : xoff.z;
    const op_t* W = Wcat + (size_t)t * 16 * NCG * D;
    const int rowb = D * 2 + 16;
    const int m0 = (blockIdx.x * SK_WPB + (tid >> 6)) * 16;
    const int li = lane & 15, lg = lane >> 4;
    const int row = min(m0 + li, M - 1);
    const long long prow = rowmap ? rowmap[row] : row;
    const op_t* x = X + prow * ldx + xo + 8 * lg;
    // the wave's X fragments are requested FIRST (KS x 16 B per lane), the adapter rows behind them in batches of unrolled loads:
    // a `load; store to LDS` loop makes hipcc wait vmcnt(0) in every iteration — 12 serial L2 round trips, which WAS this kernel's
    // duration (9.6 us at 64 views for 19 MB of X) while the X loads had not even been issued
    opx8 f[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k) f[k] = *(const opx8*)(x + 32 * k);
    {
        constexpr int CPR = 4 * KS;                                  // 16-B chunks per row of Wcat (D = 32 KS)
        constexpr int TOT = 16 * NCG * CPR, NT = 64 * SK_WPB;
        constexpr int PER = (TOT + NT - 1) / NT;                     // chunks per thread
        constexpr int BATCH = PER < 12 ? PER : 12;
#pragma unroll
        for (int b0 = 0; b0 < PER; b0 += BATCH) {
            u32x4 v[BATCH];
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int q = min(tid + (b0 + j) * NT, TOT - 1);     // (clamped: the guard sits on the store, the load stays branch-free)
                const int r = q / CPR, c = q - r * CPR;
                v[j] = *(const u32x4*)(W + (size_t)r * D + c * 8);
            }
#pragma unroll
            for (int j = 0; j < BATCH; ++j) {
                const int q = tid + (b0 + j) * NT;
                const int r = q / CPR, c = q - r * CPR;
                if (b0 + j < PER && q < TOT) *(u32x4*)(smem + r * rowb + c * 16) = v[j];
            }
        }
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
            acc[c] = MFMA16(w, f[k], acc[c], 0, 0, 0);
        }
    }
    if (m0 < M && m0 + li < M) {
#pragma unroll
        for (int c = 0; c < NCG; ++c)
            *(u32x2*)(out + prow * ldo + (t * NCG + c) * 16 + 4 * lg) =
                u32x2{pack_op2(acc[c][0] * scale, acc[c][1] * scale), pack_op2(acc[c][2] * scale, acc[c][3] * scale)};
    }
}

#endif  // TTL_OPERAND_FP32 (skinny)

// ---- weight gradients: partial[prod][chunk][r][D] = S_chunk^T · G_chunk over 256-token chunks
constexpr int WG_CH = 256;   // tokens per chunk
constexpr int WG_BN = 128;   // result columns per block

#ifdef TTL_OPERAND_FP32
// ---- strict (fp32) build: the same chunked product in fp32 FMAs.  Block = one 256-token chunk x 128 result columns; the S chunk
// [256][R] sits in LDS, thread t owns column n0 + t % 128 and the rows j = (t / 128) * R/2 .. of the result.
template <int R>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradList L, int M, int D, float* __restrict__ partial, int nch, int prod0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sS = (float*)smem;      // [256][R]
    const int tid = threadIdx.x;
    const int ch = blockIdx.x, cb = blockIdx.y, prod = prod0 + blockIdx.z;
    const int m0 = ch * WG_CH, n0 = cb * WG_BN;
    const op_t* S = L.p[prod].S; const long long lds_ = L.p[prod].lds;
    const op_t* G = L.p[prod].G; const long long ldg = L.p[prod].ldg;
    for (int q = tid; q < WG_CH * R; q += 256) {
        const int rr = q / R, c = q - rr * R;
        sS[q] = (m0 + rr < M) ? S[(long long)(m0 + rr) * lds_ + c] : 0.f;
    }
    __syncthreads();
    const int col = n0 + (tid & 127), j0 = (tid >> 7) * (R / 2);
    float acc[R / 2] = {};
    const int rows = min(WG_CH, M - m0);
    for (int m = 0; m < rows; ++m) {
        const float g = G[(long long)(m0 + m) * ldg + col];
#pragma unroll
        for (int j = 0; j < R / 2; ++j) acc[j] = fmaf(sS[m * R + j0 + j], g, acc[j]);
    }
    float* po = partial + L.p[prod].poff + (size_t)ch * R * D;
#pragma unroll
    for (int j = 0; j < R / 2; ++j) po[(size_t)(j0 + j) * D + col] = acc[j];
}
#else
__device__ __forceinline__ int wg_u(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
// Byte offset of row `row`, byte `b` of the S chunk image [256][R] (rows of 2 R bytes).  A fragment read (ds_read_tr16_b64) touches, per
// half-wave, 8 bytes x 4 in each of the rows {0-3, 8-11} (+4 for the second read, +16 for the other half-wave) of a 32-row group: with
// plain rows, rows r and r + 8 lie 256 B (R = 16) or 512 B (R = 32) apart — the same 64 banks, a 2-way conflict on every S read
// (rocprofv3 SQ_LDS_BANK_CONFLICT: 1.5e5 per launch, profiles/r05_pmc_summary_fp16.txt).  Rows with bit 3 set therefore swap the halves
// of their 256-B window (R = 16: address bit 7) / of their own 64-B row (R = 32: bit 5); the staging stores apply the same map.
template <int R>
__device__ __forceinline__ int wg_s_off(int row, int b) { return (row * (R * 2) + b) ^ (((row >> 3) & 1) << (R == 16 ? 7 : 5)); }

template <int R>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradList L, int M, int D, float* __restrict__ partial, int nch, int prod0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sG = smem;                       // [256][128] bf16, 256-B rows, 16-B chunk c at c ^ (u(row) << 1)
    char* sS = smem + WG_CH * WG_BN * 2;   // [256][R] bf16, plain
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ch = blockIdx.x, cb = blockIdx.y, prod = prod0 + blockIdx.z;
    const int m0 = ch * WG_CH, n0 = cb * WG_BN;
    // operands of this product: result [R][D] = S^T · G over the M rows
    const op_t* S = L.p[prod].S; const long long lds_ = L.p[prod].lds;
    const op_t* G = L.p[prod].G; const long long ldg = L.p[prod].ldg;
    // stage G tile: 256 rows x 16 chunks of 16 B
    // all of a thread's 16 + R/8 tile chunks are requested before the first one is written to LDS (rows beyond M: clamped address,
    // zero selected afterwards): the `load; store` loop this replaces waited vmcnt(0) per iteration — 16 serial HBM round trips,
    // ~24 of the launch's 29 us at 64 views
    {
        u32x4 gv[16], sv[R / 8];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = tid + 256 * j, r = q >> 4, c = q & 15;
            gv[j] = *(const u32x4*)(G + (long long)min(m0 + r, M - 1) * ldg + n0 + c * 8);
        }
#pragma unroll
        for (int j = 0; j < R / 8; ++j) {
            const int q = tid + 256 * j, r = q / (R / 8), c = q - r * (R / 8);
            sv[j] = *(const u32x4*)(S + (long long)min(m0 + r, M - 1) * lds_ + c * 8);
        }
        const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q = tid + 256 * j, r = q >> 4, c = q & 15;
            *(u32x4*)(sG + r * 256 + ((c ^ (wg_u(r) << 1)) << 4)) = (m0 + r < M) ? gv[j] : zero;
        }
#pragma unroll
        for (int j = 0; j < R / 8; ++j) {
            const int q = tid + 256 * j, r = q / (R / 8), c = q - r * (R / 8);
            *(u32x4*)(sS + wg_s_off<R>(r, c << 4)) = (m0 + r < M) ? sv[j] : zero;
        }
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
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sS + wg_s_off<R>(rlo, (a * 16 + 4 * tp) * 2)));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(sS + wg_s_off<R>(rhi, (a * 16 + 4 * tp) * 2)));
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
    float* po = partial + L.p[prod].poff + (size_t)ch * R * D;
#pragma unroll
    for (int a = 0; a < R / 16; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                po[(size_t)(a * 16 + 4 * lg + e) * D + n0 + 32 * wave + 16 * b + li] = acc[a][b][e];
}

#endif  // TTL_OPERAND_FP32 (wgrad)

// 64 outputs x 4 chunk-slices per block: slice q sums chunks q, q+4, ... in order, the four slice sums are added in a
// fixed order -> deterministic; one thread walking all ~50 chunks serially left the 9.8 MB read latency-bound (16 us)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nch, int D, int r, const WgradList L,
                                                           const float* __restrict__ scaler_f, int* __restrict__ scaler_i) {
    __shared__ float part[4][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    const int prod = blockIdx.y;
    const int rows = L.p[prod].rows;        // r, or 2r for two products that share their G operand (rows r.. belong to out2)
    float a0 = 0.f, a1 = 0.f;
    if (i < rows * D) {
        const float* p = partial + L.p[prod].poff + i;
        const size_t cs = (size_t)rows * D;
        int c = slice;
        // (four iterations of loads in flight at a time; the additions keep their order: a0 takes chunks c, c + 8, ..., a1 c + 4, c + 12, ...)
        for (; c + 28 < nch; c += 32) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = p[(size_t)(c + 4 * j) * cs];
#pragma unroll
            for (int j = 0; j < 8; j += 2) { a0 += t[j]; a1 += t[j + 1]; }
        }
        for (; c + 4 < nch; c += 8) { a0 += p[(size_t)c * cs]; a1 += p[(size_t)(c + 4) * cs]; }
        for (; c < nch; c += 4) a0 += p[(size_t)c * cs];
    }
    part[slice][threadIdx.x & 63] = a0 + a1;
    __syncthreads();
    if (slice != 0 || i >= rows * D) return;
    float s = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    // scaler.unscale_: undo the backward's loss scale; any inf/nan makes the optimizer skip the WHOLE step (deyo.py:186-188)
    if (scaler_f) s *= scaler_f[1];
    if (scaler_i && !isfinite(s)) atomicOr(scaler_i, 1);
    int j = i / D;
    const int d = i - j * D;
    float* out = L.p[prod].out;
    if (j >= r) { j -= r; out = L.p[prod].out2; }
    if (L.p[prod].transpose) out[(size_t)d * r + j] = s;   // dB [D][r]
    else out[(size_t)j * D + d] = s;                        // dA [r][D]
}

}  // namespace

hipError_t launch_lora_refresh(const LoraPtrs& P, int D, int r, op_t* wqkv_ext, int ldw, op_t* wqkvT_ext, int ldwt, op_t* a_cat,
                               op_t* bT_cat, op_t* wo_ext, op_t* woT_ext, int ldwo, op_t* acat_o, op_t* btcat_o, hipStream_t s) {
    int3 slot; int k = 0;
    slot.x = P.A[0] ? k++ : -1; slot.y = P.A[1] ? k++ : -1; slot.z = P.A[2] ? k++ : -1;
    hipLaunchKernelGGL(refresh_kernel, dim3((D * r + 255) / 256), dim3(256), 0, s, P, slot, D, r, wqkv_ext, ldw, wqkvT_ext, ldwt, a_cat,
                       bT_cat, wo_ext, woT_ext, ldwo, acat_o, btcat_o);
    return hipGetLastError();
}

#ifndef TTL_OPERAND_FP32
template <int NCG, int KS>
static hipError_t skinny_launch(const op_t* X, long long ldx, int3 xoff, int ntg, const op_t* W, int D, float scale, op_t* out,
                                long long ldo, int M, hipStream_t s, const int* rowmap) {
    const int smem = 16 * NCG * (D * 2 + 16);
    static std::atomic<uint64_t> done{0};
    if (hipError_t e = ensure_smem((const void*)skinny_kernel<NCG, KS>, smem, done); e != hipSuccess) return e;
    hipLaunchKernelGGL((skinny_kernel<NCG, KS>), dim3((M + 16 * SK_WPB - 1) / (16 * SK_WPB), ntg), dim3(64 * SK_WPB), smem, s, X, ldx, xoff, W, D, scale,
                       out, ldo, M, rowmap);
    return hipGetLastError();
}

#endif

hipError_t launch_lora_skinny(const op_t* X, long long ldx, const int* xoff, int ntg, const op_t* Wcat, int D, int r, float scale,
                              op_t* out, long long ldo, int M, hipStream_t s, const int* rowmap) {
    if (ntg < 1 || ntg > 3) return hipErrorInvalidValue;
#ifdef TTL_DIAG_SKIP       // timing-only ablation of the episode (tools/class_cost_ab.sh): bit 0 = no big skinny products, bit 1 = no wgrad
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 1) && M >= 4096 && diag_skip_now(cnt, 360)) return hipSuccess; }
#endif
    int3 xo = {xoff[0], ntg > 1 ? xoff[1] : 0, ntg > 2 ? xoff[2] : 0};
#ifdef TTL_OPERAND_FP32
    if (D % 64 || D > 64 * SK32_MAXC) return hipErrorInvalidValue;
    hipLaunchKernelGGL(skinny_f32_kernel, dim3((M + 4 * SK32_ROWS - 1) / (4 * SK32_ROWS), ntg), dim3(256), 0, s, X, ldx, xo, r, Wcat, D, scale,
                       out, ldo, M, rowmap);
    return hipGetLastError();
#else
    int ncg = r / 16;
    const int ks = D / 32;
    // Adapters that read the SAME columns of X (the forward's down-projections U_t = s x A_t^T of q / k / v: every xoff 0) are one
    // product with ntg * r output columns: X is read once instead of once per adapter (19.4 MB per adapter at 64 views), the
    // stacked rows of Wcat already lie behind each other, and the output columns k*r + c come out in place.
    if (ntg > 1 && xo.y == xo.x && (ntg < 3 || xo.z == xo.x) && 16 * ntg * ncg * (D * 2 + 16) <= 150 * 1024) {
        ncg *= ntg;
        ntg = 1;
    }
#define SK(N_, K_) if (ncg == N_ && ks == K_ && D % 32 == 0) return skinny_launch<N_, K_>(X, ldx, xo, ntg, Wcat, D, scale, out, ldo, M, s, rowmap)
    SK(1, 24); SK(2, 24); SK(3, 24); SK(4, 24); SK(6, 24);      // ViT-B/16, r = 16 / 32 (x 1-3 adapters on the same X)
    SK(1, 32); SK(2, 32); SK(3, 32); SK(4, 32);                 // ViT-L/14 (6 x 16 rows of D = 1024 do not fit the LDS: two passes)
    SK(1, 16); SK(2, 16); SK(3, 16); SK(4, 16); SK(6, 16);      // text tower of ViT-B/16 (D = 512)
    SK(1, 4);  SK(2, 4); SK(3, 4); SK(4, 4); SK(6, 4);          // reduced test geometry (D = 128)
#undef SK
    return hipErrorInvalidValue;
#endif
}

int lora_wgrad_chunks(int M) { return (M + WG_CH - 1) / WG_CH; }
// dynamic LDS of wgrad_kernel<R>: the G tile + the S tile in the operand type (16-bit builds); the S chunk in fp32 (strict build)
static constexpr int wgrad_smem(int R) { return sizeof(op_t) == 4 ? WG_CH * R * 4 : WG_CH * WG_BN * 2 + WG_CH * R * 2; }

hipError_t launch_lora_wgrad(const WgradList& L0, int M, int D, int r, float* partial, hipStream_t s, const float* scaler_f, int* scaler_i) {
    if (D % WG_BN || L0.n < 1 || L0.n > WGRAD_MAX) return hipErrorInvalidValue;
#ifdef TTL_DIAG_SKIP
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 2) && M >= 4096 && diag_skip_now(cnt, 120)) return hipSuccess; }
#endif
    const int nch = lora_wgrad_chunks(M);
    // Two products over the SAME G rows whose S columns lie side by side (dA_q = dU_q^T x1 and dA_v = dU_v^T x1: G = x1, the dU
    // columns adjacent in the K-extension of dqkv) can run as ONE product with 2r result rows, so that G — 19.4 MB at 64 views — is
    // read once (TTL_WGRAD_MERGE=1; rank 16 only).  Measured in situ (round 4, tools/_diag/r04m.sh): lora class 0.218 -> 0.225 ms per
    // episode, episode time unchanged — the 32-row tile halves the resident blocks per CU and the kernel is bound by its
    // load -> barrier -> multiply round trips, not by the bytes: OFF by default.  Merged products first, so each kernel
    // instantiation covers a contiguous range of the list.
    WgradList L = {};
    bool used[WGRAD_MAX] = {};
    static const int merge = TTL_EXPERIMENT("TTL_WGRAD_MERGE", 0);
    int nm = 0;
    if (r == 16 && merge)
        for (int i = 0; i < L0.n; ++i) {
            if (used[i]) continue;
            for (int j = i + 1; j < L0.n; ++j) {
                if (used[j] || L0.p[j].G != L0.p[i].G || L0.p[j].ldg != L0.p[i].ldg || L0.p[j].lds != L0.p[i].lds ||
                    L0.p[j].transpose != L0.p[i].transpose)
                    continue;
                const bool ij = L0.p[j].S == L0.p[i].S + r, ji = L0.p[i].S == L0.p[j].S + r;
                if (!ij && !ji) continue;
                WgradProd m = ij ? L0.p[i] : L0.p[j];
                m.out2 = ij ? L0.p[j].out : L0.p[i].out;
                m.rows = 2 * r;
                L.p[L.n++] = m;
                used[i] = used[j] = true;
                ++nm;
                break;
            }
        }
    for (int i = 0; i < L0.n; ++i)
        if (!used[i]) { L.p[L.n] = L0.p[i]; L.p[L.n].out2 = nullptr; L.p[L.n].rows = r; ++L.n; }
    size_t off = 0;
    for (int i = 0; i < L.n; ++i) { L.p[i].poff = off; off += (size_t)nch * L.p[i].rows * D; }
    auto launch = [&](auto kern, int R, int prod0, int count) -> hipError_t {
        if (count <= 0) return hipSuccess;
        const int SMEM = wgrad_smem(R);
        hipLaunchKernelGGL(kern, dim3(nch, D / WG_BN, count), dim3(256), SMEM, s, L, M, D, partial, nch, prod0);
        return hipGetLastError();
    };
    static std::atomic<uint64_t> done16{0}, done32{0};
    hipError_t e = ensure_smem((const void*)wgrad_kernel<16>, wgrad_smem(16), done16);
    if (e == hipSuccess) e = ensure_smem((const void*)wgrad_kernel<32>, wgrad_smem(32), done32);
    if (e != hipSuccess) return e;
    if (r == 16) {
        if ((e = launch(wgrad_kernel<32>, 32, 0, nm)) != hipSuccess) return e;
        if ((e = launch(wgrad_kernel<16>, 16, nm, L.n - nm)) != hipSuccess) return e;
    } else if (r == 32) {
        if ((e = launch(wgrad_kernel<32>, 32, 0, L.n)) != hipSuccess) return e;
    } else return hipErrorInvalidValue;
    int maxrows = r;      // result rows of the widest product (2r only when two products were merged)
    for (int i = 0; i < L.n; ++i) maxrows = L.p[i].rows > maxrows ? L.p[i].rows : maxrows;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((maxrows * D + 63) / 64, L.n), dim3(256), 0, s, partial, nch, D, r, L, scaler_f, scaler_i);
    return hipGetLastError();
}
