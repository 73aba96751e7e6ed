// Row-panel GEMM with LayerNorm as its epilogue:   h[M, 768] = resid + A[M, K] · B[768, K]^T + bias   (fp32 residual stream)
//                                                  y[M, 768] = LayerNorm(h) * gamma + beta          (operand type: the next GEMM's A)
// for the N = D = 768 projections of the image tower (out_proj -> LayerNorm 2, fc2 -> LayerNorm 1 of the next layer;
// HF modeling_clip.py:358-383 via clip/custom_clip.py:62-71 of the reference): one launch instead of GEMM + LayerNorm.
//
// Why a third big-M kernel (profiles/r06_experiments.txt r06a, r06e): the LayerNorm-forward launches are HBM-bound byte movement
// (58 MB, 10.5 us each, 23 per episode, 99 % exposed with three episodes in flight: 5.9 % of the step), and a 256-column tile cannot
// normalise a 768-column row.  Here a workgroup owns WHOLE rows: tile 128 x 768 x 32 on four waves, wave w = rows 32 w .. 32 w + 31 of
// the tile x all 768 columns = 24 MFMA 32x32x16 tiles = 384 accumulator registers (one wave per SIMD, 512-register budget).  A row's
// 768 values live in the 32 lanes of one half-wave, so mean / rstd are lane-local sums + 5 cross-lane steps; no statistics pass, no
// second launch, and the fp32 stream is written once and never re-read by a LayerNorm kernel.
//
//   * operands by LDS-DMA only (buffer_load ... lds, 1-KiB pieces = 16 image rows of 64 B; rows past M read as zeros through the
//     buffer range check): B (weights, L2-resident: 1.2 MB) ring of TWO 48-KiB stages, K-tile kt+1 requested during step kt;
//     A (activations, first touch from HBM) ring of THREE 8-KiB stages, K-tile kt+2 requested during step kt   (2 x 48 + 3 x 8 = 120 KiB)
//   * LDS image: 64-B rows (32 k), 16-B chunk c of row r at position c ^ ((r >> 2) & 3) (swizzle on the SOURCE address of the DMA):
//     a 16-lane group of a ds_read_b128 fragment read touches 16 distinct 16-B slots of the 256-B bank window
//   * weight image rows permuted: image row 32 j + c holds output column 24 c + j, so lane c owns 24 ADJACENT columns of every
//     row it holds (96-B fp32 / 48-B operand-type runs in the epilogue)
//   * B fragments are streamed (one ds_read_b128 per MFMA: 25 reads per k16 and wave = 400 LDS cycles of the 768 MFMA cycles)
//   * epilogue, three passes over the accumulators: (1) + residual (bias was the accumulators' initial value), store h, row sums;
//     (2) sum of squared deviations (two-pass variance, like ln_fwd_persist_kernel); (3) normalise, gamma / beta, store y
//     (and mean / rstd when the backward will need them)
#include <stdlib.h>

#include <atomic>

#include "kernels.hpp"

#ifndef TTL_OPERAND_FP32

namespace {

constexpr int RBM = 128, RBN = 768, RBK = 32, RNT = 256;
constexpr int RA_STAGE = RBM * 64;        // 8 KiB
constexpr int RB_STAGE = RBN * 64;        // 48 KiB
constexpr int RA_RING = 3, RB_RING = 2;
constexpr int ROW_SMEM = RA_RING * RA_STAGE + RB_RING * RB_STAGE;      // 120 KiB

template <int N>
__device__ __forceinline__ void row_wait_vm() {     // s_waitcnt vmcnt(N) lgkmcnt(0)
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

// sum over the 32 lanes that share lane >> 5 (xor offsets 1 .. 16 stay inside a half-wave)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct RowArgs {
    const op_t* A; int lda;
    const op_t* B; int ldb;          // [768][ldb]
    int M, K;
    const float* bias;               // [768] or null
    const float* resid; int ldr;     // [M][ldr] or null
    float* C; int ldc;               // fp32 stream out
    const float* gamma; const float* beta; float eps;
    op_t* Y; int ldy;                // LayerNorm(C) in the operand type
    float* mean; float* rstd;        // [M] or null
};

__global__ __launch_bounds__(RNT) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_row_ln_kernel(const RowArgs a, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, lh = lane >> 5;
    const int M = a.M, nk = a.K / RBK;
    char* const A0 = smem;
    char* const B0 = smem + RA_RING * RA_STAGE;
    // ---- fragment addresses: MFMA 32x32x16 operand = image row l32 of the slab, 16-B chunk 2 s + lh of k16 sub-step s
    const int swz = ((lh ^ ((l32 >> 2) & 3)) << 4);                 // chunk (2 s + lh) ^ ((row >> 2) & 3): s toggles byte bit 5
    const int fA0 = (wave * 32 + l32) * 64 + swz;
    const int fB0 = l32 * 64 + swz;
    // ---- DMA: piece p = 16 image rows; lane (r16, pos) fills chunk position pos of image row 16 p + r16 with global chunk pos ^ swizzle(row)
    const int r16 = lane >> 2, pos = lane & 3;
    const int cs = pos ^ ((r16 >> 2) & 3);
    constexpr int RSRC = 0x00020000;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)(((size_t)(M - 1) * a.lda + a.K) * sizeof(op_t)), RSRC);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, (int)(((size_t)(RBN - 1) * a.ldb + a.K) * sizeof(op_t)), RSRC);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)a.resid, 0, a.resid ? (int)((size_t)M * a.ldr * 4) : 0, RSRC);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)a.C, 0, (int)((size_t)M * a.ldc * 4), RSRC);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)a.Y, 0, (int)((size_t)M * a.ldy * sizeof(op_t)), RSRC);
    const int voA = (int)((r16 * a.lda + cs * 8) * sizeof(op_t));
    const int voB = (int)((24 * r16 * a.ldb + cs * 8) * sizeof(op_t));     // image row 16 p + r16 holds column 24 (16 (p & 1) + r16) + (p >> 1)
    // A stage: 8 pieces, wave w carries pieces w and w + 4; B stage: 48 pieces, wave w carries pieces w, w + 4, ..., w + 44
    auto dma_a = [&](char* stage, int row0, int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = wave + 4 * i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(stage + p * 1024), 16, voA,
                                                     (int)(((size_t)(row0 + 16 * p) * a.lda + (size_t)kt * RBK) * sizeof(op_t)), 0, 0);
        }
    };
    auto dma_b_piece = [&](char* stage, int kt, int i) {       // i = 0 .. 11
        const int p = wave + 4 * i;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(stage + p * 1024), 16, voB,
                                                 (int)(((size_t)(384 * (p & 1) + (p >> 1)) * a.ldb + (size_t)kt * RBK) * sizeof(op_t)), 0, 0);
    };

    f32x16 acc[24];
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * RBM;
        // ---- prologue: B(0), A(0), A(1)
#pragma unroll
        for (int i = 0; i < 12; ++i) dma_b_piece(B0, 0, i);
        dma_a(A0, row0, 0);
        if (nk > 1) dma_a(A0 + RA_STAGE, row0, 1);
        {   // bias = the accumulators' initial value (lane's columns 24 l32 + j)
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                const float bv = a.bias ? a.bias[24 * l32 + j] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = bv;
            }
        }
        char *aC = A0, *aN = A0 + RA_STAGE, *aNN = A0 + 2 * RA_STAGE, *bC = B0, *bN = B0 + RB_STAGE;
        for (int kt = 0; kt < nk; ++kt) {
            // K-tile kt of A and B has landed when at most the 2 pieces of A(kt+1) are still in flight (issue order: B(kt+1) then A(kt+2))
            if (kt + 1 < nk) row_wait_vm<2>(); else row_wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            const bool hb = kt + 1 < nk, ha = kt + 2 < nk;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const opx8 xa = *(const opx8*)(aC + (fA0 ^ (s << 5)));
                const char* pb = bC + (fB0 ^ (s << 5));
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    opx8 wf[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) wf[q] = *(const opx8*)(pb + (4 * g + q) * 2048);
                    if (hb) dma_b_piece(bN, kt + 1, 6 * s + g);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[4 * g + q] = MFMA32(xa, wf[q], acc[4 * g + q], 0, 0, 0);
                }
            }
            if (ha) dma_a(aNN, row0, kt + 2);
            { char* t = aC; aC = aN; aN = aNN; aNN = t; t = bC; bC = bN; bN = t; }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue.  Register r of column tile j: row 32 wave + 8 (r >> 2) + 4 lh + (r & 3), column 24 l32 + j.
        const int rbase = row0 + wave * 32 + 4 * lh;
        const int voC = (int)((4 * lh * 0 + 24 * l32) * 4);       // per-lane column offset (bytes) in an fp32 row
        float s1[16];
        // pass 1: + residual, store the fp32 stream, row sums
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = rbase + 8 * (r >> 2) + (r & 3);
            float sum = 0.f;
            if (a.resid) {
                u32x4 rv[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) rv[c] = __builtin_amdgcn_raw_buffer_load_b128(rsR, voC + 16 * c, (int)((size_t)m * a.ldr * 4), 0);
#pragma unroll
                for (int c = 0; c < 6; ++c) {
                    const f32x4 f = __builtin_bit_cast(f32x4, rv[c]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[4 * c + e][r] += f[e];
                }
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const f32x4 o = {acc[4 * c][r], acc[4 * c + 1][r], acc[4 * c + 2][r], acc[4 * c + 3][r]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, voC + 16 * c, (int)((size_t)m * a.ldc * 4), 0);
                sum += (o[0] + o[1]) + (o[2] + o[3]);
            }
            s1[r] = sum;
        }
        float mu[16], rs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) mu[r] = half_sum(s1[r]) * (1.0f / RBN);
        // pass 2: two-pass variance
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 24; ++j) { const float d = acc[j][r] - mu[r]; q += d * d; }
            rs[r] = q;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) rs[r] = rsqrtf(half_sum(rs[r]) * (1.0f / RBN) + a.eps);
        if (a.mean && l32 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rbase + 8 * (r >> 2) + (r & 3);
                if (m < M) { a.mean[m] = mu[r]; a.rstd[m] = rs[r]; }
            }
        }
        // pass 3: normalise in chunks of 8 columns (one 16-B store of 8 operand values per row and chunk)
        const int voY = (int)(24 * l32 * sizeof(op_t));
#pragma unroll
        for (int jc = 0; jc < 3; ++jc) {
            float g[8], b[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { g[e] = a.gamma[24 * l32 + 8 * jc + e]; b[e] = a.beta[24 * l32 + 8 * jc + e]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rbase + 8 * (r >> 2) + (r & 3);
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (acc[8 * jc + e][r] - mu[r]) * rs[r] * g[e] + b[e];
                const u32x4 pk = {pack_op2(o[0], o[1]), pack_op2(o[2], o[3]), pack_op2(o[4], o[5]), pack_op2(o[6], o[7])};
                __builtin_amdgcn_raw_buffer_store_b128(pk, rsY, voY + 16 * jc, (int)((size_t)m * a.ldy * sizeof(op_t)), 0);
            }
        }
        __builtin_amdgcn_s_barrier();      // the next tile's prologue overwrites the stages the last K-step read
    }
}

}  // namespace

hipError_t launch_gemm_row_ln(const GemmArgs& g, const float* gamma, const float* beta, float eps, op_t* Y, int ldy, float* mean, float* rstd,
                              hipStream_t s) {
    if (g.N != RBN || g.K % RBK || g.K < 2 * RBK || g.M < 1 || (g.lda & 7) || (g.ldb & 7) || (ldy & 7) || (g.ldc & 3) || (g.resid && (g.ldr & 3)))
        return hipErrorInvalidValue;
    const size_t lim = (size_t)1 << 31;
    if ((size_t)g.M * g.lda * sizeof(op_t) >= lim || (size_t)g.M * g.ldc * 4 >= lim || (size_t)g.M * ldy * sizeof(op_t) >= lim ||
        (g.resid && (size_t)g.M * g.ldr * 4 >= lim))
        return hipErrorInvalidValue;
    static std::atomic<uint64_t> done{0};
    hipError_t e = ensure_smem((const void*)gemm_row_ln_kernel, ROW_SMEM, done);
    if (e != hipSuccess) return e;
    RowArgs a;
    a.A = g.A; a.lda = g.lda; a.B = g.B; a.ldb = g.ldb; a.M = g.M; a.K = g.K; a.bias = g.bias; a.resid = g.resid; a.ldr = g.ldr;
    a.C = (float*)g.C; a.ldc = g.ldc; a.gamma = gamma; a.beta = beta; a.eps = eps; a.Y = Y; a.ldy = ldy; a.mean = mean; a.rstd = rstd;
    const int ntiles = (g.M + RBM - 1) / RBM;
    const int cus = device_cu_count();
    if (!cus) return hipErrorInvalidDevice;
    hipLaunchKernelGGL(gemm_row_ln_kernel, dim3(ntiles < cus ? ntiles : cus), dim3(RNT), ROW_SMEM, s, a, ntiles);
    return hipGetLastError();
}

#ifdef TTL_ROW_PROBE      // standalone probe library (tools/r06_row_probe.sh): C entry without the rest of libttl_hip
extern "C" __attribute__((visibility("default"))) int ttl_row_probe(const void* A, int lda, const void* B, int ldb, int M, int K, const float* bias,
                                                                    const float* resid, int ldr, float* C, int ldc, const float* gamma, const float* beta,
                                                                    float eps, void* Y, int ldy, float* mean, float* rstd, void* stream) {
    GemmArgs g = {};
    g.A = (const op_t*)A; g.lda = lda; g.B = (const op_t*)B; g.ldb = ldb; g.M = M; g.N = RBN; g.K = K; g.bias = bias; g.resid = resid; g.ldr = ldr;
    g.C = C; g.ldc = ldc;
    return (int)launch_gemm_row_ln(g, gamma, beta, eps, (op_t*)Y, ldy, mean, rstd, (hipStream_t)stream);
}
#endif

#endif  // TTL_OPERAND_FP32
