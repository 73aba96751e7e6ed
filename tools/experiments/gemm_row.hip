// EXPERIMENT (round 6, profiles/r06_experiments.txt r06f) — NOT part of libttl_hip: built only by tools/r06_row_probe.py into a standalone probe
// library.  Numerically right (h to 2e-6, LayerNorm output to one operand rounding, mean / rstd to 5e-7 against fp32 torch), 2x slower than
// the GEMM + LayerNorm launches it would replace; the measured reasons are in r06f.  Kept as the starting point of a next attempt.
//
// Row-panel GEMM with LayerNorm as its epilogue:   h[M, 768] = resid + A[M, K] · B[768, K]^T + bias   (fp32 residual stream)
//                                                  y[M, 768] = LayerNorm(h) * gamma + beta          (operand type: the next GEMM's A)
// for the N = D = 768 projections of the image tower (out_proj -> LayerNorm 2, fc2 -> LayerNorm 1 of the next layer;
// HF modeling_clip.py:358-383 via clip/custom_clip.py:62-71 of the reference): one launch instead of GEMM + LayerNorm.
//
// Why a third big-M kernel (profiles/r06_experiments.txt r06a, r06e): the LayerNorm-forward launches are HBM-bound byte movement
// (58 MB, 10.5 us each, 23 per episode, 99 % exposed with three episodes in flight: 5.9 % of the step), and a workgroup that holds a
// 256-column tile cannot normalise a 768-column row.  Here a workgroup owns WHOLE rows: a panel of 128 rows x 768 columns, computed as two
// column halves of 128 x 384 one after the other (a 128 x 768 accumulator tile is 384 registers per lane on four waves — hipcc spills
// it by the thousand; 128 x 384 is 192, all in AccVGPRs) with the LayerNorm of the whole row behind the second half:
//   half 0: K loop, + residual, store h[:, 0:384], row sums          (A panel read from HBM)
//   half 1: K loop, + residual, store h[:, 384:768], row sums        (A panel read again: 196 KB per workgroup, L2-hot)
//   mean; re-read the panel's first half of h (196 KB, written by these very lanes, L2-hot) into registers; two-pass variance over both
//   halves; normalise, gamma / beta, store y (and mean / rstd when the backward will need them)
// The fp32 stream is written once and never read by a LayerNorm kernel; no statistics pass, no second launch.
//
//   * tile 128 x 384 x 64 on four waves: wave w = rows 32 w .. 32 w + 31 x 384 columns = 12 MFMA 32x32x16 tiles; per k16 sub-step one A
//     fragment + 12 B fragments (13 ds_read_b128 for 12 MFMAs)
//   * operands by LDS-DMA only (buffer_load ... lds, 1-KiB pieces = 8 image rows of 128 B; rows past M read as zeros through the buffer
//     range check): B (weights, L2-resident) ring of TWO 48-KiB stages, K-tile kt+1 requested under the MFMAs of step kt; A ring of
//     THREE 16-KiB stages, K-tile kt+2 requested at the end of step kt   (2 x 48 + 3 x 16 = 144 KiB)
//   * LDS image as in gemm_huge.hip: 128-B rows, 16-B chunk c of row r at position c ^ ((r >> 1) & 7), swizzle applied on the SOURCE
//     address of the DMA; weight rows permuted (image row 32 j + c of a half holds its column 128 (j >> 2) + 4 c + (j & 3)) so that a lane
//     owns three runs of 4 ADJACENT columns and the 32 lanes of one epilogue load / store cover 512 (fp32) / 256 (operand type) contiguous bytes
//   * the residual tile comes through registers UNDER the K loop in four slices of 4 accumulator rows per half (at the end of the tile
//     it would be 38.7 MB per launch read by every workgroup at once with the matrix pipe idle); steps 0 .. 3 of a half are peeled
//   * counted waits: per step a wave issues [B(kt+1) x 12 under the MFMAs][A(kt+2) x 4][residual slice x 12 in the prologue and steps 0 .. 2]; the top
//     of the next step waits until only the youngest 4 (+ 12) are in flight (s_waitcnt vmcnt), one raw s_barrier per K-step
//   * the loop body is phase-shifted behind the barrier ([sub-step 3 of K-tile kt-1][0][1][2 ; wait ; barrier], as in gemm_huge.hip)
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "kernels.hpp"

#ifndef TTL_OPERAND_FP32

#ifndef TTL_ROW_DIAG
#define TTL_ROW_DIAG 0      // timing-only ablations (tools/r06_row_probe.py): 1 no residual stream, 2 no LayerNorm epilogue, 4 no stores of h, 8 no K loop
#endif

namespace {

constexpr int RBM = 128, RBN = 768, RBH = 384, RBK = 64, RNT = 256;
constexpr int RA_STAGE = RBM * 128;       // 16 KiB
constexpr int RB_STAGE = RBH * 128;       // 48 KiB
constexpr int ROW_SMEM = 3 * RA_STAGE + 2 * RB_STAGE;      // 144 KiB

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
    char* const B0 = smem + 3 * RA_STAGE;
    // ---- fragment addresses: MFMA 32x32x16 operand = image row l32 of the slab, 16-B chunk 2 s + lh of k16 sub-step s (s = 0 .. 3)
    const int sw = (l32 >> 1) & 7;
    const int fA0 = (wave * 32 + l32) * 128 + ((lh ^ sw) << 4);
    const int fB0 = l32 * 128 + ((lh ^ sw) << 4);
    // ---- DMA piece p = image rows 8 p .. 8 p + 7; lane (r8, p8) fills chunk position p8 of row 8 p + r8 with global chunk p8 ^ swizzle(row);
    // a wave carries pieces p = wave + 4 i only, so the swizzle ((4 (p & 1) + (r8 >> 1)) & 7) is a per-lane constant
    const int r8 = lane >> 3, p8 = lane & 7;
    const int cs = p8 ^ ((((wave & 1) << 2) + (r8 >> 1)) & 7);
    constexpr int RSRC = 0x00020000;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)(((size_t)(M - 1) * a.lda + a.K) * sizeof(op_t)), RSRC);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, (int)(((size_t)(RBN - 1) * a.ldb + a.K) * sizeof(op_t)), RSRC);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)a.resid, 0, a.resid ? (int)((size_t)M * a.ldr * 4) : 0, RSRC);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)a.C, 0, (int)((size_t)M * a.ldc * 4), RSRC);
    const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc((void*)a.Y, 0, (int)((size_t)M * a.ldy * sizeof(op_t)), RSRC);
    const int voA = (int)((r8 * a.lda + cs * 8) * sizeof(op_t));
    // image row 8 p + r8 = 32 j + c (j = p >> 2, c = 8 (p & 3) + r8) of a half holds its column 128 (j >> 2) + 4 c + (j & 3):
    // global row 384 h + 128 (p >> 4) + 32 (p & 3) + ((p >> 2) & 3) + 4 r8
    const int voB = (int)((4 * r8 * a.ldb + cs * 8) * sizeof(op_t));
    auto dma_a = [&](char* stage, int row0, int kt) {          // 16 pieces: wave w carries w, w + 4, w + 8, w + 12
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(stage + (wave + 4 * i) * 1024), 16, voA,
                                                     (int)(((size_t)(row0 + 8 * (wave + 4 * i)) * a.lda + (size_t)kt * RBK) * sizeof(op_t)), 0, 0);
    };
    auto dma_b_piece = [&](char* stage, int half, int kt, int i) {       // 48 pieces: wave w carries w + 4 i, i = 0 .. 11  (p & 3 = w, p >> 2 = i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(stage + (wave + 4 * i) * 1024), 16, voB,
                                                 (int)(((size_t)(RBH * half + 128 * (i >> 2) + 32 * wave + (i & 3)) * a.ldb + (size_t)kt * RBK) * sizeof(op_t)), 0, 0);
    };

    f32x16 acc[12];
    // per-lane byte offsets of the epilogue's buffer operations: the lane's row inside its group of 8 (4 lh) and its columns of a half
    // (128 c + 4 l32 + (0 .. 3), c = 0 .. 2: + 512 c bytes in fp32).  Everything else of an address is WAVE-UNIFORM and goes into the
    // scalar offset — a per-lane row in the scalar operand makes hipcc wrap every load / store in a waterfall loop
    const int voC = (int)((4 * lh * a.ldc + 4 * l32) * 4);
    const int voR = (int)((4 * lh * a.ldr + 4 * l32) * 4);
    const int voY = (int)((4 * lh * a.ldy + 4 * l32) * sizeof(op_t));
    using T = std::true_type; using F = std::false_type;
#define IC(v) std::integral_constant<int, (v)>{}
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int row0 = tile * RBM;
        const int rowu = row0 + wave * 32;                 // accumulator register r is row rowu + 4 lh + 8 (r >> 2) + (r & 3)
        float s1[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[r] = 0.f;
        for (int half = 0; half < 2; ++half) {
            const int cb = RBH * half * 4;                 // byte offset of the half inside an fp32 row
            // ---- prologue: B(0), A(0), A(1)
#pragma unroll
            for (int i = 0; i < 12; ++i) dma_b_piece(B0, half, 0, i);
            dma_a(A0, row0, 0);
            dma_a(A0 + RA_STAGE, row0, 1);
            {   // bias = the accumulators' initial value (lane's columns 384 half + 12 l32 + j)
#pragma unroll
                for (int j = 0; j < 12; ++j) {
                    const float bv = a.bias ? a.bias[RBH * half + 128 * (j >> 2) + 4 * l32 + (j & 3)] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][r] = bv;
                }
            }
            char *aC = A0, *aN = A0 + RA_STAGE, *aNN = A0 + 2 * RA_STAGE, *bC = B0, *bN = B0 + RB_STAGE;
            // The residual tile of the half (196 KB) comes through registers UNDER the K loop in four slices of 4 accumulator rows
            // (12 x 16 B per lane each): slice 0 requested in the prologue and added at the end of step 0, slice g requested at the end of
            // step g - 1 and added at the end of step g.  vmcnt counts in order, so a slice is forced to have landed one K-step after its
            // request (the next B tile is younger): two possible short stalls per half instead of one per K-step.
            u32x4 rv[12];
            auto res_issue = [&](auto g_) {                // accumulator rows 4 g .. 4 g + 3
                constexpr int g = decltype(g_)::value;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = rowu + 8 * g + q;
#pragma unroll
                    for (int c = 0; c < 3; ++c) rv[3 * q + c] = __builtin_amdgcn_raw_buffer_load_b128(rsR, voR + 512 * c, (int)((size_t)m * a.ldr * 4 + cb), 0);
                }
            };
            auto res_add = [&](auto g_) {
                constexpr int g = decltype(g_)::value;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const f32x4 f = __builtin_bit_cast(f32x4, rv[3 * q + c]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[4 * c + e][4 * g + q] += f[e];
                    }
            };
            const bool use_res = a.resid && !(TTL_ROW_DIAG & 1);
            if (use_res) res_issue(IC(0));
            // fragment sets alternate per k16 sub-step (2 x 13 x 4 registers).  The loop body is phase-shifted behind the barrier, like
            // gemm_huge.hip's: [sub-step 3 of K-tile kt-1][0][1][2 ; wait ; barrier] — the MFMAs of a K-tile's last sub-step run on
            // fragments read BEFORE the barrier while the first fragments of the next K-tile arrive, so only a half's very first read is exposed
            opx8 xa[2], wf[2][12];
            auto frags = [&](const char* sa, const char* sb, int s2, int set) {
                xa[set] = *(const opx8*)(sa + (fA0 ^ (s2 << 5)));
                const char* pb = sb + (fB0 ^ (s2 << 5));
#pragma unroll
                for (int j = 0; j < 12; ++j) wf[set][j] = *(const opx8*)(pb + j * 4096);
            };
            auto mma = [&](int set) {
#pragma unroll
                for (int j = 0; j < 12; ++j) acc[j] = MFMA32(xa[set], wf[set][j], acc[j], 0, 0, 0);
            };
            // 12 MFMAs with the 13 fragment reads of the next sub-step (one or two behind each of the first MFMAs) and NV DMA pieces behind them
            auto mix = [&](auto nv_) {
                constexpr int NV = decltype(nv_)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
#pragma unroll
                for (int j = 0; j < 5; ++j) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                if constexpr (NV > 0) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            // one K-step.  FRONT: the body starts with sub-step 3 of K-tile kt - 1 (not for kt = 0); RC: residual slice added at its end (-1: none),
            // RI: slice requested at its end; HB / HA: K-tiles kt + 1 (B) / kt + 2 (A) exist
            auto step = [&](int kt, auto front_, auto rc_, auto ri_, auto hb_, auto ha_) {
                constexpr bool FRONT = decltype(front_)::value, HB = decltype(hb_)::value, HA = decltype(ha_)::value;
                constexpr int RC = decltype(rc_)::value, RI = decltype(ri_)::value;
                if constexpr (FRONT) {
                    frags(aC, bC, 0, 0);
                    if constexpr (HB) { dma_b_piece(bN, half, kt + 1, 0); dma_b_piece(bN, half, kt + 1, 1); dma_b_piece(bN, half, kt + 1, 2); }
                    mma(1);
                    mix(IC(HB ? 3 : 0));
                } else {
                    frags(aC, bC, 0, 0);
                    if constexpr (HB) { dma_b_piece(bN, half, kt + 1, 0); dma_b_piece(bN, half, kt + 1, 1); dma_b_piece(bN, half, kt + 1, 2); }
                }
                frags(aC, bC, 1, 1);
                if constexpr (HB) { dma_b_piece(bN, half, kt + 1, 3); dma_b_piece(bN, half, kt + 1, 4); dma_b_piece(bN, half, kt + 1, 5); }
                mma(0);
                mix(IC(HB ? 3 : 0));
                frags(aC, bC, 2, 0);
                if constexpr (HB) { dma_b_piece(bN, half, kt + 1, 6); dma_b_piece(bN, half, kt + 1, 7); dma_b_piece(bN, half, kt + 1, 8); }
                mma(1);
                mix(IC(HB ? 3 : 0));
                frags(aC, bC, 3, 1);
                if constexpr (HB) { dma_b_piece(bN, half, kt + 1, 9); dma_b_piece(bN, half, kt + 1, 10); dma_b_piece(bN, half, kt + 1, 11); }
                mma(0);
                mix(IC(HB ? 3 : 0));
                if constexpr (RC >= 0) {
                    if (use_res) { row_wait_vm<HB ? 12 : 0>(); res_add(IC(RC >= 0 ? RC : 0)); }
                }
                if constexpr (HA) dma_a(aNN, row0, kt + 2);
                if constexpr (RI >= 0) {
                    if (use_res) res_issue(IC(RI >= 0 ? RI : 0));
                }
                if constexpr (HB) {      // top of the next step: K-tile kt + 1 has landed when only A(kt+2) and a requested slice are still in flight
                    if (RI >= 0 && use_res) row_wait_vm<(HA ? 4 : 0) + 12>(); else row_wait_vm<(HA ? 4 : 0)>();
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                }
                { char* t = aC; aC = aN; aN = aNN; aNN = t; t = bC; bC = bN; bN = t; }
            };
            if (use_res) row_wait_vm<4 + 12>(); else row_wait_vm<4>();       // B(0), A(0) landed; A(1) and the first slice may fly
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            step(0, F{}, IC(0), IC(1), T{}, T{});
            step(1, T{}, IC(1), IC(2), T{}, T{});
            step(2, T{}, IC(2), IC(3), T{}, T{});
            step(3, T{}, IC(3), IC(-1), T{}, T{});
            int kt = 4;
            for (; kt + 2 < ((TTL_ROW_DIAG & 8) ? 6 : nk); ++kt) step(kt, T{}, IC(-1), IC(-1), T{}, T{});
            step(kt, T{}, IC(-1), IC(-1), T{}, F{}); ++kt;
            step(kt, T{}, IC(-1), IC(-1), F{}, F{});
            mma(1);                                   // sub-step 3 of the last K-tile
            __builtin_amdgcn_sched_barrier(0);
            // ---- this half of the fp32 stream leaves (register r of column tile j: row rowu + 4 lh + 8 (r >> 2) + (r & 3), column 384 half + 128 (j >> 2) + 4 l32 + (j & 3))
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rowu + 8 * (r >> 2) + (r & 3);
                float sum = 0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const f32x4 o = {acc[4 * c][r], acc[4 * c + 1][r], acc[4 * c + 2][r], acc[4 * c + 3][r]};
                    if (!(TTL_ROW_DIAG & 4)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsC, voC + 512 * c, (int)((size_t)m * a.ldc * 4 + cb), 0);
                    sum += (o[0] + o[1]) + (o[2] + o[3]);
                }
                s1[r] += sum;
            }
            __builtin_amdgcn_s_barrier();      // the next prologue overwrites the stages the last K-step read
        }
        // ---- LayerNorm of the panel's rows.  The accumulators hold the SECOND half; the first half comes back from L2 (these lanes wrote it:
        // stores drained by the wait, loads with GLC so that no stale L1 line answers)
        if (TTL_ROW_DIAG & 2) continue;
        row_wait_vm<0>();
        u32x4 fh[16][3];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = rowu + 8 * (r >> 2) + (r & 3);
#pragma unroll
            for (int c = 0; c < 3; ++c) fh[r][c] = __builtin_amdgcn_raw_buffer_load_b128(rsC, voC + 512 * c, (int)((size_t)m * a.ldc * 4), 1);
        }
        float mu[16], rs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) mu[r] = half_sum(s1[r]) * (1.0f / RBN);
#pragma unroll
        for (int r = 0; r < 16; ++r) {          // two-pass variance, like ln_fwd_persist_kernel
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 12; ++j) { const float d = acc[j][r] - mu[r]; q += d * d; }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x4 f = __builtin_bit_cast(f32x4, fh[r][c]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = f[e] - mu[r]; q += d * d; }
            }
            rs[r] = q;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) rs[r] = rsqrtf(half_sum(rs[r]) * (1.0f / RBN) + a.eps);
        if (a.mean && l32 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = rowu + 4 * lh + 8 * (r >> 2) + (r & 3);
                if (m < M) { a.mean[m] = mu[r]; a.rstd[m] = rs[r]; }
            }
        }
        // normalise: per row, half and column run 4 operand values = one 8-B store (32 lanes: 256 contiguous bytes)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const f32x4 g = *(const f32x4*)(a.gamma + RBH * h2 + 128 * c + 4 * l32), b = *(const f32x4*)(a.beta + RBH * h2 + 128 * c + 4 * l32);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = rowu + 8 * (r >> 2) + (r & 3);
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = h2 ? acc[4 * c + e][r] : __builtin_bit_cast(f32x4, fh[r][c])[e];
                        o[e] = (x - mu[r]) * rs[r] * g[e] + b[e];
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_op2(o[0], o[1]), pack_op2(o[2], o[3])}, rsY, voY + (int)(128 * c * sizeof(op_t)),
                                                          (int)(((size_t)m * a.ldy + RBH * h2) * sizeof(op_t)), 0);
                }
            }
    }
#undef IC
}

}  // namespace

hipError_t launch_gemm_row_ln(const GemmArgs& g, const float* gamma, const float* beta, float eps, op_t* Y, int ldy, float* mean, float* rstd,
                              hipStream_t s) {
    if (g.N != RBN || g.K % RBK || g.K < 6 * RBK || g.M < 1 || (g.lda & 7) || (g.ldb & 7) || (ldy & 3) || (g.ldc & 3) || (g.resid && (g.ldr & 3)) ||
        !gamma || !beta || !Y || !g.C || (mean == nullptr) != (rstd == nullptr))
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

#ifdef TTL_ROW_PROBE      // standalone probe library (tools/r06_row_probe.py): C entry without the rest of libttl_hip
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
