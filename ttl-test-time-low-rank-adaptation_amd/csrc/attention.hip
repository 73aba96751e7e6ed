// Fused attention for head dim 64 (CLIP ViT-B/16: T=197, ViT-L/14: T=257; text tower: T=77, causal).
//   forward : o = softmax(q k^T / 8) v, row logsumexp saved            (HF modeling_clip.py:259-277)
//   backward: dq, dk, dv with P recomputed from the saved logsumexp     (autograd of the same)
//
// One workgroup per (view, head) — one wave per 32-row block of the sequence (attn_fwd_w_kernel, the backward kernels), 4 waves for
// short sequences — or, for the 64-view forward launches, one persistent workgroup per CU walking (view, head) problems with the
// next problem's tiles prefetched (attn_fwd_p_kernel); the whole K/V (or Q/dO) of the head lives in LDS.
// MFMA 32x32x16 bf16 everywhere.  Orientation is chosen so that NO accumulator ever crosses
// lanes or LDS (guide §3 "accumulator tile as the next MFMA's operand"):
//   forward / dQ pass : S^T = K Q^T (query on the lane) -> softmax is lane-local ->
//                       O^T = V^T P^T,  dQ^T = K^T dS^T  take P^T / dS^T straight from registers;
//                       V^T / K^T fragments come from ds_read_b64_tr_b16 on the row-major tile.
//   dK/dV pass        : S = Q K^T (key on the lane) -> dV^T = dO^T P, dK^T = Q^T dS likewise.
// LDS tiles are [rows][64] bf16 (128 B rows) filled by global_load_lds_dwordx4; 16-B chunk c of
// row r sits at slot c ^ f(r), f(r) = ((r>>1)&1)<<2 | ((r>>2)&3): conflict-free for the
// ds_read_b128 row reads AND for the transposed 4x16 block reads.
#include <stdlib.h>

#include <atomic>

#include "kernels.hpp"

namespace {

__device__ __forceinline__ int swz(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }

// byte offset of element (row, col) in a swizzled [rows][64] bf16 tile
__device__ __forceinline__ int tile_off(int row, int col) {
    return row * 128 + ((((col >> 3) ^ swz(row)) << 4) | ((col & 7) << 1));
}

// DMA rows [0,TP) of a [T][64] bf16 matrix (row stride ld elements) into a swizzled LDS tile;
// rows >= T replicate row T-1 (finite data; their results are masked).
template <int NKT, int NW = 4>
__device__ __forceinline__ void stage_tile(char* lds, const op_t* g, int ld, int T, int tid, int wave) {
    constexpr int NTHR = 64 * NW, NPASS = (NKT * 256 + NTHR - 1) / NTHR;   // TP rows x 8 chunks of 16 B
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
        int q = i * NTHR + tid, r = q >> 3, p = q & 7;
        if (NKT * 256 % NTHR != 0 && q >= NKT * 256) break;   // whole waves drop out (256 % 64 == 0)
        int c = p ^ swz(r);
        const op_t* src = g + (size_t)min(r, T - 1) * ld + c * 8;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + (i * NTHR + wave * 64) * 16), 16, 0, 0);
    }
}

// A-operand fragment of X^T for k-step (rows kb..kb+15 of the row-major tile X, columns
// cb..cb+31), in the k order of an accumulator-derived B operand:
//   element j of lane half h  <->  row kb + 8*(j>>2) + 4*h + (j&3),  column cb + (lane&31)
__device__ __forceinline__ opx8 tr_frag(const char* tile, int kb, int cb, int lane) {
    const int h = lane >> 5, grp = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const int col = cb + 16 * grp + 4 * p;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + tile_off(kb + 4 * h + q, col)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + tile_off(kb + 8 + 4 * h + q, col)));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(opx8, v);
}

// row-read fragment: rows rb + (lane&31), 16-B chunk 2*ks + (lane>>5)
__device__ __forceinline__ opx8 row_frag(const char* tile, int rb, int ks, int lane) {
    const int r = rb + (lane & 31), c = 2 * ks + (lane >> 5);
    return *(const opx8*)(tile + r * 128 + ((c ^ swz(r)) << 4));
}

__device__ __forceinline__ opx8 global_frag(const op_t* g, int ld, int row, int ks, int lane) {
    return *(const opx8*)(g + (size_t)row * ld + 16 * ks + 8 * (lane >> 5));
}

// registers 8s..8s+7 of a 32x32 accumulator -> bf16 operand fragment of k-step s
__device__ __forceinline__ opx8 acc_frag(const f32x16& a, int s) {
    u32x4 v = {pack_op2(a[8 * s], a[8 * s + 1]), pack_op2(a[8 * s + 2], a[8 * s + 3]),
               pack_op2(a[8 * s + 4], a[8 * s + 5]), pack_op2(a[8 * s + 6], a[8 * s + 7])};
    return __builtin_bit_cast(opx8, v);
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// store a [64 x 32] transposed accumulator pair (rows = head-dim, column = this lane's token).
// A lane holds 4 consecutive head-dim values (8 B) of its token per register group g; the other half-wave holds the NEXT 8 B of the
// same token.  v_permlane32_swap between groups g and g+1 gives the lower half-wave 16 contiguous bytes of group g and the upper one
// 16 contiguous bytes of group g+1 (guide T21): 4 x 16-B stores per lane instead of 8 x 8-B, each instruction still one piece per
// token row, so half the row-line visits on the store path (the per-token pieces are 1.5 KB apart: a store instruction touches
// 32 lines whatever its width).  Both lanes of a pair belong to the same token: they are masked together.
#ifndef TTL_ATTN_WIDE_STORE
#define TTL_ATTN_WIDE_STORE 1
#endif
__device__ __forceinline__ void store_ot(op_t* dst_row, const f32x16 (&o)[2], float mul, int lane) {
#if TTL_ATTN_WIDE_STORE
    const int up = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            uint32_t a0 = pack_op2(o[dt][4 * g] * mul, o[dt][4 * g + 1] * mul), a1 = pack_op2(o[dt][4 * g + 2] * mul, o[dt][4 * g + 3] * mul);
            uint32_t b0 = pack_op2(o[dt][4 * g + 4] * mul, o[dt][4 * g + 5] * mul), b1 = pack_op2(o[dt][4 * g + 6] * mul, o[dt][4 * g + 7] * mul);
            // vdst = group g, src = group g+1: lanes 32-63 of vdst <-> lanes 0-31 of src
            auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            // lower half: [own g | upper's g] = dims 8g .. 8g+7;  upper half: [lower's g+1 | own g+1] = dims 8(g+1) .. 8(g+1)+7
            *(u32x4*)(dst_row + 32 * dt + 8 * (g + up)) = u32x4{r0[0], r1[0], r0[1], r1[1]};
        }
#else
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int dh = 32 * dt + 8 * g + 4 * (lane >> 5);
            *(u32x2*)(dst_row + dh) = u32x2{pack_op2(o[dt][4 * g] * mul, o[dt][4 * g + 1] * mul),
                                            pack_op2(o[dt][4 * g + 2] * mul, o[dt][4 * g + 3] * mul)};
        }
#endif
}

constexpr float SCALE = 0.125f;  // head_dim^-0.5, head_dim = 64

// ------------------------------------------------------------------------------ forward
template <int NKT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const op_t* __restrict__ qkv, const QkvLayout L, op_t* __restrict__ out,
                                                       int ldo, float* __restrict__ lse, int T, int H, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = NKT * 32;
    char* sK = smem;
    char* sV = smem + TP * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const int ld = L.tok;
    const op_t* qg = qkv + (size_t)img * L.view + (size_t)head * L.head;
    const op_t *kg = qg + L.k_off, *vg = qg + L.v_off;
    stage_tile<NKT>(sK, kg, ld, T, tid, wave);
    stage_tile<NKT>(sV, vg, ld, T, tid, wave);
    __syncthreads();

    const int nqb = (T + 31) >> 5;
    constexpr float C2 = SCALE * 1.4426950408889634f;   // softmax in base 2: p = exp2(s*C2 - m*C2)
    for (int qb = wave; qb < nqb; qb += 4) {
        const int qrow = min(qb * 32 + (lane & 31), T - 1);
        opx8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = global_frag(qg, ld, qrow, ks, lane);
        f32x16 st[NKT];
        float mx = -INFINITY;   // max of the RAW scores (the scale is positive: same arg-max)
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x16 a = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                a = MFMA32(row_frag(sK, 32 * kt, ks, lane), qf[ks], a, 0, 0, 0);
            if (32 * kt + 32 > T) {   // wave-uniform: only the tile(s) holding keys >= T pay for the mask
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * kt + acc_row(r, lane) >= T) a[r] = -INFINITY;
            }
            if (causal && kt >= qb) {   // keys after the query (the lane's query is qb*32 + lane%32)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * kt + acc_row(r, lane) > qb * 32 + (lane & 31)) a[r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, a[r]);
            st[kt] = a;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mc = mx * C2;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], C2, -mc));
                st[kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 32, 64);
        f32x16 o[2] = {};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                opx8 pf = acc_frag(st[kt], s);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    o[dt] = MFMA32(tr_frag(sV, 32 * kt + 16 * s, 32 * dt, lane), pf, o[dt], 0, 0, 0);
            }
        const int q = qb * 32 + (lane & 31);
        if (q < T) {
            store_ot(out + (size_t)(img * T + q) * ldo + head * 64, o, 1.0f / sum, lane);
            if (lse && lane < 32) lse[((size_t)img * H + head) * T + q] = mx * SCALE + __logf(sum);
        }
    }
}

// ------------------------------------------------------------------------------ forward, one wave per query block
// NKT waves per workgroup, wave w owns query rows 32w..32w+31: all query blocks of a (view, head)
// run at once, and the scores are processed in chunks of CH key tiles with an online softmax so a
// wave needs ~125 VGPRs (the whole-row version needs 256): two 7-wave workgroups per CU = 3.5 waves
// per SIMD hide the LDS / exp latency that the 4-wave version (2 per SIMD) exposed.
template <int NKT, int CH>
__global__ __launch_bounds__(64 * NKT, 4) void attn_fwd_w_kernel(const op_t* __restrict__ qkv, const QkvLayout L, op_t* __restrict__ out,
                                                             int ldo, float* __restrict__ lse, int T, int H, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = NKT * 32, NTHR = 64 * NKT;
    char* sK = smem;
    char* sV = smem + TP * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const int ld = L.tok;
    const op_t* qg = qkv + (size_t)img * L.view + (size_t)head * L.head;
    const op_t *kg = qg + L.k_off, *vg = qg + L.v_off;
    // this wave's Q fragments first: their latency hides under the K/V staging
    const int q = wave * 32 + (lane & 31);
    const int qrow = min(q, T - 1);
    opx8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = global_frag(qg, ld, qrow, ks, lane);
    // stage K and V: TP*8 chunks of 16 B each, NTHR chunks per pass -> 4 passes
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c16 = i * NTHR + tid, r = c16 >> 3, p = c16 & 7;
        int c = p ^ swz(r);
        size_t go = (size_t)min(r, T - 1) * ld + c * 8;
        __builtin_amdgcn_global_load_lds(GLB_PTR(kg + go), LDS_PTR(sK + (i * NTHR + wave * 64) * 16), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(vg + go), LDS_PTR(sV + (i * NTHR + wave * 64) * 16), 16, 0, 0);
    }
    __syncthreads();
    if (wave * 32 >= T) return;   // (never for NKT = ceil(T/32); kept for safety — after the only barrier)

    constexpr float C2 = SCALE * 1.4426950408889634f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2] = {};
#pragma unroll
    for (int c0 = 0; c0 < NKT; c0 += CH) {
        f32x16 st[CH];
        float mx = m_run;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int kt = c0 + j;
            if (kt < NKT) {
                f32x16 a = {};
                if (causal && kt > wave) {   // every key of this tile comes after every query of the wave: nothing to compute
#pragma unroll
                    for (int r = 0; r < 16; ++r) a[r] = -INFINITY;
                    st[j] = a;
                    continue;
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) a = MFMA32(row_frag(sK, 32 * kt, ks, lane), qf[ks], a, 0, 0, 0);
                if (32 * kt + 32 > T) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (32 * kt + acc_row(r, lane) >= T) a[r] = -INFINITY;
                }
                if (causal && kt >= wave) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (32 * kt + acc_row(r, lane) > q) a[r] = -INFINITY;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, a[r]);
                st[j] = a;
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mc = mx * C2;
        const float alpha = __builtin_amdgcn_exp2f(m_run * C2 - mc);   // 0 on the first chunk (m_run = -inf)
        m_run = mx;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (c0 + j < NKT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f(fmaf(st[j][r], C2, -mc));
                    st[j][r] = p;
                    sum += p;
                }
            }
        sum += __shfl_xor(sum, 32, 64);
        l_run = l_run * alpha + sum;
        if (c0 > 0) {
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        }
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (c0 + j < NKT && !(causal && c0 + j > wave)) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    opx8 pf = acc_frag(st[j], s);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
                        o[dt] = MFMA32(tr_frag(sV, 32 * (c0 + j) + 16 * s, 32 * dt, lane), pf, o[dt], 0, 0, 0);
                }
            }
    }
    if (q < T) {
        store_ot(out + (size_t)(img * T + q) * ldo + head * 64, o, 1.0f / l_run, lane);
        if (lse && lane < 32) lse[((size_t)img * H + head) * T + q] = m_run * SCALE + __logf(l_run);
    }
}

// Timing-only ablations of attn_fwd_p_kernel (results wrong on purpose): 1 = no arithmetic, 2 = no prefetch of the next problem
#ifndef TTL_ATTN_DIAG
#define TTL_ATTN_DIAG 0
#endif
// (-DTTL_DIAG_SKIP=bits, tools/class_cost_ab.sh: timing-only ablation of the EPISODE, results wrong on purpose: bit 0 = the dense
// forward launches stop being issued after the warm-up, bit 1 = the dense backward launches)

// 16 B per lane, global -> LDS (M0 = wave-uniform LDS base, lane-linear destination), invisible to hipcc's wait-count pass
__device__ __forceinline__ void dma16_untracked(const void* g, const char* lds) {
    const uint32_t l = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)lds);
    // M0 is a reserved register that cannot be named as a clobber: saved and restored around the instruction
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(l)
                 : "memory");
}

// ------------------------------------------------------------------------------ forward, persistent (vision towers)
// One workgroup per CU walks (view, head) problems p = blockIdx.x, + gridDim.x, ...; the K/V tiles of problem i+1 are
// requested by LDS-DMA into the second half of the LDS while problem i is multiplied, so the kernel streams q/k/v at the
// rate the memory side delivers them instead of alternating a load phase and a compute phase over 1.5 rounds of blocks
// (attn_fwd_w_kernel: 768 blocks on 512 slots).  NKT waves, wave w owns query rows 32w..32w+31 as above; with one block per CU
// a wave may use 256 VGPRs, so CH = NKT (whole score row in registers, no online rescale) is affordable for T <= 224.
// LDS addresses are per-lane constants + immediates (the swizzle of a row depends on its low 4 bits only, tiles start at
// multiples of 16 rows).  Not causal (the text tower keeps attn_fwd_w_kernel).  Chunked (online) softmax in this kernel, 2 / 3 / 4
// key tiles per chunk: 0.380 / 0.321 / 0.329 ms of attention forward per episode against 0.311 with the whole row (CH = NKT).
// Holding the upper waves (w + 4, the SIMD partners of w) back by 0.5-1.8 k cycles so that one wave's softmax runs beside the other's
// matrix phases: no change (0.308-0.312): what is left is the rate the q/k/v segments arrive at.  Giving the workgroups of XCD x
// the views whose q/k/v rows XCD x's tiles of the QKV GEMM have just written (producer / consumer on the same L2): no change either.
template <int NKT, int CH>
__global__ __launch_bounds__(64 * NKT, 1) void attn_fwd_p_kernel(const op_t* __restrict__ qkv, const QkvLayout L, op_t* __restrict__ out,
                                                             int ldo, float* __restrict__ lse, int T, int H, int nprob, int xmap) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = NKT * 32, NTHR = 64 * NKT, TILE = TP * 128, BUF = 2 * TILE, QOFF = 2 * BUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = L.tok;
    const int h = lane >> 5, l31 = lane & 31;
    // row-read offsets of the K tile (row l31 of a 32-row block, 16-B chunk 2*ks + h); the q rows of this wave likewise
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * 128 + (((2 * ks + h) ^ swz(l31)) << 4);
    // transposed-read offsets of the V tile: [dt][lo/hi] (tr_frag with kb = 0, cb = 32*dt)
    int voff[2][2];
    {
        const int grp = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) voff[dt][hh] = TILE + tile_off(8 * hh + 4 * h + qq, 32 * dt + 16 * grp + 4 * pp);
    }
    // K/V: this thread's 4 chunks each (chunk c16 = i*NTHR + tid -> row c16 >> 3, slot c16 & 7); q: every wave fetches ITS OWN
    // 32 rows (4 instructions of 8 rows), so re-filling the q tile needs no barrier, only the wave's own reads retired.
    // The DMA is issued from inline asm: with the builtin, hipcc's wait-count pass puts vmcnt(0) in front of the first
    // ds_read_b64_tr_b16 that follows (it cannot tell that the transposed read touches the OTHER half of the LDS), which
    // would serialise prefetch and PV.  There is no ordinary load in the loop, so the pass has nothing to wait for; the
    // waits that matter are the explicit counted ones at the top of the problem loop.
    auto stage = [&](const op_t* qg, char* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c16 = i * NTHR + tid, r = c16 >> 3, pz = c16 & 7;
            const int c = pz ^ swz(r);
            const size_t go = (size_t)min(r, T - 1) * ld + c * 8;
            dma16_untracked(qg + L.k_off + go, buf + (i * NTHR + wave * 64) * 16);
            dma16_untracked(qg + L.v_off + go, buf + TILE + (i * NTHR + wave * 64) * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 32 * wave + 8 * i + (lane >> 3), c = (lane & 7) ^ swz(r);
            dma16_untracked(qg + (size_t)min(r, T - 1) * ld + c * 8, smem + QOFF + (32 * wave + 8 * i) * 128);
        }
    };
    const int q = wave * 32 + l31;
    // xmap (TTL_ATTN_XCD_MAP=1, experiment): blocks b, b+8, ... share an XCD (round-robin dispatch); XCD x then walks the problems of
    // views [x n/8, (x+1) n/8) — the rows whose out_proj tiles the big-M GEMM gives to XCD x (gemm_big.hip tile_of) — so that the
    // attention output is still in the L2 that reads it as the next launch's A operand.  p_step / p_end describe the block's walk.
    int p = blockIdx.x, p_step = gridDim.x, p_end = nprob;
    if (xmap && (gridDim.x & 7) == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, nb = gridDim.x >> 3;
        const int lo = (int)((long long)x * nprob / 8), hi = (int)((long long)(x + 1) * nprob / 8);
        p = lo + j; p_step = nb; p_end = hi;
        if (p >= p_end) return;
    }
    int img = p / H, head = p - img * H;
    stage(qkv + (size_t)img * L.view + (size_t)head * L.head, smem);

    constexpr float C2 = SCALE * 1.4426950408889634f;
    for (int it = 0;; ++it) {
        const int boff = (it & 1) * BUF;
        // q/K/V of this problem landed: the first problem's requests are waited for here, every later problem's at the end of
        // the previous iteration, AHEAD of that iteration's output stores (vmcnt(0) there retires the 12 DMA operations per lane
        // without assuming how many store instructions the compiler emits; the stores then stay in flight across the barrier)
        if (it == 0) __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8) | (0 << 14));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        opx8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const opx8*)(smem + QOFF + 32 * wave * 128 + koff[ks]);
        // next problem: K/V into the other half of the LDS (everyone is past the barrier, so nobody reads that half any
        // more), q over this wave's own rows once they are in registers
        const int pn = p + p_step;
        const bool more = pn < p_end;
        int imgn = img, headn = head;
        if (more) {
            imgn = pn / H; headn = pn - imgn * H;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (TTL_ATTN_DIAG != 2) stage(qkv + (size_t)imgn * L.view + (size_t)headn * L.head, smem + (boff ^ BUF));
        }
        int kb[4], vb[2][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kb[ks] = koff[ks] + boff;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) vb[dt][hh] = voff[dt][hh] + boff;

        float m_run = -INFINITY, l_run = 0.f;
        f32x16 o[2] = {};
#pragma unroll
        for (int c0 = 0; c0 < (TTL_ATTN_DIAG == 1 ? 0 : NKT); c0 += CH) {
            f32x16 st[CH];
            float mx = m_run;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int kt = c0 + j;
                if (kt < NKT) {
                    f32x16 a = {};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) a = MFMA32(*(const opx8*)(smem + kb[ks] + 32 * kt * 128), qf[ks], a, 0, 0, 0);
                    if (kt == NKT - 1 && 32 * NKT > T) {     // only the last key tile holds padded keys
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            if (32 * kt + acc_row(r, lane) >= T) a[r] = -INFINITY;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, a[r]);
                    st[j] = a;
                }
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float mc = mx * C2;
            float alpha = 1.f;
            if (CH < NKT) alpha = __builtin_amdgcn_exp2f(m_run * C2 - mc);   // 0 on the first chunk (m_run = -inf)
            m_run = mx;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (c0 + j < NKT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float e = __builtin_amdgcn_exp2f(fmaf(st[j][r], C2, -mc));
                        st[j][r] = e;
                        sum += e;
                    }
                }
            sum += __shfl_xor(sum, 32, 64);
            l_run = (CH < NKT) ? l_run * alpha + sum : sum;
            if (CH < NKT && c0 > 0) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (c0 + j < NKT) {
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const opx8 pf = acc_frag(st[j], s);
                        const int kbyte = (32 * (c0 + j) + 16 * s) * 128;
#pragma unroll
                        for (int dt = 0; dt < 2; ++dt) {
                            typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
                            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + vb[dt][0] + kbyte));
                            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + vb[dt][1] + kbyte));
                            typedef __attribute__((ext_vector_type(8))) short s16x8;
                            s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            o[dt] = MFMA32(__builtin_bit_cast(opx8, v), pf, o[dt], 0, 0, 0);
                        }
                    }
                }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8) | (0 << 14));     // next problem's q/K/V landed (see the loop top)
        __builtin_amdgcn_sched_barrier(0);
        if (TTL_ATTN_DIAG == 3) {       // timing-only: no output stores (accumulators kept alive)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) asm volatile("" ::"v"(o[dt]));
            asm volatile("" ::"v"(l_run), "v"(m_run));
        } else if (q < T) {
            store_ot(out + (size_t)(img * T + q) * ldo + head * 64, o, 1.0f / l_run, lane);
            if (lse && lane < 32) lse[((size_t)img * H + head) * T + q] = m_run * SCALE + __logf(l_run);
        }
        if (!more) break;
        p = pn; img = imgn; head = headn;
    }
}

// ------------------------------------------------------------------------------ forward, persistent + software-pipelined (round 4)
// Same outer structure as attn_fwd_p_kernel (one workgroup per CU walks (view, head) problems, the next problem's q/K/V arrive by
// LDS-DMA meanwhile), but the inner loop is a per-key-tile pipeline instead of three whole-row phases:
//   iteration t:   S(t+1) = K(t+1) Q^T   (4 MFMAs, issued FIRST: they run on the matrix pipe ...)
//                  softmax of S(t)        (... while the VALU forms P(t) = exp2(S(t) C2 - m C2), its row sum and the packed fragments)
//                  O^T += V(t)^T P(t)^T   (4 MFMAs)
// with the K fragments of tile t+2 and the V^T fragments of tile t+1 requested from LDS one iteration ahead.  attn_fwd_p_kernel
// spends a problem as QK phase (every MFMA behind the LDS read it has just issued), a 56-deep v_max3 chain, an exp phase with a
// 112-deep add chain, then PV: 14 k cycles per problem where the VALU issue floor of its own instruction mix is ~6 k.
// Running maximum with a deferred rescale (guide T13): m only moves when a tile's maximum exceeds it by more than THR (in exp2
// units), so after the first tile the 32-register rescale of O almost never runs; P stays <= 2^THR (relative precision of the
// 16-bit P operand unchanged), l is summed in fp32, O / l and lse = m SCALE + log l come out as before up to fp32 rounding order.
constexpr float ATTN_DEFER_THR = 4.0f;      // exp2 units: P <= 16

__device__ __forceinline__ float max3f(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <int NKT>
__global__ __launch_bounds__(64 * NKT, 1) void attn_fwd_s_kernel(const op_t* __restrict__ qkv, const QkvLayout L, op_t* __restrict__ out,
                                                             int ldo, float* __restrict__ lse, int T, int H, int nprob) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = NKT * 32, NTHR = 64 * NKT, TILE = TP * 128, BUF = 2 * TILE, QOFF = 2 * BUF;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = L.tok;
    const int h = lane >> 5, l31 = lane & 31;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = l31 * 128 + (((2 * ks + h) ^ swz(l31)) << 4);
    int voff[2][2];
    {
        const int grp = (lane >> 4) & 1, qq = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) voff[dt][hh] = TILE + tile_off(8 * hh + 4 * h + qq, 32 * dt + 16 * grp + 4 * pp);
    }
    auto stage = [&](const op_t* qg, char* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c16 = i * NTHR + tid, r = c16 >> 3, pz = c16 & 7;
            const int c = pz ^ swz(r);
            const size_t go = (size_t)min(r, T - 1) * ld + c * 8;
            dma16_untracked(qg + L.k_off + go, buf + (i * NTHR + wave * 64) * 16);
            dma16_untracked(qg + L.v_off + go, buf + TILE + (i * NTHR + wave * 64) * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 32 * wave + 8 * i + (lane >> 3), c = (lane & 7) ^ swz(r);
            dma16_untracked(qg + (size_t)min(r, T - 1) * ld + c * 8, smem + QOFF + (32 * wave + 8 * i) * 128);
        }
    };
    const int q = wave * 32 + l31;
    int p = blockIdx.x;
    int img = p / H, head = p - img * H;
    stage(qkv + (size_t)img * L.view + (size_t)head * L.head, smem);

    constexpr float C2 = SCALE * 1.4426950408889634f;
    constexpr float THR_RAW = ATTN_DEFER_THR / C2;       // the threshold in raw-score units
    for (int it = 0;; ++it) {
        const int boff = (it & 1) * BUF;
        if (it == 0) __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8) | (0 << 14));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        opx8 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const opx8*)(smem + QOFF + 32 * wave * 128 + koff[ks]);
        const int pn = p + gridDim.x;
        const bool more = pn < nprob;
        int imgn = img, headn = head;
        if (more) {
            imgn = pn / H; headn = pn - imgn * H;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (TTL_ATTN_DIAG != 2) stage(qkv + (size_t)imgn * L.view + (size_t)headn * L.head, smem + (boff ^ BUF));
        }
        const char* kbase = smem + boff;
        auto load_k = [&](opx8 (&kf)[4], int kt) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *(const opx8*)(kbase + koff[ks] + 32 * kt * 128);
        };
        auto load_v = [&](opx8 (&vf)[2][2], int kt) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int kbyte = (32 * kt + 16 * s2) * 128;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(kbase + voff[dt][0] + kbyte));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(kbase + voff[dt][1] + kbyte));
                    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    vf[s2][dt] = __builtin_bit_cast(opx8, v);
                }
        };
        auto qk = [&](const opx8 (&kf)[4]) {
            f32x16 a = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) a = MFMA32(kf[ks], qf[ks], a, 0, 0, 0);
            return a;
        };
        opx8 kf[4], vf[2][2];
        load_k(kf, 0);
        f32x16 s_cur = qk(kf);
        // max3f (inline asm) reads these registers: hipcc pads MFMA -> VALU hazards for its own instructions only (guide 5.7 item 2),
        // so the wait states go INSIDE an asm statement that owns the tile.  Tiles kt >= 1 were multiplied a whole iteration earlier.
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s_cur));
        if (NKT > 1) load_k(kf, 1);
        load_v(vf, 0);
        float m_run = -INFINITY;
        float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;       // four partial row sums: no 112-deep add chain
        f32x16 o[2] = {};
#pragma unroll
        for (int kt = 0; kt < (TTL_ATTN_DIAG == 1 ? 0 : NKT); ++kt) {
            f32x16 s_next = {};
            if (kt + 1 < NKT) {
                s_next = qk(kf);                             // tile kt+1 on the matrix pipe while the VALU does tile kt
                if (kt + 2 < NKT) load_k(kf, kt + 2);
            }
            if (kt == NKT - 1 && 32 * NKT > T) {             // only the last key tile holds padded keys
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (32 * kt + acc_row(r, lane) >= T) s_cur[r] = -INFINITY;
            }
            // tile maximum: a tree of v_max3_f32, not a chain.  (fmaxf on MFMA outputs makes hipcc canonicalise every input with a
            // v_max_f32 x, x first — 119 extra VALU instructions per problem here; the scores are never signalling NaNs.)
            const float m0 = max3f(s_cur[0], s_cur[1], s_cur[2]), m1 = max3f(s_cur[3], s_cur[4], s_cur[5]);
            const float m2 = max3f(s_cur[6], s_cur[7], s_cur[8]), m3 = max3f(s_cur[9], s_cur[10], s_cur[11]);
            const float m4 = max3f(s_cur[12], s_cur[13], s_cur[14]);
            float mx = max3f(max3f(m0, m1, m2), max3f(m3, m4, s_cur[15]), -INFINITY);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));          // the other half-wave holds the other 16 keys of this query
            if (kt == 0) {
                m_run = mx;
            } else if (__any(mx > m_run + THR_RAW)) {        // wave-uniform and rare after the first tile
                const float m_new = (mx > m_run + THR_RAW) ? mx : m_run;
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * C2);     // 1 for the lanes whose maximum did not move
                m_run = m_new;
                l0 *= alpha; l1 *= alpha; l2 *= alpha; l3 *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
            }
            const float mc = m_run * C2;
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const float e0 = __builtin_amdgcn_exp2f(fmaf(s_cur[r], C2, -mc)), e1 = __builtin_amdgcn_exp2f(fmaf(s_cur[r + 1], C2, -mc));
                const float e2 = __builtin_amdgcn_exp2f(fmaf(s_cur[r + 2], C2, -mc)), e3 = __builtin_amdgcn_exp2f(fmaf(s_cur[r + 3], C2, -mc));
                s_cur[r] = e0; s_cur[r + 1] = e1; s_cur[r + 2] = e2; s_cur[r + 3] = e3;
                l0 += e0; l1 += e1; l2 += e2; l3 += e3;
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const opx8 pf = acc_frag(s_cur, s2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) o[dt] = MFMA32(vf[s2][dt], pf, o[dt], 0, 0, 0);
            }
            if (kt + 1 < NKT) load_v(vf, kt + 1);
            s_cur = s_next;
        }
        float l_run = (l0 + l1) + (l2 + l3);
        l_run += __shfl_xor(l_run, 32, 64);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt((0 & 15) | (7 << 4) | (15 << 8) | (0 << 14));     // next problem's q/K/V landed (see attn_fwd_p_kernel)
        __builtin_amdgcn_sched_barrier(0);
        if (TTL_ATTN_DIAG == 3) {       // timing-only: no output stores (accumulators kept alive)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) asm volatile("" ::"v"(o[dt]));
            asm volatile("" ::"v"(l_run), "v"(m_run));
        } else if (q < T) {
            store_ot(out + (size_t)(img * T + q) * ldo + head * 64, o, 1.0f / l_run, lane);
            if (lse && lane < 32) lse[((size_t)img * H + head) * T + q] = m_run * SCALE + __logf(l_run);
        }
        if (!more) break;
        p = pn; img = imgn; head = headn;
    }
}

// ------------------------------------------------------------------------------ backward: dQ
template <int NKT, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(const op_t* __restrict__ qkv, const QkvLayout L,
                                                          const op_t* __restrict__ out, const op_t* __restrict__ dout,
                                                          int ldo, const float* __restrict__ lse,
                                                          op_t* __restrict__ dqkv, int ldd, int T, int H, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = NKT * 32;
    char* sK = smem;
    char* sV = smem + TP * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const int ld = L.tok;
    const op_t* qg = qkv + (size_t)img * L.view + (size_t)head * L.head;
    const op_t *kg = qg + L.k_off, *vg = qg + L.v_off;
    const op_t* og = out + (size_t)img * T * ldo + head * 64;
    const op_t* dog = dout + (size_t)img * T * ldo + head * 64;
    stage_tile<NKT, NW>(sK, kg, ld, T, tid, wave);
    stage_tile<NKT, NW>(sV, vg, ld, T, tid, wave);
    __syncthreads();

    const int nqb = (T + 31) >> 5;
    for (int qb = wave; qb < nqb; qb += NW) {
        const int q = qb * 32 + (lane & 31);
        const int qrow = min(q, T - 1);
        opx8 qf[4], dof[4];
        float delta = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = global_frag(qg, ld, qrow, ks, lane);
            dof[ks] = global_frag(dog, ldo, qrow, ks, lane);
            opx8 of = global_frag(og, ldo, qrow, ks, lane);
#pragma unroll
            for (int j = 0; j < 8; ++j) delta += (float)dof[ks][j] * (float)of[j];
        }
        delta += __shfl_xor(delta, 32, 64);
        // p = exp(s/8 - lse) as exp2(s*C2 - lse*log2 e): one fma + v_exp_f32 per element (the forward's form)
        constexpr float LOG2E = 1.4426950408889634f, C2 = SCALE * LOG2E;
        const float l = lse[((size_t)img * H + head) * T + qrow] * LOG2E;
        f32x16 dq[2] = {};
#pragma unroll 1
        for (int kt = 0; kt < NKT; ++kt) {
            if (causal && kt > qb) break;   // keys after every query of this block
            f32x16 s = {}, dp = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = MFMA32(row_frag(sK, 32 * kt, ks, lane), qf[ks], s, 0, 0, 0);
                dp = MFMA32(row_frag(sV, 32 * kt, ks, lane), dof[ks], dp, 0, 0, 0);
            }
            // only the tile that holds keys >= T (the last one) and, in the causal tower, the tiles at or past the query block
            // need the per-element mask: a wave-uniform branch keeps the compares and selects out of the other tiles
            if (32 * kt + 32 > T || (causal && kt >= qb)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = 32 * kt + acc_row(r, lane);
                    float p = (key < T && !(causal && key > q)) ? __builtin_amdgcn_exp2f(fmaf(s[r], C2, -l)) : 0.f;
                    s[r] = p * (dp[r] - delta) * TTL_DS_PRESCALE;  // dS^T (pre-scaled by a power of two in the fp16 build: common.hpp)
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(fmaf(s[r], C2, -l)) * (dp[r] - delta) * TTL_DS_PRESCALE;
            }
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                opx8 dsf = acc_frag(s, sb);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = MFMA32(tr_frag(sK, 32 * kt + 16 * sb, 32 * dt, lane), dsf,
                                                                     dq[dt], 0, 0, 0);
            }
        }
        if (q < T) store_ot(dqkv + (size_t)(img * T + q) * ldd + head * 64, dq, SCALE * (1.0f / TTL_DS_PRESCALE), lane);
    }
}

// ------------------------------------------------------------------------------ backward: dK, dV
template <int NKT, bool NEED_DK, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dkv_kernel(const op_t* __restrict__ qkv, const QkvLayout L,
                                                           const op_t* __restrict__ out,
                                                           const op_t* __restrict__ dout, int ldo,
                                                           const float* __restrict__ lse, op_t* __restrict__ dqkv,
                                                           int ldd, int T, int H, int causal) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int TP = NKT * 32;
    char* sQ = smem;
    char* sDO = smem + TP * 128;
    float* sLse = (float*)(smem + 2 * TP * 128);
    float* sDelta = sLse + TP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const int ld = L.tok;
    const op_t* qg = qkv + (size_t)img * L.view + (size_t)head * L.head;
    const op_t *kg = qg + L.k_off, *vg = qg + L.v_off;
    const op_t* og = out + (size_t)img * T * ldo + head * 64;
    const op_t* dog = dout + (size_t)img * T * ldo + head * 64;
    stage_tile<NKT, NW>(sQ, qg, ld, T, tid, wave);
    stage_tile<NKT, NW>(sDO, dog, ldo, T, tid, wave);
    for (int t = tid; t < TP; t += 64 * NW) {
        int row = min(t, T - 1);
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            opx8 a = *(const opx8*)(dog + (size_t)row * ldo + 8 * c);
            opx8 b = *(const opx8*)(og + (size_t)row * ldo + 8 * c);
#pragma unroll
            for (int j = 0; j < 8; ++j) d += (float)a[j] * (float)b[j];
        }
        sDelta[t] = d;
        sLse[t] = lse[((size_t)img * H + head) * T + row] * 1.4426950408889634f;     // in base 2: p = exp2(s*C2 - lse*log2 e)
    }
    __syncthreads();

    const int nkb = (T + 31) >> 5;
    for (int kb = wave; kb < nkb; kb += NW) {
        const int key = kb * 32 + (lane & 31);
        const int krow = min(key, T - 1);
        opx8 kf[4], vf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = global_frag(kg, ld, krow, ks, lane);
            vf[ks] = global_frag(vg, ld, krow, ks, lane);
        }
        f32x16 dv[2] = {}, dk[2] = {};
#pragma unroll 1
        for (int qt = (causal ? kb : 0); qt < NKT; ++qt) {   // causal: queries before this key block never see it
            f32x16 s = {}, dp = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = MFMA32(row_frag(sQ, 32 * qt, ks, lane), kf[ks], s, 0, 0, 0);
                dp = MFMA32(row_frag(sDO, 32 * qt, ks, lane), vf[ks], dp, 0, 0, 0);
            }
            f32x16 ds;
            // (per-element mask only in the query tile that holds rows >= T and, causal, in the tiles at or before the key block)
            if (32 * qt + 32 > T || (causal && qt <= kb)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int qi = 32 * qt + acc_row(r, lane);
                    float p = (qi < T && !(causal && key > qi)) ? __builtin_amdgcn_exp2f(fmaf(s[r], SCALE * 1.4426950408889634f, -sLse[qi])) : 0.f;
                    s[r] = p;
                    ds[r] = p * (dp[r] - sDelta[qi]) * TTL_DS_PRESCALE;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qi = 32 * qt + acc_row(r, lane);
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[r], SCALE * 1.4426950408889634f, -sLse[qi]));
                    s[r] = p;
                    ds[r] = p * (dp[r] - sDelta[qi]) * TTL_DS_PRESCALE;
                }
            }
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                opx8 pf = acc_frag(s, sb);
                opx8 dsf = acc_frag(ds, sb);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = MFMA32(tr_frag(sDO, 32 * qt + 16 * sb, 32 * dt, lane), pf,
                                                                     dv[dt], 0, 0, 0);
                    if (NEED_DK)
                        dk[dt] = MFMA32(tr_frag(sQ, 32 * qt + 16 * sb, 32 * dt, lane),
                                                                         dsf, dk[dt], 0, 0, 0);
                }
            }
        }
        if (key < T) {
            const int D = H * 64;     // dqkv is row-major [tokens][dq | dk | dv | ...]: the A operand of the dX GEMM
            op_t* base = dqkv + (size_t)(img * T + key) * ldd + head * 64;
            store_ot(base + 2 * D, dv, 1.0f, lane);
            if (NEED_DK) store_ot(base + D, dk, SCALE * (1.0f / TTL_DS_PRESCALE), lane);
        }
    }
}

// ------------------------------------------------------------------------------ forward, CLS query only
// Last encoder layer of the image tower: only its CLS row reaches the head, so only query 0 of every (view, head)
// is needed: o_0 = softmax(q_0 K^T / 8) V.  Same rounding points as the dense kernel (P rounded to the operand type
// before the PV product, row sum taken before rounding); writes row 0 of `out` and lse[.., 0] in place.
// NIT > 0: 32 * NIT >= T, and the thread's K / V chunks of ALL passes are requested up front (2 NIT x 16 B in flight per lane): with one
// load per loop iteration hipcc waits vmcnt(0) in every iteration, and the kernel was 14 serial round trips long (9.9 us at 64
// views for 39 MB).  NIT == 0: any T, loads inside the loops.
template <int NIT>
__global__ __launch_bounds__(256) void attn_fwd_cls_kernel(const op_t* __restrict__ qkv, const QkvLayout L, op_t* __restrict__ out, int ldo,
                                                           float* __restrict__ lse, int T, int H, const int* __restrict__ qpos,
                                                           int causal) {
    __shared__ float sq[64], sp[320], sred[32][64];
    __shared__ float smax[4], ssum[4];
    const int tid = threadIdx.x;
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const int ld = L.tok;
    const op_t* base = qkv + (size_t)img * L.view + (size_t)head * L.head;
    const op_t *kbase = base + L.k_off, *vbase = base + L.v_off;
    const int qp = qpos ? qpos[img] : 0;              // the pooled query: CLS, or the end-of-text token (text tower)
    const int Tk = causal ? qp + 1 : T;               // keys it can see
    if (tid < 64) sq[tid] = op_to_f32(base[(size_t)qp * ld + tid]);
    __syncthreads();
    const int c = tid & 7, grp = tid >> 3;            // 8 lanes per key, lane c owns head-dim chunk c
    float qc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qc[e] = sq[8 * c + e];
    opx8 kpre[NIT > 0 ? NIT : 1], vpre[NIT > 0 ? NIT : 1];
    if constexpr (NIT > 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int jr = min(32 * it + grp, Tk - 1);
            kpre[it] = *(const opx8*)(kbase + (size_t)jr * ld + 8 * c);
            vpre[it] = *(const opx8*)(vbase + (size_t)jr * ld + 8 * c);
        }
    }
    // pass 1: raw scores -> LDS, running max
    float mx = -INFINITY;
#pragma unroll
    for (int j0 = 0; j0 < (NIT > 0 ? 32 * NIT : Tk); j0 += 32) {
        if (NIT > 0 && j0 >= Tk) break;
        const int j = j0 + grp;
        const int jr = j < Tk ? j : Tk - 1;
        opx8 kf;
        if constexpr (NIT > 0) kf = kpre[j0 / 32]; else kf = *(const opx8*)(kbase + (size_t)jr * ld + 8 * c);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(qc[e], (float)kf[e], s);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) s += __shfl_xor(s, o, 64);
        if (j < Tk) { if (c == 0) sp[j] = s; mx = fmaxf(mx, s); }
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) smax[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
    // pass 2: p = exp((s - max)/8), sum, o += round(p) * v
    float sum = 0.f, o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
    for (int j0 = 0; j0 < (NIT > 0 ? 32 * NIT : Tk); j0 += 32) {
        if (NIT > 0 && j0 >= Tk) break;
        const int j = j0 + grp;
        if (j < Tk) {
            const float pj = __expf((sp[j] - mx) * SCALE);
            if (c == 0) sum += pj;
            const float pb = op_to_f32(f32_to_op(pj));
            opx8 vf;
            if constexpr (NIT > 0) vf = vpre[j0 / 32]; else vf = *(const opx8*)(vbase + (size_t)j * ld + 8 * c);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(pb, (float)vf[e], o[e]);
        }
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) ssum[tid >> 6] = sum;
#pragma unroll
    for (int e = 0; e < 8; ++e) sred[grp][8 * c + e] = o[e];
    __syncthreads();
    sum = (ssum[0] + ssum[1]) + (ssum[2] + ssum[3]);
    if (tid < 64) {
        float v = 0.f;
#pragma unroll 8
        for (int g = 0; g < 32; ++g) v += sred[g][tid];
        out[((size_t)img * T + qp) * ldo + head * 64 + tid] = f32_to_op(v / sum);
        if (tid == 0 && lse) lse[((size_t)img * H + head) * T + qp] = mx * SCALE + __logf(sum);
    }
}

// ------------------------------------------------------------------------------ backward, CLS query only
// Top layer: the loss reads only the CLS token, so d(out) is non-zero for query 0 alone and the
// whole backward of a (view, head) collapses to rank-1 work (SURVEY appendix A with q = q_0):
//   p_j = exp(q0.k_j/8 - lse0), dp_j = do0.v_j, ds_j = p_j (dp_j - do0.o0),
//   dq_0 = sum_j ds_j k_j / 8,  dk_j = ds_j q0 / 8,  dv_j = p_j do0;   dq_t = 0 for t > 0.
template <bool NEED_DK, int NIT>     // NIT as in attn_fwd_cls_kernel
__global__ __launch_bounds__(256) void attn_bwd_cls_kernel(const op_t* __restrict__ qkv, const QkvLayout L, const op_t* __restrict__ out,
                                                           int ldo, const op_t* __restrict__ dout_cls,
                                                           const float* __restrict__ lse, op_t* __restrict__ dqkv, int ldd,
                                                           int T, int H, const int* __restrict__ qpos, int causal) {
    // 8 lanes per key, lane c of the group owns head-dim chunk c (8 values = one 16-B access): every
    // load / store instruction touches whole 128-B rows
    __shared__ float sq[64], sdo[64], sred[32][64];
    __shared__ float sdelta;
    const int tid = threadIdx.x;
    const int img = blockIdx.x / H, head = blockIdx.x - img * H;
    const int ld = L.tok;
    const op_t* base = qkv + (size_t)img * L.view + (size_t)head * L.head;
    const op_t *kbase = base + L.k_off, *vbase = base + L.v_off;
    const int D = H * 64;                  // (dout_cls and dqkv are row-major)
    const int qp = qpos ? qpos[img] : 0;   // the one query with a non-zero d(out): CLS, or end-of-text
    if (tid < 64) {
        sq[tid] = op_to_f32(base[(size_t)qp * ld + tid]);
        float d = op_to_f32(dout_cls[(size_t)img * D + head * 64 + tid]);
        sdo[tid] = d;
        float prod = wave_sum(d * op_to_f32(out[((size_t)img * T + qp) * ldo + head * 64 + tid]));
        if (tid == 0) sdelta = prod;
    }
    __syncthreads();
    const float l0 = lse[((size_t)img * H + head) * T + qp];
    const float delta = sdelta;
    const int c = tid & 7, grp = tid >> 3;            // 32 key groups per pass
    float qc[8], dc[8], dq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { qc[e] = sq[8 * c + e]; dc[e] = sdo[8 * c + e]; dq[e] = 0.f; }
    opx8 kpre[NIT > 0 ? NIT : 1], vpre[NIT > 0 ? NIT : 1];
    if constexpr (NIT > 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int jr = min(32 * it + grp, T - 1);
            kpre[it] = *(const opx8*)(kbase + (size_t)jr * ld + 8 * c);
            vpre[it] = *(const opx8*)(vbase + (size_t)jr * ld + 8 * c);
        }
    }
#pragma unroll
    for (int j0 = 0; j0 < (NIT > 0 ? 32 * NIT : T); j0 += 32) {
        if (NIT > 0 && j0 >= T) break;
        const int j = j0 + grp;
        const bool ok = j < T;
        const int jr = ok ? j : T - 1;
        opx8 kf, vf;
        if constexpr (NIT > 0) { kf = kpre[j0 / 32]; vf = vpre[j0 / 32]; }
        else { kf = *(const opx8*)(kbase + (size_t)jr * ld + 8 * c); vf = *(const opx8*)(vbase + (size_t)jr * ld + 8 * c); }
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { s = fmaf(qc[e], (float)kf[e], s); dp = fmaf(dc[e], (float)vf[e], dp); }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) { s += __shfl_xor(s, o, 64); dp += __shfl_xor(dp, o, 64); }
        const bool live = ok && !(causal && j > qp);
        const float p = live ? __expf(s * SCALE - l0) : 0.f;
        // the MFMA path rounds P and dS to the operand type before the second products; same points here
        const float pb = op_to_f32(f32_to_op(p));
        const float ds = live ? op_to_f32(f32_to_op(p * (dp - delta) * TTL_DS_PRESCALE)) * (1.0f / TTL_DS_PRESCALE) : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dq[e] = fmaf(ds, (float)kf[e], dq[e]);
        if (ok) {
            op_t* o = dqkv + (size_t)(img * T + j) * ldd + head * 64 + 8 * c;
            u32x4 dv, dk;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dv[e] = pack_op2(pb * dc[2 * e], pb * dc[2 * e + 1]);
                dk[e] = pack_op2(ds * qc[2 * e] * SCALE, ds * qc[2 * e + 1] * SCALE);
            }
            *(u32x4*)(o + 2 * D) = dv;
            if (NEED_DK) *(u32x4*)(o + D) = dk;
            if (j != qp) *(u32x4*)o = u32x4{0u, 0u, 0u, 0u};
        }
    }
    // dq_qp[8c+e] = sum over the 32 key groups
#pragma unroll
    for (int e = 0; e < 8; ++e) sred[grp][8 * c + e] = dq[e];
    __syncthreads();
    if (tid < 64) {
        float v = 0.f;
#pragma unroll 8
        for (int g = 0; g < 32; ++g) v += sred[g][tid];
        dqkv[((size_t)img * T + qp) * ldd + head * 64 + tid] = f32_to_op(v * SCALE);
    }
}


template <int NKT, int CH>
hipError_t fwd_w(const op_t* qkv, QkvLayout ld, op_t* out, int ldo, float* lse, int n, int T, int H, hipStream_t s, int causal) {
    constexpr int SMEM = 2 * NKT * 32 * 128;
    static std::atomic<uint64_t> done{0};
    if (hipError_t e = ensure_smem((const void*)attn_fwd_w_kernel<NKT, CH>, SMEM, done); e != hipSuccess) return e;
    hipLaunchKernelGGL((attn_fwd_w_kernel<NKT, CH>), dim3(n * H), dim3(64 * NKT), SMEM, s, qkv, ld, out, ldo, lse, T, H, causal);
    return hipGetLastError();
}

template <int NKT, int CH>
hipError_t fwd_p(const op_t* qkv, QkvLayout ld, op_t* out, int ldo, float* lse, int n, int T, int H, hipStream_t s) {
    constexpr int SMEM = 5 * NKT * 32 * 128;     // two (K, V) pairs + q
    static std::atomic<uint64_t> done{0};
    if (hipError_t e = ensure_smem((const void*)attn_fwd_p_kernel<NKT, CH>, SMEM, done); e != hipSuccess) return e;
    const int cus = device_cu_count();
    if (!cus) return hipErrorInvalidDevice;
    const int nprob = n * H;
    const int grid = nprob < cus ? nprob : cus;
    static const int xmap = TTL_EXPERIMENT("TTL_ATTN_XCD_MAP", 0);
    hipLaunchKernelGGL((attn_fwd_p_kernel<NKT, CH>), dim3(grid), dim3(64 * NKT), SMEM, s, qkv, ld, out, ldo, lse, T, H, nprob, xmap);
    return hipGetLastError();
}

template <int NKT>
hipError_t fwd_s(const op_t* qkv, QkvLayout ld, op_t* out, int ldo, float* lse, int n, int T, int H, hipStream_t s) {
    constexpr int SMEM = 5 * NKT * 32 * 128;     // two (K, V) pairs + q
    static std::atomic<uint64_t> done{0};
    if (hipError_t e = ensure_smem((const void*)attn_fwd_s_kernel<NKT>, SMEM, done); e != hipSuccess) return e;
    const int cus = device_cu_count();
    if (!cus) return hipErrorInvalidDevice;
    const int nprob = n * H;
    const int grid = nprob < cus ? nprob : cus;
    hipLaunchKernelGGL((attn_fwd_s_kernel<NKT>), dim3(grid), dim3(64 * NKT), SMEM, s, qkv, ld, out, ldo, lse, T, H, nprob);
    return hipGetLastError();
}

template <int NKT>
hipError_t fwd_t(const op_t* qkv, QkvLayout ld, op_t* out, int ldo, float* lse, int n, int T, int H, hipStream_t s, int causal) {
    constexpr int SMEM = 2 * NKT * 32 * 128;
    static std::atomic<uint64_t> done{0};
    if (hipError_t e = ensure_smem((const void*)attn_fwd_kernel<NKT>, SMEM, done); e != hipSuccess) return e;
    hipLaunchKernelGGL((attn_fwd_kernel<NKT>), dim3(n * H), dim3(256), SMEM, s, qkv, ld, out, ldo, lse, T, H, causal);
    return hipGetLastError();
}

#ifndef TTL_ATTN_NW_DQ
#define TTL_ATTN_NW_DQ 0
#endif
#ifndef TTL_ATTN_NW_DKV
#define TTL_ATTN_NW_DKV 0
#endif
template <int NKT>
hipError_t bwd_t(const op_t* qkv, QkvLayout ld, const op_t* out, const op_t* dout, int ldo, const float* lse,
                 op_t* dqkv, int ldd, int n, int T, int H, int need_dk, hipStream_t s, int causal) {
    // waves per (sequence, head) block: 0 = one per 32-row block (NKT waves; used), or a fixed count.  In situ on
    // ViT-B/16 (tools/attn_ab.sh, attention-backward ms per episode): 4/4 waves 0.273, NKT/4 0.229, 8/8 0.222, NKT/NKT 0.205
    constexpr int NWQ = (TTL_ATTN_NW_DQ == 0) ? NKT : TTL_ATTN_NW_DQ;
    constexpr int NWK = (TTL_ATTN_NW_DKV == 0) ? NKT : TTL_ATTN_NW_DKV;
    constexpr int SMEM_A = 2 * NKT * 32 * 128;
    constexpr int SMEM_B = 2 * NKT * 32 * 128 + 2 * NKT * 32 * 4;
    static std::atomic<uint64_t> done_q{0}, done_k{0}, done_v{0};
    hipError_t e = ensure_smem((const void*)attn_bwd_dq_kernel<NKT, NWQ>, SMEM_A, done_q);
    if (e == hipSuccess) e = ensure_smem((const void*)attn_bwd_dkv_kernel<NKT, true, NWK>, SMEM_B, done_k);
    if (e == hipSuccess) e = ensure_smem((const void*)attn_bwd_dkv_kernel<NKT, false, NWK>, SMEM_B, done_v);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<NKT, NWQ>), dim3(n * H), dim3(64 * NWQ), SMEM_A, s, qkv, ld, out, dout, ldo, lse, dqkv,
                       ldd, T, H, causal);
    if (need_dk)
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<NKT, true, NWK>), dim3(n * H), dim3(64 * NWK), SMEM_B, s, qkv, ld, out, dout, ldo,
                           lse, dqkv, ldd, T, H, causal);
    else
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<NKT, false, NWK>), dim3(n * H), dim3(64 * NWK), SMEM_B, s, qkv, ld, out, dout, ldo,
                           lse, dqkv, ldd, T, H, causal);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_attention_fwd(const op_t* qkv, QkvLayout ld_qkv, op_t* out, int ld_out, float* lse, int n, int T, int H,
                                hipStream_t s, int causal) {
#ifdef TTL_DIAG_SKIP
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 1) && n * H >= 512 && diag_skip_now(cnt, 440)) return hipSuccess; }
#endif
    int nkt = (T + 31) / 32;
    if (nkt <= 1) return fwd_t<1>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s, causal);
    if (nkt <= 2) return fwd_t<2>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s, causal);
    if (nkt == 3) return fwd_w<3, 3>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s, causal);   // text tower: T = 77
    if (nkt <= 4) return fwd_t<4>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s, causal);
    static const int variant = TTL_EXPERIMENT("TTL_ATTN_VARIANT", 4);
    if (variant == 5 && !causal && n * H >= 512 && nkt == 7)      // persistent + per-tile software pipeline (round 4)
        return fwd_s<7>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s);
    if (variant == 4 && !causal && n * H >= 512) {     // persistent, K/V of the next problem prefetched (big launches only)
        if (nkt == 7) return fwd_p<7, 7>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s);     // T = 257: five tiles of 36 KiB do not fit
    }
    if (variant >= 1) {
        if (nkt == 7 && variant == 2) return fwd_w<7, 1>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s, causal);
        if (nkt == 7 && variant == 3) return fwd_w<7, 3>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s, causal);
        if (nkt == 7) return fwd_w<7, 2>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s, causal);
        if (nkt == 9) return fwd_w<9, 2>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s, causal);
    }
    if (nkt <= 7) return fwd_t<7>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s, causal);
    if (nkt <= 9) return fwd_t<9>(qkv, ld_qkv, out, ld_out, lse, n, T, H, s, causal);
    return hipErrorInvalidValue;
}

hipError_t launch_attention_bwd(const op_t* qkv, QkvLayout ld_qkv, const op_t* out, const op_t* dout, int ld_o,
                                const float* lse, op_t* dqkv, int ld_dqkv, int n, int T, int H, int need_dk,
                                hipStream_t s, int causal) {
#ifdef TTL_DIAG_SKIP
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 2) && n * H >= 512 && diag_skip_now(cnt, 80)) return hipSuccess; }
#endif
    int nkt = (T + 31) / 32;
    if (nkt <= 1) return bwd_t<1>(qkv, ld_qkv, out, dout, ld_o, lse, dqkv, ld_dqkv, n, T, H, need_dk, s, causal);
    if (nkt <= 2) return bwd_t<2>(qkv, ld_qkv, out, dout, ld_o, lse, dqkv, ld_dqkv, n, T, H, need_dk, s, causal);
    if (nkt <= 4) return bwd_t<4>(qkv, ld_qkv, out, dout, ld_o, lse, dqkv, ld_dqkv, n, T, H, need_dk, s, causal);
    if (nkt <= 7) return bwd_t<7>(qkv, ld_qkv, out, dout, ld_o, lse, dqkv, ld_dqkv, n, T, H, need_dk, s, causal);
    if (nkt <= 9) return bwd_t<9>(qkv, ld_qkv, out, dout, ld_o, lse, dqkv, ld_dqkv, n, T, H, need_dk, s, causal);
    return hipErrorInvalidValue;
}

hipError_t launch_attention_fwd_cls(const op_t* qkv, QkvLayout ld_qkv, op_t* out, int ld_out, float* lse, int n, int T, int H,
                                    hipStream_t s, const int* qpos, int causal) {
    if (T > 320) return hipErrorInvalidValue;
    const int nit = (T + 31) / 32;
#define FWD_CLS(N_) hipLaunchKernelGGL(attn_fwd_cls_kernel<N_>, dim3(n * H), dim3(256), 0, s, qkv, ld_qkv, out, ld_out, lse, T, H, qpos, causal)
    if (nit <= 3) FWD_CLS(3); else if (nit <= 7) FWD_CLS(7); else if (nit <= 9) FWD_CLS(9); else FWD_CLS(0);
#undef FWD_CLS
    return hipGetLastError();
}

hipError_t launch_attention_bwd_cls(const op_t* qkv, QkvLayout ld_qkv, const op_t* out, int ld_o, const op_t* dout_cls,
                                    const float* lse, op_t* dqkv, int ld_dqkv, int n, int T, int H, int need_dk,
                                    hipStream_t s, const int* qpos, int causal) {
    const int nit = (T + 31) / 32;
#define BWD_CLS(DK_, N_) hipLaunchKernelGGL((attn_bwd_cls_kernel<DK_, N_>), dim3(n * H), dim3(256), 0, s, qkv, ld_qkv, out, ld_o, dout_cls, lse, dqkv, \
                                           ld_dqkv, T, H, qpos, causal)
    if (need_dk) { if (nit <= 3) BWD_CLS(true, 3); else if (nit <= 7) BWD_CLS(true, 7); else if (nit <= 9) BWD_CLS(true, 9); else BWD_CLS(true, 0); }
    else { if (nit <= 3) BWD_CLS(false, 3); else if (nit <= 7) BWD_CLS(false, 7); else if (nit <= 9) BWD_CLS(false, 9); else BWD_CLS(false, 0); }
#undef BWD_CLS
    return hipGetLastError();
}
