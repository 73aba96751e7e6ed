// Big-M bf16 MFMA GEMM, 256-column tiles:  C[M,N] = A[M,K] · B[N,K]^T  (+ fused epilogue), M >= 1024.
//
// Same products as gemm.hip (HF modeling_clip.py:309-311 q/k/v, :333 out_proj, :346-350 fc1/fc2 and their
// dgrad counterparts); this kernel exists because the 160x128 tile of gemm.hip stages 1 byte of operand per
// 71 FLOP and the L2 -> LDS path (one TA per CU) is what bounds it.  Here:
//   * block tile (32*MT) x 256 x 64, 8 waves as 2 (M) x 4 (N), every wave owns (16*MT) x 64 outputs;
//     MT = 5: 160 x 256 -> 98 FLOP per staged byte, MT = 7: 224 x 256 -> 119
//   * ONE block per CU (512 threads); LDS ring of STAGES K-tiles (MT = 5: 3 x 52 KiB = 156 of the 160 KiB),
//     filled by global_load_lds_dwordx4 behind a COUNTED s_waitcnt vmcnt(N) + raw s_barrier: the DMA of
//     K-tile kt+2 is in flight while kt is multiplied and kt+1 lands (guide §5 "Pipelining across barriers")
//   * fragments are double-buffered in registers by 32-deep K half: the LDS reads of one half are issued
//     between the MFMAs of the other (sched_group_barrier), one barrier per K-tile
//   * LDS image, swizzle, weight-row permutation and the epilogue are those of gemm.hip (gemm_epilogue.hpp)
//   * bias / residual folded into the accumulators' initial value (bias through a 1 KiB LDS-DMA piece ahead of the tile's first
//     K-tile, residual read at the top of the tile): the epilogue issues stores only
//   * persistent tile loop (grid = min(tiles, resident slots)): the first two K-tiles of the NEXT tile are
//     requested before the epilogue of the current one, so a tile's prologue latency hides under the stores
//   * tile order: the 8 XCDs own contiguous row-panel ranges; inside an XCD column-major (all row panels of a
//     256-column weight slice before the next slice: slice + panels fit the 4 MiB L2) or row-major
//
// Tried against this kernel and measured slower (kept out of the tree):
//   * the same 160 x 256 tile as TWO independent 4-wave blocks per CU (each wave 160 x 64, 32-deep K-tiles so that a 3-stage ring
//     fits 78 KiB per block, conflict-free 64-B-row LDS image): meant to hide one block's epilogue under the other's MFMAs;
//     QKV 56.9 vs 43.8 us, fc2 89 vs 65 us on cold operands, 3.30 vs 2.82 ms of GEMM per episode in situ — the speed of round 1's
//     160 x 128 kernel.  Half-line (64 B per row) DMA pieces and twice the barriers per K cost more than the asynchrony buys.
//   * 224 x 256 and 256 x 256 tiles (two LDS stages only), 160 x 256 with two stages: -12 ... -15 %.  Repeated with a spill-free
//     256 x 256 variant (one fragment set instead of two, no scratch traffic in the K loop; 128 FLOP per staged byte instead of
//     98): QKV 52 vs 43 us, fc1 75 vs 62 us on cold operands, GEMM time per episode 2.85 vs 2.77 ms.  With two stages the next
//     K-tile can only be requested once the barrier has released the stage just read, so the CU's L1 request pipeline drains at
//     every K-step (12 B/clk staged instead of 17.7 with a third stage keeping requests queued): the ring depth matters more
//     than the bytes per FLOP, and 160 KiB of LDS hold three stages only up to 160 + 256 rows.
//   * a 256 x 256 tile whose third stage lives in REGISTERS (K-tile s+2 requested with ordinary global loads, 32 VGPRs per lane,
//     written into the LDS stage just read behind the step's barrier; one flat (tile, K-tile) pipeline, 220 VGPRs, no spills):
//     QKV 57-59 us (160 x 256 ring: 43), dO 30 vs 19 us: the VGPR -> LDS hop (8 ds_write_b128 per lane and K-step beside the
//     fragment reads) costs more than the extra stage and the 30 % fewer staged bytes buy.
//   * bf16 outputs accumulated transposed (MFMA(w, x), weight rows permuted so a lane owns 8 consecutive columns of one row):
//     half as many store instructions (16 B each, 16 rows x 64 B per instruction instead of 4 rows x 128 B): GEMM time per
//     episode 2.88 vs 2.76 ms in situ, 277 vs 285 images/s — the epilogue is bound by lines touched, not by store instructions;
//     whole 128-B lines per row stay.
//   * starting half of the blocks 1-4 us late (de-synchronising the epilogue store bursts): the delay is simply exposed.
//   * (round 5) what the vendor library runs on these shapes: rocprofv3 of torch.matmul shows hipBLASLt serving N >= 2304, K = 768 with
//     MT256x256x64 on FOUR waves (128 x 128 per wave, accumulators in AGPRs, global -> VGPR prefetch two K-tiles deep, stream-K) at
//     46.5 us on the QKV shape against 52-54 us here, i.e. 15-18 % less CU-time per launch — and with three episodes in flight CU-time
//     per tile is what a launch costs, not its round count (a grid of 224 blocks: +23 % GEMM time one at a time, same images/s).
//     A first HIP-source cut of that design (256 threads, compiler-scheduled ds_read / ds_write / MFMA, one K-tile of prefetch
//     registers) measured 106 us on the QKV shape; hand-pipelined it reached 57 us, and with LDS-DMA rings instead of staging registers
//     49.4 us (this kernel 52.9, library 48.2 on that lease): that third cut is gemm_huge.hip and takes the q/k/v launches
//     (+1.1 % images/s in situ; fc1 gains nothing there and stays here) — profiles/r05_experiments.txt r05i-r05n.
#include <stdlib.h>

#include <atomic>

#include "gemm_epilogue.hpp"
#include "kernels.hpp"

namespace {

constexpr int BK = 64;
constexpr int BN = 256;
constexpr int NTHR = 512;

// Diagnostic ablations (tools/gemm_big_ablate.sh builds libttl_hip_bdiagN.so; results are WRONG on purpose, only the
// timing is read): 1 = no DMA in the steady loop, 2 = no MFMA, 3 = LDS fragment reads from a fixed address (hoisted),
// 4 = no barrier, 5 = no epilogue stores.  The product build has TTL_GEMM_DIAG == 0 and none of this exists in it.
#ifndef TTL_GEMM_DIAG
#define TTL_GEMM_DIAG 0
#endif
#if TTL_GEMM_DIAG == 4
#define BARRIER() ((void)0)
#else
#define BARRIER() __builtin_amdgcn_s_barrier()
#endif

// NS slots of {one non-MFMA op, then its share of the NM MFMAs}; the first NV slots are VMEM (the DMA pieces:
// requested as early as possible), the rest LDS reads
// Build-time schedule knobs (tools/gemm_big_ablate.sh NAME=VALUE builds variants for in-situ A/B):
//   TTL_BIG_SCHED  0 = pinned interleave below, 1 = the compiler's own order
//   TTL_BIG_MFIRST 1 = behind the barrier the MFMAs lead and the LDS reads follow (hipcc puts an lgkmcnt(0) in front of
//                  the first MFMA after a barrier whatever it reads; with a read already issued that wait costs its latency)
//   TTL_BIG_PRIO   1 = s_setprio 1 around the MFMA halves
#ifndef TTL_BIG_SCHED
#define TTL_BIG_SCHED 0
#endif
#ifndef TTL_BIG_MFIRST
#define TTL_BIG_MFIRST 1
#endif
#ifndef TTL_BIG_PRIO
#define TTL_BIG_PRIO 0
#endif
#ifndef TTL_BIG_EPI_OVERLAP
#define TTL_BIG_EPI_OVERLAP 1
#endif
template <int S, int NS, int NV, int NM, bool MFIRST = false>
struct Mix {
    static __device__ __forceinline__ void run() {
        if constexpr (TTL_BIG_SCHED == 0 && S < NS) {
            constexpr int m = NM / NS + (S < NM % NS ? 1 : 0);
            if constexpr (MFIRST && m > 0) __builtin_amdgcn_sched_group_barrier(0x008, m, 0);
            __builtin_amdgcn_sched_group_barrier(S < NV ? 0x010 : 0x100, 1, 0);
            if constexpr (!MFIRST && m > 0) __builtin_amdgcn_sched_group_barrier(0x008, m, 0);
            Mix<S + 1, NS, NV, NM, MFIRST>::run();
        }
    }
};

// s_waitcnt vmcnt(N) lgkmcnt(0) through the builtin (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]),
// so hipcc's own wait-count pass knows the LDS reads are retired and does not add a second wait behind the barrier
template <int N>
__device__ __forceinline__ void wait_dma() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

// Epilogue of this kernel: bias and residual were folded into the accumulators' initial value, so the only memory
// operations here are stores (an ordinary load next to the next tile's in-flight LDS-DMA makes hipcc wait vmcnt(0),
// which drains that prefetch).  Register r of sub-tile (mt, nt) is row 16*mt + 4*lg + r, column 4*li + nt.
constexpr int EPI_GELU_C2 = 100;   // internal: EPI_GELU with the second (pre-activation) output, kept branch-free
constexpr int EPI_OP_HM = 101;     // internal: EPI_OP into a head-major q/k/v buffer (GemmArgs::hm_T, kernels.hpp QkvLayout)
constexpr int EPI_GELU_BWD_BIG = 105;   // EPI_GELU_BWD (MLP dgrad: C = product * quick_gelu'(u)); u of the tile is read at the TOP of the tile
                                        // into 8 * MT registers per lane and multiplied in at the end: 128-row tiles (MT = 4) keep that spill-free

// Head-major q/k/v (EPI_OP_HM): the lane's 4 columns n0..n0+3 lie in one head, so their offset inside a view's block is a
// per-tile constant (hm_col_base, computed once per tile) and a row adds view*3*D*T + t*64 with view = m / T by a multiply-high
// (magic checked on the host for every row a launch can store).  Per (mt, r) a 16-lane group still writes one whole 128-B
// line — row t of the (view, plane, head) tile — and consecutive tokens are now ADJACENT lines.
__device__ __forceinline__ size_t hm_col_base(const GemmArgs& a, int n0) {
    const int Dm = a.N / 3;
    const int plane = (n0 >= Dm) + (n0 >= 2 * Dm);
    const int rem = n0 - plane * Dm;
    return ((size_t)plane * Dm + (size_t)(rem & ~63)) * a.hm_T + (rem & 63);
}

template <int EPI, int MT>
__device__ __forceinline__ void big_epilogue_row(const GemmArgs& a, const f32x4 (&acc)[MT][4], int mt, int rbase, size_t n0, int lg,
                                                 const u32x2 (*auxr)[4] = nullptr) {
    {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t m = (size_t)(rbase + mt * 16 + 4 * lg + r);
            float v0 = acc[mt][0][r], v1 = acc[mt][1][r], v2 = acc[mt][2][r], v3 = acc[mt][3][r];
            if constexpr (EPI == EPI_OP_HM) {
                const unsigned view = __umulhi((unsigned)m, a.hm_magic), t = (unsigned)m - view * (unsigned)a.hm_T;
                st_out((u32x2*)((op_t*)a.C + (size_t)view * a.N * a.hm_T + n0 + (size_t)t * 64), u32x2{pack_op2(v0, v1), pack_op2(v2, v3)});
            } else if constexpr (EPI == EPI_F32 || EPI == EPI_RESID_F32) {
                st_out((f32x4*)((float*)a.C + m * a.ldc + n0), f32x4{v0, v1, v2, v3});
            } else {
                if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_C2) {
                    if constexpr (EPI == EPI_GELU_C2) __builtin_nontemporal_store(u32x2{pack_op2(v0, v1), pack_op2(v2, v3)}, (u32x2*)(a.C2 + m * a.ldc2 + n0));
                    v0 = quick_gelu_f(v0); v1 = quick_gelu_f(v1); v2 = quick_gelu_f(v2); v3 = quick_gelu_f(v3);
                    __builtin_nontemporal_store(u32x2{pack_op2(v0, v1), pack_op2(v2, v3)}, (u32x2*)((op_t*)a.C + m * a.ldc + n0));
                } else {
                    if constexpr (EPI == EPI_GELU_BWD_BIG) {
                        const u32x2 t = auxr[mt][r];
                        v0 *= quick_gelu_grad_f(op_lo(t[0])); v1 *= quick_gelu_grad_f(op_hi(t[0]));
                        v2 *= quick_gelu_grad_f(op_lo(t[1])); v3 *= quick_gelu_grad_f(op_hi(t[1]));
                    }
                    st_out((u32x2*)((op_t*)a.C + m * a.ldc + n0), u32x2{pack_op2(v0, v1), pack_op2(v2, v3)});
                }
            }
        }
    }
}

template <int EPI, int MT>
__device__ __forceinline__ void big_epilogue(const GemmArgs& a, const f32x4 (&acc)[MT][4], int rbase, size_t n0, int lg,
                                             const u32x2 (*auxr)[4] = nullptr) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) big_epilogue_row<EPI, MT>(a, acc, mt, rbase, n0, lg, auxr);
}

struct TileMap {
    int ntm, ntn;     // row tiles, column tiles
    int order;        // 0: 1-D XCD chunks, n fastest; 1: XCD row ranges, column-major inside; 2: XCD row ranges, row-major
    int ntiles;       // order 0: ntm*ntn; else 8 * max_x(rows_x * ntn) slots (some empty)
};

// slot -> (row tile, column tile); false: the slot is empty
__device__ __forceinline__ bool tile_of(const TileMap& tm, int slot, int& rt, int& ct) {
    if (tm.order == 0) {
        const int t = xcd_remap(slot, tm.ntiles);
        rt = t / tm.ntn; ct = t - rt * tm.ntn;
        return true;
    }
    const int x = slot & 7, j = slot >> 3;
    const int r0 = x * tm.ntm / 8, r1 = (x + 1) * tm.ntm / 8, nr = r1 - r0;
    if (j >= nr * tm.ntn) return false;
    if (tm.order == 1) { ct = j / nr; rt = r0 + (j - ct * nr); }
    else { const int jr = j / tm.ntn; rt = r0 + jr; ct = j - jr * tm.ntn; }
    return true;
}

#ifdef TTL_CLOCK_STAMPS
__device__ TtlClockStamp g_big_stamps[TTL_STAMP_SLOTS];
#endif

template <int MT, int STAGES, int EPI>
__global__ __launch_bounds__(NTHR, 2) void gemm_big_kernel(const GemmArgs a, const TileMap tm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(STAGES == 2 || STAGES == 3, "ring depth");
    constexpr int BM = 32 * MT, WM = 16 * MT;
    constexpr int NAP = BM / 8;               // 1 KiB DMA pieces (one wave instruction: 8 rows x 128 B) of the A tile
    constexpr int NPIECE = NAP + BN / 8;
    constexpr int NPW = (NPIECE + 7) / 8;     // pieces of the waves that carry the most
    constexpr int NFULL = NPIECE % 8;         // waves [0, NFULL) carry NPW pieces, the others NPW - 1 (0: all NPW)
    constexpr int NUNI = NFULL ? NPW - 1 : NPW;   // pieces every wave carries
    constexpr int A_BYTES = BM * 128;
    constexpr int STAGE = A_BYTES + BN * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 15, lg = lane >> 4;
    const int M = a.M;
    const int nk = a.K / BK;

    // ---- fragment addresses: row (tile row + li) of a 128-B-per-row image, 16-B chunk (4s + lg) ^ swizzle(row)
    const int swA = (li >> 1) & 7;
    const int fA0 = (wm * WM + li) * 128 + ((lg ^ swA) << 4);
    const int fW0 = A_BYTES + (wn * 64 + li) * 128 + ((lg ^ swA) << 4);
    auto load_frags = [&](const char* sb, int s, opx8 (&xf)[MT], opx8 (&wf)[4]) {
        if (TTL_GEMM_DIAG == 3) sb = smem;
        const char* pa = sb + (fA0 ^ (s << 6));
        const char* pw = sb + (fW0 ^ (s << 6));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[mt] = *(const opx8*)(pa + mt * 2048);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const opx8*)(pw + nt * 2048);
    };

    // ---- DMA pieces of this wave: piece j = i*8 + wave covers image rows 8j' .. 8j'+7 of A (j < NAP) or B
    uint32_t poff[NPW];
    const char* pbase[NPW];
    int plds[NPW];
    auto setup = [&](int row0, int col0) {
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int j = i * 8 + wave;
            const int jj = (j < NAP) ? j : j - NAP;
            const int r = jj * 8 + (lane >> 3), p = lane & 7;
            const int c = p ^ ((r >> 1) & 7);
            if (j < NAP) {
                const int gr = min(row0 + r, M - 1);
                poff[i] = (uint32_t)(((size_t)gr * a.lda + c * 8) * sizeof(op_t));
                pbase[i] = (const char*)a.A;
                plds[i] = j * 1024;
            } else {
                const int n = (r & ~63) + 4 * (r & 15) + ((r >> 4) & 3);   // physical LDS row r holds this output column
                poff[i] = (uint32_t)(((size_t)(col0 + n) * a.ldb + c * 8) * sizeof(op_t));
                pbase[i] = (const char*)a.B;
                plds[i] = A_BYTES + jj * 1024;
            }
        }
    };
    auto issue_odd = [&](int kt, char* sb) {    // the piece only waves [0, NFULL) carry: wave-uniform branch, kept out of
        if constexpr (NFULL != 0) {            // the scheduled region
            if (TTL_GEMM_DIAG == 1 && kt >= 2) return;
            if (wave < NFULL)
                __builtin_amdgcn_global_load_lds(GLB_PTR(pbase[NPW - 1] + (size_t)kt * (BK * sizeof(op_t)) + poff[NPW - 1]),
                                                 LDS_PTR(sb + plds[NPW - 1]), 16, 0, 0);
        }
    };
    auto issue_uni = [&](int kt, char* sb) {
        if (TTL_GEMM_DIAG == 1 && kt >= 2) return;
#pragma unroll
        for (int i = 0; i < NUNI; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(pbase[i] + (size_t)kt * (BK * sizeof(op_t)) + poff[i]), LDS_PTR(sb + plds[i]), 16, 0,
                                             0);
    };
    // wait until at most `tiles` of this wave's K-tiles of DMA are still in flight (+ all LDS reads retired)
    auto wait_tiles1 = [&]() {
        if constexpr (NFULL != 0) { if (wave < NFULL) wait_dma<NPW>(); else wait_dma<NPW - 1>(); }
        else wait_dma<NPW>();
    };
    // the same while the previous tile's epilogue stores (at least NST of them, issued AFTER the K-tile waited for and
    // before the one left in flight) may still be unacknowledged: vmcnt counts loads and stores together, in order
    // (+ the 4*MT ordinary loads the MLP-dgrad tile issues at its top for u: younger still, and counting them keeps this wait from
    // also retiring K-tile 1 and the stores.  The residual tile's loads are not counted: their values are needed right behind the
    // barrier anyway.)
    constexpr int NST = 4 * MT + (EPI == EPI_GELU_BWD_BIG ? 4 * MT : 0);
    static_assert(NPW + NST < 64, "vmcnt is a 6-bit counter");
    auto wait_tiles1_st = [&]() {
        if constexpr (NFULL != 0) { if (wave < NFULL) wait_dma<NPW + NST>(); else wait_dma<NPW - 1 + NST>(); }
        else wait_dma<NPW + NST>();
    };
    // bias of the tile's 256 columns: ONE LDS-DMA piece (wave 4, which carries one piece fewer than waves 0-3) into the KiB
    // behind the ring, requested BEFORE the tile's first K-tile, so every counted wait that retires K-tile 0 retires it too;
    // the lanes read their 4 columns behind the top-of-tile barrier.  (An ordinary load next to in-flight LDS-DMA makes hipcc
    // wait vmcnt(0); a VGPR-destination inline-asm load is only order-pinned, not allocation-pinned — guide §5.7 item 1.)
    char* const sbias = smem + STAGES * STAGE;
    auto fetch_bias = [&](int col0) {
        if (a.bias && wave == 4)
            __builtin_amdgcn_global_load_lds(GLB_PTR((const char*)(a.bias + col0) + lane * 16), LDS_PTR(sbias), 16, 0, 0);
    };

    f32x4 acc[MT][4];
    opx8 xf0[MT], wf0[4], xf1[MT], wf1[4];
    auto mma = [&](const opx8 (&xf)[MT], const opx8 (&wf)[4]) {
        if (TTL_GEMM_DIAG == 2) {   // keep the LDS reads alive without the matrix pipe
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(xf[mt]));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) asm volatile("" ::"v"(wf[nt]));
            return;
        }
        if (TTL_BIG_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MFMA16(xf[mt], wf[nt], acc[mt][nt], 0, 0, 0);
        if (TTL_BIG_PRIO) __builtin_amdgcn_s_setprio(0);
    };

    int slot = blockIdx.x;
    int rt, ct;
    bool have = tile_of(tm, slot, rt, ct);
    while (!have) {     // (empty slots of an uneven XCD row split)
        slot += gridDim.x;
        if (slot >= tm.ntiles) return;
        have = tile_of(tm, slot, rt, ct);
    }
    int row0 = rt * BM, col0 = ct * BN;
    TTL_STAMP_DECL;
    TTL_STAMP_BEGIN();
    setup(row0, col0);
    char* s0 = smem;
    char* s1 = smem + STAGE;
    char* s2 = smem + (STAGES == 3 ? 2 * STAGE : 0);
    // ring prologue of the first tile
    fetch_bias(col0);
    issue_odd(0, s0); issue_uni(0, s0);
    if constexpr (STAGES == 3) { issue_odd(1, s1); issue_uni(1, s1); }

    bool first = true;
    for (;;) {
        // residual tile (EPI_RESID_F32): read at the TOP of the tile, where it overlaps the DMA prologue, instead of at the
        // end where all blocks of a launch would read theirs at once with the matrix pipe idle
        f32x4 rsd[EPI == EPI_RESID_F32 ? MT : 1][4];
        if constexpr (EPI == EPI_RESID_F32) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    rsd[mt][r] = *(const f32x4*)(a.resid + (size_t)(row0 + wm * WM + mt * 16 + 4 * lg + r) * a.ldr + col0 + wn * 64 + 4 * li);
        }
        u32x2 auxr[EPI == EPI_GELU_BWD_BIG ? MT : 1][4];
        if constexpr (EPI == EPI_GELU_BWD_BIG) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    auxr[mt][r] = *(const u32x2*)(a.aux + (size_t)(row0 + wm * WM + mt * 16 + 4 * lg + r) * a.ldaux + col0 + wn * 64 + 4 * li);
        }
        char *cur = s0, *nxt = s1, *nn = s2;
        // K-tile 0 landed (STAGES == 3: K-tile 1 may still fly), and with it the bias piece (older than K-tile 0)
        if constexpr (STAGES == 3) { if (first) wait_tiles1(); else wait_tiles1_st(); }
        else { if (first) wait_dma<0>(); else wait_dma<NST>(); }
        BARRIER();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (STAGES == 2) { issue_odd(1, nxt); issue_uni(1, nxt); }
        f32x4 biasv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias) biasv = *(const f32x4*)(sbias + (wn * 64 + 4 * li) * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if constexpr (EPI == EPI_RESID_F32)
                    acc[mt][nt] = f32x4{rsd[mt][0][nt] + biasv[nt], rsd[mt][1][nt] + biasv[nt], rsd[mt][2][nt] + biasv[nt], rsd[mt][3][nt] + biasv[nt]};
                else
                    acc[mt][nt] = f32x4{biasv[nt], biasv[nt], biasv[nt], biasv[nt]};
            }
        load_frags(cur, 0, xf0, wf0);
        TTL_STAMP_K0(first);

        // ---- steady state: step kt multiplies K-tile kt; requests kt+2 (3 stages: at the top, into the stage freed by
        // barrier kt-1; 2 stages: behind barrier kt, into the stage just read); one barrier per step
        for (int kt = 0; kt + 2 < nk; ++kt) {
            if constexpr (STAGES == 3) {
                issue_odd(kt + 2, nn);
                issue_uni(kt + 2, nn);
            }
            load_frags(cur, 1, xf1, wf1);
            mma(xf0, wf0);
            if constexpr (STAGES == 3) Mix<0, NUNI + MT + 4, NUNI, 4 * MT>::run();
            else Mix<0, MT + 4, 0, 4 * MT>::run();
            __builtin_amdgcn_sched_barrier(0);
            // K-tile kt+1 landed; h1 reads of kt retired
            if constexpr (STAGES == 3) { if (kt == 0 && !first) wait_tiles1_st(); else wait_tiles1(); }
            else wait_dma<0>();
            BARRIER();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (STAGES == 2) { issue_odd(kt + 2, cur); issue_uni(kt + 2, cur); }
            load_frags(nxt, 0, xf0, wf0);
            mma(xf1, wf1);
            if constexpr (STAGES == 3) Mix<0, MT + 4, 0, 4 * MT, TTL_BIG_MFIRST != 0>::run();
            else Mix<0, NUNI + MT + 4, NUNI, 4 * MT>::run();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (STAGES == 3) { char* t = cur; cur = nxt; nxt = nn; nn = t; }
            else { char* t = cur; cur = nxt; nxt = t; }
        }
        // ---- step nk-2: nothing left to request for this tile
        {
            load_frags(cur, 1, xf1, wf1);
            mma(xf0, wf0);
            Mix<0, MT + 4, 0, 4 * MT>::run();
            __builtin_amdgcn_sched_barrier(0);
            wait_dma<0>();
            BARRIER();
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- next tile of this block: request its first K-tiles now (every stage but `nxt` is free behind the barrier
        // above: `cur` was read before it, `nn` two steps ago), so they land under step nk-1 and the epilogue
        int nslot = slot + gridDim.x, nrt = 0, nct = 0;
        bool more = false;
        while (nslot < tm.ntiles && !(more = tile_of(tm, nslot, nrt, nct))) nslot += gridDim.x;
        const int erow0 = row0, ecol0 = col0;
        if (more) {
            row0 = nrt * BM; col0 = nct * BN;
            setup(row0, col0);
            fetch_bias(col0);
            issue_odd(0, cur); issue_uni(0, cur);
            if constexpr (STAGES == 3) { issue_odd(1, nn); issue_uni(1, nn); }
        }
        // ---- step nk-1, row-major: the last K-tile's eight MFMAs of output row block mt, then that row block's epilogue
        // (pack / activation / stores) while the next row block multiplies (TTL_BIG_EPI_OVERLAP: 0 = all MFMAs, then the
        // whole epilogue with the matrix pipe idle)
        {
            load_frags(nxt, 0, xf0, wf0);
            mma(xf1, wf1);
            Mix<0, MT + 4, 0, 4 * MT>::run();
            __builtin_amdgcn_sched_barrier(0);
            load_frags(nxt, 1, xf1, wf1);
            size_t ecol = (size_t)(ecol0 + wn * 64 + 4 * li);      // column argument of the epilogue (EPI_OP_HM: offset inside a view's block)
            if constexpr (EPI == EPI_OP_HM) ecol = hm_col_base(a, (int)ecol);
            if constexpr (TTL_BIG_EPI_OVERLAP && TTL_GEMM_DIAG == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MFMA16(xf0[mt], wf0[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MFMA16(xf1[mt], wf1[nt], acc[mt][nt], 0, 0, 0);
                    if (mt > 0) big_epilogue_row<EPI, MT>(a, acc, mt - 1, erow0 + wm * WM, ecol, lg, auxr);
                }
                big_epilogue_row<EPI, MT>(a, acc, MT - 1, erow0 + wm * WM, ecol, lg, auxr);
            } else {
                mma(xf0, wf0);
                mma(xf1, wf1);
            }
        }
        if (TTL_GEMM_DIAG == 5) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) asm volatile("" ::"v"(acc[mt][nt]));
        } else if (!(TTL_BIG_EPI_OVERLAP && TTL_GEMM_DIAG == 0)) {
            size_t ecol = (size_t)(ecol0 + wn * 64 + 4 * li);
            if constexpr (EPI == EPI_OP_HM) ecol = hm_col_base(a, (int)ecol);
            big_epilogue<EPI, MT>(a, acc, erow0 + wm * WM, ecol, lg, auxr);
        }
        TTL_STAMP_K1(first);       // (gemm_big: the first tile's K loop AND its epilogue, which is interleaved with step nk-1)
        if (!more) break;
        slot = nslot;
        first = false;
        // ring order of the next tile: K-tile 0 sits in `cur`, K-tile 1 in `nn` (3 stages); everyone must be done
        // reading `nxt` (step nk-1) before K-tile 2 goes there: the barrier at the top of the tile orders that
        if constexpr (STAGES == 3) { s0 = cur; s1 = nn; s2 = nxt; }
        else { s0 = cur; s1 = nxt; }
    }
    TTL_STAMP_END(g_big_stamps);
}

template <int MT, int STAGES, int EPI>
hipError_t launch_big_t(const GemmArgs& a, int order, int max_blocks, hipStream_t s) {
    constexpr int BM = 32 * MT;
    constexpr int SMEM = STAGES * (BM + BN) * 128 + 1024;   // ring + the bias slot
    static std::atomic<uint64_t> done{0};
    hipError_t e = ensure_smem((const void*)gemm_big_kernel<MT, STAGES, EPI>, SMEM, done);
    if (e != hipSuccess) return e;
    TileMap tm;
    tm.ntm = (a.M + BM - 1) / BM; tm.ntn = a.N / BN; tm.order = order;
    if (order == 0 || tm.ntm < 8) { tm.order = 0; tm.ntiles = tm.ntm * tm.ntn; }
    else { tm.ntiles = 8 * ((tm.ntm + 7) / 8) * tm.ntn; }
    int grid = tm.ntiles < max_blocks ? tm.ntiles : max_blocks;
    if (tm.order != 0) grid = (grid / 8) * 8 ? (grid / 8) * 8 : 8;   // whole XCD rounds: slot & 7 must stay the block's XCD label
    hipLaunchKernelGGL((gemm_big_kernel<MT, STAGES, EPI>), dim3(grid), dim3(NTHR), SMEM, s, a, tm);
    return hipGetLastError();
}

template <int EPI>
hipError_t launch_big_v(const GemmArgs& a, int mt, int stages, int order, int max_blocks, hipStream_t s) {
    if (mt == 5 && stages == 3) return launch_big_t<5, 3, EPI>(a, order, max_blocks, s);
    if (mt == 5 && stages == 2) return launch_big_t<5, 2, EPI>(a, order, max_blocks, s);
    if (mt == 7) return launch_big_t<7, 2, EPI>(a, order, max_blocks, s);
    if (mt == 8) return launch_big_t<8, 2, EPI>(a, order, max_blocks, s);
    return hipErrorInvalidValue;
}

}  // namespace

unsigned qkv_hm_magic(int T, int limit) {
    if (T < 1 || limit < 1) return 0;
    const uint64_t magic = (1ull << 32) / (uint64_t)T + 1;
    if (magic >> 32) return 0;                                    // T == 1
    for (uint64_t m = 0; m < (uint64_t)limit; ++m)
        if (((m * magic) >> 32) != m / (uint64_t)T) return 0;
    return (unsigned)magic;
}

// rows the kernel may store for a launch of M rows with row tiles of 32*mt (unguarded epilogue)
#ifdef TTL_CLOCK_STAMPS
extern "C" __attribute__((visibility("default"))) int ttl_diag_clock_stamps_big(void* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_big_stamps), sizeof(TtlClockStamp) * TTL_STAMP_SLOTS);
}
#endif

bool gemm_big_applicable(GemmEpi epi, const GemmArgs& a) {
    if (epi == EPI_PATCH) return false;                       // scattered output rows: guarded kernel of gemm.hip
    // MLP dgrad: its epilogue reads u (8*MT registers fetched ahead in gemm.hip).  Tried here with u fetched behind the last
    // barrier: 82 us per launch vs 83-87 us on gemm.hip's kernel (the 77 MB read stays exposed either way) at the price of spills.
    if (epi == EPI_GELU_BWD) {      // 128-row tiles with u read at the top of the tile (TTL_GEMM_BIG_DGRAD=0: gemm.hip's 160 x 128 kernel)
        static const bool on = TTL_EXPERIMENT("TTL_GEMM_BIG_DGRAD", 1) != 0;
        if (!on || !a.aux || (size_t)((a.M + 127) / 128) * 128 > (size_t)a.padded) return false;
    }
    if (a.M < 1024 || a.N % BN || a.K % BK || a.K / BK < 3) return false;
    if (a.amap || a.cmap || a.c2map || a.splits > 1) return false;
    return true;
}

hipError_t launch_gemm_big(GemmEpi epi, const GemmArgs& a, hipStream_t s) {
    if (gemm_huge_applicable(epi, a)) return launch_gemm_huge(epi, a, s);     // q/k/v, fc1: 256 x 256 tiles on four waves (gemm_huge.hip)
    // Tile height (TTL_GEMM_BIG_MT), ring depth, tile order, resident blocks: tuned in situ, see DESIGN.md §3.1
    static const int mt_env = TTL_EXPERIMENT("TTL_GEMM_BIG_MT", 5);
    static const int st_env = TTL_EXPERIMENT("TTL_GEMM_BIG_STAGES", 0);
    static const int order_env = TTL_EXPERIMENT("TTL_GEMM_BIG_ORDER", -1);
    static const int blocks_env = TTL_EXPERIMENT("TTL_GEMM_BIG_BLOCKS", 0);
    const int cus = device_cu_count();
    if (!cus) return hipErrorInvalidDevice;
    int mt = mt_env;
    // TTL_GEMM_BIG_MT_WIDE (experiment r05m): another tile height for the wide, short-K launches only (QKV, fc1, MLP dgrad is separate)
    static const int mt_wide = TTL_EXPERIMENT("TTL_GEMM_BIG_MT_WIDE", 0);
    if (mt_wide && a.N >= 2304 && a.K <= 1024) mt = mt_wide;
    if ((size_t)((a.M + 32 * mt - 1) / (32 * mt)) * 32 * mt > (size_t)a.padded) mt = 5;   // a.padded = rows every output buffer has
    if ((size_t)((a.M + 159) / 160) * 160 > (size_t)a.padded) return hipErrorInvalidValue;
    const int stages = (mt == 5) ? (st_env ? st_env : 3) : 2;
    const int ntn = a.N / BN, ntm = (a.M + 32 * mt - 1) / (32 * mt);
    // column-major inside an XCD when its tiles take more than one round (the weight slice stays in L2 while the
    // row panels cycle); row-major when everything is resident at once (a panel's column tiles run side by side)
    int order = order_env >= 0 ? order_env : ((ntm * ntn > cus) ? 1 : 2);
    const int max_blocks = blocks_env ? blocks_env : cus;
    switch (epi) {
        case EPI_F32: return launch_big_v<EPI_F32>(a, mt, stages, order, max_blocks, s);
        case EPI_OP:
            if (a.hm_T) {   // head-major q/k/v: N = 3*D with whole heads per 64 columns, row -> view by the checked multiplier
                if (a.N % 192 || !a.hm_magic || a.hm_T < 1) return hipErrorInvalidValue;
                return launch_big_v<EPI_OP_HM>(a, mt, stages, order, max_blocks, s);
            }
            return launch_big_v<EPI_OP>(a, mt, stages, order, max_blocks, s);
        case EPI_RESID_F32: return launch_big_v<EPI_RESID_F32>(a, mt, stages, order, max_blocks, s);
        case EPI_GELU_BWD: {
            const int ntm4 = (a.M + 127) / 128;
            return launch_big_t<4, 3, EPI_GELU_BWD_BIG>(a, (ntm4 * ntn > cus) ? 1 : 2, max_blocks, s);
        }
        case EPI_GELU: return a.C2 ? launch_big_v<EPI_GELU_C2>(a, mt, stages, order, max_blocks, s) : launch_big_v<EPI_GELU>(a, mt, stages, order, max_blocks, s);
        default: break;
    }
    return hipErrorInvalidValue;
}
