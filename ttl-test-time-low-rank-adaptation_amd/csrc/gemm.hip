// bf16 MFMA GEMM for the ViT projections:  C[M,N] = A[M,K] · B[N,K]^T  (+ fused epilogue)
//
// Replaces the nn.Linear / Conv2d matmuls the reference runs through torch (HF
// modeling_clip.py:202-218 patch conv, :309-311 q/k/v, :333 out_proj, :346-350 fc1/fc2) and
// their autograd dgrad counterparts.  gfx950 design:
//   * block tile 160 x 128 x 64, 4 waves (2x2), two blocks per CU (32 x 64, 2 waves, 4-stage ring for small M),
//     v_mfma_f32_16x16x32_bf16, fp32 accumulate
//   * operands staged HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS
//     stages, next tile's DMA in flight under the current tile's MFMAs
//   * LDS image is lane-linear (DMA constraint), bank conflicts removed by XOR-swizzling the
//     SOURCE chunk and the ds_read_b128 chunk with the same involution (guide rule 21)
//   * the weight tile's rows are permuted in LDS (physical row 16*nt + i of a 64-column slab holds
//     output column 4*i + nt) so that a lane's four n-subtile accumulators are 4 CONTIGUOUS output
//     columns: every store instruction writes 4 rows x 256 B (fp32) / 128 B (bf16) of whole lines
//   * 1-D grid, XCD-aware tile order (n fastest: the A panel of a row tile stays in one L2)
#include <stdlib.h>

#include <atomic>

#include "kernels.hpp"
#include "gemm_epilogue.hpp"

namespace {

constexpr int BK = 64;

// Diagnostic ablations of the main loop (tools/gemm_ablate.sh builds libttl_hip_diagN.so; results are WRONG on
// purpose, only the timing is read): 1 = no DMA in the steady loop, 2 = no MFMA, 3 = no LDS fragment reads,
// 4 = no barrier.  The product build has TTL_GEMM_DIAG == 0 and none of this exists in it.
#ifndef TTL_GEMM_DIAG
#define TTL_GEMM_DIAG 0
#endif

// compile-time unrolled scheduling hints: NP x { MFMA x MPER, (first NP0 rounds) DS_READ x DPER, VMEM x 1 }
template <int I, int NP, int NP0, int MPER, int DPER>
struct SchedLoop {
    static __device__ __forceinline__ void run() {
        __builtin_amdgcn_sched_group_barrier(0x008, MPER, 0);
        if constexpr (I < NP0) __builtin_amdgcn_sched_group_barrier(0x100, DPER, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        SchedLoop<I + 1, NP, NP0, MPER, DPER>::run();
    }
};
template <int NP, int NP0, int MPER, int DPER>
struct SchedLoop<NP, NP, NP0, MPER, DPER> {
    static __device__ __forceinline__ void run() {}
};

// BM x (64*WNW) block tile, WMW x WNW waves, every wave owns (BM/WMW) x 64 outputs.
// STAGES == 2: one K-tile in flight, plain __syncthreads().  STAGES > 2 (small-M calls, which are
// latency-bound: a dozen blocks on the whole chip): STAGES-1 tiles of DMA in flight behind a counted
// s_waitcnt vmcnt(N) + raw s_barrier; tiles past the end re-load the last tile into a dead stage so
// that ONE immediate serves the whole loop.
template <int BM, int WMW, int WNW, int STAGES, int EPI, bool GUARD>
__global__ __launch_bounds__(64 * WMW * WNW, 2) void gemm_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NTHR = 64 * WMW * WNW;
    constexpr int BN = 64 * WNW;
    constexpr int WM = BM / WMW;      // rows per wave
    constexpr int MT = WM / 16;       // 16-row sub-tiles per wave
    constexpr int NA = BM * 8 / NTHR; // DMA instructions per thread for the A tile
    constexpr int NB = BN * 8 / NTHR;
    static_assert(WM % 16 == 0 && (BM * 8) % NTHR == 0 && (BN * 8) % NTHR == 0, "tile shape");
    constexpr int A_BYTES = BM * BK * 2;
    constexpr int STAGE = A_BYTES + BN * BK * 2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNW, wn = wave % WNW;
    const int li = lane & 15, lg = lane >> 4;

    const int ntn = a.N / BN;
    const int M = a.M;
    int row0, col0;
    if (a.xc > 0) {
        // 2-D XCD partition (blocks b and b+8 share an XCD): the 8 XCDs form an (8/xc) x xc grid over
        // (row tiles) x (column tiles), so an XCD's slice of the weight matrix, N/xc x K, stays in its 4 MiB
        // L2 for the whole launch while the activation panels stream through it once per XCD column.
        // With a 1-D order every XCD walks ALL of B once per row tile: at N*K*2 B > 4 MiB (fc1 / fc2 of
        // ViT-B) that is an LRU-thrashing cyclic sweep served from the Infinity Cache.
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int xr_n = 8 / a.xc, xr = x / a.xc, xcol = x - xr * a.xc;
        const int ntm = (M + BM - 1) / BM;
        const int r0 = xr * ntm / xr_n, r1 = (xr + 1) * ntm / xr_n;
        const int cols = ntn / a.xc;
        if (j >= (r1 - r0) * cols) return;   // uneven row split: the spare blocks of this XCD have no tile
        const int jr = j / cols;
        row0 = (r0 + jr) * BM;
        col0 = (xcol * cols + (j - jr * cols)) * BN;
    } else {
        const int bid = xcd_remap(blockIdx.x, gridDim.x);
        row0 = (bid / ntn) * BM;
        col0 = (bid % ntn) * BN;
    }
    // split-K (small-M fp32 partials): slice blockIdx.y covers K/splits and writes its own [M][ldc] slab
    const int nk = a.K / BK / a.splits;
    const size_t koff = (size_t)blockIdx.y * nk * BK;

    // ---- per-thread DMA sources (row fixed for the whole K loop): 32-bit lane offsets from a
    // wave-uniform base, so stepping K is scalar arithmetic (global_load_lds saddr + voffset form)
    uint32_t aoff[NA], boff[NB];
    const char* abase = (const char*)a.A + koff * sizeof(op_t);
    const char* bbase = (const char*)a.B + koff * sizeof(op_t);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int q = i * NTHR + tid, r = q >> 3, p = q & 7;
        int c = p ^ ((r >> 1) & 7);
        int gr = min(row0 + r, M - 1);
        if constexpr (GUARD) { if (a.amap) gr = a.amap[gr]; }
        aoff[i] = (uint32_t)(((size_t)gr * a.lda + c * 8) * sizeof(op_t));
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int q = i * NTHR + tid, r = q >> 3, p = q & 7;
        int c = p ^ ((r >> 1) & 7);
        int n = (r & ~63) + 4 * (r & 15) + ((r >> 4) & 3);  // physical LDS row r holds this output column
        boff[i] = (uint32_t)(((size_t)(col0 + n) * a.ldb + c * 8) * sizeof(op_t));
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE;
        const char* ak = abase + (size_t)kt * (BK * sizeof(op_t));
        const char* bk = bbase + (size_t)kt * (BK * sizeof(op_t));
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(ak + aoff[i]), LDS_PTR(base + (i * NTHR + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(bk + boff[i]), LDS_PTR(base + A_BYTES + (i * NTHR + wave * 64) * 16), 16, 0, 0);
    };

    // ---- per-lane fragment addresses ----
    const int swA = (li >> 1) & 7;
    int offA[MT], offW[4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) offA[mt] = (wm * WM + mt * 16 + li) * 128;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) offW[nt] = A_BYTES + (wn * 64 + 16 * nt + li) * 128;

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int NP = NA + NB;            // DMA pieces per thread per K-tile
    constexpr int NP0 = (NP + 1) / 2;      // issued under the first 32-deep half, the rest under the second
    auto stage_piece = [&](int i, int kt, char* base) {
        if (TTL_GEMM_DIAG == 1) return;
        if (i < NA)
            __builtin_amdgcn_global_load_lds(GLB_PTR(abase + (size_t)kt * (BK * sizeof(op_t)) + aoff[i]),
                                             LDS_PTR(base + (i * NTHR + wave * 64) * 16), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds(GLB_PTR(bbase + (size_t)kt * (BK * sizeof(op_t)) + boff[i - NA]),
                                             LDS_PTR(base + A_BYTES + ((i - NA) * NTHR + wave * 64) * 16), 16, 0, 0);
    };
    auto load_frags = [&](const char* base, int s, opx8 (&xf)[MT], opx8 (&wf)[4]) {
        if (TTL_GEMM_DIAG == 3) base = smem;   // loop-invariant address: hoisted out of the K loop by the compiler
        const int cA = ((4 * s + lg) ^ swA) << 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xf[mt] = *(const opx8*)(base + offA[mt] + cA);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const opx8*)(base + offW[nt] + cA);
    };
    auto mma = [&](const opx8 (&xf)[MT], const opx8 (&wf)[4]) {
        if (TTL_GEMM_DIAG == 2) {   // keep the LDS reads alive without the matrix pipe
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" ::"v"(xf[mt]));
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) asm volatile("" ::"v"(wf[nt]));
            return;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = MFMA16(xf[mt], wf[nt], acc[mt][nt], 0, 0, 0);
    };
    // MLP-dgrad epilogue multiplies by gelu'(u): fetch u (8 B per lane per row) BEFORE the K loop so its
    // latency hides under the MFMAs instead of sitting at the end of the tile (40 VGPRs)
    u32x2 auxr[(EPI == EPI_GELU_BWD && !GUARD) ? MT : 1][4];
    if constexpr (EPI == EPI_GELU_BWD && !GUARD) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                auxr[mt][r] = *(const u32x2*)(a.aux + (size_t)(row0 + wm * WM + mt * 16 + 4 * lg + r) * a.ldaux + col0 + wn * 64 + 4 * li);
    }
    if constexpr (STAGES == 2) {
        stage(0, 0);
        // steady state: tile kt is consumed while tile kt+1's DMA is issued BETWEEN the MFMAs (the
        // matrix pipe never waits behind a burst of DMA issue); the last tile is peeled (no prefetch)
        for (int kt = 0; kt + 1 < nk; ++kt) {
            if (TTL_GEMM_DIAG != 4) __syncthreads();  // tile kt landed (vmcnt(0) + barrier); everyone is done with the other stage
            const char* base = smem + (kt & 1) * STAGE;
            char* nxt = smem + ((kt + 1) & 1) * STAGE;
            opx8 xf0[MT], wf0[4], xf1[MT], wf1[4];
            load_frags(base, 0, xf0, wf0);
            load_frags(base, 1, xf1, wf1);
#pragma unroll
            for (int i = 0; i < NP0; ++i) stage_piece(i, kt + 1, nxt);
            mma(xf0, wf0);
#pragma unroll
            for (int i = NP0; i < NP; ++i) stage_piece(i, kt + 1, nxt);
            mma(xf1, wf1);
            // Main-loop order, measured IN SITU (tools/gemm_macro_ab.sh TTL_GEMM_SCHED 0 1 2 + quick_bench.py, GEMM ms per episode):
            //   0 (used)  DMA pieces spread between the MFMAs ........ 3.52
            //   1         fragments, the DMA as one burst, MFMAs ..... 3.57
            //   2         the compiler's own order ................... 3.70
            // A back-to-back microbenchmark of one shape (inputs and outputs resident in L2 / Infinity Cache,
            // tools/gemm_ablate.py) ranks them the other way round (1 is 4-16 % ahead there): judge in situ.
#ifndef TTL_GEMM_SCHED
#define TTL_GEMM_SCHED 0
#endif
            if constexpr (TTL_GEMM_SCHED == 1) {          // all fragments, the DMA as one burst, all MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MT + 4), 0);
                __builtin_amdgcn_sched_group_barrier(0x010, NP, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT * 4, 0);
            } else if constexpr (TTL_GEMM_SCHED == 2) {   // the compiler's own order
            } else {
            // pin the interleave: fragments of the first half, then {MFMAs, one DMA piece, a few reads}
            __builtin_amdgcn_sched_group_barrier(0x100, MT + 4, 0);
            SchedLoop<0, NP, NP0, (2 * MT * 4) / NP, (MT + 4 + NP0 - 1) / NP0>::run();
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * MT * 4 - NP * ((2 * MT * 4) / NP), 0);
            }
        }
        {
            __syncthreads();
            const char* base = smem + ((nk - 1) & 1) * STAGE;
            opx8 xf0[MT], wf0[4], xf1[MT], wf1[4];
            load_frags(base, 0, xf0, wf0);
            load_frags(base, 1, xf1, wf1);
            mma(xf0, wf0);
            mma(xf1, wf1);
        }
    } else {
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t) stage(min(t, nk - 1), t);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * NP) : "memory");  // tile kt landed
            __builtin_amdgcn_s_barrier();  // ... for every wave; and all are done reading stage (kt-1) % STAGES
            const char* base = smem + (kt % STAGES) * STAGE;
            char* nxt = smem + ((kt + STAGES - 1) % STAGES) * STAGE;
            const int kn = min(kt + STAGES - 1, nk - 1);
            opx8 xf0[MT], wf0[4], xf1[MT], wf1[4];
            load_frags(base, 0, xf0, wf0);
            load_frags(base, 1, xf1, wf1);
#pragma unroll
            for (int i = 0; i < NP; ++i) stage_piece(i, kn, nxt);
            mma(xf0, wf0);
            mma(xf1, wf1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the dummy tail loads before the block retires
    }
    GemmArgs e = a;
    if (EPI == EPI_F32 && a.splits > 1) e.C = (float*)a.C + (size_t)blockIdx.y * M * a.ldc;
    if constexpr (EPI == EPI_GELU_BWD && !GUARD)
        gemm_epilogue<EPI, MT, GUARD>(e, acc, row0 + wm * WM, col0 + wn * 64 + 4 * li, lg, M, auxr);
    else
        gemm_epilogue<EPI, MT, GUARD>(e, acc, row0 + wm * WM, col0 + wn * 64 + 4 * li, lg, M);
}


template <int BM, int WMW, int WNW, int EPI, bool GUARD = true, int STAGES = 2>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
    constexpr int BN = 64 * WNW;
    constexpr int SMEM = STAGES * (BM + BN) * BK * 2;
    static std::atomic<uint64_t> done{0};     // once per (kernel, device)
    if (hipError_t e = ensure_smem((const void*)gemm_kernel<BM, WMW, WNW, STAGES, EPI, GUARD>, SMEM, done); e != hipSuccess) return e;
    if (a.N % BN) return hipErrorInvalidValue;
    int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
    GemmArgs b = a;
    b.xc = 0;
    int nblk = ntm * ntn;
    static const int xcd2d = TTL_EXPERIMENT("TTL_GEMM_XCD2D", 2);   // 0 = 1-D order; n = slice limit n*1.5 MiB (+1 % in situ)
    if (xcd2d && a.M >= 1024 && a.splits == 1 && ntm >= 8) {
        // columns of the XCD grid: halve the weight slice until it sits comfortably in a 4 MiB L2
        const size_t limit = (size_t)xcd2d * 1536 * 1024;
        int xc = 1;
        // Only wide outputs (N >= 1536): at N = 768 the six column tiles of a row panel already run side by side on one
        // XCD, and splitting them makes two XCDs fetch the (long-K) activation panel.  HBM-side read bytes per launch
        // (rocprofv3 FETCH_SIZE): QKV 96 -> 75 MB, fc1 179 -> 118, MLP dgrad 257 -> 196 with the split; out_proj / fc2
        // 111 -> 140 and the dX GEMMs 102 -> 154 if it were applied to them too.
        while (a.N >= 1536 && xc < 8 && (size_t)a.N * a.K * sizeof(op_t) / xc > limit && ntn % (2 * xc) == 0) xc *= 2;
        const int xr_n = 8 / xc;
        b.xc = xc;
        nblk = 8 * ((ntm + xr_n - 1) / xr_n) * (ntn / xc);
    }
    hipLaunchKernelGGL((gemm_kernel<BM, WMW, WNW, STAGES, EPI, GUARD>), dim3(nblk, a.splits), dim3(64 * WMW * WNW), SMEM, s, b);
    return hipGetLastError();
}

// Tile choice (measured in situ on the 64-view ViT-B/16 episode, tools/quick_bench.py):
//   160x128x64, 4 waves, 2 blocks/CU, unguarded epilogue ........ 3.85 ms of GEMM per image  <- used
//   same with the per-row guard in the epilogue .................. 4.21 ms (hipcc re-waits vmcnt(0) per store block)
//   320x128x64, 8 waves, 1 block/CU .............................. 4.35 ms (nothing overlaps its prologue/epilogue)
//   160x128x32, 2/3/4-stage LDS ring, counted vmcnt, register
//   double-buffered fragments, 2-3 blocks/CU ..................... 4.37-4.48 ms (twice the barriers per K)
//   256x128x64, 8 waves in two groups one barrier apart (one loads
//   fragments + issues DMA while the other issues MFMAs), 3 stages  +5% on the N >= 2304 shapes, +15% overall
//   256x256x64 / 256x128x64 (2- and 3-stage), 8 waves, 1 block/CU  0.91 / 0.83 / 0.80 of the 160x128 rate on the
//   QKV shape, worse on N = 768: 150-600 blocks quantise badly on 256 CUs and nothing hides a block's prologue;
//   in situ, used for the N >= 2304 GEMMs only: 3.21 (256x256) / 3.30 (256x128) vs 3.02 ms of big-M GEMM per episode
// Ablations of the 160x128 loop on M=12800,N=2304,K=768 (tools/gemm_ablate.py): product 55.6 us; without the
// DMA 38.0 us; without the MFMAs 43.9 us (= 14.5 TB/s of L2->LDS staging, the guide's L2-resident LDS-gather
// rate is 17-19 TB/s); without the LDS fragment reads 52.5 us; without the barrier 52.4 us.  The kernel is
// bound by L2->LDS staging at this tile's 71 FLOP per staged byte, not by LDS reads, barriers or MFMA issue.
//   A through a 3-stage LDS ring + B straight from L2 into registers in MFMA operand layout (four waves side by
//   side, 160 x 32 each; two K-tiles in flight per block instead of one; bit-identical results): 4.32 ms vs 3.47
//   in situ.  A register-destination B load touches 16 lines x 64 B per instruction (half lines, twice the
//   line requests of the DMA), and the 2-column-per-lane epilogue halves the store width.
//   cache-policy bits on the operand DMA (global_load_lds aux): sc0 on A neutral, nt on A -16 % (the panel IS re-read
//   by the other column tiles); non-temporal residual loads in the epilogue -2 %.
// M = 12608 gives 79 row tiles, so N = 768 / 2304 / 3072 launch 474 / 1422 / 1896 blocks = 0.93 /
// 2.78 / 3.70 rounds of the 512 resident slots (>= 93% of whole rounds; 128x128 gives 77% at N = 768).
// The DMA-only ablation of the 128x128 loop already moves ~20 TB/s L2->LDS, i.e. the tile's
// 64-71 FLOP per staged byte is near the L2->LDS ceiling: the next step is a larger block tile.
// Small-M calls (1-view inference, pooled-row GEMMs) are latency-bound: a 4-stage ring (3 K-tiles of DMA in
// flight), the guarded epilogue and a small tile (see launch_v).
template <int EPI>
hipError_t launch_v(const GemmArgs& a, hipStream_t s) {
    static const int variant = TTL_EXPERIMENT("TTL_GEMM_VARIANT", 2);
    if (a.M < 1024) {
        // Small-M calls (1-view inference, pooled-row GEMMs of the last layer and of its backward: M = 64..257) are
        // latency chains on a handful of blocks, so the tile is SMALL to spread them over more CUs.  In situ, ms of
        // this class per episode (20 launches, tools/gemm_macro_ab.sh): 128x128 4-stage 0.32, 64x128 0.29 (M<=64) /
        // 0.26 (all), 64x64 0.23, 32x128 0.25, 32x64 0.215 (used; 6 or 8 stages no better), 16x64 / 16x128 0.26.
        if (variant == 8) return launch_t<128, 2, 2, EPI, true, 4>(a, s);
        return launch_t<32, 2, 1, EPI, true, 4>(a, s);
    }
    if (variant == 0) return launch_t<128, 2, 2, EPI>(a, s);
    if (a.padded && EPI != EPI_PATCH) {
        if (variant == 1) return launch_t<320, 4, 2, EPI, false>(a, s);
        return launch_t<160, 2, 2, EPI, false>(a, s);
    }
    return launch_t<160, 2, 2, EPI, true>(a, s);
}

}  // namespace

bool gemm_takes_big(GemmEpi epi, const GemmArgs& a) {
    static const bool use_big = TTL_EXPERIMENT("TTL_GEMM_BIG", 1) != 0;
    return use_big && a.padded && gemm_big_applicable(epi, a) && (size_t)((a.M + 159) / 160) * 160 <= (size_t)a.padded;
}

hipError_t launch_gemm(GemmEpi epi, const GemmArgs& a0, hipStream_t s) {
    if (a0.M <= 0 || a0.N % 128 || a0.K % BK || a0.K <= 0 || (a0.lda & 7) || (a0.ldb & 7)) return hipErrorInvalidValue;
    if ((a0.amap || a0.cmap || a0.c2map) && a0.M >= 1024) return hipErrorInvalidValue;   // row maps: guarded small-M kernels only
#ifdef TTL_DIAG_SKIP       // timing-only ablation of the episode (tools/class_cost_ab.sh): bit 0 = no small-M launches, bit 1 = no big-M ones
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 1) && a0.M < 1024 && diag_skip_now(cnt, 1200)) return hipSuccess; }
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 2) && a0.M >= 1024 && diag_skip_now(cnt, 2200)) return hipSuccess; }
#endif
    GemmArgs a = a0;
    a.splits = 1;
    // Small-M, long-K, fp32-output calls (fc2 / dx of the 1-view inference and of the CLS-only top-layer
    // backward) put a dozen blocks on 256 CUs: slice K over blockIdx.y into fp32 partial slabs and sum
    // them in a fixed order (deterministic; no float atomics).
    if (a.M < 1024 && (epi == EPI_F32 || epi == EPI_RESID_F32) && a.ws && a.K >= 1536) {
        int sp = 8;
        while (sp > 1 && ((a.K / BK) % sp || (size_t)sp * a.M * a.N * sizeof(float) > a.ws_bytes)) sp >>= 1;
        if (sp > 1) {
            GemmArgs p = a;
            p.splits = sp; p.C = a.ws; p.ldc = a.N; p.bias = nullptr; p.resid = nullptr; p.cmap = nullptr;
            hipError_t e = launch_v<EPI_F32>(p, s);
            if (e != hipSuccess) return e;
            return launch_splitk_reduce(a.ws, sp, a.M, a.N, epi == EPI_RESID_F32 ? a.resid : nullptr, a.ldr, a.bias, (float*)a.C,
                                        a.ldc, s, a.cmap);
        }
    }
    // big-M launches: 256-column tiles (gemm_big.hip) unless switched off (TTL_GEMM_BIG=0: the 160x128 kernel below)
    if (gemm_takes_big(epi, a)) return launch_gemm_big(epi, a, s);
    if (a.hm_T) return hipErrorInvalidValue;   // head-major q/k/v output exists in the big-M epilogue only (ask gemm_takes_big first)
    switch (epi) {
        case EPI_F32: return launch_v<EPI_F32>(a, s);
        case EPI_OP: return launch_v<EPI_OP>(a, s);
        case EPI_RESID_F32: return launch_v<EPI_RESID_F32>(a, s);
        case EPI_GELU: return launch_v<EPI_GELU>(a, s);
        case EPI_PATCH: return launch_v<EPI_PATCH>(a, s);
        case EPI_GELU_BWD: return launch_v<EPI_GELU_BWD>(a, s);
    }
    return hipErrorInvalidValue;
}
