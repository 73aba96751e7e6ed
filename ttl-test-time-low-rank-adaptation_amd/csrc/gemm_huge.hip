// Wide, short-K big-M GEMM on 256 x 256 x 64 tiles and FOUR waves:  C[M,N] = A[M,K] · B[N,K]^T (+ fused epilogue) for launches with
// N >= 2304 and K <= 1024.  It runs the q/k/v projection (HF modeling_clip.py:309-311, head-major output); it also carries fc1's
// epilogues (modeling_clip.py:346-348, quick_gelu + the pre-activation), which stay on gemm_big.hip by default (gemm_huge_applicable).
//
// Why a second big-M kernel: these launches have 12-16 K-steps per tile, so what a tile costs is set by the bytes staged per FLOP and
// by the per-tile prologue / epilogue as much as by the K loop.  gemm_big.hip's 160 x 256 tile stages 1 byte per 98 FLOP on eight
// waves (80 x 64 outputs per wave: 147 KiB of LDS fragment reads per K-step); here a wave owns 128 x 128 outputs (4 x 4 MFMA
// 32x32x16 tiles, 256 accumulator registers): 1 staged byte per 128 FLOP and 131 KiB of fragment reads per K-step for 1.6x the FLOP.
// Measured on the 64-view ViT-B/16 shapes (fp16, M = 12 608, cold operands; row-major outputs: profiles/r05_experiments.txt r05n;
// with the episode's epilogues: tools/gemm_huge_bench.py): q/k/v 49.4 us against 52.9 us (vendor library 48.2), head-major 53.9
// against 55.7; fc1 70.0 against 67.7, with both outputs 79.1 against 70.7 (600 tiles = three rounds of 256, and 128 store
// instructions per lane behind which the next tile's first wait has to drain most of them: vmcnt counts to 63).  In situ, three
// episodes in flight: q/k/v +1.1 % images/s, fc1 +0.2 % (profiles/r05_huge_sweep_fp16.txt).
//
//   * operands by LDS-DMA only (buffer_load_dwordx4 ... lds: one per-lane offset per operand, everything else wave-uniform; rows past
//     M read as zeros through the buffer range check), no staging registers:
//       A (activations, first touch from HBM): ring of THREE 32-KiB slots, K-tile kt+2 requested during step kt
//       B (weights, L2-resident):              ring of TWO slots, K-tile kt+1 requested during step kt              (3 + 2 = 160 KiB)
//     both behind ONE counted s_waitcnt vmcnt(8) lgkmcnt(0) + raw s_barrier per K-step (the 8 pieces of A(kt+2) stay in flight)
//   * a K-step is four k16 sub-steps of 16 MFMAs; fragment sets alternate per sub-step (2 x 32 registers); the loop body is
//     phase-shifted to start behind the barrier: [sub-step 3 of K-tile kt-1][0][1][2 ; wait ; barrier], each sub-step =
//     8 x {MFMA, ds_read_b128 of the next sub-step's fragments} + 4 x {2 MFMA, one DMA piece} (an LDS-DMA is not moved across an LDS
//     read by hipcc, so the pieces follow the reads; bunching the 16 pieces of a step in one sub-step costs 6 % per K-step)
//   * LDS image: 128-B rows, 16-B chunk c of row r at position c ^ ((r >> 1) & 7) (swizzle applied on the SOURCE address of the DMA):
//     conflict-free for the 32-row ds_read_b128 fragment reads; weight rows permuted so that a lane owns 4 adjacent output columns
//   * bias through one 1-KiB DMA piece into the third A slot (free until step 0 requests K-tile 2) and folded into the
//     accumulators' initial value; persistent tile loop, the next tile's bias and first K-tiles requested before the last 16 MFMAs
//     and the stores of the current one
//   * stores through buffer descriptors: rows >= M are dropped by the range check (no padding requirement on the outputs)
//   * q/k/v: taken when the launch has tiles for >= 85 % of the CUs (gemm_huge_applicable)
//   * the N = D projections (out_proj, fc2 and their dgrads: EPI_RESID_F32 / EPI_F32 / EPI_OP, 150 tiles at 64 views): taken when
//     the context was told that other episodes share the GPU (GemmArgs::concurrent >= 2, ttl_ctx_set_concurrency) and the launch has
//     tiles for half of the CUs: 72 us against 60.5 us one at a time, 10.8 against 14.3 ms of CU-time, +2.5 % images/s with three
//     episodes in flight.  The residual tile is folded into the accumulators' initial value in two halves of 32 rows.
// History of the design (first cut with compiler-scheduled register staging 106 us, hand-pipelined register staging 57-60 us, all-DMA
// with bunched requests 56 us): profiles/r05_experiments.txt r05l, r05n.
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "kernels.hpp"

#ifndef TTL_GEMM_NT_GELU
#define TTL_GEMM_NT_GELU 2      // as gemm_epilogue.hpp: fc1's g (1) and u (2) are stored non-temporally
#endif

namespace {

constexpr int BM = 256, BN = 256, BK = 64, NT = 256;
constexpr int SLOT = 32768;
constexpr int EPI_GELU_C2 = 100;   // internal: EPI_GELU with the second (pre-activation) output
constexpr int EPI_OP_HM = 101;     // internal: EPI_OP into a head-major q/k/v buffer (GemmArgs::hm_T, kernels.hpp QkvLayout)
constexpr int EPI_DGELU = 105;     // EPI_GELU_BWD (MLP dgrad: C = product * quick_gelu'(u)); u of the tile is read behind the last barrier

template <int N>
__device__ __forceinline__ void wait_vm() {     // s_waitcnt vmcnt(N) lgkmcnt(0)  (gfx9 encoding, as gemm_big.hip)
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
}

// 16 MFMAs of a sub-step: the 8 fragment reads lead, one per MFMA, in the order they are consumed; NV (0 or 4) DMA pieces follow
// TTL_HUGE_SHAPE_PROBE (diagnostic builds only, tools/r06_shape_probe.sh; WRONG products): the same loop — fragment reads, DMA pieces,
// waits, barriers, epilogue — with every v_mfma_f32_32x32x16 replaced by two v_mfma_f32_16x16x32 of the same operand registers into two
// quarters of its accumulator: equal MFMA cycles, equal LDS / VMEM traffic.  Times what the OTHER bf16/f16 MFMA shape would do to the
// clock the chip holds in THIS loop (MI355X_MICROARCH.md DVFS give-back item 7) before anyone rewrites the fragment layout for it.
#ifdef TTL_HUGE_SHAPE_PROBE
#define HUGE_MPS 2      // MFMA instructions per 32x32x16 of the real kernel
#define ACC(mt, j, r) acc[mt][j][(r) >> 2][(r) & 3]
#else
#define HUGE_MPS 1
#define ACC(mt, j, r) acc[mt][j][r]
#endif

template <int NV>
__device__ __forceinline__ void mix() {
#pragma unroll
    for (int g = 0; g < 8; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, HUGE_MPS, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
    if constexpr (NV >= 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { __builtin_amdgcn_sched_group_barrier(0x008, 2 * HUGE_MPS, 0); __builtin_amdgcn_sched_group_barrier(0x010, 1, 0); }
    } else {
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * HUGE_MPS, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// slot -> tile.  order 0: the XCDs own contiguous ranges of the row-major tile list (xcd_remap); order 1 / 2: XCD x (= slot & 7) owns
// the row panels [x * ntm / 8, (x + 1) * ntm / 8) and walks them column-major (all its panels of one 256-column weight slice before the
// next slice) / row-major; slots past an XCD's last tile are empty (the launch has 8 * max_x(tiles of x) slots).
__device__ __forceinline__ bool huge_tile_of(int order, int slot, int ntm, int ntn, int& rt, int& ct) {
    if (order == 0) {
        const int t = xcd_remap(slot, ntm * ntn);
        rt = t / ntn; ct = t - rt * ntn;
        return true;
    }
    const int x = slot & 7, j = slot >> 3;
    const int r0 = x * ntm / 8, nr = (x + 1) * ntm / 8 - r0;
    if (j >= nr * ntn) return false;
    if (order == 1) { ct = j / nr; rt = r0 + (j - ct * nr); }
    else { const int jr = j / ntn; rt = r0 + jr; ct = j - jr * ntn; }
    return true;
}

#ifdef TTL_CLOCK_STAMPS
__device__ TtlClockStamp g_huge_stamps[TTL_STAMP_SLOTS];
#endif

template <int EPI>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_huge_kernel(const GemmArgs a, int ntm, int ntn, int order, int nslots) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l32 = lane & 31, lh = lane >> 5;
    const int M = a.M, nk = a.K / BK, ntiles = nslots;
    // ---- fragment addresses: MFMA 32x32x16 operand = row l32 of the wave's slab, 16-B chunk 2 s + lh of K sub-step s
    const int sw = (l32 >> 1) & 7;
    const int fA0 = (wm * 128 + l32) * 128 + ((lh ^ sw) << 4);
    const int fW0 = (wn * 128 + l32) * 128 + ((lh ^ sw) << 4);
    // ---- DMA piece j = 4 i + wave (i = 0..7) of a slot = image rows 8 j .. 8 j + 7; lane (r8, p8) fills chunk position p8 of row
    // 8 j + r8 with global chunk p8 ^ swizzle(row); the swizzle depends on j only through its parity = the wave's
    const int r8 = lane >> 3, p8 = lane & 7;
    const int cs = p8 ^ ((((wave & 1) << 2) + (r8 >> 1)) & 7);
    constexpr int RSRC = 0x00020000;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, (int)(((size_t)(M - 1) * a.lda + a.K) * sizeof(op_t)), RSRC);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, (int)(((size_t)(a.N - 1) * a.ldb + a.K) * sizeof(op_t)), RSRC);
    const __amdgpu_buffer_rsrc_t rsBias = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, a.bias ? a.N * 4 : 0, RSRC);
    // outputs: rows >= M lie past num_records (row-major) or get an out-of-range offset (head-major) and are dropped
    // (head-major: whole [T][64] tiles of ceil(M / T) views — a trailing partial view's rows lie inside its own view block)
    constexpr bool F32OUT = (EPI == EPI_F32 || EPI == EPI_RESID_F32);
    constexpr int ESZ = F32OUT ? 4 : (int)sizeof(op_t);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
        a.C, 0, (int)(EPI == EPI_OP_HM ? (size_t)((M + a.hm_T - 1) / a.hm_T) * a.hm_T * a.N * sizeof(op_t) : (size_t)M * a.ldc * ESZ), RSRC);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)a.resid, 0, (int)(EPI == EPI_RESID_F32 ? (size_t)M * a.ldr * 4 : 0), RSRC);
    const __amdgpu_buffer_rsrc_t rsC2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.C2, 0, (int)(EPI == EPI_GELU_C2 ? (size_t)M * a.ldc2 * sizeof(op_t) : 0), RSRC);
    const __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc((void*)a.aux, 0, (int)(EPI == EPI_DGELU ? (size_t)M * a.ldaux * sizeof(op_t) : 0), RSRC);
    const int voA = (int)((r8 * a.lda + cs * 8) * sizeof(op_t));
    const int voB = (int)((4 * (8 * wave + r8) * a.ldb + cs * 8) * sizeof(op_t));   // image row 32 nt + c of a wave's 128 columns holds column 4 c + nt
    char* const A0 = smem;
    char* const B0 = smem + 3 * SLOT;
    auto dma_a = [&](char* slot, int row0, int kt, int i0 = 0, int i1 = 8) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(slot + (4 * i + wave) * 1024), 16, voA,
                                                     (int)(((size_t)(row0 + 8 * (4 * i + wave)) * a.lda + (size_t)kt * BK) * sizeof(op_t)), 0, 0);
    };
    auto dma_b = [&](char* slot, int col0, int kt, int i0 = 0, int i1 = 8) {
#pragma unroll
        for (int i = i0; i < i1; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(slot + (4 * i + wave) * 1024), 16, voB,
                                                     (int)(((size_t)(col0 + (i >> 2) * 128 + (i & 3)) * a.ldb + (size_t)kt * BK) * sizeof(op_t)), 0, 0);
    };
    auto dma_bias = [&](int col0) {     // (null bias: num_records 0, the piece lands as zeros)
        if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsBias, LDS_PTR(A0 + 2 * SLOT), 16, lane * 16, col0 * 4, 0, 0);
    };

#ifdef TTL_HUGE_SHAPE_PROBE
    f32x4 acc[4][4][4];
#else
    f32x16 acc[4][4];
#endif
    opx8 xf[2][4], wf[2][4];
    auto frags = [&](const char* sa, const char* sb, int s, int set) {
        const char* pa = sa + (fA0 ^ (s << 5));
        const char* pw = sb + (fW0 ^ (s << 5));
        xf[set][0] = *(const opx8*)pa;
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[set][j] = *(const opx8*)(pw + j * 4096);
#pragma unroll
        for (int mt = 1; mt < 4; ++mt) xf[set][mt] = *(const opx8*)(pa + mt * 4096);
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#ifdef TTL_HUGE_SHAPE_PROBE
                acc[mt][j][2 * set] = MFMA16(xf[set][mt], wf[set][j], acc[mt][j][2 * set], 0, 0, 0);
                acc[mt][j][2 * set + 1] = MFMA16(xf[set][mt], wf[set][j], acc[mt][j][2 * set + 1], 0, 0, 0);
#else
                acc[mt][j] = MFMA32(xf[set][mt], wf[set][j], acc[mt][j], 0, 0, 0);
#endif
            }
    };

    int slot = blockIdx.x;
    int row0, col0;
    {
        int rt, ct;
        while (slot < ntiles && !huge_tile_of(order, slot, ntm, ntn, rt, ct)) slot += gridDim.x;
        if (slot >= ntiles) return;
        row0 = rt * BM; col0 = ct * BN;
    }
    TTL_STAMP_DECL;
    TTL_STAMP_BEGIN();
    dma_bias(col0); dma_b(B0, col0, 0); dma_a(A0, row0, 0); dma_a(A0 + SLOT, row0, 1);
    bool first = true;
    for (;;) {
        // K-tile 0 of A and B and the bias have landed; K-tile 1 of A may still fly.  Behind a previous tile its 64 (GELU + u: 128)
        // stores and K-tile 1 are younger than what is waited for: vmcnt counts loads and stores together, in order, up to 63
        // residual tile (EPI_RESID_F32), folded into the accumulators' initial value in two halves of 32 rows x 4 columns per lane: the first
        // half is requested here, under the wait for K-tile 0, the second behind the requests of step 0
        [[maybe_unused]] f32x4 rs[2][16];
        [[maybe_unused]] const int voR = (int)(((wm * 128 + 4 * lh) * a.ldr + wn * 128 + 4 * l32) * 4);
        auto ld_res = [&](int g) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rs[h][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rsR, voR, (int)(((size_t)(row0 + (2 * g + h) * 32 + 8 * (r >> 2) + (r & 3)) * a.ldr + col0) * 4), 0));
        };
        if constexpr (EPI == EPI_RESID_F32) ld_res(0);
        if (first) wait_vm<8>(); else wait_vm<63>();
        __builtin_amdgcn_s_barrier();
        char *aC = A0, *aN = A0 + SLOT, *aNN = A0 + 2 * SLOT, *bC = B0, *bN = B0 + SLOT;
        auto rotate = [&]() { char* t = aC; aC = aN; aN = aNN; aNN = t; t = bC; bC = bN; bN = t; };
        [[maybe_unused]] f32x4 resid_bv = {0.f, 0.f, 0.f, 0.f};
        {   // the bias leaves the third slot before K-tile 2 is requested into it (one extra barrier per tile)
            const f32x4 bv = *(const f32x4*)(aNN + (wn * 128 + 4 * l32) * 4);
            frags(aC, bC, 0, 0);
            wait_vm<63>();
            __builtin_amdgcn_s_barrier();
            if constexpr (EPI != EPI_RESID_F32) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) ACC(mt, j, r) = bv[j];
            } else {
                auto init_half = [&](int g) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) ACC(2 * g + h, j, r) = bv[j] + rs[h][r][j];
                };
                init_half(0);
                resid_bv = bv;
            }
        }
        // one K-step.  front: the body starts with sub-step 3 of K-tile kt-1 (not for kt = 0); hb / ha: K-tiles kt+1 (B) / kt+2 (A) exist
        auto body = [&](int kt, auto hb_, auto ha_, auto front_) {
            constexpr bool hb = decltype(hb_)::value, ha = decltype(ha_)::value, front = decltype(front_)::value;
            if constexpr (front) {
                frags(aC, bC, 0, 0);
                if constexpr (hb) dma_b(bN, col0, kt + 1, 0, 4);
                mma(1);
                mix<hb ? 4 : 0>();
            } else {
                if constexpr (hb) dma_b(bN, col0, kt + 1, 0, 4);
                if constexpr (EPI == EPI_RESID_F32) {      // step 0: the second half of the residual tile behind the first requests
                    ld_res(1);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) ACC(2 + h, j, r) = resid_bv[j] + rs[h][r][j];
                }
            }
            frags(aC, bC, 1, 1);
            if constexpr (hb) dma_b(bN, col0, kt + 1, 4, 8);
            mma(0);
            mix<hb ? 4 : 0>();
            frags(aC, bC, 2, 0);
            if constexpr (ha) dma_a(aNN, row0, kt + 2, 0, 4);
            mma(1);
            mix<ha ? 4 : 0>();
            frags(aC, bC, 3, 1);
            if constexpr (ha) dma_a(aNN, row0, kt + 2, 4, 8);
            mma(0);
            mix<ha ? 4 : 0>();
            if constexpr (ha) wait_vm<8>(); else wait_vm<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            rotate();
        };
        TTL_STAMP_K0(first);
        body(0, std::true_type{}, std::true_type{}, std::false_type{});
        int kt = 1;
        for (; kt + 2 < nk; ++kt) body(kt, std::true_type{}, std::true_type{}, std::true_type{});
        body(kt, std::true_type{}, std::false_type{}, std::true_type{}); ++kt;
        body(kt, std::false_type{}, std::false_type{}, std::true_type{});
        TTL_STAMP_K1(first);
        // ---- every slot is free: the next tile's bias and first K-tiles go out before the last MFMAs and the stores of this one
        int nslot = slot + gridDim.x;
        int nrow0 = 0, ncol0 = 0;
        {
            int rt = 0, ct = 0;
            while (nslot < ntiles && !huge_tile_of(order, nslot, ntm, ntn, rt, ct)) nslot += gridDim.x;
            nrow0 = rt * BM; ncol0 = ct * BN;
        }
        if (nslot < ntiles) { dma_bias(ncol0); dma_b(B0, ncol0, 0); dma_a(A0, nrow0, 0); }
        mma(1);
        __builtin_amdgcn_sched_barrier(0);
        // MLP dgrad: the lane's 64 x 4 pre-activations of this tile (rows >= M: zeros).  Requested behind the last MFMAs (the fragment
        // registers are free then: 128 more live registers do not fit beside them); the wait for them covers the next tile's first pieces
        [[maybe_unused]] u32x2 ur[4][16];
        if constexpr (EPI == EPI_DGELU) {
            const int vou = (int)(((wm * 128 + 4 * lh) * a.ldaux + wn * 128 + 4 * l32) * sizeof(op_t));
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ur[mt][r] = __builtin_amdgcn_raw_buffer_load_b64(rsU, vou, (int)(((size_t)(row0 + mt * 32 + 8 * (r >> 2) + (r & 3)) * a.ldaux + col0) * sizeof(op_t)), 0);
        }
        // ---- epilogue: register r of sub-tile (mt, j) is row 32 mt + 8 (r >> 2) + 4 lh + (r & 3), column 4 l32 + j of the wave's slab
        const int n0 = col0 + wn * 128 + 4 * l32;
        int vo, vo2 = 0;
        if constexpr (EPI == EPI_OP_HM) {
            // the lane's 4 columns lie in one head: offset inside a view's block (as gemm_big.hip hm_col_base)
            const int Dm = a.N / 3;
            const int plane = (n0 >= Dm) + (n0 >= 2 * Dm);
            const int rem = n0 - plane * Dm;
            vo = (plane * Dm + (rem & ~63)) * a.hm_T + (rem & 63);
        } else {
            vo = (int)(((wm * 128 + 4 * lh) * a.ldc + wn * 128 + 4 * l32) * ESZ);
            if constexpr (EPI == EPI_GELU_C2) vo2 = (int)(((wm * 128 + 4 * lh) * a.ldc2 + wn * 128 + 4 * l32) * sizeof(op_t));
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);      // (keeps the accumulator reads of 4 rows, not 64, live at once)
                float v0 = ACC(mt, 0, r), v1 = ACC(mt, 1, r), v2 = ACC(mt, 2, r), v3 = ACC(mt, 3, r);
                const int mrow = mt * 32 + 8 * (r >> 2) + (r & 3);
                if constexpr (EPI == EPI_OP_HM) {
                    const unsigned m = (unsigned)(row0 + wm * 128 + 4 * lh + mrow);
                    const unsigned view = __umulhi(m, a.hm_magic), t = m - view * (unsigned)a.hm_T;
                    const unsigned off = (view * (unsigned)a.N * (unsigned)a.hm_T + (unsigned)vo + t * 64u) * (unsigned)sizeof(op_t);
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_op2(v0, v1), pack_op2(v2, v3)}, rsC, (int)(m < (unsigned)M ? off : 0x80000000u), 0, 0);
                } else if constexpr (F32OUT) {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v0, v1, v2, v3}), rsC, vo,
                                                           (int)(((size_t)(row0 + mrow) * a.ldc + col0) * 4), 0);
                } else {
                    if constexpr (EPI == EPI_GELU_C2)
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_op2(v0, v1), pack_op2(v2, v3)}, rsC2, vo2,
                                                              (int)(((size_t)(row0 + mrow) * a.ldc2 + col0) * sizeof(op_t)), TTL_GEMM_NT_GELU == 2 ? 2 : 0);
                    if constexpr (EPI == EPI_GELU || EPI == EPI_GELU_C2) {
                        v0 = quick_gelu_f(v0); v1 = quick_gelu_f(v1); v2 = quick_gelu_f(v2); v3 = quick_gelu_f(v3);
                    }
                    if constexpr (EPI == EPI_DGELU) {
                        const u32x2 t = ur[mt][r];
                        v0 *= quick_gelu_grad_f(op_lo(t[0])); v1 *= quick_gelu_grad_f(op_hi(t[0]));
                        v2 *= quick_gelu_grad_f(op_lo(t[1])); v3 *= quick_gelu_grad_f(op_hi(t[1]));
                    }
                    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_op2(v0, v1), pack_op2(v2, v3)}, rsC, vo,
                                                          (int)(((size_t)(row0 + mrow) * a.ldc + col0) * sizeof(op_t)),
                                                          ((EPI == EPI_GELU || EPI == EPI_GELU_C2) && TTL_GEMM_NT_GELU) ? 2 : 0);
                }
            }
        if (nslot >= ntiles) break;
        dma_a(A0 + SLOT, nrow0, 1);
        slot = nslot; row0 = nrow0; col0 = ncol0; first = false;
    }
    TTL_STAMP_END(g_huge_stamps);
}

template <int EPI>
hipError_t launch_huge_t(const GemmArgs& a, int max_blocks, hipStream_t s) {
    static std::atomic<uint64_t> done{0};
    hipError_t e = ensure_smem((const void*)gemm_huge_kernel<EPI>, 5 * SLOT, done);
    if (e != hipSuccess) return e;
    const int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
    // tile order: measured neutral (profiles/r05_experiments.txt r05p: what the other orders save is re-fetched from the Infinity Cache)
    static const int order_env = TTL_EXPERIMENT("TTL_GEMM_HUGE_ORDER", 0);
    const int order = (ntm >= 8) ? order_env : 0;
    const int nslots = order ? 8 * ((ntm + 7) / 8) * ntn : ntm * ntn;
    int grid = nslots < max_blocks ? nslots : max_blocks;
    if (order) grid = (grid / 8) * 8 ? (grid / 8) * 8 : 8;      // whole XCD rounds: slot & 7 must stay the block's XCD label
    hipLaunchKernelGGL((gemm_huge_kernel<EPI>), dim3(grid), dim3(NT), 5 * SLOT, s, a, ntm, ntn, order, nslots);
    return hipGetLastError();
}

}  // namespace

#ifdef TTL_CLOCK_STAMPS
// diagnostic export: the stamps of the LAST gemm_huge launch's workgroups (host buffer of TTL_STAMP_SLOTS x 8 uint64)
extern "C" __attribute__((visibility("default"))) int ttl_diag_clock_stamps_huge(void* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_huge_stamps), sizeof(TtlClockStamp) * TTL_STAMP_SLOTS);
}
#endif

bool gemm_huge_applicable(GemmEpi epi, const GemmArgs& a) {
    // TTL_GEMM_HUGE: 0 = off (everything on gemm_big.hip), 1 = q/k/v and fc1, 2 = q/k/v-shaped (EPI_OP) only (default), 3 = fc1 only.
    // In situ (tools/r05_huge_sweep.sh, fp16, three episodes in flight): q/k/v +1.1 % images/s and the GEMM class one at a time
    // 2.883 -> 2.850 ms; fc1 +0.2 % and 2.883 -> 2.99 ms (600 tiles = three rounds of 256, two 77-MB outputs per launch): fc1 stays
    // on gemm_big.hip, its epilogues here are kept for the tests and for other shapes.
    static const int mode = ttl_switch(SW_GEMM_HUGE);
    // TTL_GEMM_HUGE_DGRAD=1: the MLP dgrad (EPI_GELU_BWD, N = F) too — measured slower (95.9 vs 86.1 us one at a time, no change in situ:
    // its u tile has to be read behind the last MFMAs, profiles/r05_experiments.txt r05q): off
    static const int dgrad = TTL_EXPERIMENT("TTL_GEMM_HUGE_DGRAD", 0);
    if (mode <= 0) return false;
    // The N = 768 / 1024 launches (out_proj, fc2 and their dgrads: fp32 outputs, residual): 150 tiles of 256 x 256 leave 106 CUs idle and
    // take 40 % longer than the 160 x 256 kernel's 237 tiles one at a time — but 11-31 % less CU-time, and with other episodes in flight
    // (GemmArgs::concurrent >= 2: the caller said so, ttl_ctx_set_concurrency) those CUs run their kernels: +2.5 % images/s at three
    // episodes in flight, +1.1 % at two; below tiles for half of the CUs it loses (32 views: -1 %, 8 views: -8 %).
    // TTL_GEMM_HUGE_NARROW: -1 = that rule (default), 0 = never, 1 = whenever the launch has the tiles.  profiles/r05_narrow_ab_fp16.txt
    static const int narrow = ttl_switch(SW_GEMM_HUGE_NARROW);
    if ((epi == EPI_F32 || epi == EPI_RESID_F32 || epi == EPI_OP) && !a.hm_T && a.N < 2304) {
        if (narrow == 0 || (narrow < 0 && a.concurrent < 2)) return false;
        if (a.M < 1024 || a.N % BN || a.K % BK || a.K / BK < 3 || a.amap || a.cmap || a.c2map || a.splits > 1) return false;
        const size_t lim2 = (size_t)1 << 31;
        if ((size_t)a.M * a.lda * sizeof(op_t) >= lim2 || (size_t)a.N * a.ldb * sizeof(op_t) >= lim2 || (size_t)a.M * a.ldc * 4 >= lim2) return false;
        if (epi == EPI_RESID_F32 && (!a.resid || (size_t)a.M * a.ldr * 4 >= lim2)) return false;
        static const int mink = TTL_EXPERIMENT("TTL_GEMM_HUGE_NARROW_MINK", 0);
        static const int maxk = TTL_EXPERIMENT("TTL_GEMM_HUGE_NARROW_MAXK", 1 << 30);
        const long cus2 = device_cu_count();
        return cus2 > 0 && (long)((a.M + BM - 1) / BM) * (a.N / BN) * 2 >= cus2 && a.K >= mink && a.K <= maxk;
    }
    if (epi == EPI_GELU_BWD) { if (!dgrad || !a.aux || (size_t)a.M * a.ldaux * sizeof(op_t) >= ((size_t)1 << 31)) return false; }
    else if (epi != EPI_OP && epi != EPI_GELU) return false;
    else if ((mode == 2 && epi != EPI_OP) || (mode == 3 && epi != EPI_GELU)) return false;
    if (a.M < 1024 || a.N < 2304 || a.N % BN || a.K % BK || a.K / BK < 3 || a.K > 1024) return false;
    if (a.amap || a.cmap || a.c2map || a.splits > 1) return false;
    // The launch has to fill (most of) one round of 256 x 256 tiles over the CUs: below that the 160 x 256 kernel's smaller tiles keep
    // more CUs busy (8 views: 63 tiles here, 90 there: 1157 vs 1168 images/s).  Partial LATER rounds are not held against it: ViT-L/14's
    // 780 tiles (76 % of four rounds) are 5.7 % slower one at a time (107.9 vs 102.1 us) and still +0.8 % images/s with three episodes
    // in flight, where a launch costs tiles x time per tile (tools/r05_huge_fill_ab.sh).  TTL_GEMM_HUGE_MIN_FILL: percent of one round.
    static const int min_fill = ttl_switch(SW_GEMM_HUGE_MIN_FILL);
    const long cus = device_cu_count();
    if (cus <= 0) return false;
    const long tiles = (long)((a.M + BM - 1) / BM) * (a.N / BN);
    if (tiles * 100 < cus * min_fill) return false;
    // 32-bit buffer offsets: every operand / output extent stays below 2 GiB
    const size_t lim = (size_t)1 << 31;
    if ((size_t)a.M * a.lda * sizeof(op_t) >= lim || (size_t)a.N * a.ldb * sizeof(op_t) >= lim) return false;
    if (a.hm_T) { if (((size_t)a.M + a.hm_T) * a.N * sizeof(op_t) >= lim) return false; }
    else if ((size_t)a.M * a.ldc * sizeof(op_t) >= lim || (a.C2 && (size_t)a.M * a.ldc2 * sizeof(op_t) >= lim)) return false;
    return true;
}

hipError_t launch_gemm_huge(GemmEpi epi, const GemmArgs& a, hipStream_t s) {
    static const int blocks_env = TTL_EXPERIMENT("TTL_GEMM_HUGE_BLOCKS", 0);
    const int cus = device_cu_count();
    if (!cus) return hipErrorInvalidDevice;
    const int max_blocks = blocks_env > 0 ? blocks_env : cus;
    if (epi == EPI_OP) {
        if (a.hm_T) {
            if (a.N % 192 || !a.hm_magic || a.hm_T < 1) return hipErrorInvalidValue;
            return launch_huge_t<EPI_OP_HM>(a, max_blocks, s);
        }
        return launch_huge_t<EPI_OP>(a, max_blocks, s);
    }
    if (epi == EPI_GELU) return a.C2 ? launch_huge_t<EPI_GELU_C2>(a, max_blocks, s) : launch_huge_t<EPI_GELU>(a, max_blocks, s);
    if (epi == EPI_GELU_BWD) return launch_huge_t<EPI_DGELU>(a, max_blocks, s);
    if (epi == EPI_F32) return launch_huge_t<EPI_F32>(a, max_blocks, s);
    if (epi == EPI_RESID_F32) return launch_huge_t<EPI_RESID_F32>(a, max_blocks, s);
    return hipErrorInvalidValue;
}
