// Epilogue shared by the GEMM kernels (gemm.hip: 160x128 / 32x64 tiles; gemm_big.hip: 32*MT x 256 tiles).
#pragma once
#include "kernels.hpp"

namespace {

// Output stores.  Measured (rocprofv3 --pmc FETCH_SIZE): the 58-155 MB of output a launch writes evict the
// weight slice / activation panels from the 4 MiB L2s, e.g. fc1 fetches 118 MB where 24 MB are algorithmic;
// non-temporal stores cut that to 64 MB but the GEMM is not faster (the re-reads hit the Infinity Cache) and
// the CONSUMER kernels lose their Infinity-Cache hits (attention forward +11 %, episode +3.7 %): plain stores stay.
#ifndef TTL_GEMM_NT_STORE
#define TTL_GEMM_NT_STORE 0
#endif
// Exception (TTL_GEMM_NT_GELU: 1 = g, 2 = g and u; 2 used): fc1's outputs, 77-155 MB per launch.  g is consumed once by
// fc2, u only by the backward; storing them non-temporally cuts the launch's HBM-side traffic (all big-M GEMMs: 158 ->
// 149 MB per launch) and is +0.8 % images/s with one and with three episodes in flight.
#ifndef TTL_GEMM_NT_GELU
#define TTL_GEMM_NT_GELU 2
#endif
template <typename V>
__device__ __forceinline__ void st_out(V* p, V v) {
#if TTL_GEMM_NT_STORE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// ---- epilogue shared by both kernels: accumulator register r of sub-tile (mt, nt) is row
// 16*mt + 4*lg + r, column 4*li + nt of the wave's 64-column slab: per (mt, r) a lane owns 4
// contiguous columns, so one store instruction writes 4 rows x 256 B (fp32) / 128 B (bf16).
// GUARD = false: the caller guarantees that every row of the last row tile exists in all output /
// residual / aux buffers (the context pads its arena), so the epilogue is straight-line code: with
// a per-row branch hipcc re-waits vmcnt(0) in every store block and the stores serialise.
template <int EPI, int MT, bool GUARD>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, const f32x4 (&acc)[MT][4], int rbase, int n0, int lg, int M,
                                              const u32x2 (*auxr)[4] = nullptr) {
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bias = *(const float4*)(a.bias + n0);
    // Guarded epilogues that ADD something per row (residual / position embedding): all of the lane's rows are requested first, with
    // clamped row indices — a load behind the per-row guard makes hipcc wait vmcnt(0) in front of every store block, i.e. one serial
    // round trip per row (20 of them in the patch-embedding launch, 4 in every small-M residual launch).
    constexpr bool PRE = GUARD && (EPI == EPI_RESID_F32 || EPI == EPI_PATCH);
    float4 pre[PRE ? MT : 1][4];
    int prow[PRE ? MT : 1][4];
    if constexpr (PRE) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = min(rbase + mt * 16 + 4 * lg + r, M - 1);
                prow[mt][r] = a.cmap ? a.cmap[m] : m;
            }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (EPI == EPI_PATCH) {
                    const int m = min(rbase + mt * 16 + 4 * lg + r, M - 1);
                    const int img = m / a.G2, p = m - img * a.G2;
                    pre[mt][r] = *(const float4*)(a.pos + (size_t)(1 + p) * a.N + n0);
                } else pre[mt][r] = *(const float4*)(a.resid + (size_t)prow[mt][r] * a.ldr + n0);
            }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = rbase + mt * 16 + 4 * lg + r;
            if (GUARD && m >= M) continue;
            int pc = m, p2 = m;          // physical rows of C / resid and of C2 (row maps: small-M launches only)
            if constexpr (PRE) { pc = prow[mt][r]; if (a.c2map) p2 = a.c2map[m]; }
            else if constexpr (GUARD) { if (a.cmap) pc = a.cmap[m]; if (a.c2map) p2 = a.c2map[m]; }
            float v0 = acc[mt][0][r] + bias.x, v1 = acc[mt][1][r] + bias.y, v2 = acc[mt][2][r] + bias.z, v3 = acc[mt][3][r] + bias.w;
            if constexpr (EPI == EPI_F32 || EPI == EPI_RESID_F32 || EPI == EPI_PATCH) {
                size_t orow = pc;
                if constexpr (EPI == EPI_PATCH) {
                    int img = m / a.G2, p = m - img * a.G2;
                    orow = (size_t)img * a.T + 1 + p;
                    float4 t;
                    if constexpr (PRE) t = pre[mt][r]; else t = *(const float4*)(a.pos + (size_t)(1 + p) * a.N + n0);
                    v0 += t.x; v1 += t.y; v2 += t.z; v3 += t.w;
                }
                if constexpr (EPI == EPI_RESID_F32) {
                    float4 t;
                    if constexpr (PRE) t = pre[mt][r]; else t = *(const float4*)(a.resid + (size_t)pc * a.ldr + n0);
                    v0 += t.x; v1 += t.y; v2 += t.z; v3 += t.w;
                }
                st_out((f32x4*)((float*)a.C + orow * a.ldc + n0), f32x4{v0, v1, v2, v3});
            } else {
                if constexpr (EPI == EPI_GELU) {
                    if (a.C2) {
                        if (TTL_GEMM_NT_GELU == 2) __builtin_nontemporal_store(u32x2{pack_op2(v0, v1), pack_op2(v2, v3)}, (u32x2*)(a.C2 + (size_t)p2 * a.ldc2 + n0));
                        else st_out((u32x2*)(a.C2 + (size_t)p2 * a.ldc2 + n0), u32x2{pack_op2(v0, v1), pack_op2(v2, v3)});
                    }
                    v0 = quick_gelu_f(v0); v1 = quick_gelu_f(v1); v2 = quick_gelu_f(v2); v3 = quick_gelu_f(v3);
                }
                if constexpr (EPI == EPI_GELU_BWD) {
                    u32x2 t = auxr ? auxr[mt][r] : *(const u32x2*)(a.aux + (size_t)m * a.ldaux + n0);
                    v0 *= quick_gelu_grad_f(op_lo(t[0])); v1 *= quick_gelu_grad_f(op_hi(t[0]));
                    v2 *= quick_gelu_grad_f(op_lo(t[1])); v3 *= quick_gelu_grad_f(op_hi(t[1]));
                }
                if (TTL_GEMM_NT_GELU && EPI == EPI_GELU)
                    __builtin_nontemporal_store(u32x2{pack_op2(v0, v1), pack_op2(v2, v3)}, (u32x2*)((op_t*)a.C + (size_t)pc * a.ldc + n0));
                else
                st_out((u32x2*)((op_t*)a.C + (size_t)pc * a.ldc + n0), u32x2{pack_op2(v0, v1), pack_op2(v2, v3)});
            }
        }
    }
}

}  // namespace
