// bf16 MFMA GEMM for the ViT projections:  C[M,N] = A[M,K] · B[N,K]^T  (+ fused epilogue)
//
// Replaces the nn.Linear / Conv2d matmuls the reference runs through torch (HF
// modeling_clip.py:202-218 patch conv, :309-311 q/k/v, :333 out_proj, :346-350 fc1/fc2) and
// their autograd dgrad counterparts.  gfx950 design:
//   * block tile BM x 128 x 64, 4 waves (2x2), v_mfma_f32_16x16x32_bf16, fp32 accumulate
//   * operands staged HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), two LDS
//     stages, next tile's DMA in flight under the current tile's MFMAs
//   * LDS image is lane-linear (DMA constraint), bank conflicts removed by XOR-swizzling the
//     SOURCE chunk and the ds_read_b128 chunk with the same involution (guide rule 21)
//   * the weight tile is the MFMA A operand with its rows permuted so that every lane ends up
//     with 16 CONTIGUOUS output columns -> 64 B (fp32) / 32 B (bf16) stores per lane per row
//   * 1-D grid, XCD-aware tile order (n fastest: the A panel of a row tile stays in one L2)
#include "kernels.hpp"

namespace {

constexpr int BN = 128;
constexpr int BK = 64;

template <int BM, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WM = BM / 2;       // rows per wave
    constexpr int MT = WM / 16;      // 16-row sub-tiles per wave
    constexpr int NA = BM * 8 / 256; // DMA instructions per thread for the A tile
    constexpr int NB = BN * 8 / 256;
    constexpr int A_BYTES = BM * BK * 2;
    constexpr int STAGE = A_BYTES + BN * BK * 2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;

    const int ntn = a.N / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int row0 = (bid / ntn) * BM;
    const int col0 = (bid % ntn) * BN;
    const int M = a.M;

    // ---- per-thread DMA sources (row fixed for the whole K loop) ----
    const bf16_t* asrc[NA];
    const bf16_t* bsrc[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        int q = i * 256 + tid, r = q >> 3, p = q & 7;
        int c = p ^ ((r >> 1) & 7);
        int gr = min(row0 + r, M - 1);
        asrc[i] = a.A + (size_t)gr * a.lda + c * 8;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        int q = i * 256 + tid, r = q >> 3, p = q & 7;
        int c = p ^ (((r >> 1) & 1) | (((r >> 4) & 3) << 1));
        bsrc[i] = a.B + (size_t)(col0 + r) * a.ldb + c * 8;
    }
    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(asrc[i] + kt * BK), LDS_PTR(base + (i * 256 + wave * 64) * 16),
                                             16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; ++i)
            __builtin_amdgcn_global_load_lds(GLB_PTR(bsrc[i] + kt * BK),
                                             LDS_PTR(base + A_BYTES + (i * 256 + wave * 64) * 16), 16, 0, 0);
    };

    // ---- per-lane fragment addresses ----
    const int swA = (li >> 1) & 7;
    const int swW = ((li >> 1) & 1) | (((li >> 2) & 3) << 1);
    int offA[MT], offW[4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) offA[mt] = (wm * WM + mt * 16 + li) * 128;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) offW[nt] = A_BYTES + (wn * 64 + 16 * (li >> 2) + 4 * nt + (li & 3)) * 128;

    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // tile kt landed (vmcnt(0) + barrier); everyone is done with the other stage
        if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
        const char* base = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int cA = ((4 * s + lg) ^ swA) << 4;
            const int cW = ((4 * s + lg) ^ swW) << 4;
            bf16x8 xf[MT], wf[4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) xf[mt] = *(const bf16x8*)(base + offA[mt] + cA);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) wf[nt] = *(const bf16x8*)(base + offW[nt] + cW);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0);
        }
    }

    // ---- epilogue: lane (lg, li) owns rows m(mt) and 16 contiguous columns n0 .. n0+15 ----
    const int n0 = col0 + wn * 64 + 16 * lg;
    float bias[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) bias[j] = 0.f;
    if (a.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 b = *(const float4*)(a.bias + n0 + 4 * j);
            bias[4 * j] = b.x; bias[4 * j + 1] = b.y; bias[4 * j + 2] = b.z; bias[4 * j + 3] = b.w;
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = row0 + wm * WM + mt * 16 + li;
        if (m >= M) continue;
        float v[16];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * nt + r] = acc[mt][nt][r] + bias[4 * nt + r];

        if constexpr (EPI == EPI_F32 || EPI == EPI_RESID_F32 || EPI == EPI_PATCH) {
            size_t orow = m;
            if constexpr (EPI == EPI_PATCH) {
                int img = m / a.G2, p = m - img * a.G2;
                orow = (size_t)img * a.T + 1 + p;
                const float* pp = a.pos + (size_t)(1 + p) * a.N + n0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 t = *(const float4*)(pp + 4 * j);
                    v[4 * j] += t.x; v[4 * j + 1] += t.y; v[4 * j + 2] += t.z; v[4 * j + 3] += t.w;
                }
            }
            if constexpr (EPI == EPI_RESID_F32) {
                const float* rp = a.resid + (size_t)m * a.ldr + n0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 t = *(const float4*)(rp + 4 * j);
                    v[4 * j] += t.x; v[4 * j + 1] += t.y; v[4 * j + 2] += t.z; v[4 * j + 3] += t.w;
                }
            }
            float* cp = (float*)a.C + orow * a.ldc + n0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(float4*)(cp + 4 * j) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        } else {
            if constexpr (EPI == EPI_GELU) {
                if (a.C2) {
                    bf16_t* up = a.C2 + (size_t)m * a.ldc2 + n0;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        *(u32x4*)(up + 8 * j) = u32x4{pack_bf16x2(v[8 * j], v[8 * j + 1]), pack_bf16x2(v[8 * j + 2], v[8 * j + 3]),
                                                      pack_bf16x2(v[8 * j + 4], v[8 * j + 5]), pack_bf16x2(v[8 * j + 6], v[8 * j + 7])};
                }
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = quick_gelu_f(v[j]);
            }
            if constexpr (EPI == EPI_GELU_BWD) {
                const bf16_t* up = a.aux + (size_t)m * a.ldaux + n0;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    u32x4 t = *(const u32x4*)(up + 8 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[8 * j + 2 * e] *= quick_gelu_grad_f(bf16lo(t[e]));
                        v[8 * j + 2 * e + 1] *= quick_gelu_grad_f(bf16hi(t[e]));
                    }
                }
            }
            bf16_t* cp = (bf16_t*)a.C + (size_t)m * a.ldc + n0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *(u32x4*)(cp + 8 * j) = u32x4{pack_bf16x2(v[8 * j], v[8 * j + 1]), pack_bf16x2(v[8 * j + 2], v[8 * j + 3]),
                                              pack_bf16x2(v[8 * j + 4], v[8 * j + 5]), pack_bf16x2(v[8 * j + 6], v[8 * j + 7])};
        }
    }
}

template <int BM, int EPI>
hipError_t launch_t(const GemmArgs& a, hipStream_t s) {
    constexpr int SMEM = 2 * (BM + BN) * BK * 2;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<BM, EPI>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    int ntm = (a.M + BM - 1) / BM, ntn = a.N / BN;
    hipLaunchKernelGGL((gemm_kernel<BM, EPI>), dim3(ntm * ntn), dim3(256), SMEM, s, a);
    return hipGetLastError();
}

}  // namespace

hipError_t launch_gemm(GemmEpi epi, const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N % BN || a.K % BK || a.K <= 0 || (a.lda & 7) || (a.ldb & 7)) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_F32: return launch_t<128, EPI_F32>(a, s);
        case EPI_BF16: return launch_t<128, EPI_BF16>(a, s);
        case EPI_RESID_F32: return launch_t<128, EPI_RESID_F32>(a, s);
        case EPI_GELU: return launch_t<128, EPI_GELU>(a, s);
        case EPI_PATCH: return launch_t<128, EPI_PATCH>(a, s);
        case EPI_GELU_BWD: return launch_t<128, EPI_GELU_BWD>(a, s);
    }
    return hipErrorInvalidValue;
}
