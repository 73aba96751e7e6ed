// fp32 GEMM of libttl_hip_strict.so — the TEST-ONLY strict-precision build (common.hpp, -DTTL_OPERAND_FP32).
//
// Same contract as gemm.hip / gemm_big.hip (kernels.hpp: GemmArgs, GemmEpi — C[M,N] = A[M,K] · B[N,K]^T with the fused epilogues of
// the ViT projections: HF modeling_clip.py:202-218 patch conv, :309-311 q/k/v, :333 out_proj, :346-350 fc1 / fc2 and their dgrads),
// but every operand buffer holds fp32 and the products run on v_mfma_f32_32x32x2_f32: no operand rounding anywhere, so the launch
// sequences of api.hip can be compared with the reference's fp32 CPU path at 1e-5 (logits) / 1e-4 (gradients) instead of through
// the 2-8e-3 noise of a 16-bit forward.  Simple on purpose: 128 x 128 x 16 tiles through LDS, one tile per block, every row guarded,
// no split-K, no head-major output.  Never benched, never the default; the product builds do not contain this file.
#include "kernels.hpp"

#ifndef TTL_OPERAND_FP32
#error "strict_gemm.hip belongs to the fp32 (strict) build only"
#endif

namespace {

constexpr int BM = 128, BN = 128, BK = 16, LDT = BM + 4;

// accumulator register v of a 32x32 MFMA result: row (v & 3) + 8 (v >> 2) + 4 (lane >> 5), column lane & 31
__device__ __forceinline__ int acc_row(int v, int lane) { return (v & 3) + 8 * (v >> 2) + 4 * (lane >> 5); }

template <int EPI>
__global__ __launch_bounds__(256) void sgemm_kernel(const GemmArgs a) {
    __shared__ float sA[BK][LDT], sB[BK][LDT];     // k-major: a fragment read is 32 consecutive floats per half-wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;       // 2 x 2 waves, 64 x 64 outputs each
    const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;
    const int M = a.M;
    // loader: thread -> (tile row tid / 2, k half tid & 1): 2 x 16 B of A and of B per K-tile
    const int lr = tid >> 1, kh = tid & 1;
    int ar = min(row0 + lr, M - 1);
    if (a.amap) ar = a.amap[ar];
    const float* ap = a.A + (size_t)ar * a.lda + 8 * kh;
    const float* bp = a.B + (size_t)(col0 + lr) * a.ldb + 8 * kh;
    float4 ra0, ra1, rb0, rb1;
    auto gload = [&](int kt) {
        ra0 = *(const float4*)(ap + kt * BK); ra1 = *(const float4*)(ap + kt * BK + 4);
        rb0 = *(const float4*)(bp + kt * BK); rb1 = *(const float4*)(bp + kt * BK + 4);
    };
    auto sstore = [&]() {
        const int k0 = 8 * kh;
        sA[k0 + 0][lr] = ra0.x; sA[k0 + 1][lr] = ra0.y; sA[k0 + 2][lr] = ra0.z; sA[k0 + 3][lr] = ra0.w;
        sA[k0 + 4][lr] = ra1.x; sA[k0 + 5][lr] = ra1.y; sA[k0 + 6][lr] = ra1.z; sA[k0 + 7][lr] = ra1.w;
        sB[k0 + 0][lr] = rb0.x; sB[k0 + 1][lr] = rb0.y; sB[k0 + 2][lr] = rb0.z; sB[k0 + 3][lr] = rb0.w;
        sB[k0 + 4][lr] = rb1.x; sB[k0 + 5][lr] = rb1.y; sB[k0 + 6][lr] = rb1.z; sB[k0 + 7][lr] = rb1.w;
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[mi][ni][v] = 0.f;
    const int nk = a.K / BK;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();            // everyone is done reading the previous K-tile
        sstore();
        __syncthreads();
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + (lane >> 5);
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = sA[k][wm * 64 + i * 32 + (lane & 31)];
                bf[i] = sB[k][wn * 64 + i * 32 + (lane & 31)];
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
        }
    }
    // ---- epilogue (the formulas of gemm_epilogue.hpp, element by element)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = col0 + wn * 64 + ni * 32 + (lane & 31);
        const float bias = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int m = row0 + wm * 64 + mi * 32 + acc_row(v, lane);
                if (m >= M) continue;
                const size_t pc = a.cmap ? a.cmap[m] : m;     // physical row of C / resid
                float val = acc[mi][ni][v] + bias;
                if constexpr (EPI == EPI_F32) ((float*)a.C)[pc * a.ldc + n] = val;
                else if constexpr (EPI == EPI_OP) ((op_t*)a.C)[pc * a.ldc + n] = val;
                else if constexpr (EPI == EPI_RESID_F32) ((float*)a.C)[pc * a.ldc + n] = a.resid[pc * a.ldr + n] + val;
                else if constexpr (EPI == EPI_GELU) {
                    if (a.C2) a.C2[(size_t)(a.c2map ? a.c2map[m] : m) * a.ldc2 + n] = val;
                    ((op_t*)a.C)[pc * a.ldc + n] = quick_gelu_f(val);
                } else if constexpr (EPI == EPI_PATCH) {
                    const int img = m / a.G2, p = m - img * a.G2;
                    ((float*)a.C)[((size_t)img * a.T + 1 + p) * a.ldc + n] = val + a.pos[(size_t)(1 + p) * a.N + n];
                } else if constexpr (EPI == EPI_GELU_BWD)
                    ((op_t*)a.C)[pc * a.ldc + n] = val * quick_gelu_grad_f(a.aux[(size_t)m * a.ldaux + n]);
            }
    }
}

template <int EPI>
hipError_t launch_e(const GemmArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(sgemm_kernel<EPI>, dim3(a.N / BN, (a.M + BM - 1) / BM), dim3(256), 0, s, a);
    return hipGetLastError();
}

}  // namespace

// the 16-bit builds' big-M kernel (and its head-major q/k/v epilogue) does not exist here
bool gemm_big_applicable(GemmEpi, const GemmArgs&) { return false; }
bool gemm_takes_big(GemmEpi, const GemmArgs&) { return false; }
hipError_t launch_gemm_big(GemmEpi, const GemmArgs&, hipStream_t) { return hipErrorInvalidValue; }

hipError_t launch_gemm(GemmEpi epi, const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N % 128 || a.K % 64 || a.K <= 0 || (a.lda & 7) || (a.ldb & 7)) return hipErrorInvalidValue;   // gemm.hip's contract
    if (a.hm_T) return hipErrorInvalidValue;
    switch (epi) {
        case EPI_F32: return launch_e<EPI_F32>(a, s);
        case EPI_OP: return launch_e<EPI_OP>(a, s);
        case EPI_RESID_F32: return launch_e<EPI_RESID_F32>(a, s);
        case EPI_GELU: return launch_e<EPI_GELU>(a, s);
        case EPI_PATCH: return launch_e<EPI_PATCH>(a, s);
        case EPI_GELU_BWD: return launch_e<EPI_GELU_BWD>(a, s);
    }
    return hipErrorInvalidValue;
}
