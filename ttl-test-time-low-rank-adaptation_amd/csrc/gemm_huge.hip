// EXPERIMENT (round 5, r05l; off unless TTL_GEMM_HUGE=1): 256 x 256 x 64 tiles on FOUR waves per CU, every wave a 128 x 128 output
// block (64 f32x4 accumulators = 256 registers: one wave per SIMD may use all 512), operands prefetched global -> registers one
// K-tile ahead and written to a double-buffered LDS image with ds_write_b128.  Why: rocprofv3 of torch.matmul on the episode's
// shapes (profiles/r05_experiments.txt r05i) shows that hipBLASLt serves N >= 2304, K = 768 with exactly this shape
// (MT256x256x64, 256 threads, stream-K) at 15-18 % less CU-time per launch than gemm_big_kernel<5,3,*> — 128 instead of 98 FLOP
// per staged byte, and with three episodes in flight CU-time per tile is what a launch costs (r05j / r05k), not its round count.
// The LDS image, swizzle, weight-row permutation and store layout are those of gemm.hip / gemm_big.hip.
#include <stdlib.h>

#include <atomic>

#include "gemm_epilogue.hpp"
#include "kernels.hpp"

namespace {

constexpr int HBM_ = 256, HBN = 256, HBK = 64, HTHR = 256;
constexpr int H_ABYTES = HBM_ * 128, H_STAGE = H_ABYTES + HBN * 128;      // 64 KiB per K-tile

constexpr int HEPI_OP = 1, HEPI_OP_HM = 101, HEPI_GELU = 3, HEPI_GELU_C2 = 100;

__device__ __forceinline__ size_t hm_base(const GemmArgs& a, int n0) {
    const int Dm = a.N / 3;
    const int plane = (n0 >= Dm) + (n0 >= 2 * Dm);
    const int rem = n0 - plane * Dm;
    return ((size_t)plane * Dm + (size_t)(rem & ~63)) * a.hm_T + (rem & 63);
}

template <int EPI>
__global__ __launch_bounds__(HTHR) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_huge_kernel(const GemmArgs a, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int M = a.M, nk = a.K / HBK;
    const int ntiles = ntm * ntn;

    // fragment addresses (gemm_big.hip): row (tile row + li) of a 128-B-per-row image, 16-B chunk (4s + lg) ^ swizzle(row)
    const int swA = (li >> 1) & 7;
    const int fA0 = (wm * 128 + li) * 128 + ((lg ^ swA) << 4);
    const int fW0 = H_ABYTES + (wn * 128 + li) * 128 + ((lg ^ swA) << 4);
    // staging: chunk q = i*256 + tid -> image row q >> 3, 16-B piece q & 7 (global: contiguous 128 B per row; LDS: swizzled slot)
    const int sr = tid >> 3, sp = tid & 7;
    const int lds_st = sr * 128 + ((sp ^ ((sr >> 1) & 7)) << 4);          // + i * 4096 (rows + 32: same swizzle)

    for (int slot = blockIdx.x; slot < ntiles; slot += gridDim.x) {
        const int t = xcd_remap(slot, ntiles);
        const int rt = t / ntn, ct = t - rt * ntn;
        const int row0 = rt * HBM_, col0 = ct * HBN;
        // A rows: 32-bit element offsets (the last row tile clamps to row M - 1); B rows: wave-uniform strides from ONE per-lane base
        uint32_t ao[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ao[i] = (uint32_t)min(row0 + i * 32 + sr, M - 1) * (uint32_t)a.lda + (uint32_t)(sp * 8);
        // physical LDS row r = 32 i + sr holds output column (r & ~63) + 4 (r & 15) + ((r >> 4) & 3)
        const op_t* gb0 = a.B + (size_t)(col0 + 4 * (sr & 15) + (sr >> 4)) * a.ldb + sp * 8;
        u32x4 ra[8], rb[8];
        auto gload = [&](int kt) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ra[i] = *(const u32x4*)(a.A + ao[i] + kt * HBK);
                rb[i] = *(const u32x4*)(gb0 + (size_t)((i >> 1) * 64 + (i & 1) * 2) * a.ldb + kt * HBK);
            }
        };
        auto lstore = [&](char* sb) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { *(u32x4*)(sb + lds_st + i * 4096) = ra[i]; *(u32x4*)(sb + H_ABYTES + lds_st + i * 4096) = rb[i]; }
        };
        f32x4 acc[8][2][4];
        {
            f32x4 bv[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            if (a.bias) { bv[0] = *(const f32x4*)(a.bias + col0 + wn * 128 + 4 * li); bv[1] = *(const f32x4*)(a.bias + col0 + wn * 128 + 64 + 4 * li); }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][h][nt] = f32x4{bv[h][nt], bv[h][nt], bv[h][nt], bv[h][nt]};
        }
        __syncthreads();                 // the previous tile's last reads of both LDS buffers are done
        gload(0);
        lstore(smem);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const char* cur = smem + (kt & 1) * H_STAGE;
            if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                opx8 xf[8], wf[2][4];
                const char* pa = cur + (fA0 ^ (s << 6));
                const char* pw = cur + (fW0 ^ (s << 6));
#pragma unroll
                for (int mt = 0; mt < 8; ++mt) xf[mt] = *(const opx8*)(pa + mt * 2048);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) wf[h][nt] = *(const opx8*)(pw + h * 8192 + nt * 2048);
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[mt][h][nt] = MFMA16(xf[mt], wf[h][nt], acc[mt][h][nt], 0, 0, 0);
            }
            if (kt + 1 < nk) lstore(smem + ((kt + 1) & 1) * H_STAGE);
            __syncthreads();
        }
        // ---- epilogue: register r of (mt, h, nt) is row 16 mt + 4 lg + r, column 64 h + 4 li + nt of the wave's 128 x 128 block
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            size_t n0 = (size_t)(col0 + wn * 128 + h * 64 + 4 * li);
            if constexpr (EPI == HEPI_OP_HM) n0 = hm_base(a, (int)n0);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t m = (size_t)(row0 + wm * 128 + mt * 16 + 4 * lg + r);
                    float v0 = acc[mt][h][0][r], v1 = acc[mt][h][1][r], v2 = acc[mt][h][2][r], v3 = acc[mt][h][3][r];
                    if constexpr (EPI == HEPI_OP_HM) {
                        const unsigned view = __umulhi((unsigned)m, a.hm_magic), tk = (unsigned)m - view * (unsigned)a.hm_T;
                        *(u32x2*)((op_t*)a.C + (size_t)view * a.N * a.hm_T + n0 + (size_t)tk * 64) = u32x2{pack_op2(v0, v1), pack_op2(v2, v3)};
                    } else if constexpr (EPI == HEPI_GELU || EPI == HEPI_GELU_C2) {
                        if constexpr (EPI == HEPI_GELU_C2) __builtin_nontemporal_store(u32x2{pack_op2(v0, v1), pack_op2(v2, v3)}, (u32x2*)(a.C2 + m * a.ldc2 + n0));
                        v0 = quick_gelu_f(v0); v1 = quick_gelu_f(v1); v2 = quick_gelu_f(v2); v3 = quick_gelu_f(v3);
                        __builtin_nontemporal_store(u32x2{pack_op2(v0, v1), pack_op2(v2, v3)}, (u32x2*)((op_t*)a.C + m * a.ldc + n0));
                    } else {
                        *(u32x2*)((op_t*)a.C + m * a.ldc + n0) = u32x2{pack_op2(v0, v1), pack_op2(v2, v3)};
                    }
                }
        }
    }
}

template <int EPI>
hipError_t launch_huge_t(const GemmArgs& a, hipStream_t s) {
    constexpr int SMEM = 2 * H_STAGE;
    static std::atomic<uint64_t> done{0};
    hipError_t e = ensure_smem((const void*)gemm_huge_kernel<EPI>, SMEM, done);
    if (e != hipSuccess) return e;
    const int cus = device_cu_count();
    if (!cus) return hipErrorInvalidDevice;
    const int ntm = (a.M + HBM_ - 1) / HBM_, ntn = a.N / HBN;
    const int grid = ntm * ntn < cus ? ntm * ntn : cus;
    hipLaunchKernelGGL((gemm_huge_kernel<EPI>), dim3(grid), dim3(HTHR), SMEM, s, a, ntm, ntn);
    return hipGetLastError();
}

}  // namespace

// big-M launches with wide outputs and short K (QKV, fc1): rows the unguarded epilogue may store = round_up(M, 256) <= a.padded
bool gemm_huge_applicable(GemmEpi epi, const GemmArgs& a) {
    static const int on = [] { const char* v = getenv("TTL_GEMM_HUGE"); return v ? atoi(v) : 0; }();
    if (!on) return false;
    if (epi != EPI_OP && epi != EPI_GELU) return false;
    if (a.M < 1024 || a.N % HBN || a.N < 2304 || a.K % HBK || a.K > 1024) return false;
    if (a.amap || a.cmap || a.c2map || a.splits > 1) return false;
    if ((size_t)((a.M + HBM_ - 1) / HBM_) * HBM_ > (size_t)a.padded) return false;
    return true;
}

hipError_t launch_gemm_huge(GemmEpi epi, const GemmArgs& a, hipStream_t s) {
    if (epi == EPI_OP) return a.hm_T ? launch_huge_t<HEPI_OP_HM>(a, s) : launch_huge_t<HEPI_OP>(a, s);
    if (epi == EPI_GELU) return a.C2 ? launch_huge_t<HEPI_GELU_C2>(a, s) : launch_huge_t<HEPI_GELU>(a, s);
    return hipErrorInvalidValue;
}
