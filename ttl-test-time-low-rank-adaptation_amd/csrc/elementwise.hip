// HBM-bound helpers around the GEMMs: casts, im2col of the N x 3 x S x S view batch, CLS rows,
// LayerNorm forward/backward (wave per row, fp32 statistics).
// Reference ops replaced: nn.LayerNorm (HF modeling_clip.py:605,358-360,607) and its autograd;
// Conv2d input unfolding (:202-218).
#include <stdlib.h>

#include "kernels.hpp"

namespace {

__global__ void cast_kernel(const float* __restrict__ src, op_t* __restrict__ dst, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    size_t stride = (size_t)gridDim.x * blockDim.x * 8;
    for (; i + 8 <= n; i += stride) {
        float4 a = *(const float4*)(src + i), b = *(const float4*)(src + i + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        st_op8(dst + i, v);
    }
    // tail (n % 8) handled by the first thread
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (size_t j = n & ~(size_t)7; j < n; ++j) dst[j] = f32_to_op(src[j]);
}

__global__ void cast_rows_kernel(const float* __restrict__ src, int rows, int cols, op_t* __restrict__ dst, int ld) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = (size_t)rows * cols;
    if (i >= n) return;
    int r = (int)(i / cols), c = (int)(i - (size_t)r * cols);
    dst[(size_t)r * ld + c] = f32_to_op(src[i]);
}

// 32x32 LDS-tiled transpose + cast
__global__ void transpose_kernel(const float* __restrict__ src, int R, int C, op_t* __restrict__ dst, int ld) {
    __shared__ float tile[32][33];
    int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8) {
        int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int c = c0 + j, r = r0 + tx;  // dst[c][r]
        if (c < C && r < R) dst[(size_t)c * ld + r] = f32_to_op(tile[tx][j]);
    }
}

// One thread converts 8 consecutive pixels of one image row: 32 B coalesced reads, 16 B writes
// landing in the patch-major im2col matrix the patch GEMM consumes.
__global__ void im2col_kernel(const float* __restrict__ x, op_t* __restrict__ out, int n, int S, int P, int Kp) {
    const int G = S / P;
    const int W8 = S / 8;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n * 3 * S * W8;
    if (t >= total) return;
    int xs = (int)(t % W8) * 8;
    size_t rest = t / W8;
    int y = (int)(rest % S); rest /= S;
    int c = (int)(rest % 3);
    int img = (int)(rest / 3);
    const float* src = x + (((size_t)img * 3 + c) * S + y) * S + xs;
    float4 a = *(const float4*)src, b = *(const float4*)(src + 4);
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    int gy = y / P, py = y - gy * P;
    if (P % 8 == 0) {
        int gx = xs / P, px = xs - gx * P;
        op_t* d = out + ((size_t)(img * G + gy) * G + gx) * Kp + (c * P + py) * P + px;
        st_op8(d, v);
    } else {  // P = 14 (ViT-L/14): 8-pixel groups straddle patches
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int xx = xs + e, gx = xx / P, px = xx - gx * P;
            out[((size_t)(img * G + gy) * G + gx) * Kp + (c * P + py) * P + px] = f32_to_op(v[e]);
        }
    }
}

__global__ void cls_rows_kernel(float* __restrict__ h, const float* __restrict__ cls, const float* __restrict__ pos,
                                int n, int T, int D) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * D) return;
    int img = i / D, d = i - img * D;
    h[(size_t)img * T * D + d] = cls[d] + pos[d];
}

// ---- LayerNorm: one wave per row, D <= 1024, D % 4 == 0.  Lane l owns float4 chunks l, l+64, ...
constexpr int LN_MAXC = 4;

// RPW rows per wave: all of a wave's row loads are requested before the first reduction (more bytes in flight per wave).
// Measured in situ (tools/ew_macro_ab.sh TTL_LN_RPW 1 2 4): 0.55 / 0.57 / 0.60 ms of this class per episode — one row per
// wave (more waves in flight) is the fastest; the kernel runs at 4.4 TB/s of its 58 MB (profiles/r02_hbm_kernels.json).
#ifndef TTL_LN_RPW
#define TTL_LN_RPW 1
#endif
template <int RPW>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long long row_stride,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y32, op_t* __restrict__ y16, int ld16,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int D, float eps, const int* __restrict__ rowmap) {
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    int lane = threadIdx.x & 63;
    const int nch = D >> 2;
    float4 v[RPW][LN_MAXC];
    float s[RPW];
    // gamma / beta travel with the row (clamped index, no branch): loaded behind the reductions they were a second, serial round trip
    float4 gam[LN_MAXC], bet[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = min(lane + 64 * i, nch - 1);
        gam[i] = *(const float4*)(gamma + 4 * c); bet[i] = *(const float4*)(beta + 4 * c);
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = min(row0 + r, rows - 1);
        const float* xr = x + (size_t)(rowmap ? rowmap[row] : row) * row_stride;
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            int c = lane + 64 * i;
#if defined(TTL_LN_VARIANT) && TTL_LN_VARIANT == 3
            if (c < nch) { const float* pp = xr + 4 * c; v[r][i] = make_float4(__builtin_nontemporal_load(pp), __builtin_nontemporal_load(pp + 1), __builtin_nontemporal_load(pp + 2), __builtin_nontemporal_load(pp + 3)); }
#else
            if (c < nch) v[r][i] = *(const float4*)(xr + 4 * c);
#endif
            else v[r][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) s[r] += (v[r][i].x + v[r][i].y) + (v[r][i].z + v[r][i].w);   // (zero beyond nch)
    }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int row = row0 + r;
        if (row >= rows) break;
        float mu = wave_sum(s[r]) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            int c = lane + 64 * i;
            if (c < nch) {
                float a = v[r][i].x - mu, b = v[r][i].y - mu, cc = v[r][i].z - mu, d = v[r][i].w - mu;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        float rs = rsqrtf(wave_sum(q) / (float)D + eps);
        if (lane == 0) {
            if (mean) mean[row] = mu;
            if (rstd) rstd[row] = rs;
        }
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            int c = lane + 64 * i;
            if (c < nch) {
                const float4 g = gam[i], b = bet[i];
                float o0 = (v[r][i].x - mu) * rs * g.x + b.x, o1 = (v[r][i].y - mu) * rs * g.y + b.y;
                float o2 = (v[r][i].z - mu) * rs * g.z + b.z, o3 = (v[r][i].w - mu) * rs * g.w + b.w;
                if (y32) *(float4*)(y32 + (size_t)row * D + 4 * c) = make_float4(o0, o1, o2, o3);
                if (y16) st_op4(y16 + (size_t)row * ld16 + 4 * c, o0, o1, o2, o3);
            }
        }
    }
}

// TTL_LN_VARIANT (tools/ew_macro_ab.sh): 0 = one row per wave, one wave per row (rounds 1-3); 2 = persistent waves (TTL_LN_PBLK
// blocks of 4 per CU) walking rows with the NEXT row's loads in flight while the current row is reduced and written (round 4, used
// for the big launches); 3 = variant 0 with non-temporal loads of x.  In situ, layernorm / elementwise class per episode:
// 0.545 ms (0), 0.516 / 0.513 / 0.525 (2 with 4 / 8 / 2 blocks per CU), 0.562 (3): profiles/r04_ln_variants.txt.
#ifndef TTL_LN_VARIANT
#define TTL_LN_VARIANT 2
#endif
#if TTL_LN_VARIANT == 2
// TTL_LN_HOLD_GB 0: gamma / beta are re-read per row (L1 hits) instead of held in 32 registers, and the kernel is capped at 64 VGPRs —
// what is left per SIMD beside two big-M GEMM waves (2 x 224 of 512), so that its blocks can run on a CU another episode's GEMM holds
#ifndef TTL_LN_HOLD_GB
#define TTL_LN_HOLD_GB 1
#endif
#if TTL_LN_HOLD_GB
#define LN_PERSIST_ATTR
#else
#define LN_PERSIST_ATTR __attribute__((amdgpu_waves_per_eu(8)))
#endif
template <int NC>     // NC float4 chunks per lane: 3 for D <= 768 (ViT-B, exact: no masks), 4 up to D = 1024
__global__ __launch_bounds__(256) LN_PERSIST_ATTR void ln_fwd_persist_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, op_t* __restrict__ y16, int ld16,
                                                             float* __restrict__ mean, float* __restrict__ rstd, int rows, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int nw = gridDim.x * 4, w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nch = D >> 2;
#if TTL_LN_HOLD_GB
    float4 g[NC], b[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = min(lane + 64 * i, nch - 1);
        g[i] = *(const float4*)(gamma + 4 * c); b[i] = *(const float4*)(beta + 4 * c);
    }
#define LN_G(i) g[i]
#define LN_B(i) b[i]
#else
#define LN_G(i) (*(const float4*)(gp + 4 * c))
#define LN_B(i) (*(const float4*)(bp + 4 * c))
#endif
    float4 cur[NC], nxt[NC];
    int row = w;
    if (row >= rows) return;
#pragma unroll
    for (int i = 0; i < NC; ++i) cur[i] = *(const float4*)(x + (size_t)row * D + 4 * min(lane + 64 * i, nch - 1));
    for (; row < rows; row += nw) {
        const int rn = min(row + nw, rows - 1);
#pragma unroll
        for (int i = 0; i < NC; ++i) nxt[i] = *(const float4*)(x + (size_t)rn * D + 4 * min(lane + 64 * i, nch - 1));
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            if (lane + 64 * i >= nch) cur[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            s += (cur[i].x + cur[i].y) + (cur[i].z + cur[i].w);
        }
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i)
            if (lane + 64 * i < nch) {
                const float a0 = cur[i].x - mu, a1 = cur[i].y - mu, a2 = cur[i].z - mu, a3 = cur[i].w - mu;
                q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
        if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#if !TTL_LN_HOLD_GB
        const float *gp = gamma, *bp = beta;
        asm volatile("" : "+s"(gp), "+s"(bp));      // keeps the loop-invariant gamma / beta loads from being hoisted back into 24 registers
#endif
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const float4 gg = LN_G(i), bb = LN_B(i);
                const float o0 = (cur[i].x - mu) * rs * gg.x + bb.x, o1 = (cur[i].y - mu) * rs * gg.y + bb.y;
                const float o2 = (cur[i].z - mu) * rs * gg.z + bb.z, o3 = (cur[i].w - mu) * rs * gg.w + bb.w;
                st_op4(y16 + (size_t)row * ld16 + 4 * c, o0, o1, o2, o3);
            }
        }
#pragma unroll
        for (int i = 0; i < NC; ++i) cur[i] = nxt[i];
    }
}
#endif

// dx = rstd * (dxh - mean(dxh) - xh * mean(dxh * xh)),  dxh = dy * gamma,  xh = (x - mean) * rstd
template <int NC>     // float4 chunks per lane: 3 for D <= 768 (54 VGPRs: fits beside two big-M GEMM waves per SIMD), 4 up to D = 1024 (66)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ dres,
                                                     float* __restrict__ o32, op_t* __restrict__ o16, int rows, int D,
                                                     long long xs, long long os, int stat_stride, int dres_T,
                                                     const int* __restrict__ pool, long long os16) {
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int lane = threadIdx.x & 63;
    const int nch = D >> 2;
    const float mu = mean[(size_t)row * stat_stride], rs = rstd[(size_t)row * stat_stride];
    // dres_T > 0: the residual gradient is non-zero only on the pooled row of every sequence (token 0 =
    // CLS for the image tower, pool[v] = end-of-text position for the text tower) and is stored
    // compactly as [rows / dres_T][D]
    const float* dres_row = nullptr;
    if (dres) {
        if (dres_T > 0) {
            const int v = row / dres_T, t = row - v * dres_T;
            if (t == (pool ? pool[v] : 0)) dres_row = dres + (size_t)v * D;
        } else dres_row = dres + (size_t)row * os;
    }
    // every load of the row is requested before the first use, with clamped chunk indices and selects instead of branches (a guard
    // around a load makes hipcc wait vmcnt(0) right behind it: the guarded version of this kernel was seven serial round trips per row)
    float4 d[NC], xv[NC], g[NC], rr[NC];
    const float* rsrc = dres_row ? dres_row : dy + (size_t)row * D;     // (no residual gradient: any readable row, zero selected below)
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = min(lane + 64 * i, nch - 1);
        d[i] = *(const float4*)(dy + (size_t)row * D + 4 * c);
        xv[i] = *(const float4*)(x + (size_t)row * xs + 4 * c);
        g[i] = *(const float4*)(gamma + 4 * c);
        rr[i] = *(const float4*)(rsrc + 4 * c);
    }
    float4 dxh[NC], xh[NC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        int c = lane + 64 * i;
        if (c < nch) {
            dxh[i] = make_float4(d[i].x * g[i].x, d[i].y * g[i].y, d[i].z * g[i].z, d[i].w * g[i].w);
            xh[i] = make_float4((xv[i].x - mu) * rs, (xv[i].y - mu) * rs, (xv[i].z - mu) * rs, (xv[i].w - mu) * rs);
            s1 += (dxh[i].x + dxh[i].y) + (dxh[i].z + dxh[i].w);
            s2 += (dxh[i].x * xh[i].x + dxh[i].y * xh[i].y) + (dxh[i].z * xh[i].z + dxh[i].w * xh[i].w);
        }
    }
    const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        int c = lane + 64 * i;
        if (c < nch) {
            const float4 r = dres_row ? rr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            float o0 = r.x + rs * (dxh[i].x - m1 - xh[i].x * m2), o1 = r.y + rs * (dxh[i].y - m1 - xh[i].y * m2);
            float o2 = r.z + rs * (dxh[i].z - m1 - xh[i].z * m2), o3 = r.w + rs * (dxh[i].w - m1 - xh[i].w * m2);
            if (o32) *(float4*)(o32 + (size_t)row * os + 4 * c) = make_float4(o0, o1, o2, o3);
            if (o16) st_op4(o16 + (size_t)row * os16 + 4 * c, o0, o1, o2, o3);
        }
    }
}

// text tower embedding (HF CLIPTextEmbeddings): h[p][t][:] = tok[ids[p][t]][:] + pos[t][:]
__global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ ids, const float* __restrict__ tok,
                                                         const float* __restrict__ pos, float* __restrict__ h, int rows, int T, int D) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // one float4 each
    const int nch = D >> 2;
    if (i >= (size_t)rows * nch) return;
    const int row = (int)(i / nch), c = (int)(i - (size_t)row * nch);
    const float4 a = *(const float4*)(tok + (size_t)ids[row] * D + 4 * c);
    const float4 b = *(const float4*)(pos + (size_t)(row % T) * D + 4 * c);
    *(float4*)(h + (size_t)row * D + 4 * c) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// dst[v][0..cols) = src[(v*T + pool[v]) * ld + 0..cols)   (pooled rows of every sequence -> compact)
template <typename Tp>
__global__ __launch_bounds__(256) void gather_rows_kernel(const Tp* __restrict__ src, long long ld, const int* __restrict__ pool,
                                                          int T, Tp* __restrict__ dst, int n, int cols) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)n * cols) return;
    const int v = (int)(i / cols), c = (int)(i - (size_t)v * cols);
    dst[i] = src[((size_t)v * T + pool[v]) * ld + c];
}

// blockIdx = (chunk, selected view j, entry): dst block j of the entry = its src block idx[j]
__global__ __launch_bounds__(256) void gather_view_blocks_kernel(const GatherTable t, const long long* __restrict__ idx) {
    const GatherEntry e = t.e[blockIdx.z];
    const int j = blockIdx.y;
    const char* src = (const char*)e.src + (size_t)idx[j] * e.stride_bytes;
    char* dst = (char*)e.dst + (size_t)j * e.block_bytes;
    const size_t nth = (size_t)gridDim.x * 256, i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (((e.block_bytes | e.stride_bytes | (size_t)e.src | (size_t)e.dst) & 15) == 0) {
        for (size_t i = i0; i < e.block_bytes / 16; i += nth) ((u32x4*)dst)[i] = ((const u32x4*)src)[i];
    } else {
        for (size_t i = i0; i < e.block_bytes / 4; i += nth) ((uint32_t*)dst)[i] = ((const uint32_t*)src)[i];
    }
}

// dst[c][r] = src[r][c]  (small fp32 matrices: logits between [views,prompts] and [prompts,views])
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, int R, int C, float* __restrict__ dst) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)R * C) return;
    const int c = (int)(i / R), r = (int)(i - (size_t)c * R);
    dst[i] = src[(size_t)r * C + c];
}

// one block per row: dst[v][:] = src[v][:] / ||src[v]|| (or a plain copy), dstT[:][v] likewise
__global__ __launch_bounds__(256) void unit_rows_kernel(const float* __restrict__ src, int n, int E, int normalize,
                                                        float* __restrict__ dst, float* __restrict__ dstT) {
    __shared__ float red[4];
    const int v = blockIdx.x;
    float nn = 0.f;
    for (int e = threadIdx.x; e < E; e += 256) { float t = src[(size_t)v * E + e]; nn += t * t; }
    nn = wave_sum(nn);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nn;
    __syncthreads();
    const float nrm = normalize ? sqrtf((red[0] + red[1]) + (red[2] + red[3])) : 1.f;
    for (int e = threadIdx.x; e < E; e += 256) {
        const float t = normalize ? src[(size_t)v * E + e] / nrm : src[(size_t)v * E + e];
        dst[(size_t)v * E + e] = t;
        dstT[(size_t)e * n + v] = t;
    }
}

__global__ void splitk_reduce_kernel(const float* __restrict__ part, int splits, int M, int N, const float* __restrict__ resid,
                                     int ldr, const float* __restrict__ bias, float* __restrict__ out, int ldc,
                                     const int* __restrict__ cmap) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= (size_t)M * N) return;
    int m = (int)(i / N), n = (int)(i - (size_t)m * N);
    const int pm = cmap ? cmap[m] : m;   // physical row of resid / out
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (resid) acc = *(const float4*)(resid + (size_t)pm * ldr + n);
    if (bias) { float4 b = *(const float4*)(bias + n); acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w; }
    // the slices are requested together (splits <= 8) and added in index order; one load per loop iteration made hipcc wait
    // vmcnt(0) eight times in a row
    float4 p[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) p[s] = *(const float4*)(part + (size_t)min(s, splits - 1) * M * N + i);
#pragma unroll
    for (int s = 0; s < 8; ++s)
        if (s < splits) { acc.x += p[s].x; acc.y += p[s].y; acc.z += p[s].z; acc.w += p[s].w; }
    for (int s = 8; s < splits; ++s) {
        float4 q = *(const float4*)(part + (size_t)s * M * N + i);
        acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
    }
    *(float4*)(out + (size_t)pm * ldc + n) = acc;
}

}  // namespace

hipError_t launch_splitk_reduce(const float* part, int splits, int M, int N, const float* resid, int ldr, const float* bias,
                                float* out, int ldc, hipStream_t s, const int* cmap) {
    size_t n4 = (size_t)M * N / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, splits, M, N, resid, ldr,
                       bias, out, ldc, cmap);
    return hipGetLastError();
}

hipError_t launch_cast_f32_op(const float* src, op_t* dst, size_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    size_t blocks = (n / 8 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(cast_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}

hipError_t launch_cast_rows_f32_op(const float* src, int rows, int cols, op_t* dst, int ld, hipStream_t s) {
    size_t n = (size_t)rows * cols;
    hipLaunchKernelGGL(cast_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, rows, cols, dst, ld);
    return hipGetLastError();
}

hipError_t launch_transpose_f32_op(const float* src, int R, int C, op_t* dst, int ld, hipStream_t s) {
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, s, src, R, C, dst, ld);
    return hipGetLastError();
}

hipError_t launch_im2col(const float* x, op_t* patches, int n, int S, int P, int Kp, hipStream_t s) {
    if (S % 8) return hipErrorInvalidValue;
    size_t total = (size_t)n * 3 * S * (S / 8);
    hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, patches, n, S, P, Kp);
    return hipGetLastError();
}

// ---- episode start, image tower: CLS rows + pre-LayerNorm + LayerNorm 1 of layer 0 in ONE pass over the embedded rows (round 6).
// HF modeling_clip.py:187-196 (class embedding + position embedding), :854 (pre_layrnorm) and :359 (layer_norm1 of encoder layer 0) via
// clip/custom_clip.py:62-71 of the reference.  Three launches before (cls_rows_kernel, ln_fwd_kernel with an fp32 output in place,
// ln_fwd_persist_kernel): the rows were written, read, written, read again.  One wave per row; lane l owns float4 chunks l, l + 64, ...
// and every sum is taken in the order the two LayerNorm kernels take it, so h, x1 and the statistics are BIT-identical to the
// three-launch sequence (tests/test_gpu_kernels.py::test_embed_layernorms_equal_the_three_launches).
//   row r of image r / T: x = (r % T == 0) ? cls + pos[0] : h[r] (the patch rows the EPI_PATCH GEMM wrote, position embedding included)
//   y = LN(x; g0, b0) -> h[r] (fp32: the residual stream);   x1[r] = LN(y; g1, b1) in the operand type (+ mean / rstd of that second LayerNorm)
__global__ __launch_bounds__(256) void embed_ln2_kernel(float* __restrict__ h, const float* __restrict__ cls, const float* __restrict__ pos, int T,
                                                        const float* __restrict__ g0, const float* __restrict__ b0,
                                                        const float* __restrict__ g1, const float* __restrict__ b1,
                                                        op_t* __restrict__ y16, int ld16, float* __restrict__ mean, float* __restrict__ rstd,
                                                        int rows, int D, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const int nch = D >> 2;
    const bool is_cls = (row % T) == 0;
    float4 v[LN_MAXC];
#pragma unroll
    for (int i = 0; i < LN_MAXC; ++i) {
        const int c = lane + 64 * i;
        if (c < nch) {
            if (is_cls) {
                const float4 a = *(const float4*)(cls + 4 * c), p = *(const float4*)(pos + 4 * c);
                v[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
            } else v[i] = *(const float4*)(h + (size_t)row * D + 4 * c);
        } else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const float* gamma = pass ? g1 : g0;
        const float* beta = pass ? b1 : b0;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);      // (zero beyond nch)
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            if (lane + 64 * i < nch) {
                const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
        if (pass && lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + 64 * i;
            if (c < nch) {
                const float4 g = *(const float4*)(gamma + 4 * c), b = *(const float4*)(beta + 4 * c);
                const float o0 = (v[i].x - mu) * rs * g.x + b.x, o1 = (v[i].y - mu) * rs * g.y + b.y;
                const float o2 = (v[i].z - mu) * rs * g.z + b.z, o3 = (v[i].w - mu) * rs * g.w + b.w;
                if (pass) st_op4(y16 + (size_t)row * ld16 + 4 * c, o0, o1, o2, o3);
                else { *(float4*)(h + (size_t)row * D + 4 * c) = make_float4(o0, o1, o2, o3); v[i] = make_float4(o0, o1, o2, o3); }
            }
        }
    }
}

hipError_t launch_embed_layernorms(float* h, const float* cls, const float* pos, int T, const float* g0, const float* b0, const float* g1,
                                   const float* b1, op_t* y16, int ld16, float* mean, float* rstd, int rows, int D, float eps, hipStream_t s) {
    if (D % 4 || D > 256 * LN_MAXC || T < 1 || rows < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(embed_ln2_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, h, cls, pos, T, g0, b0, g1, b1, y16, ld16, mean, rstd, rows, D, eps);
    return hipGetLastError();
}

hipError_t launch_cls_rows(float* h, const float* cls, const float* pos, int n, int T, int D, hipStream_t s) {
    hipLaunchKernelGGL(cls_rows_kernel, dim3((n * D + 255) / 256), dim3(256), 0, s, h, cls, pos, n, T, D);
    return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, long long row_stride, const float* gamma, const float* beta, float* y_f32,
                            op_t* y_bf16, int ld_bf16, float* mean, float* rstd, int rows, int D, float eps,
                            hipStream_t s, const int* rowmap) {
    if (D % 4 || D > 256 * LN_MAXC) return hipErrorInvalidValue;
#ifdef TTL_DIAG_SKIP       // timing-only ablation of the episode (tools/class_cost_ab.sh): bit 0 = no big LayerNorm forward launches
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 1) && rows >= 4096 && diag_skip_now(cnt, 1320)) return hipSuccess; }
#endif
#if TTL_LN_VARIANT == 2
    if (rows >= 4096 && !y_f32 && y_bf16 && !rowmap && row_stride == D) {
        const int cus = device_cu_count();
        if (!cus) return hipErrorInvalidDevice;
        static const int pblk = [] { const int b = TTL_EXPERIMENT("TTL_LN_PBLK", 8); return b < 1 ? 1 : b; }();
        if (D <= 768) hipLaunchKernelGGL(ln_fwd_persist_kernel<3>, dim3(cus * pblk), dim3(256), 0, s, x, gamma, beta, y_bf16, ld_bf16, mean, rstd, rows, D, eps);
        else hipLaunchKernelGGL(ln_fwd_persist_kernel<4>, dim3(cus * pblk), dim3(256), 0, s, x, gamma, beta, y_bf16, ld_bf16, mean, rstd, rows, D, eps);
        return hipGetLastError();
    }
#endif
    // several rows per wave only where there are rows to spare (big launches); small ones stay one row per wave
    if (TTL_LN_RPW > 1 && rows >= 4096)
        hipLaunchKernelGGL((ln_fwd_kernel<TTL_LN_RPW>), dim3((rows + 4 * TTL_LN_RPW - 1) / (4 * TTL_LN_RPW)), dim3(256), 0, s, x, row_stride,
                           gamma, beta, y_f32, y_bf16, ld_bf16, mean, rstd, rows, D, eps, rowmap);
    else
        hipLaunchKernelGGL((ln_fwd_kernel<1>), dim3((rows + 3) / 4), dim3(256), 0, s, x, row_stride, gamma, beta, y_f32, y_bf16,
                           ld_bf16, mean, rstd, rows, D, eps, rowmap);
    return hipGetLastError();
}

hipError_t launch_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                const float* gamma, const float* dres, float* out_f32, op_t* out_bf16, int rows,
                                int D, hipStream_t s, long long x_stride, long long o_stride, int stat_stride,
                                int dres_T, const int* pool, long long ld_bf16) {
#ifdef TTL_DIAG_SKIP       // bit 1 = no big LayerNorm backward launches (after the warm-up)
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 2) && rows >= 4096 && diag_skip_now(cnt, 240)) return hipSuccess; }
#endif
    if (D % 4 || D > 256 * LN_MAXC) return hipErrorInvalidValue;
    auto kern = D <= 768 ? ln_bwd_kernel<3> : ln_bwd_kernel<4>;
    hipLaunchKernelGGL(kern, dim3((rows + 3) / 4), dim3(256), 0, s, dy, x, mean, rstd, gamma, dres, out_f32,
                       out_bf16, rows, D, x_stride ? x_stride : (long long)D, o_stride ? o_stride : (long long)D,
                       stat_stride, dres_T, pool, ld_bf16 ? ld_bf16 : (o_stride ? o_stride : (long long)D));
    return hipGetLastError();
}

hipError_t launch_text_embed(const int* ids, const float* tok, const float* pos, float* h, int rows, int T, int D, hipStream_t s) {
    if (D % 4) return hipErrorInvalidValue;
    size_t n4 = (size_t)rows * (D / 4);
    hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, ids, tok, pos, h, rows, T, D);
    return hipGetLastError();
}

hipError_t launch_gather_rows_f32(const float* src, long long ld, const int* pool, int T, float* dst, int n, int cols, hipStream_t s) {
    size_t tot = (size_t)n * cols;
    hipLaunchKernelGGL((gather_rows_kernel<float>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, ld, pool, T, dst, n, cols);
    return hipGetLastError();
}

hipError_t launch_gather_rows_op(const op_t* src, long long ld, const int* pool, int T, op_t* dst, int n, int cols, hipStream_t s) {
    size_t tot = (size_t)n * cols;
    hipLaunchKernelGGL((gather_rows_kernel<op_t>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, ld, pool, T, dst, n, cols);
    return hipGetLastError();
}

hipError_t launch_gather_view_blocks(const GatherTable& t, const long long* idx, int n_sel, hipStream_t s) {
    if (t.n < 1 || t.n > GATHER_MAX || n_sel < 1) return hipErrorInvalidValue;
    unsigned long long big = 0;
    for (int i = 0; i < t.n; ++i) {
        if ((t.e[i].block_bytes | t.e[i].stride_bytes) & 3) return hipErrorInvalidValue;
        big = t.e[i].block_bytes > big ? t.e[i].block_bytes : big;
    }
    unsigned gx = (unsigned)((big / 16 + 1023) / 1024);      // ~4 16-byte pieces per thread of the largest block
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    hipLaunchKernelGGL(gather_view_blocks_kernel, dim3(gx, n_sel, t.n), dim3(256), 0, s, t, idx);
    return hipGetLastError();
}

hipError_t launch_transpose_f32(const float* src, int R, int C, float* dst, hipStream_t s) {
    size_t tot = (size_t)R * C;
    hipLaunchKernelGGL(transpose_f32_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, src, R, C, dst);
    return hipGetLastError();
}

hipError_t launch_unit_rows(const float* src, int n, int E, int normalize, float* dst, float* dstT, hipStream_t s) {
    hipLaunchKernelGGL(unit_rows_kernel, dim3(n), dim3(256), 0, s, src, n, E, normalize, dst, dstT);
    return hipGetLastError();
}

hipError_t launch_fill_zero(void* p, size_t bytes, hipStream_t s) { return hipMemsetAsync(p, 0, bytes, s); }
