// PLPD filter of DeYO on the device (deyo.py:115-151): the view-destroying transforms and the
// "pseudo-label probability difference" mask, so that --filter_plpd 1 stays inside the fused, graph-capturable episode.
//
//   x' = destroy(x[filter_ids_1])        'occ'   : a window filled with the view's per-channel mean        (deyo.py:118-122)
//                                        'patch' : resize to a multiple of patch_len (antialiased bilinear), permute the
//                                                  patch_len^2 patches of every view, resize back              (deyo.py:123-130)
//                                        'pixel' : ONE pixel permutation shared by all views and channels      (deyo.py:131-134)
//   plpd_b = softmax(z[ids1[b]])[c] - softmax(z'[b])[c],  c = argmax z[ids1[b]];   keep[ids1[b]] = plpd_b > threshold   (:137-146)
//
// The permutations come from the HOST generator exactly as the reference draws them (torch.argsort(torch.rand(B, P)) /
// torch.randperm(S*S) on the CPU generator) and arrive as int32 device arrays; the kernels are pure byte movement except for the
// resize, which restates ATen's separable antialiased bilinear (aten/src/ATen/native/cpu/UpSampleKernel.cpp,
// _compute_indices_min_size_weights_aa + basic_loop_aa_*: horizontal pass, then vertical, fp32 weights normalised per output
// index).  With image_size % patch_len == 0 (e.g. 224 / 4; the reference's default --patch_len 6 is NOT such a case) both resizes are the identity and 'patch' is a
// pure gather: bit-identical to the torch chain; otherwise within 1e-6 (summation / contraction order), tests/test_gpu_plpd.py.
// Every kernel is guarded by the DEVICE-side selection count *n_sel (the host sizes the launch for the count it expects).
#include "kernels.hpp"

namespace {

// mean over the S*S pixels of (view idx[b], channel c) -> mean[b*3 + c]; one block per (b, c)
__global__ __launch_bounds__(256) void plpd_mean_kernel(const float* __restrict__ x, const long long* __restrict__ idx, const int* __restrict__ n_sel,
                                                        int S, float* __restrict__ mean) {
    __shared__ float red[4];
    const int b = blockIdx.x / 3, c = blockIdx.x - 3 * b;
    if (b >= *n_sel) return;
    const float* src = x + ((size_t)idx[b] * 3 + c) * S * S;
    float s = 0.f;
    for (int i = threadIdx.x; i < S * S / 4; i += 256) { const float4 v = *(const float4*)(src + 4 * i); s += (v.x + v.y) + (v.z + v.w); }
    for (int i = (S * S / 4) * 4 + threadIdx.x; i < S * S; i += 256) s += src[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) mean[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)(S * S);
}

// out[b][c][y][x] = inside the window ? mean[b][c] : x[idx[b]][c][y][x]
__global__ __launch_bounds__(256) void plpd_occ_kernel(const float* __restrict__ x, const long long* __restrict__ idx, const int* __restrict__ n_sel,
                                                       int S, int r0, int c0, int sz, const float* __restrict__ mean, float* __restrict__ out,
                                                       size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int px = (int)(i % S), py = (int)((i / S) % S);
    const size_t bc = i / ((size_t)S * S);
    const int b = (int)(bc / 3), c = (int)(bc - 3 * (size_t)b);
    if (b >= *n_sel) return;
    const bool in = py >= r0 && py < r0 + sz && px >= c0 && px < c0 + sz;
    out[i] = in ? mean[bc] : x[(((size_t)idx[b] * 3 + c) * S + py) * S + px];
}

// out[b][c][p] = x[idx[b]][c][perm[p]]      (x_prime[:, :, torch.randperm(S*S)])
__global__ __launch_bounds__(256) void plpd_pixel_kernel(const float* __restrict__ x, const long long* __restrict__ idx, const int* __restrict__ n_sel,
                                                         int S, const int* __restrict__ perm, float* __restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int p = (int)(i % ((size_t)S * S));
    const size_t bc = i / ((size_t)S * S);
    const int b = (int)(bc / 3), c = (int)(bc - 3 * (size_t)b);
    if (b >= *n_sel) return;
    out[i] = x[((size_t)idx[b] * 3 + c) * S * S + perm[p]];
}

// patch shuffle on a [B][3][St][St] image (St = pl*h): output patch (p1, p2) of view b is source patch perm[b][p1*pl + p2];
// src_idx != null: the source is x[src_idx[b]] (the gather by filter_ids_1 folded in, St == S)
__global__ __launch_bounds__(256) void plpd_patch_kernel(const float* __restrict__ src, const long long* __restrict__ src_idx,
                                                         const int* __restrict__ n_sel, int St, int pl, const int* __restrict__ perm,
                                                         float* __restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int px = (int)(i % St), py = (int)((i / St) % St);
    const size_t bc = i / ((size_t)St * St);
    const int b = (int)(bc / 3), c = (int)(bc - 3 * (size_t)b);
    if (b >= *n_sel) return;
    const int h = St / pl;
    const int p1 = py / h, y = py - p1 * h, p2 = px / h, xx = px - p2 * h;
    const int q = perm[b * pl * pl + p1 * pl + p2], q1 = q / pl, q2 = q - q1 * pl;
    const size_t sb = src_idx ? (size_t)src_idx[b] : (size_t)b;
    out[i] = src[((sb * 3 + c) * St + (q1 * h + y)) * St + (q2 * h + xx)];
}

// ATen's antialiased bilinear along ONE axis: output index o of `nout` from `nin` inputs.
struct AaTaps { int xmin, xsize; float scale, center, invscale; };
__device__ __forceinline__ AaTaps aa_taps(int o, int nin, int nout) {
    AaTaps t;
    t.scale = (float)nin / (float)nout;                               // area_pixel_compute_scale, align_corners = false
    const float support = t.scale >= 1.f ? t.scale : 1.f;             // (interp_size / 2) * scale, interp_size = 2
    t.invscale = t.scale >= 1.f ? 1.f / t.scale : 1.f;
    t.center = t.scale * ((float)o + 0.5f);
    t.xmin = max((int)(t.center - support + 0.5f), 0);
    t.xsize = min((int)(t.center + support + 0.5f), nin) - t.xmin;
    return t;
}
__device__ __forceinline__ float aa_w(const AaTaps& t, int j) {
    const float a = fabsf(((float)(j + t.xmin) - t.center + 0.5f) * t.invscale);
    return a < 1.f ? 1.f - a : 0.f;
}
// one pass: axis == 0 resizes the width (rows of `win` -> rows of `wout`), axis == 1 the height.  src_idx as above (first pass only).
__global__ __launch_bounds__(256) void plpd_resize_kernel(const float* __restrict__ src, const long long* __restrict__ src_idx,
                                                          const int* __restrict__ n_sel, int hin, int win, int hout, int wout, int axis,
                                                          float* __restrict__ out, size_t total) {
#pragma clang fp contract(off)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int ox = (int)(i % wout), oy = (int)((i / wout) % hout);
    const size_t bc = i / ((size_t)wout * hout);
    const int b = (int)(bc / 3), c = (int)(bc - 3 * (size_t)b);
    if (b >= *n_sel) return;
    const size_t sb = src_idx ? (size_t)src_idx[b] : (size_t)b;
    const float* plane = src + (sb * 3 + c) * (size_t)hin * win;
    const AaTaps t = axis == 0 ? aa_taps(ox, win, wout) : aa_taps(oy, hin, hout);
    float tot = 0.f;
    for (int j = 0; j < t.xsize; ++j) tot += aa_w(t, j);
    float acc = 0.f;
    for (int j = 0; j < t.xsize; ++j) {
        const float w = aa_w(t, j) / tot;
        const float v = axis == 0 ? plane[(size_t)oy * win + t.xmin + j] : plane[(size_t)(t.xmin + j) * win + ox];
        acc = j == 0 ? v * w : acc + v * w;
    }
    out[i] = acc;
}

// torch.argmax's order (deyo.py:138): NaN counts as the maximum, the first index wins among equals.  A candidate row that is all
// -inf or holds a NaN does reach this kernel (the rank-based top-k of head_loss.hip gives a NaN entropy rank 0): the arg-max must
// still be an index < K.  INT_MAX marks "no element seen" (lanes beyond K) and never wins.
__device__ __forceinline__ bool argmax_better(float v, int a, float mx, int am) {
    if (a == 0x7fffffff) return false;
    if (am == 0x7fffffff) return true;
    const bool vn = v != v, mn = mx != mx;
    if (vn || mn) return vn && (!mn || a < am);
    return v > mx || (v == mx && a < am);
}

// keep[ids1[b]] = softmax(z[ids1[b]])[c] - softmax(z'[b])[c] > thr,  c = argmax z[ids1[b]] (first maximum, like torch.argmax);
// one block per candidate view; `keep` [N] was zeroed by the launcher
__global__ __launch_bounds__(256) void plpd_keep_kernel(const float* __restrict__ z, const float* __restrict__ zp, const long long* __restrict__ idx,
                                                        const int* __restrict__ n_sel, int K, float thr, unsigned char* __restrict__ keep,
                                                        float* __restrict__ plpd_out) {
    __shared__ float red[4];
    __shared__ int redi[4];
    const int b = blockIdx.x;
    if (b >= *n_sel) return;
    const float* row = z + (size_t)idx[b] * K;
    const float* rowp = zp + (size_t)b * K;
    float mx = -INFINITY, mxp = -INFINITY;
    int am = 0x7fffffff;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float v = row[k];
        if (argmax_better(v, k, mx, am)) { mx = v; am = k; }
        mxp = fmaxf(mxp, rowp[k]);
    }
    // block arg-max with the lowest index among equals
    for (int o = 32; o > 0; o >>= 1) {
        const float v = __shfl_xor(mx, o, 64); const int a = __shfl_xor(am, o, 64);
        if (argmax_better(v, a, mx, am)) { mx = v; am = a; }
        mxp = fmaxf(mxp, __shfl_xor(mxp, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = mx; redi[threadIdx.x >> 6] = am; }
    __syncthreads();
    float bm = red[0]; int ba = redi[0];
    for (int w = 1; w < 4; ++w) if (argmax_better(red[w], redi[w], bm, ba)) { bm = red[w]; ba = redi[w]; }
    if ((unsigned)ba >= (unsigned)K) ba = 0;      // (K >= 1: unreachable; the read of rowp[ba] below must stay inside the row whatever the data)
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mxp;
    __syncthreads();
    const float bmp = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.f, sp = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) { s += expf(row[k] - bm); sp += expf(rowp[k] - bmp); }
    s = wave_sum(s); sp = wave_sum(sp);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float den = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sp;
    __syncthreads();
    const float denp = (red[0] + red[1]) + (red[2] + red[3]);
    if (threadIdx.x == 0) {
        const float d = 1.0f / den - expf(rowp[ba] - bmp) / denp;      // softmax(z)[c] = exp(0) / den at the arg-max
        keep[idx[b]] = d > thr ? 1 : 0;
        if (plpd_out) plpd_out[b] = d;
    }
}

}  // namespace

size_t plpd_views_workspace_floats(int n_max, int S, int aug, int patch_len) {
    if (aug == PLPD_OCC) return (size_t)n_max * 3;
    if (aug == PLPD_PATCH && patch_len > 0 && S % patch_len) return 2 * (size_t)n_max * 3 * S * S;
    return 0;
}

hipError_t launch_plpd_views(const float* x, const long long* idx, const int* n_sel, int n_max, int S, const PlpdArgs& p, float* out,
                             float* ws, hipStream_t s) {
    if (n_max < 1 || S < 1) return hipErrorInvalidValue;
    const size_t total = (size_t)n_max * 3 * S * S;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (p.aug == PLPD_OCC) {
        if (p.occ_size < 1 || p.row_start < 0 || p.col_start < 0 || p.row_start + p.occ_size > S || p.col_start + p.occ_size > S) return hipErrorInvalidValue;
        hipLaunchKernelGGL(plpd_mean_kernel, dim3(n_max * 3), dim3(256), 0, s, x, idx, n_sel, S, ws);
        hipLaunchKernelGGL(plpd_occ_kernel, dim3(grid), dim3(256), 0, s, x, idx, n_sel, S, p.row_start, p.col_start, p.occ_size, ws, out, total);
    } else if (p.aug == PLPD_PIXEL) {
        if (!p.perm) return hipErrorInvalidValue;
        hipLaunchKernelGGL(plpd_pixel_kernel, dim3(grid), dim3(256), 0, s, x, idx, n_sel, S, p.perm, out, total);
    } else if (p.aug == PLPD_PATCH) {
        if (!p.perm || p.patch_len < 1 || p.patch_len > S) return hipErrorInvalidValue;
        const int St = (S / p.patch_len) * p.patch_len;
        if (St == S) {       // both resizes are the identity: one gather
            hipLaunchKernelGGL(plpd_patch_kernel, dim3(grid), dim3(256), 0, s, x, idx, n_sel, S, p.patch_len, p.perm, out, total);
        } else {
            float *a = ws, *b = ws + total;      // (St < S: every intermediate fits n_max*3*S*S)
            const size_t t1 = (size_t)n_max * 3 * S * St, t2 = (size_t)n_max * 3 * St * St;
            auto g = [](size_t t) { return dim3((unsigned)((t + 255) / 256)); };
            hipLaunchKernelGGL(plpd_resize_kernel, g(t1), dim3(256), 0, s, x, idx, n_sel, S, S, S, St, 0, a, t1);            // width  S -> St
            hipLaunchKernelGGL(plpd_resize_kernel, g(t2), dim3(256), 0, s, a, nullptr, n_sel, S, St, St, St, 1, b, t2);      // height S -> St
            hipLaunchKernelGGL(plpd_patch_kernel, g(t2), dim3(256), 0, s, b, nullptr, n_sel, St, p.patch_len, p.perm, a, t2);
            const size_t t3 = (size_t)n_max * 3 * St * S;
            hipLaunchKernelGGL(plpd_resize_kernel, g(t3), dim3(256), 0, s, a, nullptr, n_sel, St, St, St, S, 0, b, t3);      // width  St -> S
            hipLaunchKernelGGL(plpd_resize_kernel, grid, dim3(256), 0, s, b, nullptr, n_sel, St, S, S, S, 1, out, total);    // height St -> S
        }
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_plpd_keep(const float* logits, const float* logits_prime, const long long* idx, const int* n_sel, int n_max, int N, int K,
                            float threshold, unsigned char* keep, float* plpd_out, hipStream_t s) {
    hipError_t e = hipMemsetAsync(keep, 0, (size_t)N, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(plpd_keep_kernel, dim3(n_max), dim3(256), 0, s, logits, logits_prime, idx, n_sel, K, threshold, keep, plpd_out);
    return hipGetLastError();
}
