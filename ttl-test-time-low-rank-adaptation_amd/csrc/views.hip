// GPU-side view generator: one decoded uint8 image -> the N x 3 x S x S fp32 batch
// the patch-embed im2col consumes, normalised with CLIP mean/std.
// Replaces the host pipeline of data/datautils.py:98-157 (AugMixAugmenter with an empty aug_list, Q13):
// base view = Resize(S, bicubic) + CenterCrop(S); other views = RandomResizedCrop(S) (bilinear) +
// RandomHorizontalFlip; then ToTensor + Normalize (ttl.py:225-241).
//
// Byte work, so the bar is bit-exactness with what the reference's host path produces.  torchvision's
// PIL backend ends in Pillow's ImagingResample: two separable 8-bit passes (horizontal, then vertical)
// with 22-bit fixed-point taps computed in double precision, each pass rounded and clipped to uint8.
// Two kernels restate that:
//   coef_kernel   per (view, axis, output index): tap window + integer taps, in fp64 with contraction off
//                 so every intermediate rounds like the host's C code;
//   views_kernel  one thread per output pixel: for each vertical tap row it forms the horizontal-pass
//                 byte (clip8 of the integer sum) and accumulates the vertical pass — the uint8
//                 intermediate image never exists in memory, at the cost of recomputing each
//                 intermediate byte for the ~2-4 output rows that use it (the source crop stays in L2).
// Output = ((u8 / 255) - mean) / std with IEEE fp32 divisions, like ToTensor + Normalize.
#include "kernels.hpp"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

#pragma clang fp contract(off)
__device__ double filt64(double x, int bicubic) {
    if (x < 0.0) x = -x;
    if (!bicubic) return x < 1.0 ? 1.0 - x : 0.0;
    const double a = -0.5;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// geometry of one (view, axis): input extent, output extent of the resize, first output index used
struct AxisGeom { int in_size, out_size, first, origin; };

__device__ AxisGeom axis_geom(const int* box, int axis, int H, int W, int S) {
    const int base = (box[4] >> 1) & 1;
    AxisGeom g;
    if (base) {   // transforms.Resize(S): shorter side -> S, longer int(S*long/short); CenterCrop offset round((n-S)/2)
        const int nh = (W <= H) ? (int)((double)S * H / W) : S, nw = (W <= H) ? S : (int)((double)S * W / H);
        g.in_size = axis ? W : H;
        g.out_size = axis ? nw : nh;
        g.first = (int)rint((g.out_size - S) / 2.0);
        g.origin = 0;
    } else {
        g.in_size = axis ? box[3] : box[2];
        g.out_size = S;
        g.first = 0;
        g.origin = axis ? box[1] : box[0];
    }
    return g;
}

// table layout per (view, axis, o): [0] = first source index (absolute), [1] = tap count, [2..] = taps
#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void coef_kernel(const int* __restrict__ boxes, int H, int W, int S, int kstride,
                                                   int* __restrict__ table) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    const int axis = blockIdx.y, v = blockIdx.z;
    if (o >= S) return;
    const int* box = boxes + 5 * v;
    const int bicubic = (box[4] >> 1) & 1;
    const AxisGeom g = axis_geom(box, axis, H, W, S);
    const double scale = (double)g.in_size / g.out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = (bicubic ? 2.0 : 1.0) * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (g.first + o + 0.5) * scale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > g.in_size) xmax = g.in_size;
    xmax -= xmin;
    if (xmax > kstride - 2) xmax = kstride - 2;   // cannot happen when kstride comes from ttl_make_views_workspace_bytes
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += filt64((x + xmin - center + 0.5) * ss, bicubic);
    int* t = table + ((size_t)(v * 2 + axis) * S + o) * kstride;
    t[0] = g.origin + xmin;
    t[1] = xmax;
    for (int x = 0; x < xmax; ++x) {
        double k = filt64((x + xmin - center + 0.5) * ss, bicubic);
        if (ww != 0.0) k /= ww;
        t[2 + x] = k < 0 ? (int)(-0.5 + k * (1 << PRECISION_BITS)) : (int)(0.5 + k * (1 << PRECISION_BITS));
    }
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

__global__ __launch_bounds__(256) void views_kernel(const unsigned char* __restrict__ img, int W, const int* __restrict__ boxes,
                                                    const int* __restrict__ table, int kstride, int S, float3 mean, float3 stdv,
                                                    float* __restrict__ out) {
    // one thread per output pixel of the view, 256 consecutive pixels of the row-major [S][S] plane per block: no idle lanes
    // whatever S is (one block per output row left 32 of 256 lanes idle at S = 224)
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    const int v = blockIdx.z;
    if (pix >= S * S) return;
    const int oy = pix / S, ox = pix - oy * S;
    const int flip = boxes[5 * v + 4] & 1;
    const int oxs = flip ? (S - 1 - ox) : ox;   // hflip of the resized crop == mirrored output column
    const int* ty = table + ((size_t)(v * 2 + 0) * S + oy) * kstride;
    const int* tx = table + ((size_t)(v * 2 + 1) * S + oxs) * kstride;
    const int y0 = ty[0], ny = ty[1], x0 = tx[0], nx = tx[1];
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int y = 0; y < ny; ++y) {
        const unsigned char* row = img + ((size_t)(y0 + y) * W + x0) * 3;
        int h0 = 1 << (PRECISION_BITS - 1), h1 = h0, h2 = h0;
        for (int x = 0; x < nx; ++x) {
            const int k = tx[2 + x];
            h0 += (int)row[3 * x] * k;
            h1 += (int)row[3 * x + 1] * k;
            h2 += (int)row[3 * x + 2] * k;
        }
        const int ky = ty[2 + y];
        a0 += clip8(h0) * ky;
        a1 += clip8(h1) * ky;
        a2 += clip8(h2) * ky;
    }
    const size_t plane = (size_t)S * S, o = (size_t)v * 3 * plane + (size_t)oy * S + ox;
    out[o] = ((float)clip8(a0) / 255.0f - mean.x) / stdv.x;
    out[o + plane] = ((float)clip8(a1) / 255.0f - mean.y) / stdv.y;
    out[o + 2 * plane] = ((float)clip8(a2) / 255.0f - mean.z) / stdv.z;
}

}  // namespace

int views_kstride(int H, int W, int S) {
    // widest tap window: bicubic (radius 2) at the largest down-scale either axis can see, + [first, count]
    const int m = H > W ? H : W;
    double scale = (double)m / S;
    if (scale < 1.0) scale = 1.0;
    return (int)ceil(2.0 * scale) * 2 + 1 + 2 + 1;
}

hipError_t launch_make_views(const unsigned char* img, int H, int W, const int* boxes, int n, int S, const float* mean,
                             const float* stdv, float* out, int* table, int kstride, hipStream_t s) {
    hipLaunchKernelGGL(coef_kernel, dim3((S + 255) / 256, 2, n), dim3(256), 0, s, boxes, H, W, S, kstride, table);
    hipLaunchKernelGGL(views_kernel, dim3((S * S + 255) / 256, 1, n), dim3(256), 0, s, img, W, boxes, table, kstride, S,
                       make_float3(mean[0], mean[1], mean[2]), make_float3(stdv[0], stdv[1], stdv[2]), out);
    return hipGetLastError();
}
