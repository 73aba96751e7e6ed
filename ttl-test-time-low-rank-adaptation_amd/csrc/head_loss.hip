// Logit head, entropy / confidence-selection loss (forward + analytic backward), AdamW, reset.
// All fp32 — these decide the selection mask, which must be bit-exact w.r.t. the reference.
//
//   head  : CLS -> post_layernorm -> visual_projection -> f/||f|| -> exp(logit_scale) f̂ t̂^T
//           (HF modeling_clip.py:646-651,751; clip/custom_clip.py:680-687) and its backward
//   loss  : deyo.py:85-90,102-113,159-181 (weighted entropy) / ttl.py:50-61 (TPT avg-entropy)
//   adamw : torch.optim.AdamW over the flat LoRA buffer (ttl.py:218, deyo.py:187)
//   reset : LoRA_AB.reset + optimizer.load_state_dict (clip/custom_clip.py:202-215, ttl.py:344)
#include "kernels.hpp"

namespace {

// Slices of a mat-vec's reduction per block (HB = 64 * slices threads).  Every head kernel is a serial walk of I / slices
// terms per thread behind L2 round trips (25 us for a 64 x 768 x 512 product with 4 slices, the SAME 24 us for one view).
// Measured (round 3, tools/hip_variant.sh head_loss TTL_HEAD_SLICES=..., three leases of bench.py each): 16 slices cut the
// class from 0.158 to 0.098 ms per episode one at a time, 8 slices to 0.114 — and move the rate with three episodes in flight
// by +0.3 % (289.4 vs 288.6 images/s, inside the spread): the class is only 45 % exposed.  The fp32 summation order of the
// logits changes with the slice count, so the default stays at round 2's 4 (the fixtures' adapted logits sit on sign-like
// AdamW steps and move at the 1e-3 level with any perturbation of the last bit).
#ifndef TTL_HEAD_SLICES
#define TTL_HEAD_SLICES 4
#endif
constexpr int NS = TTL_HEAD_SLICES;
static_assert(NS >= 4 && NS % 4 == 0 && NS <= 16, "TTL_HEAD_SLICES must be 4, 8, 12 or 16 (matvec64 reduces the slices four at a time)");
constexpr int HB = 64 * NS;  // threads per head block

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t = fmaxf(t, red[i]);
    return t;
}

// ---- head forward / backward mat-vecs.  All four are  out[v][o] = post( sum_i in[v][i] * W[i][o] )  with W row-major
// [I][O] (coalesced over o).  One block = 64 outputs x 4 slices of the reduction (256 threads): each thread walks I/4
// terms with four independent chains, the slices meet in LDS.  (One thread per output walking all I terms left
// 128 blocks of a 64-view call latency-bound at 23-33 us per launch.)
constexpr int HO = 64;   // outputs per block

template <typename F>
__device__ __forceinline__ float matvec64(const float* __restrict__ in_lds, const float* __restrict__ W, int I, int O, int o, int slice,
                                          float (*part)[HO], F&& post_unused) {
    (void)post_unused;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (o < O) {
        const float* w = W + o;
        int i = slice;
        // (8-fold unrolling for 32 loads in flight per thread was tried: the class got 2x SLOWER in situ)
#pragma unroll 2
        for (; i + 3 * NS < I; i += 4 * NS) {
            a0 = fmaf(in_lds[i], w[(size_t)i * O], a0);
            a1 = fmaf(in_lds[i + NS], w[(size_t)(i + NS) * O], a1);
            a2 = fmaf(in_lds[i + 2 * NS], w[(size_t)(i + 2 * NS) * O], a2);
            a3 = fmaf(in_lds[i + 3 * NS], w[(size_t)(i + 3 * NS) * O], a3);
        }
        for (; i < I; i += NS) a0 = fmaf(in_lds[i], w[(size_t)i * O], a0);
    }
    part[slice][threadIdx.x & (HO - 1)] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < NS; k += 4)      // fixed order: deterministic
        t += (part[k][threadIdx.x & (HO - 1)] + part[k + 1][threadIdx.x & (HO - 1)]) +
             (part[k + 2][threadIdx.x & (HO - 1)] + part[k + 3][threadIdx.x & (HO - 1)]);
    return t;
}

// grid (ceil(E/64), n): f[v][e] = sum_d y[v][d] * WpT[d][e]
__global__ __launch_bounds__(HB) void head_proj_kernel(HeadArgs a) {
    extern __shared__ float sm[];
    __shared__ float part[NS][HO];
    const int v = blockIdx.y, e = blockIdx.x * HO + (threadIdx.x & (HO - 1)), slice = threadIdx.x >> 6;
    for (int d = threadIdx.x; d < a.D; d += HB) sm[d] = a.y[(size_t)v * a.D + d];
    __syncthreads();
    const float acc = matvec64(sm, a.WpT, a.D, a.E, e, slice, part, 0);
    if (slice == 0 && e < a.E) {
        a.f[(size_t)v * a.E + e] = acc;
        if (a.feats_out) a.feats_out[(size_t)v * a.E + e] = acc;
    }
}
// grid (ceil(K/64), n): z[v][k] = scale * <f/||f||, t_k>
__global__ __launch_bounds__(HB) void head_logits_kernel(HeadArgs a) {
    extern __shared__ float sm[];
    __shared__ float red[HB / 64];
    __shared__ float part[NS][HO];
    const int v = blockIdx.y, k = blockIdx.x * HO + (threadIdx.x & (HO - 1)), slice = threadIdx.x >> 6;
    float nn = 0.f;
    for (int e = threadIdx.x; e < a.E; e += HB) { float t = a.f[(size_t)v * a.E + e]; sm[e] = t; nn += t * t; }
    const float inv = a.scale / sqrtf(block_sum(nn, red));
    const float acc = matvec64(sm, a.tfeatT, a.E, a.K, k, slice, part, 0);
    if (slice == 0 && k < a.K) a.logits[(size_t)v * a.K + k] = acc * inv;
}
// ---- head backward
// grid (ceil(E/64), n): dfh[v][e] = scale * sum_k dz[v][k] t[k][e]
__global__ __launch_bounds__(HB) void head_dfh_kernel(HeadArgs a, const float* __restrict__ dz) {
    extern __shared__ float sm[];
    __shared__ float part[NS][HO];
    const int v = blockIdx.y, e = blockIdx.x * HO + (threadIdx.x & (HO - 1)), slice = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < a.K; k += HB) sm[k] = dz[(size_t)v * a.K + k];
    __syncthreads();
    const float acc = matvec64(sm, a.tfeat, a.K, a.E, e, slice, part, 0);
    // scaler.scale(loss): the loss scale of the context's GradScaler state (1 in the bf16 build), removed again in
    // wgrad_reduce_kernel (scaler.unscale_)
    if (slice == 0 && e < a.E) a.tmp_e[(size_t)v * a.E + e] = acc * a.scale * (a.gscale ? a.gscale[0] : 1.0f);
}
// grid (ceil(D/64), n): df = (dfh - fh <fh,dfh>)/||f|| ; dy[v][d] = sum_e df[e] Wp[e][d]
__global__ __launch_bounds__(HB) void head_dy_kernel(HeadArgs a) {
    extern __shared__ float sm[];
    __shared__ float red[HB / 64];
    __shared__ float part[NS][HO];
    const int v = blockIdx.y, d = blockIdx.x * HO + (threadIdx.x & (HO - 1)), slice = threadIdx.x >> 6;
    const float* f = a.f + (size_t)v * a.E;
    const float* dfh = a.tmp_e + (size_t)v * a.E;
    float nn = 0.f;
    for (int e = threadIdx.x; e < a.E; e += HB) nn += f[e] * f[e];
    const float nrm = sqrtf(block_sum(nn, red));
    float dot = 0.f;
    for (int e = threadIdx.x; e < a.E; e += HB) dot += dfh[e] * (f[e] / nrm);
    dot = block_sum(dot, red);
    for (int e = threadIdx.x; e < a.E; e += HB) sm[e] = (dfh[e] - (f[e] / nrm) * dot) / nrm;
    __syncthreads();
    const float acc = matvec64(sm, a.Wp, a.E, a.D, d, slice, part, 0);
    if (slice == 0 && d < a.D) a.tmp_d[(size_t)v * a.D + d] = acc;
}

// ---- loss, pass 1: one block per view: row softmax statistics -> H_i, lse_i
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ z, int K, float* __restrict__ Hout,
                                                        float* __restrict__ lse) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float* zr = z + (size_t)i * K;
    float mx = -INFINITY;
    for (int k = tid; k < K; k += 256) mx = fmaxf(mx, zr[k]);
    mx = block_max(mx, red);
    float se = 0.f;
    for (int k = tid; k < K; k += 256) se += expf(zr[k] - mx);
    se = block_sum(se, red);
    const float l = mx + logf(se);
    float h = 0.f;
    for (int k = tid; k < K; k += 256) { float lp = zr[k] - l; h -= expf(lp) * lp; }
    h = block_sum(h, red);
    if (tid == 0) { Hout[i] = h; lse[i] = l; }
}

// ---- loss, pass 2 (single block): selection (bit-exact integer result from fp32 compares),
// coefficients, loss.  rank_i = #{j : H_j < H_i or (H_j == H_i and j < i)} == position in a
// stable ascending argsort.
__global__ __launch_bounds__(256) void select_kernel(const float* __restrict__ H, int N, int objective, int mode,
                                                     int ktop, float thresh, float margin, float reweight,
                                                     int reuse_idx, long long* __restrict__ idx, int* __restrict__ n_io,
                                                     float* __restrict__ loss, float* __restrict__ coef /*[N]*/,
                                                     int* __restrict__ nsel_dev, const unsigned char* __restrict__ keep) {
    __shared__ float red[4];
    __shared__ int sn;
    __shared__ int wcnt[4];
    const int tid = threadIdx.x;
    if (tid == 0) sn = 0;
    __syncthreads();
    const bool topk = (objective == 1) || (mode == 1);
    int n;
    if (objective == 1 && reuse_idx) {
        n = *n_io;
        for (int i = tid; i < N; i += 256) coef[i] = 0.f;
        __syncthreads();
        for (int j = tid; j < n; j += 256) coef[idx[j]] = 1.f;
        __syncthreads();
    } else if (topk) {
        n = ktop;  // int(N * selection_p), evaluated on the host in double like Python does
        for (int i = tid; i < N; i += 256) {
            float hi = H[i];
            int rank = 0;
            for (int j = 0; j < N; ++j) { float hj = H[j]; rank += (hj < hi) || (hj == hi && j < i); }
            bool sel = rank < n;
            coef[i] = sel ? 1.f : 0.f;
            if (sel && idx) idx[rank] = i;
        }
        __syncthreads();
    } else {
        // torch.where(H <= thresh): ascending index order
        for (int base = 0; base < N; base += 256) {
            int i = base + tid;
            bool sel = (i < N) && (H[i] <= thresh);
            unsigned long long bal = __ballot(sel);
            if ((tid & 63) == 0) wcnt[tid >> 6] = __popcll(bal);
            __syncthreads();
            int off = sn;
            for (int w = 0; w < (tid >> 6); ++w) off += wcnt[w];
            int pos = off + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
            if (i < N) coef[i] = sel ? 1.f : 0.f;
            if (sel && idx) idx[pos] = i;
            __syncthreads();
            if (tid == 0) sn += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
        }
        n = sn;
    }
    if (keep) {
        // second-stage filter (PLPD, deyo.py:144-151): drop selected views whose keep flag is 0; idx keeps
        // the first-stage order, the loss / gradient / step use the surviving set
        __syncthreads();
        float cnt = 0.f;
        for (int i = tid; i < N; i += 256) {
            if (coef[i] != 0.f && !keep[i]) coef[i] = 0.f;
            cnt += coef[i];
        }
        n = (int)(block_sum(cnt, red) + 0.5f);
    }
    if (tid == 0) { if (n_io) *n_io = n; *nsel_dev = n; }
    if (objective == 0) {
        // coeff_i = reweight / exp(H_i - margin) ; loss = mean_{i in S}(H_i * coeff_i)
        float part = 0.f;
        for (int i = tid; i < N; i += 256) {
            float c = 0.f;
            if (coef[i] != 0.f) {
                c = (reweight != 0.f) ? reweight * (1.0f / expf(H[i] - margin)) : 1.0f;
                part += H[i] * c;
                c /= (float)n;
            }
            coef[i] = c;  // c_i / n, 0 for unselected
        }
        part = block_sum(part, red);
        if (tid == 0 && loss) *loss = (n > 0) ? part / (float)n : 0.f;
    }
}

// ---- loss, pass 3 (DeYO): dz_ik = -(c_i/n) p_ik (log p_ik + H_i)
__global__ __launch_bounds__(256) void deyo_grad_kernel(const float* __restrict__ z, int K, const float* __restrict__ H,
                                                        const float* __restrict__ lse, const float* __restrict__ coef,
                                                        float* __restrict__ dz) {
    const int i = blockIdx.x;
    const float c = coef[i], h = H[i], l = lse[i];
    const float* zr = z + (size_t)i * K;
    float* dr = dz + (size_t)i * K;
    for (int k = threadIdx.x; k < K; k += 256) {
        float lp = zr[k] - l;
        dr[k] = (c != 0.f) ? -c * expf(lp) * (lp + h) : 0.f;
    }
}

// ---- TPT: avg_k = logsumexp_{i in S}(logp_ik) - log n ; loss = -sum_k avg_k e^{avg_k}
// pass A (grid over class chunks): column statistics into scratch: cmax[k], csum[k]
__global__ __launch_bounds__(256) void tpt_col_kernel(const float* __restrict__ z, int N, int K,
                                                      const float* __restrict__ lse, const float* __restrict__ sel,
                                                      const int* __restrict__ nsel, float* __restrict__ avg,
                                                      float* __restrict__ gk) {
    int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    int n = *nsel;
    if (n == 0) { avg[k] = 0.f; gk[k] = 0.f; gk[K + k] = 0.f; return; }
    float mx = -INFINITY;
    for (int i = 0; i < N; ++i) if (sel[i] != 0.f) mx = fmaxf(mx, z[(size_t)i * K + k] - lse[i]);
    float se = 0.f;
    for (int i = 0; i < N; ++i) if (sel[i] != 0.f) se += expf(z[(size_t)i * K + k] - lse[i] - mx);
    float lsek = mx + logf(se);                // logsumexp over selected views
    float a = lsek - logf((float)n);
    a = fmaxf(a, -3.4028234663852886e38f);     // clamp(min=finfo.min), ttl.py:59-60
    avg[k] = lsek;                             // keep the un-shifted lse for the weights
    gk[k] = -(1.0f + a) * expf(a);             // dL/davg_k
    // loss contribution reduced by the row kernel (needs a grid-wide sum): store -a e^a in place
    gk[K + k] = -a * expf(a);
}
// pass B: one block per view: dz_ij = g_j w_ij - p_ij sum_k g_k w_ik, w_ik = exp(logp_ik - lse_k)
__global__ __launch_bounds__(256) void tpt_grad_kernel(const float* __restrict__ z, int N, int K,
                                                       const float* __restrict__ lse, const float* __restrict__ sel,
                                                       const float* __restrict__ avg, const float* __restrict__ gk,
                                                       float* __restrict__ dz, float* __restrict__ loss) {
    __shared__ float red[4];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float* zr = z + (size_t)i * K;
    float* dr = dz + (size_t)i * K;
    if (i == 0 && loss) {
        float part = 0.f;
        for (int k = tid; k < K; k += 256) part += gk[K + k];
        part = block_sum(part, red);
        if (tid == 0) *loss = part;
    }
    if (sel[i] == 0.f) {
        for (int k = tid; k < K; k += 256) dr[k] = 0.f;
        return;
    }
    const float l = lse[i];
    float sgw = 0.f;
    for (int k = tid; k < K; k += 256) sgw += gk[k] * expf(zr[k] - l - avg[k]);
    sgw = block_sum(sgw, red);
    for (int k = tid; k < K; k += 256) {
        float lp = zr[k] - l;
        dr[k] = gk[k] * expf(lp - avg[k]) - expf(lp) * sgw;
    }
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, const int* __restrict__ nsel) {
    if (nsel && *nsel == 0) return;  // deyo.py:183: no step when nothing was selected
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    if (!isfinite(gi)) return;       // GradScaler semantics: never step on inf/nan
    float pi = p[i] * (1.0f - lr * wd);
    float mi = b1 * m[i] + (1.0f - b1) * gi;
    float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
    m[i] = mi;
    v[i] = vi;
}

__global__ void scaler_pre_step_kernel(ScalerState st, const int* __restrict__ nsel, int host_step, float b1, float b2, int dynamic,
                                       float growth, float backoff, int interval) {
    if (nsel && *nsel == 0) { st.i[SC_DO_STEP] = 0; st.i[SC_FOUND_INF] = 0; return; }   // deyo.py:183: neither step nor update
    if (st.i[SC_FOUND_INF]) {
        st.i[SC_DO_STEP] = 0;
        st.i[SC_SKIPPED] += 1;
        st.i[SC_TRACKER] = 0;
        if (dynamic) st.f[SC_SCALE] *= backoff;
    } else {
        const int t = host_step > 0 ? host_step : st.i[SC_STEP] + 1;
        st.i[SC_STEP] = t;
        st.i[SC_DO_STEP] = 1;
        st.f[SC_BC1] = (float)(1.0 - pow((double)b1, (double)t));
        st.f[SC_BC2S] = (float)sqrt(1.0 - pow((double)b2, (double)t));
        if (dynamic && ++st.i[SC_TRACKER] >= interval) { st.f[SC_SCALE] *= growth; st.i[SC_TRACKER] = 0; }
    }
    st.f[SC_INV] = 1.0f / st.f[SC_SCALE];
    st.i[SC_FOUND_INF] = 0;
}

__global__ void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 size_t n, float lr, float b1, float b2, float eps, float wd, ScalerState st) {
    if (!st.i[SC_DO_STEP]) return;       // the whole step or nothing (GradScaler.step)
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float bc1 = st.f[SC_BC1], bc2_sqrt = st.f[SC_BC2S];
    float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    float mi = b1 * m[i] + (1.0f - b1) * gi;
    float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
    m[i] = mi;
    v[i] = vi;
}

__global__ void scaler_unscale_kernel(float* __restrict__ g, size_t n, ScalerState st) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float t = g[i] * st.f[SC_INV];
    g[i] = t;
    if (!isfinite(t)) atomicOr(st.i + SC_FOUND_INF, 1);
}

__global__ void scaler_reset_step_kernel(ScalerState st) { st.i[SC_STEP] = 0; st.i[SC_DO_STEP] = 0; st.i[SC_FOUND_INF] = 0; }

__global__ void reset_kernel(float* __restrict__ p, const float* __restrict__ snap, float* __restrict__ m,
                             float* __restrict__ v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    p[i] = snap[i];
    if (m) m[i] = 0.f;
    if (v) v[i] = 0.f;
}

}  // namespace

hipError_t launch_head_fwd(const HeadArgs& a, int n, hipStream_t s) {
#ifdef TTL_DIAG_SKIP       // timing-only ablation of the episode (tools/class_cost_ab.sh): bit 0 = no head forward / backward launches
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 1) && diag_skip_now(cnt, 80)) return hipSuccess; }
#endif
    hipError_t e = launch_layernorm(a.h, (long long)a.T * a.D, a.ln_g, a.ln_b, a.y, nullptr, 0, a.cls_mean, a.cls_rstd, n, a.D,
                                    a.eps, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(head_proj_kernel, dim3((a.E + HO - 1) / HO, n), dim3(HB), a.D * sizeof(float), s, a);
    if (a.K > 0) hipLaunchKernelGGL(head_logits_kernel, dim3((a.K + HO - 1) / HO, n), dim3(HB), a.E * sizeof(float), s, a);
    return hipGetLastError();
}

hipError_t launch_head_logits(const HeadArgs& a, int n, hipStream_t s) {
    hipLaunchKernelGGL(head_logits_kernel, dim3((a.K + HO - 1) / HO, n), dim3(HB), a.E * sizeof(float), s, a);
    return hipGetLastError();
}

hipError_t launch_head_bwd(const HeadArgs& a, const float* dlogits, float* dh, op_t* dh16, int n, hipStream_t s) {
#ifdef TTL_DIAG_SKIP       // timing-only ablation of the episode (tools/class_cost_ab.sh): bit 0 = no head forward / backward launches
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 1) && diag_skip_now(cnt, 40)) return hipSuccess; }
#endif
    hipLaunchKernelGGL(head_dfh_kernel, dim3((a.E + HO - 1) / HO, n), dim3(HB), a.K * sizeof(float), s, a, dlogits);
    hipLaunchKernelGGL(head_dy_kernel, dim3((a.D + HO - 1) / HO, n), dim3(HB), a.E * sizeof(float), s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // LayerNorm backward on the CLS rows -> compact [n, D]
    return launch_layernorm_bwd(a.tmp_d, a.h, a.cls_mean, a.cls_rstd, a.ln_g, nullptr, dh, dh16, n, a.D, s,
                                (long long)a.T * a.D, (long long)a.D);
}

// scratch layout (floats): [0,N) lse | [N,2N) coef | [2N,2N+1) nsel(int) | [2N+4, 2N+4+N) H (if H_out null)
//                          | then avg[K] | gk[2K]
hipError_t launch_entropy_loss(const float* logits, int N, int K, int objective, int mode, double rho, float thresh,
                               float margin, float reweight, int reuse_idx, float* H_out, long long* idx_io, int* n_io,
                               float* loss_out, float* dlogits, float* scratch, hipStream_t s, const unsigned char* keep) {
    float* lse = scratch;
    float* coef = scratch + N;
    int* nsel = (int*)(scratch + 2 * N);
    float* H = H_out ? H_out : scratch + 2 * N + 4;
    float* avg = scratch + 3 * N + 4;
    float* gk = avg + K;
    hipLaunchKernelGGL(row_stats_kernel, dim3(N), dim3(256), 0, s, logits, K, H, lse);
    const int ktop = (int)((double)N * rho);  // Python: int(batch_entropy.size()[0] * top), ttl.py:52 / deyo.py:105
    hipLaunchKernelGGL(select_kernel, dim3(1), dim3(256), 0, s, H, N, objective, mode, ktop, thresh, margin, reweight,
                       reuse_idx, idx_io, n_io, loss_out, coef, nsel, keep);
    if (objective == 0) {
        hipLaunchKernelGGL(deyo_grad_kernel, dim3(N), dim3(256), 0, s, logits, K, H, lse, coef, dlogits);
    } else {
        hipLaunchKernelGGL(tpt_col_kernel, dim3((K + 255) / 256), dim3(256), 0, s, logits, N, K, lse, coef, nsel, avg, gk);
        hipLaunchKernelGGL(tpt_grad_kernel, dim3(N), dim3(256), 0, s, logits, N, K, lse, coef, avg, gk, dlogits, loss_out);
    }
    return hipGetLastError();
}

hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                        float wd, int step, const int* n_selected, hipStream_t s) {
    double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps,
                       wd, (float)bc1, (float)sqrt(bc2), n_selected);
    return hipGetLastError();
}

hipError_t launch_scaler_pre_step(ScalerState st, const int* n_selected, int host_step, float b1, float b2, int dynamic, float growth,
                                  float backoff, int interval, hipStream_t s) {
    hipLaunchKernelGGL(scaler_pre_step_kernel, dim3(1), dim3(1), 0, s, st, n_selected, host_step, b1, b2, dynamic, growth, backoff, interval);
    return hipGetLastError();
}

hipError_t launch_adamw_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
                            ScalerState st, hipStream_t s) {
    hipLaunchKernelGGL(adamw_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, st);
    return hipGetLastError();
}

hipError_t launch_scaler_unscale(float* g, size_t n, ScalerState st, hipStream_t s) {
    hipLaunchKernelGGL(scaler_unscale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, n, st);
    return hipGetLastError();
}

hipError_t launch_scaler_reset_step(ScalerState st, hipStream_t s) {
    hipLaunchKernelGGL(scaler_reset_step_kernel, dim3(1), dim3(1), 0, s, st);
    return hipGetLastError();
}

hipError_t launch_lora_reset(float* p, const float* snap, float* m, float* v, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, snap, m, v, n);
    return hipGetLastError();
}
