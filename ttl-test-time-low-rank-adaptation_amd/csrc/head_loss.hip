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
__device__ __forceinline__ int block_sum_int(int v, int* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += red[i];
    return t;
}

// ---- head forward / backward products (round 4).  All four are  C[v][o] = post( sum_i A'[v][i] * W[i][o] )  with W row-major
// [I][O] in fp32: CLS -> post_layernorm -> visual_projection, f/||f|| -> logits, and the two backward products
// (HF modeling_clip.py:646-651,751; clip/custom_clip.py:680-687 and their autograd).  They are tiny GEMMs (64 x 768 x 512 ...)
// whose cost was pure latency: round 3 walked 192 strided terms per thread behind L2 round trips (25 us for head_proj, the same
// 25 us for ONE view).  Now: one 16 x 16 output tile per block on the matrix pipe in EXACT fp32 (v_mfma_f32_16x16x4_f32: a
// k-ordered fp32 fma chain per output, no reduced precision — the selection mask hangs on these logits), the reduction split over
// the block's 8 waves, every wave requesting ALL of its operands before the first product (one L2 round trip), the 8 partial tiles
// summed in LDS in a fixed order (deterministic).  The per-row work in front of a product — LayerNorm statistics of the pooled
// rows, ||f||, <f^, df^> — is the block's prologue (its 16 rows, two per wave), so the separate LayerNorm launch and the
// normalisation passes are gone: head forward = 2 launches, backward = 2 + the pooled rows' LayerNorm backward.
constexpr int HM_WAVES = 8, HM_THREADS = 64 * HM_WAVES;
constexpr int HM_CB = 8;          // 16-deep k chunks a wave keeps in flight per batch (I <= 1024: one batch)
enum { HS_PROJ = 0, HS_LOGITS = 1, HS_DFH = 2, HS_DY = 3 };
constexpr int HM_LNC = 4;         // float4 chunks of a row per lane in the LayerNorm prologue (D <= 1024, as ln_fwd_kernel)

template <int STAGE, bool VEC>
__global__ __launch_bounds__(HM_THREADS) void head_mm_kernel(HeadArgs a, int n, const float* __restrict__ dz) {
    __shared__ float part[HM_WAVES][4][64];
    __shared__ float rowc[16][2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int col0 = blockIdx.x * 16, row0 = blockIdx.y * 16;
    const int I = STAGE == HS_PROJ ? a.D : STAGE == HS_DFH ? a.K : a.E;
    const int O = STAGE == HS_PROJ ? a.E : STAGE == HS_LOGITS ? a.K : STAGE == HS_DFH ? a.E : a.D;
    const float* __restrict__ W = STAGE == HS_PROJ ? a.WpT : STAGE == HS_LOGITS ? a.tfeatT : STAGE == HS_DFH ? a.tfeat : a.Wp;
    // ---- this wave's share of the reduction: chunks of 16 k; inside a chunk lane (r, q) holds k = 16c + 4q + j for product j
    // (the MFMA sums its four k's in lane-group order: any assignment of k's to (q, j) is a valid order as long as A and B agree)
    const int nchunks = (I + 15) >> 4, cpw = (nchunks + HM_WAVES - 1) / HM_WAVES;
    const int c_beg = wave * cpw, c_end = min(nchunks, c_beg + cpw);
    const int col = min(col0 + r, O - 1);
    float av[HM_CB][4], bv[HM_CB][4];
    // W operands of a batch of chunks.  No branch inside a load block (a branch around a load makes hipcc wait for the loads in
    // front of it, which turns the batch back into a serial walk): addresses are clamped into the matrix (finite weights), what
    // lies beyond I or beyond the wave's share is multiplied by an exact zero on the A side.
    auto load_w = [&](int cb) {
#pragma unroll
        for (int j = 0; j < HM_CB; ++j) {
            const int kc = 16 * min(cb + j, nchunks - 1) + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[j][e] = W[(size_t)min(kc + e, I - 1) * O + col];
        }
    };
    load_w(c_beg);      // the first batch's weights travel while the prologue computes the row constants
    // ---- prologue: per-row constants of the block's 16 rows, two rows per wave (rows >= n: the last row's, never stored)
    if (STAGE != HS_DFH) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int lr = 2 * wave + rr, v = min(row0 + lr, n - 1);
            if (STAGE == HS_PROJ) {
                // LayerNorm statistics exactly as ln_fwd_kernel forms them (lane l owns float4 chunks l, l+64, ...; fp32)
                const float* xr = a.h + (size_t)v * a.T * a.D;
                const int nch = a.D >> 2;
                float4 x[HM_LNC];
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < HM_LNC; ++i) {
                    x[i] = *(const float4*)(xr + 4 * min(lane + 64 * i, nch - 1));      // (clamped: no branch around a load)
                }
#pragma unroll
                for (int i = 0; i < HM_LNC; ++i) {
                    if (lane + 64 * i >= nch) x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
                }
                const float mu = wave_sum(s) / (float)a.D;
                float qq = 0.f;
#pragma unroll
                for (int i = 0; i < HM_LNC; ++i) {
                    if (lane + 64 * i < nch) {
                        const float d0 = x[i].x - mu, d1 = x[i].y - mu, d2 = x[i].z - mu, d3 = x[i].w - mu;
                        qq += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                    }
                }
                const float rs = rsqrtf(wave_sum(qq) / (float)a.D + a.eps);
                if (lane == 0) {
                    rowc[lr][0] = mu; rowc[lr][1] = rs;
                    if (blockIdx.x == 0 && row0 + lr < n) { a.cls_mean[v] = mu; a.cls_rstd[v] = rs; }
                }
            } else {
                // ||f|| (and, for the backward, <f, df^>) in ONE pass: the row's chunks are requested together (E <= 1024)
                const float* f = a.f + (size_t)v * a.E;
                const float* dfh = a.tmp_e + (size_t)v * a.E;
                const int nch = a.E >> 2;
                float nn = 0.f, raw = 0.f;
                float4 t[HM_LNC], d[HM_LNC];
#pragma unroll
                for (int i = 0; i < HM_LNC; ++i) {
                    const int c = min(lane + 64 * i, nch - 1);
                    t[i] = *(const float4*)(f + 4 * c);
                    if (STAGE == HS_DY) d[i] = *(const float4*)(dfh + 4 * c);
                }
#pragma unroll
                for (int i = 0; i < HM_LNC; ++i) {
                    if (lane + 64 * i < nch) {
                        nn += (t[i].x * t[i].x + t[i].y * t[i].y) + (t[i].z * t[i].z + t[i].w * t[i].w);
                        if (STAGE == HS_DY) raw += (d[i].x * t[i].x + d[i].y * t[i].y) + (d[i].z * t[i].z + d[i].w * t[i].w);
                    }
                }
                const float nrm = sqrtf(wave_sum(nn));
                if (STAGE == HS_LOGITS) {
                    if (lane == 0) rowc[lr][0] = a.scale / nrm;
                } else {   // HS_DY: df = (dfh - fh <fh, dfh>) / ||f||, fh = f / ||f||
                    const float dot = wave_sum(raw) / nrm;
                    if (lane == 0) { rowc[lr][0] = nrm; rowc[lr][1] = dot; }
                }
            }
        }
        __syncthreads();
    }
    const int row = min(row0 + r, n - 1);
    const float c0 = (STAGE == HS_DFH) ? 0.f : rowc[r][0], c1 = (STAGE == HS_PROJ || STAGE == HS_DY) ? rowc[r][1] : 0.f;
    const float* arow = STAGE == HS_PROJ ? a.h + (size_t)row * a.T * a.D : STAGE == HS_DFH ? dz + (size_t)row * a.K
                        : STAGE == HS_LOGITS ? a.f + (size_t)row * a.E : a.tmp_e + (size_t)row * a.E;
    const float* frow = a.f + (size_t)row * a.E;       // HS_DY only
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // VEC (I % 4 == 0): one 16-B load per lane and chunk, a group of four k's is valid or invalid as one.
    for (int cb = c_beg; cb < c_end; cb += HM_CB) {
        if (cb != c_beg) load_w(cb);
        // phase 1: every load of the batch; phase 2: the arithmetic on them (kept apart so that the waits come after ALL requests)
        float x4[HM_CB][4], f4[HM_CB][4], g4[HM_CB][4], b4[HM_CB][4];
#pragma unroll
        for (int j = 0; j < HM_CB; ++j) {
            const int kc = 16 * min(cb + j, nchunks - 1) + 4 * q;
            if (VEC) {
                const int kl = min(kc, I - 4);
                const float4 t = *(const float4*)(arow + kl);
                x4[j][0] = t.x; x4[j][1] = t.y; x4[j][2] = t.z; x4[j][3] = t.w;
                if (STAGE == HS_PROJ) {
                    const float4 g = *(const float4*)(a.ln_g + kl), b = *(const float4*)(a.ln_b + kl);
                    g4[j][0] = g.x; g4[j][1] = g.y; g4[j][2] = g.z; g4[j][3] = g.w;
                    b4[j][0] = b.x; b4[j][1] = b.y; b4[j][2] = b.z; b4[j][3] = b.w;
                }
                if (STAGE == HS_DY) { const float4 u = *(const float4*)(frow + kl); f4[j][0] = u.x; f4[j][1] = u.y; f4[j][2] = u.z; f4[j][3] = u.w; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = min(kc + e, I - 1);
                    x4[j][e] = arow[k];
                    if (STAGE == HS_PROJ) { g4[j][e] = a.ln_g[k]; b4[j][e] = a.ln_b[k]; }
                    if (STAGE == HS_DY) f4[j][e] = frow[k];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < HM_CB; ++j) {
            const int c = cb + j;
            const int kc = 16 * min(c, nchunks - 1) + 4 * q;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = x4[j][e];
                if (STAGE == HS_PROJ) t = (t - c0) * c1 * g4[j][e] + b4[j][e];    // LayerNorm apply, ln_fwd_kernel's expression
                if (STAGE == HS_DY) t = (t - (f4[j][e] / c0) * c1) / c0;
                av[j][e] = (c < c_end && kc + e < I) ? t : 0.f;                     // beyond I / beyond this wave's share: exact zero
            }
        }
#pragma unroll
        for (int j = 0; j < HM_CB; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][e], bv[j][e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) part[wave][e][lane] = acc[e];
    __syncthreads();
    if (wave != 0) return;
    // accumulator register e of lane (r, q) = C[row0 + 4q + e][col0 + r]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = ((part[0][e][lane] + part[1][e][lane]) + (part[2][e][lane] + part[3][e][lane])) +
                        ((part[4][e][lane] + part[5][e][lane]) + (part[6][e][lane] + part[7][e][lane]));
        const int lr = 4 * q + e, orow = row0 + lr, ocol = col0 + r;
        if (orow >= n || ocol >= O) continue;
        if (STAGE == HS_PROJ) {
            a.f[(size_t)orow * a.E + ocol] = t;
            if (a.feats_out) a.feats_out[(size_t)orow * a.E + ocol] = t;
        } else if (STAGE == HS_LOGITS) {
            a.logits[(size_t)orow * a.K + ocol] = t * rowc[lr][0];
        } else if (STAGE == HS_DFH) {
            // scaler.scale(loss): the loss scale of the context's GradScaler state (1 in the bf16 build), removed again in
            // wgrad_reduce_kernel (scaler.unscale_)
            a.tmp_e[(size_t)orow * a.E + ocol] = t * a.scale * (a.gscale ? a.gscale[0] : 1.0f);
        } else {
            a.tmp_d[(size_t)orow * a.D + ocol] = t;
        }
    }
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

// ---- DeYO loss, passes 2 + 3 in ONE launch (round 4).  One block per view.  Every block repeats the O(N) part of
// the selection for the whole batch from the N entropies (N <= SEL_NMAX: a few thousand compares) — the same fp32 compares, so the
// same bit-exact set as select_kernel — and writes its own row of dz_ik = -(c_i/n) p_ik (log p_ik + H_i); block 0 also writes the
// index list in the reference's order, the count, the loss, and clears found_inf for the backward that follows (the memset the
// episode would otherwise enqueue).  rank_i = #{j : H_j < H_i or (H_j == H_i and j < i)} == position in a stable ascending argsort.
constexpr int SEL_NMAX = 2048;
__global__ __launch_bounds__(256) void deyo_select_grad_kernel(const float* __restrict__ z, int N, int K, const float* __restrict__ H,
                                                               const float* __restrict__ lse, int mode, int ktop, float thresh,
                                                               float margin, float reweight, const unsigned char* __restrict__ keep,
                                                               long long* __restrict__ idx, int* __restrict__ n_io,
                                                               float* __restrict__ loss, int* __restrict__ nsel_dev,
                                                               int* __restrict__ clear_flag, float* __restrict__ dz) {
    __shared__ float sH[SEL_NMAX];
    __shared__ float red[4];
    __shared__ int redi[4];
    __shared__ int s_mine;
    const int i = blockIdx.x, tid = threadIdx.x;
    for (int j = tid; j < N; j += 256) sH[j] = H[j];
    if (tid == 0) s_mine = 0;
    __syncthreads();
    const bool topk = (mode == 1);
    int cnt = 0;
    float lpart = 0.f;
    for (int j = tid; j < N; j += 256) {
        const float hj = sH[j];
        bool sel;
        int pos = 0;      // topk: the rank; le_thresh: how many selected views precede j (torch.where order)
        if (topk) {
            for (int t = 0; t < N; ++t) { const float ht = sH[t]; pos += (ht < hj) || (ht == hj && t < j); }
            sel = pos < ktop;
        } else {
            sel = hj <= thresh;
            if (i == 0 && sel && idx)
                for (int t = 0; t < j; ++t) pos += sH[t] <= thresh;
        }
        if (i == 0 && sel && idx) idx[pos] = j;                 // first-stage list, before the keep filter (deyo.py:103-108)
        const bool live = sel && (!keep || keep[j]);              // second-stage filter (PLPD, deyo.py:144-151)
        cnt += live;
        if (j == i) s_mine = live;
        if (live) lpart += hj * ((reweight != 0.f) ? reweight * (1.0f / expf(hj - margin)) : 1.0f);
    }
    const int n = block_sum_int(cnt, redi);
    if (i == 0) {
        lpart = block_sum(lpart, red);
        if (tid == 0) {
            if (n_io) *n_io = n;
            *nsel_dev = n;
            if (loss) *loss = (n > 0) ? lpart / (float)n : 0.f;
            if (clear_flag) *clear_flag = 0;
        }
    }
    __syncthreads();
    float c = 0.f;
    const float h = sH[i], l = lse[i];
    if (s_mine) {       // coeff_i = reweight / exp(H_i - margin), over n (deyo.py:175-181)
        c = (reweight != 0.f) ? reweight * (1.0f / expf(h - margin)) : 1.0f;
        c /= (float)n;
    }
    const float* zr = z + (size_t)i * K;
    float* dr = dz + (size_t)i * K;
    for (int k = tid; k < K; k += 256) {
        const float lp = zr[k] - l;
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

// One AdamW element (torch.optim.AdamW, decoupled weight decay), the SAME instruction sequence in every kernel that steps: plain
// IEEE fp32 operations, no fma contraction (hipcc contracts  b1*m + (1-b1)*g  one way or the other depending on the code around it,
// which made the fused and the two-launch optimizer differ in the last bit from the second step on).
__device__ __forceinline__ float adamw_elem(float p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float wd,
                                            float bc1, float bc2_sqrt) {
#pragma clang fp contract(off)
    const float pi = p * (1.0f - lr * wd);
    const float mi = b1 * m + (1.0f - b1) * g;
    const float vi = b2 * v + (1.0f - b2) * g * g;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    m = mi;
    v = vi;
    return pi - (lr / bc1) * (mi / denom);
}

__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float bc2_sqrt, const int* __restrict__ nsel) {
    if (nsel && *nsel == 0) return;  // deyo.py:183: no step when nothing was selected
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float gi = g[i];
    if (!isfinite(gi)) return;       // GradScaler semantics: never step on inf/nan
    float mi = m[i], vi = v[i];
    p[i] = adamw_elem(p[i], gi, mi, vi, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
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
    float mi = m[i], vi = v[i];
    p[i] = adamw_elem(p[i], g[i], mi, vi, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
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

__global__ void scaler_reset_step_kernel(ScalerState st) {
    st.i[SC_STEP] = 0; st.i[SC_DO_STEP] = 0; st.i[SC_FOUND_INF] = 0; st.i[SC_STEP_A] = 0; st.i[SC_STEP_B] = 0;
}

// The operand-dtype images of LoRA element i of the flat buffer (what refresh_kernel, lora.hip, derives per layer): tensor
// ti = i / (r D) — per layer A [r,D] then B [D,r] of every adapter slot — and the element's place in the K-extension columns of the
// projection images and in the stacked A / B^T rows of the skinny products.
__device__ __forceinline__ void write_lora_images(const LoraImages& im, size_t i, float val) {
    const int per = im.r * im.D;
    const int ti = (int)(i / per), w = (int)(i - (size_t)ti * per);
    const int layer = ti / (2 * im.ntg), rem = ti - layer * 2 * im.ntg, slot = rem >> 1, isB = rem & 1;
    if (layer >= im.layers) return;
    const LoraLayerImages& L = im.L[layer];
    const int t = im.proj[slot];
    const op_t x = f32_to_op(val);
    const int D = im.D, r = im.r;
    if (t < 3) {
        int k = 0;                      // position among the enabled q/k/v adapters
        for (int s2 = 0; s2 < slot; ++s2) k += im.proj[s2] < 3;
        if (isB) {
            const int nB = w / r, jB = w - nB * r;
            L.wext[(size_t)(t * D + nB) * im.ldw + D + k * r + jB] = x;
            L.btcat[(size_t)(k * r + jB) * D + nB] = x;
        } else {
            const int jA = w / D, dA = w - jA * D;
            L.acat[(size_t)(k * r + jA) * D + dA] = x;
            L.wtext[(size_t)dA * im.ldwt + 3 * D + k * r + jA] = x;
        }
    } else {
        if (isB) {
            const int nB = w / r, jB = w - nB * r;
            L.woext[(size_t)nB * im.ldwo + D + jB] = x;
            L.btcat_o[(size_t)jB * D + nB] = x;
        } else {
            const int jA = w / D, dA = w - jA * D;
            L.acat_o[(size_t)jA * D + dA] = x;
            L.wotext[(size_t)dA * im.ldwo + D + jA] = x;
        }
    }
}

// ---- scaler.step(optimizer) + scaler.update() + AdamW in ONE launch (round 4; deyo.py:186-188, ttl.py:218).
// Every block takes the GradScaler decision itself, from words NO block of this launch writes: found_inf (OR-ed by the backward's
// gradient reduction, cleared again when the next loss is formed), *nsel, and the step count of slot `parity` — the launch writes
// the other slot, the fused episode alternates (update u reads slot u & 1; the episodic reset zeroes both).  Block 0 alone
// carries the state forward (scale, growth tracker, skip count, the canonical step count): none of it is read by the other blocks.
__global__ void adamw_fused_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                   size_t n, float lr, float b1, float b2, float eps, float wd, ScalerState st,
                                   const int* __restrict__ nsel, int parity, int dynamic, float growth, float backoff, int interval,
                                   const LoraImages im) {
    __shared__ float s_bc[2];
    const bool none = nsel && *nsel == 0;            // deyo.py:183: neither step nor update when nothing was selected
    const bool inf = st.i[SC_FOUND_INF] != 0;
    const int t = st.i[SC_STEP_A + parity] + 1;
    if (threadIdx.x == 0) {
        s_bc[0] = (float)(1.0 - pow((double)b1, (double)t));
        s_bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
        if (blockIdx.x == 0) {
            const bool step = !none && !inf;
            st.i[SC_STEP_A + (parity ^ 1)] = step ? t : t - 1;
            st.i[SC_DO_STEP] = step;
            if (step) {
                st.i[SC_STEP] = t;
                st.f[SC_BC1] = s_bc[0]; st.f[SC_BC2S] = s_bc[1];
                if (dynamic && ++st.i[SC_TRACKER] >= interval) { st.f[SC_SCALE] *= growth; st.i[SC_TRACKER] = 0; }
            } else if (!none) {       // found_inf: the WHOLE step is skipped, the scale backs off
                st.i[SC_SKIPPED] += 1;
                st.i[SC_TRACKER] = 0;
                if (dynamic) st.f[SC_SCALE] *= backoff;
            }
            if (!none) st.f[SC_INV] = 1.0f / st.f[SC_SCALE];
        }
    }
    if (none || inf) return;
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float bc1 = s_bc[0], bc2_sqrt = s_bc[1];
    float mi = m[i], vi = v[i];
    const float pn = adamw_elem(p[i], g[i], mi, vi, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
    p[i] = pn;
    m[i] = mi;
    v[i] = vi;
    if (im.layers > 0) write_lora_images(im, i, pn);      // (a skipped step leaves parameters AND images as they are)
}

// LoRA_AB.reset() + optimizer.load_state_dict(empty) (clip/custom_clip.py:202-215, ttl.py:344) + the scaler's per-image step
// counters in one launch; the loss scale PERSISTS (the reference's only cross-image state, Q14)
__global__ void episode_reset_kernel(float* __restrict__ p, const float* __restrict__ snap, float* __restrict__ m,
                                     float* __restrict__ v, size_t n, ScalerState st, const LoraImages im) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        st.i[SC_STEP] = 0; st.i[SC_DO_STEP] = 0; st.i[SC_FOUND_INF] = 0; st.i[SC_STEP_A] = 0; st.i[SC_STEP_B] = 0;
    }
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = snap[i];
    p[i] = x;
    m[i] = 0.f;
    v[i] = 0.f;
    if (im.layers > 0) write_lora_images(im, i, x);
}

// accuracy(output, target, topk=(1, 5)) of utils/tools.py:88-102 as integer hit counts for ONE prediction row, accumulated on
// the device: hits[0] += target in top-1, hits[1] += target in top-min(5,K), hits[2] += 1.  rank = #{k : z_k > z_t or (z_k == z_t
// and k < t)} (ties: lowest index first).  A target outside [0, K) never hits.
__global__ __launch_bounds__(256) void topk_hits_kernel(const float* __restrict__ z, int K, const long long* __restrict__ target,
                                                        long long* __restrict__ hits) {
    __shared__ int redi[4];
    const long long t = target[0];
    const bool valid = t >= 0 && t < K;
    int cnt = 0;
    if (valid) {
        const float zt = z[t];
        for (int k = threadIdx.x; k < K; k += 256) { const float zk = z[k]; cnt += (zk > zt) || (zk == zt && k < t); }
    }
    const int rank = block_sum_int(cnt, redi);
    if (threadIdx.x == 0) {
        hits[0] += (valid && rank < 1) ? 1 : 0;
        hits[1] += (valid && rank < (K < 5 ? K : 5)) ? 1 : 0;
        hits[2] += 1;
    }
}

__global__ void reset_kernel(float* __restrict__ p, const float* __restrict__ snap, float* __restrict__ m,
                             float* __restrict__ v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    p[i] = snap[i];
    if (m) m[i] = 0.f;
    if (v) v[i] = 0.f;
}

}  // namespace

template <int STAGE>
static void head_mm_launch(const HeadArgs& a, int n, int O, const float* dz, hipStream_t s) {
    const int I = STAGE == HS_PROJ ? a.D : STAGE == HS_DFH ? a.K : a.E;
    const dim3 grid((O + 15) / 16, (n + 15) / 16);
    if (I % 4 == 0 && I >= 4) hipLaunchKernelGGL((head_mm_kernel<STAGE, true>), grid, dim3(HM_THREADS), 0, s, a, n, dz);
    else hipLaunchKernelGGL((head_mm_kernel<STAGE, false>), grid, dim3(HM_THREADS), 0, s, a, n, dz);
}

hipError_t launch_head_fwd(const HeadArgs& a, int n, hipStream_t s) {
#ifdef TTL_DIAG_SKIP       // timing-only ablation of the episode (tools/class_cost_ab.sh): bit 0 = no head forward / backward launches
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 1) && diag_skip_now(cnt, 80)) return hipSuccess; }
#endif
    if (a.D % 4 || a.D > 256 * HM_LNC || a.E % 4 || a.E > 256 * HM_LNC) return hipErrorInvalidValue;
    head_mm_launch<HS_PROJ>(a, n, a.E, nullptr, s);          // post_layernorm of the pooled rows (prologue) + visual_projection
    if (a.K > 0) head_mm_launch<HS_LOGITS>(a, n, a.K, nullptr, s);
    return hipGetLastError();
}

hipError_t launch_head_logits(const HeadArgs& a, int n, hipStream_t s) {
    if (a.E % 4 || a.E > 256 * HM_LNC) return hipErrorInvalidValue;
    head_mm_launch<HS_LOGITS>(a, n, a.K, nullptr, s);
    return hipGetLastError();
}

hipError_t launch_head_bwd(const HeadArgs& a, const float* dlogits, float* dh, op_t* dh16, int n, hipStream_t s) {
#ifdef TTL_DIAG_SKIP       // timing-only ablation of the episode (tools/class_cost_ab.sh): bit 0 = no head forward / backward launches
    { static std::atomic<int> cnt{0}; if ((TTL_DIAG_SKIP & 1) && diag_skip_now(cnt, 40)) return hipSuccess; }
#endif
    if (a.E % 4 || a.E > 256 * HM_LNC) return hipErrorInvalidValue;
    head_mm_launch<HS_DFH>(a, n, a.E, dlogits, s);
    head_mm_launch<HS_DY>(a, n, a.D, nullptr, s);
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
                               float* loss_out, float* dlogits, float* scratch, hipStream_t s, const unsigned char* keep,
                               int* clear_flag) {
    float* lse = scratch;
    float* coef = scratch + N;
    int* nsel = (int*)(scratch + 2 * N);
    float* H = H_out ? H_out : scratch + 2 * N + 4;
    float* avg = scratch + 3 * N + 4;
    float* gk = avg + K;
    hipLaunchKernelGGL(row_stats_kernel, dim3(N), dim3(256), 0, s, logits, K, H, lse);
    const int ktop = (int)((double)N * rho);  // Python: int(batch_entropy.size()[0] * top), ttl.py:52 / deyo.py:105
    if (objective == 0 && N <= SEL_NMAX) {    // DeYO: selection + loss + gradient in one launch
        hipLaunchKernelGGL(deyo_select_grad_kernel, dim3(N), dim3(256), 0, s, logits, N, K, H, lse, mode, ktop, thresh, margin, reweight,
                           keep, idx_io, n_io, loss_out, nsel, clear_flag, dlogits);
        return hipGetLastError();
    }
    if (clear_flag) { hipError_t e = hipMemsetAsync(clear_flag, 0, sizeof(int), s); if (e != hipSuccess) return e; }
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

hipError_t launch_adamw_fused(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
                              ScalerState st, const int* n_selected, int parity, int dynamic, float growth, float backoff, int interval,
                              hipStream_t s, const LoraImages* img) {
    LoraImages none = {};
    hipLaunchKernelGGL(adamw_fused_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd, st,
                       n_selected, parity & 1, dynamic, growth, backoff, interval, img ? *img : none);
    return hipGetLastError();
}

hipError_t launch_episode_reset(float* p, const float* snap, float* m, float* v, size_t n, ScalerState st, hipStream_t s,
                                const LoraImages* img) {
    LoraImages none = {};
    hipLaunchKernelGGL(episode_reset_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, snap, m, v, n, st, img ? *img : none);
    return hipGetLastError();
}

hipError_t launch_topk_hits(const float* logits, int K, const long long* target, long long* hits, hipStream_t s) {
    hipLaunchKernelGGL(topk_hits_kernel, dim3(1), dim3(256), 0, s, logits, K, target, hits);
    return hipGetLastError();
}
