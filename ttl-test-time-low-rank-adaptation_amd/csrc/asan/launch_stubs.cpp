// HOST-ONLY launch layer for `make asan` (see asan/hip/hip_runtime.h): every launcher of kernels.hpp exists here as a host
// function that (a) for the kernels of the WEIGHT path (casts, transposes, LoRA refresh, peer-feature normalisation, reset)
// does the kernel's work on the host exactly as indexed by the kernel, and (b) for every other kernel touches the first and the
// last byte each operand / output of the launch covers — including the rows past M that the unguarded big-M GEMM epilogue
// stores and the head-major q/k/v image — so AddressSanitizer checks the extents the host glue (api.hip) passes against the
// sizes it allocated.  Results of (b) are meaningless; nothing here is linked into libttl_hip.so.
#include <math.h>

#include <initializer_list>

#include "../kernels.hpp"

namespace {

inline void rd(const void* p, size_t byte) { volatile unsigned char c = ((const volatile unsigned char*)p)[byte]; (void)c; }
inline void wr(void* p, size_t byte) { ((volatile unsigned char*)p)[byte] = 0; }
inline void span_r(const void* p, size_t bytes) { if (p && bytes) { rd(p, 0); rd(p, bytes - 1); } }
inline void span_w(void* p, size_t bytes) { if (p && bytes) { wr(p, 0); wr(p, bytes - 1); } }
// rows x cols of `esz`-byte elements at a row pitch of ld elements
inline void mat_r(const void* p, long long rows, long long cols, long long ld, int esz) {
    if (p && rows > 0 && cols > 0) { rd(p, 0); rd(p, (size_t)(((rows - 1) * ld + cols) * esz - 1)); }
}
inline void mat_w(void* p, long long rows, long long cols, long long ld, int esz) {
    if (p && rows > 0 && cols > 0) { wr(p, 0); wr(p, (size_t)(((rows - 1) * ld + cols) * esz - 1)); }
}

op_t to_op(float f) {       // bf16 round-to-nearest-even of the bits (the product build converts in hardware)
    uint32_t u;
    memcpy(&u, &f, 4);
    return (op_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

}  // namespace

// ---------------------------------------------------------------- GEMM
bool gemm_big_applicable(GemmEpi epi, const GemmArgs& a) {
    if (epi == EPI_PATCH) return false;
    if (epi == EPI_GELU_BWD && (!a.aux || (size_t)((a.M + 127) / 128) * 128 > (size_t)a.padded)) return false;   // 128-row tiles, u read per tile
    if (a.M < 1024 || a.N % 256 || a.K % 64 || a.K / 64 < 3) return false;
    if (a.amap || a.cmap || a.c2map || a.splits > 1) return false;
    return true;
}
bool gemm_takes_big(GemmEpi epi, const GemmArgs& a) {
    return a.padded && gemm_big_applicable(epi, a) && (size_t)((a.M + 159) / 160) * 160 <= (size_t)a.padded;
}
unsigned qkv_hm_magic(int T, int limit) {
    if (T < 1 || limit < 1) return 0;
    const uint64_t magic = (1ull << 32) / (uint64_t)T + 1;
    if (magic >> 32) return 0;
    for (uint64_t m = 0; m < (uint64_t)limit; ++m)
        if (((m * magic) >> 32) != m / (uint64_t)T) return 0;
    return (unsigned)magic;
}
hipError_t launch_gemm_big(GemmEpi epi, const GemmArgs& a, hipStream_t s) { return launch_gemm(epi, a, s); }

hipError_t launch_gemm(GemmEpi epi, const GemmArgs& a, hipStream_t) {
    if (a.M <= 0 || a.N % 128 || a.K % 64 || a.K <= 0 || (a.lda & 7) || (a.ldb & 7)) return hipErrorInvalidValue;
    if ((a.amap || a.cmap || a.c2map) && a.M >= 1024) return hipErrorInvalidValue;
    const bool big = gemm_takes_big(epi, a);
    if (a.hm_T && !big) return hipErrorInvalidValue;
    // the big-M kernel reads A rows clamped to M-1 but STORES whole 160-row tiles (a.padded rows exist in every output buffer)
    const long long rows_st = !big ? a.M : epi == EPI_GELU_BWD ? (long long)((a.M + 127) / 128) * 128 : (long long)((a.M + 159) / 160) * 160;
    if (a.amap) { span_r(a.amap, (size_t)a.M * 4); for (int m : {0, a.M - 1}) mat_r(a.A + (size_t)a.amap[m] * a.lda, 1, a.K, a.lda, 2); }
    else mat_r(a.A, a.M, a.K, a.lda, 2);
    mat_r(a.B, a.N, a.K, a.ldb, 2);
    span_r(a.bias, (size_t)a.N * 4);
    const bool f32out = epi == EPI_F32 || epi == EPI_RESID_F32 || epi == EPI_PATCH;
    const int esz = f32out ? 4 : 2;
    if (epi == EPI_PATCH) {
        const long long last = (long long)((a.M - 1) / a.G2) * a.T + 1 + (a.M - 1) % a.G2;
        mat_w(a.C, last + 1, a.N, a.ldc, 4);
        mat_r(a.pos, a.G2 + 1, a.N, a.N, 4);
    } else if (a.hm_T) {     // head-major q/k/v: row m = view*T + t, the image of the last stored row ends one (view, plane, head) tile
        const long long view = (rows_st - 1) / a.hm_T;
        span_w(a.C, (size_t)((view + 1) * (long long)a.N * a.hm_T) * 2);
    } else if (a.cmap) {
        span_r(a.cmap, (size_t)a.M * 4);
        for (int m : {0, a.M - 1}) mat_w((char*)a.C + (size_t)a.cmap[m] * a.ldc * esz, 1, a.N, a.ldc, esz);
    } else mat_w(a.C, rows_st, a.N, a.ldc, esz);
    if (epi == EPI_RESID_F32) {
        if (a.cmap) for (int m : {0, a.M - 1}) mat_r(a.resid + (size_t)a.cmap[m] * a.ldr, 1, a.N, a.ldr, 4);
        else mat_r(a.resid, rows_st, a.N, a.ldr, 4);
    }
    if (epi == EPI_GELU && a.C2) {
        if (a.c2map) { span_r(a.c2map, (size_t)a.M * 4); for (int m : {0, a.M - 1}) mat_w(a.C2 + (size_t)a.c2map[m] * a.ldc2, 1, a.N, a.ldc2, 2); }
        else mat_w(a.C2, rows_st, a.N, a.ldc2, 2);
    }
    if (epi == EPI_GELU_BWD) mat_r(a.aux, big ? rows_st : a.M, a.N, a.ldaux, 2);     // (the big-M kernel reads u of whole row tiles)
    if (a.M < 1024 && (epi == EPI_F32 || epi == EPI_RESID_F32) && a.ws && a.K >= 1536) span_w(a.ws, a.ws_bytes);
    return hipSuccess;
}

// ---------------------------------------------------------------- elementwise (weight path: done for real)
hipError_t launch_cast_f32_op(const float* src, op_t* dst, size_t n, hipStream_t) {
    for (size_t i = 0; i < n; ++i) dst[i] = to_op(src[i]);
    return hipSuccess;
}
hipError_t launch_transpose_f32_op(const float* src, int rows_src, int cols_src, op_t* dst, int ld_dst, hipStream_t) {
    for (int r = 0; r < rows_src; ++r)
        for (int c = 0; c < cols_src; ++c) dst[(size_t)c * ld_dst + r] = to_op(src[(size_t)r * cols_src + c]);
    return hipSuccess;
}
hipError_t launch_cast_rows_f32_op(const float* src, int rows, int cols, op_t* dst, int ld_dst, hipStream_t) {
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) dst[(size_t)r * ld_dst + c] = to_op(src[(size_t)r * cols + c]);
    return hipSuccess;
}
hipError_t launch_unit_rows(const float* src, int n, int E, int normalize, float* dst, float* dstT, hipStream_t) {
    for (int v = 0; v < n; ++v) {
        double nn = 0;
        for (int e = 0; e < E; ++e) nn += (double)src[(size_t)v * E + e] * src[(size_t)v * E + e];
        const float nrm = normalize ? (float)sqrt(nn) : 1.f;
        for (int e = 0; e < E; ++e) {
            const float t = src[(size_t)v * E + e] / nrm;
            dst[(size_t)v * E + e] = t;
            dstT[(size_t)e * n + v] = t;
        }
    }
    return hipSuccess;
}
hipError_t launch_fill_zero(void* p, size_t bytes, hipStream_t) { memset(p, 0, bytes); return hipSuccess; }
hipError_t launch_transpose_f32(const float* src, int R, int C, float* dst, hipStream_t) {
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) dst[(size_t)c * R + r] = src[(size_t)r * C + c];
    return hipSuccess;
}

// ---------------------------------------------------------------- elementwise (activation path: extents only)
hipError_t launch_im2col(const float* x, op_t* patches, int n, int S, int P, int Kp, hipStream_t) {
    span_r(x, (size_t)n * 3 * S * S * 4);
    span_w(patches, (size_t)n * (S / P) * (S / P) * Kp * 2);
    return hipSuccess;
}
hipError_t launch_cls_rows(float* h, const float* cls, const float* pos, int n, int T, int D, hipStream_t) {
    span_r(cls, (size_t)D * 4); span_r(pos, (size_t)D * 4);
    mat_w(h, (long long)(n - 1) * T + 1, D, D, 4);
    return hipSuccess;
}
hipError_t launch_embed_layernorms(float* h, const float* cls, const float* pos, int, const float* g0, const float* b0, const float* g1,
                                   const float* b1, op_t* y16, int ld16, float* mean, float* rstd, int rows, int D, float, hipStream_t) {
    span_r(cls, (size_t)D * 4); span_r(pos, (size_t)D * 4);
    for (const float* p : {g0, b0, g1, b1}) span_r(p, (size_t)D * 4);
    mat_r(h, rows, D, D, 4); mat_w(h, rows, D, D, 4); mat_w(y16, rows, D, ld16, 2);
    span_w(mean, (size_t)rows * 4); span_w(rstd, (size_t)rows * 4);
    return hipSuccess;
}
hipError_t launch_layernorm(const float* x, long long row_stride, const float* gamma, const float* beta, float* y_f32, op_t* y_bf16,
                            int ld_bf16, float* mean, float* rstd, int rows, int D, float, hipStream_t, const int* rowmap) {
    if (rowmap) { span_r(rowmap, (size_t)rows * 4); for (int r : {0, rows - 1}) mat_r(x + (size_t)rowmap[r] * row_stride, 1, D, D, 4); }
    else mat_r(x, rows, D, row_stride, 4);
    span_r(gamma, (size_t)D * 4); span_r(beta, (size_t)D * 4);
    mat_w(y_f32, rows, D, D, 4); mat_w(y_bf16, rows, D, ld_bf16, 2);
    span_w(mean, (size_t)rows * 4); span_w(rstd, (size_t)rows * 4);
    return hipSuccess;
}
hipError_t launch_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, const float* dres,
                                float* out_f32, op_t* out_bf16, int rows, int D, hipStream_t, long long x_stride, long long o_stride,
                                int stat_stride, int dres_T, const int* pool, long long ld_bf16) {
    if (!x_stride) x_stride = D;
    if (!o_stride) o_stride = D;
    if (!ld_bf16) ld_bf16 = o_stride;
    mat_r(dy, rows, D, D, 4); mat_r(x, rows, D, x_stride, 4);
    mat_r(mean, rows, 1, stat_stride, 4); mat_r(rstd, rows, 1, stat_stride, 4); span_r(gamma, (size_t)D * 4);
    if (dres_T > 0) { mat_r(dres, rows / dres_T, D, D, 4); if (pool) span_r(pool, (size_t)(rows / dres_T) * 4); }
    else mat_r(dres, rows, D, o_stride, 4);
    mat_w(out_f32, rows, D, o_stride, 4); mat_w(out_bf16, rows, D, ld_bf16, 2);
    return hipSuccess;
}
hipError_t launch_text_embed(const int* ids, const float* tok, const float* pos, float* h, int rows, int T, int D, hipStream_t) {
    span_r(ids, (size_t)rows * 4); span_r(tok, (size_t)D * 4); mat_r(pos, T, D, D, 4); mat_w(h, rows, D, D, 4);
    return hipSuccess;
}
hipError_t launch_gather_rows_f32(const float* src, long long ld, const int* pool, int T, float* dst, int n, int cols, hipStream_t) {
    span_r(pool, (size_t)n * 4);
    for (int v : {0, n - 1}) mat_r(src + ((size_t)v * T + pool[v]) * ld, 1, cols, ld, 4);
    mat_w(dst, n, cols, cols, 4);
    return hipSuccess;
}
hipError_t launch_gather_rows_op(const op_t* src, long long ld, const int* pool, int T, op_t* dst, int n, int cols, hipStream_t) {
    span_r(pool, (size_t)n * 4);
    for (int v : {0, n - 1}) mat_r(src + ((size_t)v * T + pool[v]) * ld, 1, cols, ld, 2);
    mat_w(dst, n, cols, cols, 2);
    return hipSuccess;
}
hipError_t launch_gather_view_blocks(const GatherTable& t, const long long* idx, int n_sel, hipStream_t) {
    span_r(idx, (size_t)n_sel * 8);     // (the index values come from a stubbed selection: only block 0 of every source is touched)
    for (int i = 0; i < t.n; ++i) { span_r(t.e[i].src, t.e[i].block_bytes); span_w(t.e[i].dst, (size_t)n_sel * t.e[i].block_bytes); }
    return hipSuccess;
}
hipError_t launch_splitk_reduce(const float* part, int splits, int M, int N, const float* resid, int ldr, const float* bias, float* out,
                                int ldc, hipStream_t, const int*) {
    span_r(part, (size_t)splits * M * N * 4); mat_r(resid, M, N, ldr, 4); span_r(bias, (size_t)N * 4); mat_w(out, M, N, ldc, 4);
    return hipSuccess;
}

// ---------------------------------------------------------------- attention (extents of both q/k/v layouts)
static void qkv_extent(const op_t* qkv, const QkvLayout& L, int n, int T, int H) {
    const long long last = (long long)(n - 1) * L.view + (long long)(T - 1) * L.tok + (long long)(H - 1) * L.head + 63;
    rd(qkv, 0); rd(qkv, (size_t)(last + L.v_off) * 2 + 1); rd(qkv, (size_t)(last + L.k_off) * 2 + 1);
}
hipError_t launch_attention_fwd(const op_t* qkv, QkvLayout L, op_t* out, int ld_out, float* lse, int n, int T, int H, hipStream_t, int) {
    qkv_extent(qkv, L, n, T, H);
    mat_w(out, (long long)n * T, H * 64, ld_out, 2); span_w(lse, (size_t)n * H * T * 4);
    return hipSuccess;
}
hipError_t launch_attention_bwd(const op_t* qkv, QkvLayout L, const op_t* out, const op_t* dout, int ld_o, const float* lse, op_t* dqkv,
                                int ld_dqkv, int n, int T, int H, int, hipStream_t, int) {
    qkv_extent(qkv, L, n, T, H);
    mat_r(out, (long long)n * T, H * 64, ld_o, 2); mat_r(dout, (long long)n * T, H * 64, ld_o, 2); span_r(lse, (size_t)n * H * T * 4);
    mat_w(dqkv, (long long)n * T, 3 * H * 64, ld_dqkv, 2);
    return hipSuccess;
}
hipError_t launch_attention_fwd_cls(const op_t* qkv, QkvLayout L, op_t* out, int ld_out, float* lse, int n, int T, int H, hipStream_t,
                                    const int* qpos, int) {
    qkv_extent(qkv, L, n, T, H);
    if (qpos) span_r(qpos, (size_t)n * 4);
    mat_w(out, (long long)(n - 1) * T + 1, H * 64, ld_out, 2); span_w(lse, (size_t)n * H * T * 4);
    return hipSuccess;
}
hipError_t launch_attention_bwd_cls(const op_t* qkv, QkvLayout L, const op_t* out, int ld_o, const op_t* dout_cls, const float* lse,
                                    op_t* dqkv, int ld_dqkv, int n, int T, int H, int, hipStream_t, const int* qpos, int) {
    qkv_extent(qkv, L, n, T, H);
    if (qpos) span_r(qpos, (size_t)n * 4);
    mat_r(out, (long long)(n - 1) * T + 1, H * 64, ld_o, 2); mat_r(dout_cls, n, H * 64, H * 64, 2); span_r(lse, (size_t)n * H * T * 4);
    mat_w(dqkv, (long long)n * T, 3 * H * 64, ld_dqkv, 2);
    return hipSuccess;
}

// ---------------------------------------------------------------- head / loss / optimizer
static void head_extent(const HeadArgs& a, int n) {
    mat_r(a.h, (long long)(n - 1) * a.T + 1, a.D, a.D, 4);
    span_r(a.ln_g, (size_t)a.D * 4); span_r(a.ln_b, (size_t)a.D * 4);
    span_r(a.WpT, (size_t)a.D * a.E * 4); span_r(a.Wp, (size_t)a.D * a.E * 4);
    if (a.K > 0) { span_r(a.tfeat, (size_t)a.K * a.E * 4); span_r(a.tfeatT, (size_t)a.K * a.E * 4); }
    span_w(a.cls_mean, (size_t)n * 4); span_w(a.cls_rstd, (size_t)n * 4); span_w(a.f, (size_t)n * a.E * 4);
}
hipError_t launch_head_fwd(const HeadArgs& a, int n, hipStream_t) {
    head_extent(a, n);
    if (a.K > 0) span_w(a.logits, (size_t)n * a.K * 4);
    span_w(a.feats_out, (size_t)n * a.E * 4);
    return hipSuccess;
}
hipError_t launch_head_logits(const HeadArgs& a, int n, hipStream_t) {
    span_r(a.f, (size_t)n * a.E * 4); span_r(a.tfeatT, (size_t)a.K * a.E * 4); span_w(a.logits, (size_t)n * a.K * 4);
    return hipSuccess;
}
hipError_t launch_head_bwd(const HeadArgs& a, const float* dlogits, float* dcls, op_t* dcls_bf16, int n, hipStream_t) {
    span_r(dlogits, (size_t)n * a.K * 4); span_r(a.f, (size_t)n * a.E * 4);
    span_w(a.tmp_e, (size_t)n * a.E * 4); span_w(a.tmp_d, (size_t)n * a.D * 4);
    span_w(dcls, (size_t)n * a.D * 4); span_w(dcls_bf16, (size_t)n * a.D * 2);
    if (a.gscale) span_r(a.gscale, 4);
    return hipSuccess;
}
hipError_t launch_entropy_loss(const float* logits, int N, int K, int, int, double, float, float, float, int, float* H_out, long long* idx_io,
                               int* n_io, float* loss_out, float* dlogits, float* scratch, hipStream_t, const unsigned char* keep, int* clear_flag) {
    if (clear_flag) *clear_flag = 0;
    span_r(logits, (size_t)N * K * 4); span_w(dlogits, (size_t)N * K * 4); span_w(scratch, ((size_t)4 * N + 3 * K + 16) * 4);
    span_w(H_out, (size_t)N * 4); span_w(idx_io, (size_t)N * 8); span_w(loss_out, 4); if (keep) span_r(keep, N);
    if (n_io) *n_io = N;      // "every view selected": the optimizer step below runs
    return hipSuccess;
}
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float, float, float, float, float, int, const int* nsel, hipStream_t) {
    span_w(p, n * 4); span_r(g, n * 4); span_w(m, n * 4); span_w(v, n * 4); if (nsel) span_r(nsel, 4);
    return hipSuccess;
}
hipError_t launch_scaler_pre_step(ScalerState st, const int* nsel, int, float, float, int, float, float, int, hipStream_t) {
    span_w(st.f, SC_NF * 4); span_w(st.i, SC_NI * 4); if (nsel) span_r(nsel, 4);
    return hipSuccess;
}
hipError_t launch_adamw_dev(float* p, const float* g, float* m, float* v, size_t n, float, float, float, float, float, ScalerState st, hipStream_t) {
    span_w(p, n * 4); span_r(g, n * 4); span_w(m, n * 4); span_w(v, n * 4); span_r(st.f, SC_NF * 4); span_r(st.i, SC_NI * 4);
    return hipSuccess;
}
// the images the fused reset / optimizer launches write element by element: the same extents the refresh stub touches
static void lora_image_extents(const LoraImages* im) {
    if (!im || im->layers <= 0) return;
    const size_t D = im->D, r = im->r;
    int nqkv = 0, has_o = 0;
    for (int k = 0; k < im->ntg; ++k) { if (im->proj[k] < 3) ++nqkv; else has_o = 1; }
    for (int i = 0; i < im->layers; ++i) {
        const LoraLayerImages& L = im->L[i];
        if (nqkv) {
            mat_w(L.wext, 3 * D, D + nqkv * r, im->ldw, 2); mat_w(L.wtext, D, 3 * D + nqkv * r, im->ldwt, 2);
            span_w(L.acat, nqkv * r * D * 2); span_w(L.btcat, nqkv * r * D * 2);
        }
        if (has_o) {
            mat_w(L.woext, D, D + r, im->ldwo, 2); mat_w(L.wotext, D, D + r, im->ldwo, 2);
            span_w(L.acat_o, r * D * 2); span_w(L.btcat_o, r * D * 2);
        }
    }
}
hipError_t launch_adamw_fused(float* p, const float* g, float* m, float* v, size_t n, float, float, float, float, float, ScalerState st, const int* nsel,
                              int, int, float, float, int, hipStream_t, const LoraImages* im) {
    lora_image_extents(im);
    span_w(p, n * 4); span_r(g, n * 4); span_w(m, n * 4); span_w(v, n * 4); span_w(st.f, SC_NF * 4); span_w(st.i, SC_NI * 4); if (nsel) span_r(nsel, 4);
    return hipSuccess;
}
hipError_t launch_episode_reset(float* p, const float* snap, float* m, float* v, size_t n, ScalerState st, hipStream_t, const LoraImages* im) {
    lora_image_extents(im);
    memcpy(p, snap, n * 4); memset(m, 0, n * 4); memset(v, 0, n * 4); span_w(st.i, SC_NI * 4);
    return hipSuccess;
}
hipError_t launch_topk_hits(const float* logits, int K, const long long* target, long long* hits, hipStream_t) {
    span_r(logits, (size_t)K * 4); span_r(target, 8); span_w(hits, 24);
    return hipSuccess;
}
hipError_t launch_scaler_unscale(float* g, size_t n, ScalerState st, hipStream_t) { span_w(g, n * 4); span_r(st.f, SC_NF * 4); span_w(st.i, 4); return hipSuccess; }
hipError_t launch_scaler_reset_step(ScalerState st, hipStream_t) { span_w(st.i, SC_NI * 4); return hipSuccess; }
hipError_t launch_lora_reset(float* p, const float* snap, float* m, float* v, size_t n, hipStream_t) {
    memcpy(p, snap, n * 4);
    if (m) memset(m, 0, n * 4);
    if (v) memset(v, 0, n * 4);
    return hipSuccess;
}

// ---------------------------------------------------------------- LoRA
hipError_t launch_lora_refresh(const LoraPtrs& P, int D, int r, op_t* wext, int ldw, op_t* wtext, int ldwt, op_t* acat, op_t* btcat,
                               op_t* woext, op_t* wotext, int ldwo, op_t* acat_o, op_t* btcat_o, hipStream_t) {
    int slot[3], k = 0;
    for (int t = 0; t < 3; ++t) slot[t] = P.A[t] ? k++ : -1;
    for (int i = 0; i < D * r; ++i) {     // as refresh_kernel (lora.hip) indexes it
        const int nB = i / r, jB = i - nB * r, jA = i / D, dA = i - jA * D;
        for (int t = 0; t < 3; ++t) {
            if (slot[t] < 0) continue;
            const op_t b = to_op(P.B[t][i]), a = to_op(P.A[t][i]);
            wext[(size_t)(t * D + nB) * ldw + D + slot[t] * r + jB] = b;
            btcat[(size_t)(slot[t] * r + jB) * D + nB] = b;
            acat[(size_t)(slot[t] * r + jA) * D + dA] = a;
            wtext[(size_t)dA * ldwt + 3 * D + slot[t] * r + jA] = a;
        }
        if (P.A[3]) {
            const op_t b = to_op(P.B[3][i]), a = to_op(P.A[3][i]);
            woext[(size_t)nB * ldwo + D + jB] = b;
            btcat_o[(size_t)jB * D + nB] = b;
            acat_o[(size_t)jA * D + dA] = a;
            wotext[(size_t)dA * ldwo + D + jA] = a;
        }
    }
    return hipSuccess;
}
hipError_t launch_lora_skinny(const op_t* X, long long ldx, const int* xoff, int ntg, const op_t* Wcat, int D, int r, float, op_t* out,
                              long long ldo, int M, hipStream_t, const int* rowmap) {
    if (ntg < 1 || ntg > 3) return hipErrorInvalidValue;
    if (rowmap) span_r(rowmap, (size_t)M * 4);
    for (int k = 0; k < ntg; ++k) {
        if (rowmap) for (int m : {0, M - 1}) mat_r(X + (size_t)rowmap[m] * ldx + xoff[k], 1, D, ldx, 2);
        else mat_r(X + xoff[k], M, D, ldx, 2);
    }
    mat_r(Wcat, (long long)ntg * r, D, D, 2);
    if (rowmap) for (int m : {0, M - 1}) mat_w(out + (size_t)rowmap[m] * ldo, 1, (long long)ntg * r, ldo, 2);
    else mat_w(out, M, (long long)ntg * r, ldo, 2);
    return hipSuccess;
}
int lora_wgrad_chunks(int M) { return (M + 255) / 256; }
hipError_t launch_lora_wgrad(const WgradList& L, int M, int D, int r, float* partial, hipStream_t, const float* sf, int* si) {
    if (D % 64 || L.n < 1 || L.n > WGRAD_MAX) return hipErrorInvalidValue;
    for (int p = 0; p < L.n; ++p) {
        mat_r(L.p[p].S, M, r, L.p[p].lds, 2); mat_r(L.p[p].G, M, D, L.p[p].ldg, 2);
        span_w(L.p[p].out, (size_t)r * D * 4);
    }
    span_w(partial, (size_t)L.n * lora_wgrad_chunks(M) * r * D * 4);
    if (sf) span_r(sf, SC_NF * 4);
    if (si) span_w(si, 4);
    return hipSuccess;
}

// ---------------------------------------------------------------- views
int views_kstride(int H, int W, int S) {
    const int m = H > W ? H : W;
    double scale = (double)m / S;
    if (scale < 1.0) scale = 1.0;
    return (int)ceil(2.0 * scale) * 2 + 1 + 2 + 1;
}
hipError_t launch_make_views(const unsigned char* img, int H, int W, const int* boxes, int n, int S, const float* mean, const float* stdv,
                             float* out, int* table, int kstride, hipStream_t) {
    span_r(img, (size_t)H * W * 3); span_r(boxes, (size_t)n * 5 * 4); span_r(mean, 12); span_r(stdv, 12);
    span_w(out, (size_t)n * 3 * S * S * 4); span_w(table, (size_t)n * 2 * S * kstride * 4);
    return hipSuccess;
}

// ---------------------------------------------------------------- PLPD filter (plpd.hip)
size_t plpd_views_workspace_floats(int n_max, int S, int aug, int patch_len) {
    if (aug == PLPD_OCC) return (size_t)n_max * 3;
    if (aug == PLPD_PATCH && patch_len > 0 && S % patch_len) return 2 * (size_t)n_max * 3 * S * S;
    return 0;
}
hipError_t launch_plpd_views(const float* x, const long long* idx, const int* n_sel, int n_max, int S, const PlpdArgs& p, float* out, float* ws,
                             hipStream_t) {
    if (n_max < 1 || S < 1) return hipErrorInvalidValue;
    span_r(idx, (size_t)n_max * 8); span_r(n_sel, 4);
    span_r(x, (size_t)3 * S * S * 4);                     // (which views are read depends on idx: the first one stands in)
    span_w(out, (size_t)n_max * 3 * S * S * 4);
    const size_t wsf = plpd_views_workspace_floats(n_max, S, p.aug, p.patch_len);
    if (wsf) span_w(ws, wsf * 4);
    if (p.aug == PLPD_PATCH) span_r(p.perm, (size_t)n_max * p.patch_len * p.patch_len * 4);
    if (p.aug == PLPD_PIXEL) span_r(p.perm, (size_t)S * S * 4);
    return hipSuccess;
}
hipError_t launch_plpd_keep(const float* logits, const float* logits_prime, const long long* idx, const int* n_sel, int n_max, int N, int K,
                            float, unsigned char* keep, float* plpd_out, hipStream_t) {
    span_r(logits, (size_t)N * K * 4); span_r(logits_prime, (size_t)n_max * K * 4); span_r(idx, (size_t)n_max * 8); span_r(n_sel, 4);
    span_w(keep, (size_t)N); span_w(plpd_out, (size_t)n_max * 4);
    return hipSuccess;
}
