// Host-side launchers of the HIP kernels (one translation unit per kernel family).
// All take a hipStream_t and enqueue only; none synchronises.
#pragma once
#include <atomic>

#include "common.hpp"

// ---------------------------------------------------------------- launch helpers (host)
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): `done` is the kernel's own bit mask of the
// devices already served, so a second device used from the same process gets its opt-in too.  Thread-safe.
inline hipError_t ensure_smem(const void* fn, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    done.fetch_or(bit, std::memory_order_release);
    return hipSuccess;
}
#ifdef TTL_DIAG_SKIP
// Timing-only ablations of the episode (tools/class_cost_ab.sh; never defined in the product build): a call site stops issuing its
// launches after `after` of them, i.e. once the buffers downstream hold realistic data from the warm-up (all-zero MFMA operands
// would raise the clock and flatter the remaining kernels).
inline bool diag_skip_now(std::atomic<int>& n, int after) { return n.fetch_add(1) >= after; }
#endif
// CU count of the CURRENT device (cached per device), 0 on error
inline int device_cu_count() {
    static std::atomic<int> ncu[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int v = ncu[dev & 63].load(std::memory_order_relaxed);
    if (v) return v;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
    ncu[dev & 63].store(p.multiProcessorCount, std::memory_order_relaxed);
    return p.multiProcessorCount;
}

// ---------------------------------------------------------------- GEMM (gemm.hip)
// C[M,N] (+)= A[M,K] * B[N,K]^T with bf16 operands (row-major, K contiguous), fp32 accumulate
// on v_mfma_f32_16x16x32_bf16.  Requirements: K % 64 == 0, N % 128 == 0, lda/ldb % 8 == 0.
enum GemmEpi {
    EPI_F32 = 0,       // C fp32 = acc (+ bias)
    EPI_OP = 1,      // C bf16 = acc (+ bias)
    EPI_RESID_F32 = 2, // C fp32 = resid + acc + bias
    EPI_GELU = 3,      // C bf16 = quick_gelu(acc + bias); C2 bf16 = acc + bias (if C2)
    EPI_PATCH = 4,     // C fp32 [(m/G2)*T + 1 + m%G2][n] = acc + pos[1 + m%G2][n]
    EPI_GELU_BWD = 5,  // C bf16 = acc * quick_gelu'(aux[m][n])
};
struct GemmArgs {
    const op_t* A; int lda;
    const op_t* B; int ldb;
    int M, N, K;
    void* C; int ldc;
    const float* bias;            // [N] or null
    const float* resid; int ldr;  // EPI_RESID_F32
    op_t* C2; int ldc2;         // EPI_GELU second output (pre-activation), may be null
    const op_t* aux; int ldaux; // EPI_GELU_BWD: saved pre-activation u
    const float* pos; int G2; int T; // EPI_PATCH
    float* ws; size_t ws_bytes;   // optional split-K workspace (small-M fp32-output calls)
    int splits;                   // internal: K slices of this launch (blockIdx.y)
    int padded;                   // rows every C/resid/aux/C2 buffer really has (>= round_up(M, 320): unguarded epilogue allowed); 0 = unknown
    int xc;                       // internal: columns of the 2-D XCD grid (0 = 1-D tile order)
    // Row maps (small-M launches only, M < 1024): logical row m lives at physical row map[m] of the buffer.
    // Lets the last text-tower layer run on the pooled (end-of-text) row of every prompt in place.
    const int* amap;              // A operand rows
    const int* cmap;              // C and resid rows
    const int* c2map;             // C2 rows
    // Head-major q/k/v output (EPI_OP on the big-M kernel only; hm_T == 0: plain row-major C).  Row m = view*hm_T + t, column
    // c = plane*D + head*64 + d (D = N/3) goes to C[view*3*D*hm_T + plane*D*hm_T + head*64*hm_T + t*64 + d]: every (view, plane,
    // head) is one contiguous [T][64] tile, which is what the attention kernels stage (QkvLayout below).  hm_magic: see qkv_hm_magic.
    int hm_T; unsigned hm_magic;
    // Episodes sharing the GPU with this launch's stream (ttl_ctx_set_concurrency; 0 / 1: alone).  With others in flight the idle CUs of a
    // partial round run their kernels, so a launch costs its CU-time, not its makespan: the tile choice may differ (gemm_huge_applicable).
    int concurrent;
};
hipError_t launch_gemm(GemmEpi epi, const GemmArgs& a, hipStream_t s);
// gemm_big.hip: (32*MT) x 256 tiles, 8 waves, one persistent block per CU; big-M launches whose output buffers have
// `a.padded` >= round_up(M, tile rows) rows (unguarded epilogue)
bool gemm_big_applicable(GemmEpi epi, const GemmArgs& a);
// true iff launch_gemm(epi, a) runs on the big-M kernel (a.padded set as the caller will set it): the only path with GemmArgs::hm_T
bool gemm_takes_big(GemmEpi epi, const GemmArgs& a);
hipError_t launch_gemm_big(GemmEpi epi, const GemmArgs& a, hipStream_t s);
// gemm_huge.hip: 256 x 256 tiles on four waves for the wide, short-K big-M launches (q/k/v, fc1); launch_gemm_big hands these over
// (so gemm_takes_big still answers for them; TTL_GEMM_HUGE=0: everything stays on gemm_big.hip)
bool gemm_huge_applicable(GemmEpi epi, const GemmArgs& a);
hipError_t launch_gemm_huge(GemmEpi epi, const GemmArgs& a, hipStream_t s);

// ---------------------------------------------------------------- elementwise (elementwise.hip)
hipError_t launch_cast_f32_op(const float* src, op_t* dst, size_t n, hipStream_t s);
// dst[r][c] = bf16(src[c][r])  (src [rows_src, cols_src] fp32 -> dst [cols_src, ld_dst] bf16)
hipError_t launch_transpose_f32_op(const float* src, int rows_src, int cols_src, op_t* dst, int ld_dst,
                                     hipStream_t s);
// dst rows of ld_dst elements: bf16(src[r][c]) for c < cols, untouched beyond
hipError_t launch_cast_rows_f32_op(const float* src, int rows, int cols, op_t* dst, int ld_dst,
                                     hipStream_t s);
// patches[(n*G+gy)*G+gx][c*P*P+py*P+px] = bf16(x[n][c][gy*P+py][gx*P+px]); zero pad to Kp
hipError_t launch_im2col(const float* x, op_t* patches, int n, int S, int P, int Kp, hipStream_t s);
// h[n*T + 0][:] = cls + pos[0]
hipError_t launch_cls_rows(float* h, const float* cls, const float* pos, int n, int T, int D, hipStream_t s);
// Episode start of the image tower in one pass: h[r] = LN(r % T == 0 ? cls + pos[0] : h[r]; g0, b0) (fp32, in place) and
// y16[r] = LN(h[r]; g1, b1) in the operand type (+ mean / rstd of the second LayerNorm); bit-identical to launch_cls_rows +
// launch_layernorm(in place) + launch_layernorm
hipError_t launch_embed_layernorms(float* h, const float* cls, const float* pos, int T, const float* g0, const float* b0, const float* g1,
                                   const float* b1, op_t* y16, int ld16, float* mean, float* rstd, int rows, int D, float eps, hipStream_t s);
// LayerNorm over rows of fp32 x [rows, D].  y_f32 (ld D) and/or y_bf16 (ld ld_bf16) outputs;
// mean/rstd optional saves.  row_stride: distance (elements) between consecutive input rows
// (T*D to pick CLS rows).
hipError_t launch_layernorm(const float* x, long long row_stride, const float* gamma, const float* beta,
                            float* y_f32, op_t* y_bf16, int ld_bf16, float* mean, float* rstd, int rows,
                            int D, float eps, hipStream_t s, const int* rowmap = nullptr /* input row r = x row rowmap[r] */);
// dx = LN-backward(dy; x, mean, rstd, gamma); out_f32 = dres + dx; out_bf16 = bf16(out_f32)
// x_stride / o_stride: row pitch (elements) of x and of dres/outputs (D when contiguous);
// stat_stride: pitch of mean/rstd; dres_T > 0: dres is compact [rows/dres_T][D], non-zero only on
// rows that are multiples of dres_T (CLS tokens)
hipError_t launch_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                                const float* gamma, const float* dres, float* out_f32, op_t* out_bf16,
                                int rows, int D, hipStream_t s, long long x_stride = 0, long long o_stride = 0,
                                int stat_stride = 1, int dres_T = 0, const int* pool = nullptr,
                                long long ld_bf16 = 0 /* pitch of out_bf16 when it differs from o_stride (K-extended operand buffers) */);
// text tower: h[row][:] = tok[ids[row]][:] + pos[row % T][:]   (fp32, like HF CLIPTextEmbeddings)
hipError_t launch_text_embed(const int* ids, const float* tok, const float* pos, float* h, int rows, int T, int D, hipStream_t s);
// dst[v][0..cols) = src[(v*T + pool[v]) * ld + 0..cols): the pooled row of every sequence, compact
hipError_t launch_gather_rows_f32(const float* src, long long ld, const int* pool, int T, float* dst, int n, int cols, hipStream_t s);
// Per-view blocks of several buffers at once: for every entry e and j < n_sel, dst_e block j = src_e block idx[j] (+ first bytes).
// The saved activations of the views a top-k selection kept (deyo.py:105, ttl.py:52), packed so that the backward runs on those views
// only (api.hip backward_impl).  stride_bytes = distance between consecutive views in src, block_bytes = what is copied per view and the
// distance between views in dst; all multiples of 4, the pointers 4-byte aligned (16-byte copies where everything is 16-byte aligned).
struct GatherEntry { const void* src; void* dst; unsigned long long stride_bytes; unsigned long long block_bytes; };
constexpr int GATHER_MAX = 64;
struct GatherTable { GatherEntry e[GATHER_MAX]; int n; };
hipError_t launch_gather_view_blocks(const GatherTable& t, const long long* idx, int n_sel, hipStream_t s);
hipError_t launch_gather_rows_op(const op_t* src, long long ld, const int* pool, int T, op_t* dst, int n, int cols, hipStream_t s);
// dst [C,R] = src [R,C]^T (fp32)
hipError_t launch_transpose_f32(const float* src, int R, int C, float* dst, hipStream_t s);
// dst [n,E] = rows of src (divided by their L2 norm when normalize), dstT [E,n] the same transposed
hipError_t launch_unit_rows(const float* src, int n, int E, int normalize, float* dst, float* dstT, hipStream_t s);
hipError_t launch_fill_zero(void* p, size_t bytes, hipStream_t s);
// out[m][n] = (resid ? resid[m][n] : 0) + (bias ? bias[n] : 0) + sum_s part[s][m][n]   (fixed order: deterministic)
hipError_t launch_splitk_reduce(const float* part, int splits, int M, int N, const float* resid, int ldr,
                                const float* bias, float* out, int ldc, hipStream_t s, const int* cmap = nullptr);

// ---------------------------------------------------------------- attention (attention.hip)
// Where element d of (view n, token t, head h) of q / k / v sits in a q/k/v buffer (element offsets):
//   q: n*view + t*tok + h*head + d,   k: + k_off,   v: + v_off
// row-major [n*T][ld] (q | k | v along a row, the projection GEMM's natural output): 128-B head segments at a stride of ld;
// head-major [n][3][H][T][64] (written by the big-M QKV GEMM's epilogue, GemmArgs::hm_T): contiguous [T][64] tiles.
struct QkvLayout { long long view; int tok, head; long long k_off, v_off; };
inline QkvLayout qkv_row_major(int T, int D, int ld) { return {(long long)T * ld, ld, 64, (long long)D, 2LL * D}; }
inline QkvLayout qkv_head_major(int T, int H) {
    const long long plane = (long long)H * 64 * T;
    return {3 * plane, 64, 64 * T, plane, 2 * plane};
}
// m / T == __umulhi(m, magic) for every m < limit, or 0 when no such 32-bit multiplier exists (checked exhaustively)
unsigned qkv_hm_magic(int T, int limit);
// causal != 0: key j is visible to query i only for j <= i (text tower)
hipError_t launch_attention_fwd(const op_t* qkv, QkvLayout lay, op_t* out, int ld_out, float* lse, int n, int T,
                                int H, hipStream_t s, int causal = 0);
hipError_t launch_attention_bwd(const op_t* qkv, QkvLayout lay, const op_t* out, const op_t* dout, int ld_o,
                                const float* lse, op_t* dqkv, int ld_dqkv, int n, int T, int H, int need_dk,
                                hipStream_t s, int causal = 0);
// Forward for query 0 of every sequence only (last image-tower layer): writes row n*T of `out` and lse[n][h][0].
hipError_t launch_attention_fwd_cls(const op_t* qkv, QkvLayout lay, op_t* out, int ld_out, float* lse, int n, int T, int H,
                                    hipStream_t s, const int* qpos = nullptr, int causal = 0);
// Same gradients when d(out) is non-zero only for ONE query of every sequence (the top layer): token 0
// (CLS) or, with qpos != null, token qpos[sequence] (end-of-text).  dout_cls bf16 [n][H*64]; writes dense
// dq (zero rows for every other token), dk, dv.
hipError_t launch_attention_bwd_cls(const op_t* qkv, QkvLayout lay, const op_t* out, int ld_o, const op_t* dout_cls,
                                    const float* lse, op_t* dqkv, int ld_dqkv, int n, int T, int H, int need_dk,
                                    hipStream_t s, const int* qpos = nullptr, int causal = 0);

// ---------------------------------------------------------------- head / loss / optimizer (head_loss.hip)
struct HeadArgs {
    const float* h; int T; int D; int E; int K;   // h: fp32 residual stream [n*T, D]
    const float* ln_g; const float* ln_b; float eps;
    const float* WpT;      // [D][E] fp32 (visual_projection transposed)
    const float* Wp;       // [E][D] fp32
    const float* tfeat;    // [K][E]
    const float* tfeatT;   // [E][K]
    float scale;           // exp(logit_scale)
    const float* gscale;   // device: loss scale of the backward (ScalerState.f[0]); null = 1
    float* cls_mean; float* cls_rstd; float* y; float* f; // saves [n],[n],[n,D],[n,E]
    float* logits;         // [n,K]
    float* feats_out;      // optional [n,E]
    float* tmp_e; float* tmp_d;  // scratch [n,E], [n,D] (backward)
};
hipError_t launch_head_fwd(const HeadArgs& a, int n, hipStream_t s);
// only the logit stage: a.f [n,E] (not normalised) x a.tfeatT -> a.logits [n,K]
hipError_t launch_head_logits(const HeadArgs& a, int n, hipStream_t s);
// dlogits [n,K] -> gradient of the CLS rows, compact [n,D] fp32 + bf16 copy (all other rows of
// the stream gradient are zero)
hipError_t launch_head_bwd(const HeadArgs& a, const float* dlogits, float* dcls, op_t* dcls_bf16, int n, hipStream_t s);

hipError_t launch_entropy_loss(const float* logits, int N, int K, int objective, int mode, double rho, float thresh,
                               float margin, float reweight, int reuse_idx, float* H_out, long long* idx_io,
                               int* n_io, float* loss_out, float* dlogits, float* scratch /*>= 4*N + 3*K + 16 floats*/,
                               hipStream_t s, const unsigned char* keep = nullptr,
                               int* clear_flag = nullptr /* != null: *clear_flag = 0 on the way (ScalerState found_inf before the backward) */);
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2,
                        float eps, float wd, int step, const int* n_selected, hipStream_t s);
// ---- torch.amp.GradScaler semantics (ttl.py:222, deyo.py:186-188) with the state on the device, so that a whole
// episode stays one enqueue.  f[0] loss scale, f[1] 1/scale, f[2] bias correction 1 - b1^t, f[3] sqrt(1 - b2^t);
// i[0] found_inf (OR-ed by the gradient reduction), i[1] growth tracker, i[2] optimizer steps taken since the reset,
// i[3] do_step (decision of the current update), i[4] steps skipped on inf/nan so far.
struct ScalerState { float* f; int* i; };
enum { SC_SCALE = 0, SC_INV = 1, SC_BC1 = 2, SC_BC2S = 3, SC_NF = 4 };
// i[5] / i[6]: the step count again, in two slots the fused optimizer launch alternates between (adamw_fused_kernel)
enum { SC_FOUND_INF = 0, SC_TRACKER = 1, SC_STEP = 2, SC_DO_STEP = 3, SC_SKIPPED = 4, SC_STEP_A = 5, SC_STEP_B = 6, SC_NI = 8 };
// scaler.step + scaler.update decision of one update (one thread): nothing at all when *n_selected == 0 (deyo.py:183);
// found_inf -> skip the WHOLE step, scale *= backoff, tracker = 0; else step (host_step > 0: that step count, else the
// device counter + 1), tracker += 1, scale *= growth every `interval` clean steps.  Clears found_inf.
hipError_t launch_scaler_pre_step(ScalerState st, const int* n_selected, int host_step, float b1, float b2, int dynamic,
                                  float growth, float backoff, int interval, hipStream_t s);
// AdamW over the flat buffer iff i[SC_DO_STEP], bias corrections from f[]
hipError_t launch_adamw_dev(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                            float wd, ScalerState st, hipStream_t s);
// scaler.unscale_(optimizer) on an arbitrary gradient buffer: g *= 1/scale, found_inf |= any non-finite
hipError_t launch_scaler_unscale(float* g, size_t n, ScalerState st, hipStream_t s);
hipError_t launch_scaler_reset_step(ScalerState st, hipStream_t s);
// The fused episode's forms (one launch each): launch_scaler_pre_step + launch_adamw_dev with the step count read from slot
// `parity` & 1 and written to the other slot (update u of an episode passes u; launch_episode_reset zeroes both slots);
// launch_lora_reset (m, v required) + launch_scaler_reset_step.
// `img` (optional, img.layers > 0): the operand-dtype images derived from the LoRA parameters (launch_lora_refresh's outputs) are
// written by the same launch, element by element as the parameter is: the forward that follows needs no refresh launches.
constexpr int LORA_IMG_MAX_LAYERS = 24;
struct LoraLayerImages { op_t *wext, *wtext, *acat, *btcat, *woext, *wotext, *acat_o, *btcat_o; };
struct LoraImages {
    int layers;            // trained layers with adapters (0: no image writes)
    int ntg;               // adapters per layer, in the bound buffer's order (q, k, v, out among the enabled ones)
    int D, r, ldw, ldwt, ldwo;
    int proj[4];           // per adapter slot: 0 q / 1 k / 2 v (slot k of the q/k/v images) or 3 = out_proj
    LoraLayerImages L[LORA_IMG_MAX_LAYERS];
};
hipError_t launch_adamw_fused(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd,
                              ScalerState st, const int* n_selected, int parity, int dynamic, float growth, float backoff, int interval,
                              hipStream_t s, const LoraImages* img = nullptr);
hipError_t launch_episode_reset(float* p, const float* snap, float* m, float* v, size_t n, ScalerState st, hipStream_t s,
                                const LoraImages* img = nullptr);
// utils/tools.py:88-102 accuracy(output, target, (1, 5)) for one prediction row [1,K]: hits[0..2] += {top-1 hit, top-5 hit, 1}
hipError_t launch_topk_hits(const float* logits, int K, const long long* target, long long* hits, hipStream_t s);
hipError_t launch_lora_reset(float* p, const float* snap, float* m, float* v, size_t n, hipStream_t s);

// ---------------------------------------------------------------- LoRA (lora.hip)
// fp32 LoRA parameters of ONE layer, in the order q_proj, k_proj, v_proj, out_proj (null = the projection has no adapter;
// the reference ships q and v, clip/custom_clip.py:586)
struct LoraPtrs { const float* A[4]; const float* B[4]; };
// Refresh the bf16 images derived from them (slot k = position of a q/k/v adapter among the enabled ones):
//   wqkv_ext [3D][ldw]:  cols D + k*r.. of rows t*D.. = B_t ;  wqkvT_ext [D][ldwt]: cols 3D + k*r.. = A_t^T
//   a_cat [nqkv*r][D] = rows of A_t ;  bT_cat [nqkv*r][D] = rows of B_t^T
//   out_proj: wo_ext [D][ldwo] cols D.. = B_o ; woT_ext [D][ldwo] cols D.. = A_o^T ; acat_o [r][D] = A_o ; btcat_o [r][D] = B_o^T
hipError_t launch_lora_refresh(const LoraPtrs& P, int D, int r, op_t* wqkv_ext, int ldw, op_t* wqkvT_ext, int ldwt, op_t* a_cat,
                               op_t* bT_cat, op_t* wo_ext, op_t* woT_ext, int ldwo, op_t* acat_o, op_t* btcat_o, hipStream_t s);
// out[m][k*r + c] = operand(scale * sum_d X[m][xoff[k] + d] * Wcat[k*r + c][d]),  k in [0,ntg), c in [0,r), d in [0,D)
//   (lora_down: every xoff 0; dU: dq at 0, dk at D, dv at 2D).  rowmap (optional): row m lives at physical row rowmap[m].
hipError_t launch_lora_skinny(const op_t* X, long long ldx, const int* xoff, int ntg, const op_t* Wcat, int D, int r, float scale,
                              op_t* out, long long ldo, int M, hipStream_t s, const int* rowmap = nullptr);
// LoRA weight gradients from saved activations, as a list of products  out = S^T · G  over M rows:
//   dB_t = d_t^T Us_t (S = Us_t = s*x*A_t^T [M][r], G = d_t [M][D], stored transposed [D][r]);  dA_t = dU_t^T x (S = dU_t, G = x).
// Pitches in elements (a pitch of T*ld walks the pooled row of every sequence).  partial: fp32 scratch [n][nchunk][r][D].
// scaler_f / scaler_i (ScalerState arrays, may be null): the gradients are divided by the loss scale scaler_f[0] and any
// non-finite value sets scaler_i[0] (found_inf).
constexpr int WGRAD_MAX = 8;
struct WgradProd {
    const op_t* S; long long lds; const op_t* G; long long ldg; float* out; int transpose;
    // filled in by launch_lora_wgrad: second output (rows r.. of a merged product), result rows, offset of the partials
    float* out2; int rows; size_t poff;
};
struct WgradList { WgradProd p[WGRAD_MAX]; int n; };
hipError_t launch_lora_wgrad(const WgradList& L, int M, int D, int r, float* partial, hipStream_t s,
                             const float* scaler_f = nullptr, int* scaler_i = nullptr);
int lora_wgrad_chunks(int M);

// ---------------------------------------------------------------- PLPD filter (plpd.hip; deyo.py:115-151)
enum { PLPD_OCC = 0, PLPD_PATCH = 1, PLPD_PIXEL = 2 };
struct PlpdArgs { int aug; int patch_len, occ_size, row_start, col_start; const int* perm; };
// floats of scratch launch_plpd_views needs (occ: the per-(view, channel) means; patch with S % patch_len != 0: two resize stages)
size_t plpd_views_workspace_floats(int n_max, int S, int aug, int patch_len);
// out[b] = destroy(x[idx[b]]) for b < *n_sel (device count; the launch is sized for n_max rows)
hipError_t launch_plpd_views(const float* x, const long long* idx, const int* n_sel, int n_max, int S, const PlpdArgs& p, float* out,
                             float* ws, hipStream_t s);
// keep [N] = 0, then keep[idx[b]] = softmax(z[idx[b]])[argmax] - softmax(z'[b])[same class] > threshold for b < *n_sel
hipError_t launch_plpd_keep(const float* logits, const float* logits_prime, const long long* idx, const int* n_sel, int n_max, int N, int K,
                            float threshold, unsigned char* keep, float* plpd_out, hipStream_t s);

// ---------------------------------------------------------------- view generator (views.hip)
// img uint8 HWC [H][W][3]; boxes int32 [n][5] = top,left,height,width,flags (bit0 flip, bit1 base view);
// out fp32 [n,3,S,S] normalised; table = n*2*S*kstride ints of scratch, kstride = views_kstride(H,W,S).
int views_kstride(int H, int W, int S);
hipError_t launch_make_views(const unsigned char* img, int H, int W, const int* boxes, int n, int S, const float* mean,
                             const float* stdv, float* out, int* table, int kstride, hipStream_t s);
