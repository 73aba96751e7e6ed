// libttl_hip.so — C ABI (include/ttl_hip.h): context, weight images, and the launch sequences of
// the hot path.  No torch types; the host side (ttl_amd/_lib.py) binds this with ctypes.
//
// HBM layout of a context:
//   weights  : per layer bf16 [3D][D+64] (Wqkv | LoRA B cols | 0), [D][D] Wo, [F][D] W1, [D][F] W2,
//              fp32 biases / LN affine; for the trained layers also the dgrad images
//              bf16 [D][3D+64] (Wqkv^T | LoRA A^T), Wo^T, W1^T [D][F], W2^T [F][D].
//   stream   : fp32 residual stream [M][D]; trained layers keep h_in and h_mid (LN backward).
//   saved    : per trained layer bf16 x1ext [M][D+64] (LN1 out | s·U), qkv [M][3D], attn out [M][D],
//              u [M][F] (fc1 pre-activation), fp32 lse [N][H][T], LN statistics.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <atomic>
#include <string>
#include <vector>

#include "../../include/ttl_hip.h"
#include "kernels.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail((int)e_, "%s -> %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

struct Layer {
    // forward images
    op_t* wqkv;   // [3D][ldw]
    float* bqkv;    // [3D]
    op_t* wo;     // [D][D]
    float* bo;
    op_t* w1;     // [F][D]
    float* b1;
    op_t* w2;     // [D][F]
    float* b2;
    float *ln1g, *ln1b, *ln2g, *ln2b;
    // dgrad images (trained layers only)
    op_t* wqkvT;  // [D][ldwt]
    op_t* woT;    // [D][D]   (= Wo^T: [in][out] -> rows = in)
    op_t* w1T;    // [D][F]
    op_t* w2T;    // [F][D]
    op_t* acat;   // [nqkv*r][D]  rows of A of the enabled q/k/v adapters
    op_t* btcat;  // [nqkv*r][D]  rows of B^T
    op_t *acat_o, *btcat_o;   // [r][D] out_proj adapter: A_o, B_o^T
    int ldwo;     // row pitch of wo / woT: D, or D+64 when the layer carries an out_proj adapter (K-extension columns)
    int ldat;     // row pitch of the layer's saved attention output (same rule)
    // saved activations (trained layers only)
    float* h_in; float* h_mid;
    op_t* x1ext; op_t* qkv; op_t* attn; op_t* u;
    float* lse; float *mu1, *rs1, *mu2, *rs2;
    bool qkv_hm;      // layout of `qkv` as the last forward wrote it: head-major [n][3][H][T][64] (big-M QKV GEMM) or row-major [n*T][3D]
    bool trained;     // i >= layer_lo: activations saved, dgrad images kept, gradient flows through it
    bool lora;        // layer_lo <= i <= layer_hi: carries trainable adapters (--layer_range); layers above layer_hi are
                      // frozen in the reference (B == 0 forever, Q10) but still lie on the gradient's path
    unsigned loaded;  // bitmask of loaded tensors
};

}  // namespace

struct ttl_ctx {
    ttl_config c;
    int D, F, H, T, E, L, P, S, G2, r, Kp, ldx, ldw, ldwt, nT, nS;  // nT: layers with adapters, nS: saved layers (layer_lo..L-1)
    // adapters per layer: tg = mask of TTL_LORA_*, ntg of them, nqkv among q/k/v at slots sq/sk/sv (-1 = none), has_o;
    // ext = K-extension columns of the QKV GEMMs (nqkv*r rounded up to 64), ldh = pitch of the bf16 stream-gradient buffers
    int tg, ntg, nqkv, sq, sk, sv, has_o, ext, ldh;
    int Mmax;
    ttl_ctx* parent = nullptr;              // != null: frozen weight images are the parent's (ttl_ctx_create_shared); read-only here
    std::atomic<int> refs{1};               // the owner's handle + one per context sharing this one's weight images: the memory
                                            // goes when the last of them is destroyed, whatever the order of the destroy calls
    int use_hm = 0; unsigned hm_magic = 0;   // head-major q/k/v from the big-M QKV GEMM (TTL_QKV_HEAD_MAJOR=0: row-major everywhere)
    float scaling;
    std::vector<void*> allocs;
    size_t bytes = 0;
    std::vector<Layer> layers;
    // embeddings / head
    op_t* wpatch;  // [D][Kp]
    float *cls, *pos, *preg, *preb, *postg, *postb;
    float *wp, *wpT;  // [E][D], [D][E]
    unsigned head_loaded = 0;
    // text tower (c.tower == TTL_TOWER_TEXT): token table, prompt ids, pooled (end-of-text) positions and the
    // compact copies of the pooled rows the top-layer backward reads; logits in [views, prompts] order
    int text = 0;
    float* tok = nullptr; int* ids = nullptr; int* pool = nullptr; int* poolrows = nullptr; int n_prompts = 0;
    float *hpool = nullptr, *hmid_g = nullptr, *mu2_g = nullptr, *rs2_g = nullptr; op_t* u_g = nullptr;
    float *logits_nk = nullptr, *dlogits_nk = nullptr, *dlogits_kn = nullptr;
    // peer features (image tower: class-text features; text tower: the image features of the views)
    float *tfeat, *tfeatT;
    int K = 0;
    float scale = 100.f;
    // lora binding
    float* lora_p = nullptr; float* lora_g = nullptr; size_t lora_n = 0;
    // shared activations
    op_t* patches; float* h;  // running residual stream
    std::vector<float*> h_out;   // outputs of the saved layers (h_out[i] = input of the next one)
    op_t *x1, *qkv, *attn, *x2, *g;
    float *cls_mean, *cls_rstd, *ycls, *feat, *logits, *dlogits, *head_te, *head_td;
    // backward scratch
    float *dh, *dh2, *dx; op_t *dh16, *dbig, *dattn, *dqkv;
    // top-layer backward works on the CLS rows only (compact [N, .] buffers)
    float *dcls, *dxc, *dhmc; op_t *dcls16, *dgc, *dhmc16, *doc;
    op_t* att_g;   // text tower, out_proj adapter in the top layer: the pooled rows of the saved attention output, compact [N][D+64]
    float* wg_partial;
    float* gemm_ws; size_t gemm_ws_bytes;
    float* loss_scratch; long long* idx_buf; int* n_buf; float* loss_buf; float* H_buf;
    // PLPD filter (deyo.py:115-151): keep mask + plpd values of the saving context; destroyed views + scratch of the context that runs
    // the second forward (allocated on first use); text mode: normalised features of the destroyed views and their logits
    unsigned char* keep_buf = nullptr; float* plpd_val = nullptr;
    float* plpd_x = nullptr; float* plpd_ws = nullptr; size_t plpd_ws_floats = 0;
    float *plpd_tf = nullptr, *plpd_tfT = nullptr, *plpd_lkn = nullptr, *plpd_lnk = nullptr;
    // GradScaler state (device) and policy (host): ttl.py:222, deyo.py:186-188
    ScalerState sc{nullptr, nullptr};
    float* sc_unit = nullptr;       // {1, 1, 1, 1}: what the backward reads instead of sc.f while `prescaled` is set
    bool prescaled = false;         // ttl_ctx_backward_prescaled: dlogits already carry the caller's loss scale (torch GradScaler on the autograd path)
    int sc_dynamic = 0; float sc_growth = 2.f, sc_backoff = 0.5f; int sc_interval = 2000;
    // Backward on the selected views only (top-k selections: deyo.py:105 filter_ent, ttl.py:52 TPT; TTL_BWD_COMPACT=0: off).  The loss
    // gradient is zero outside the int(n * rho) selected views, so their rows contribute nothing to any LoRA gradient: the saved
    // activations of the selected views are packed into these buffers (room for sel_cap views, sel_rows rows) and the same backward
    // runs with n = n_sel.  Image tower only.
    struct SelBuf { float *h_in, *h_mid, *lse, *mu1, *rs1, *mu2, *rs2; op_t *x1ext, *qkv, *attn, *u; };
    int sel_cap = 0, sel_rows = 0, cur_padded = 0;
    int concurrency = 1;      // episodes the caller keeps in flight on this GPU (ttl_ctx_set_concurrency): GemmArgs::concurrent
    std::vector<SelBuf> selb;                  // one per saved layer (layer_lo .. L-1)
    float *sel_mean = nullptr, *sel_rstd = nullptr, *sel_y = nullptr, *sel_f = nullptr, *sel_h = nullptr, *sel_dz = nullptr;
    bool saved = false; int saved_n = 0; int stream_views = 0;
    bool dry = false;               // sizing walk of ctx_create_impl (ttl_workspace_bytes): no HIP call is made
    bool saved_pooled = false;      // the saved forward ran its last layer on the pooled rows only (LN2 statistics of that layer: [n], not [n*T])
    // profiling
    bool prof = false; double prof_ms[TTL_NCLASS] = {}; long long prof_n[TTL_NCLASS] = {}; double gemm_flops = 0, gemm_flops_all = 0, gemm_flops_all_last = 0, gemm_bytes = 0, gemm_bytes_last = 0;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_events;
};

namespace {

template <typename Tp>
int dalloc(ttl_ctx* c, Tp** p, size_t count, bool zero = false) {
    void* q = nullptr;
    size_t bytes = count * sizeof(Tp);
    if (bytes == 0) bytes = 16;
    if (c->dry) { c->bytes += bytes; *p = nullptr; return 0; }     // ttl_workspace_bytes: the same walk, nothing allocated
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(TTL_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    if (zero) {
        e = hipMemset(q, 0, bytes);
        if (e != hipSuccess) return fail((int)e, "hipMemset failed: %s", hipGetErrorString(e));
    }
    c->allocs.push_back(q);
    c->bytes += bytes;
    *p = (Tp*)q;
    return 0;
}
#define ALLOC(ptr, count, zero)                        \
    do {                                               \
        int rc_ = dalloc(c, &(ptr), (size_t)(count), zero); \
        if (rc_) return rc_;                           \
    } while (0)

int check_config(const ttl_config* k) {
    if (!k) return fail(TTL_EINVAL, "null config");
    if (k->width < 128 || k->heads < 1 || k->mlp < 128 || k->layers < 1)      // (before any division by them)
        return fail(TTL_EINVAL, "width / heads / mlp / layers must be positive (got D=%d H=%d F=%d L=%d)", k->width, k->heads, k->mlp, k->layers);
    if (k->width % 128 || k->width > 1024 || k->width / k->heads != 64 || k->width % k->heads)
        return fail(TTL_EINVAL, "width must be a multiple of 128 with head_dim 64 (got D=%d H=%d)", k->width, k->heads);
    if (k->mlp % 128) return fail(TTL_EINVAL, "mlp must be a multiple of 128");
    if (k->rank != 16 && k->rank != 32) return fail(TTL_EINVAL, "rank must be 16 or 32 (got %d)", k->rank);
    if (k->tower != TTL_TOWER_IMAGE && k->tower != TTL_TOWER_TEXT) return fail(TTL_EINVAL, "unknown tower %d", k->tower);
    if (k->tower == TTL_TOWER_TEXT) {
        if (k->context_length < 2 || k->context_length > 128 || k->vocab_size < 2)
            return fail(TTL_EINVAL, "bad context_length %d / vocab_size %d", k->context_length, k->vocab_size);
    } else if (k->patch_size < 1 || k->image_size < k->patch_size || k->image_size % k->patch_size || k->image_size % 8)
        return fail(TTL_EINVAL, "bad image/patch size");
    if (k->layer_lo < 0 || k->layer_hi >= k->layers || k->layer_lo > k->layer_hi)
        return fail(TTL_EINVAL, "bad layer range [%d,%d] for %d layers", k->layer_lo, k->layer_hi, k->layers);
    int T = k->tower == TTL_TOWER_TEXT ? k->context_length : (k->image_size / k->patch_size) * (k->image_size / k->patch_size) + 1;
    if (T > 288) return fail(TTL_EINVAL, "token count %d > 288 unsupported", T);
    if (k->max_views < 1 || k->max_classes < 1 || k->embed < 4 || k->embed > 1024 || k->embed % 4) return fail(TTL_EINVAL, "bad capacities (embed must be a multiple of 4 in [4, 1024])");
    if (k->lora_targets < 0 || k->lora_targets > 15) return fail(TTL_EINVAL, "lora_targets must be a mask of TTL_LORA_Q|K|V|O (got %d)", k->lora_targets);
    return 0;
}

void set_geometry(ttl_ctx* c, const ttl_config* k) {
    c->c = *k;
    c->D = k->width; c->F = k->mlp; c->H = k->heads; c->E = k->embed; c->L = k->layers; c->P = k->patch_size;
    c->text = (k->tower == TTL_TOWER_TEXT);
    c->S = k->image_size; c->r = k->rank;
    if (c->text) { c->G2 = 0; c->T = k->context_length; c->Kp = 64; }
    else { c->G2 = (c->S / c->P) * (c->S / c->P); c->T = c->G2 + 1; c->Kp = round_up(3 * c->P * c->P, 64); }
    c->tg = k->lora_targets ? k->lora_targets : (TTL_LORA_Q | TTL_LORA_V);   // the reference's target_modules (custom_clip.py:586)
    c->sq = c->sk = c->sv = -1; c->nqkv = 0;
    if (c->tg & TTL_LORA_Q) c->sq = c->nqkv++;
    if (c->tg & TTL_LORA_K) c->sk = c->nqkv++;
    if (c->tg & TTL_LORA_V) c->sv = c->nqkv++;
    c->has_o = (c->tg & TTL_LORA_O) != 0;
    c->ntg = c->nqkv + c->has_o;
    c->ext = round_up(c->nqkv * c->r > 0 ? c->nqkv * c->r : 1, 64);
    c->ldx = c->D + c->ext;        // x1ext: D | nqkv*r LoRA cols | zero pad
    c->ldw = c->D + c->ext;        // wqkv rows
    c->ldwt = 3 * c->D + c->ext;   // wqkvT rows / dqkv rows
    c->ldh = c->D + (c->has_o ? 64 : 0);   // dh16 / dhmc16: D | r cols of dU_o | zero pad
    c->nT = k->layer_hi - k->layer_lo + 1;
    c->nS = k->layers - k->layer_lo;
    c->Mmax = round_up(k->max_views * c->T, 1280);  // padded: the big GEMM tiles store whole row tiles unguarded
    c->scaling = k->lora_alpha / (float)k->rank;
}

// ---- profiling helper: bracket a launch group with events on the stream
struct Prof {
    ttl_ctx* c; int cls; hipStream_t s; hipEvent_t a = nullptr, b = nullptr;
    Prof(ttl_ctx* c_, int cls_, hipStream_t s_) : c(c_), cls(cls_), s(s_) {
        if (c->prof) { (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, s); }
    }
    ~Prof() {
        if (c->prof) { (void)hipEventRecord(b, s); c->prof_events.push_back({cls, {a, b}}); }
    }
};

// Will the QKV projection `a` (EPI_OP, N = 3D) run on the big-M kernel?  Then its epilogue writes q/k/v head-major
// (contiguous [T][64] tiles per (view, plane, head)): the attention kernels read whole lines from adjacent addresses instead of
// 128-B segments at a stride of 3D elements.  Same values, same arithmetic: only the addresses differ.
bool qkv_goes_head_major(ttl_ctx* c, GemmArgs& a) {
    if (!c->use_hm || a.N != 3 * c->D) return false;
    GemmArgs p = a;
    p.padded = (a.M > c->c.max_views) ? c->Mmax : 0;
    p.hm_T = c->T; p.hm_magic = c->hm_magic;
    if (!gemm_takes_big(EPI_OP, p)) return false;
    a.hm_T = c->T; a.hm_magic = c->hm_magic;
    return true;
}

int gemm(ttl_ctx* c, GemmEpi epi, const GemmArgs& a0, hipStream_t s) {
    const bool big = a0.M >= 1024;   // launch_gemm's own split: the 160x128 kernel vs the latency-bound small-M path
    Prof p(c, big ? 0 : 6, s);
    GemmArgs a = a0;
    // token-row buffers ([n*T, .]) have Mmax = round_up(max_views*T, 1280) rows: whole row tiles may be stored unguarded.
    // The compact [n, .] buffers of the pooled last layer / CLS-only top-layer backward have exactly max_views rows:
    // guarded kernels, whatever n is.
    // (a backward on the packed views reads saved tensors that have sel_rows rows; its outputs are the full-size scratch buffers)
    a.padded = (a0.M > c->c.max_views) ? (c->cur_padded ? c->cur_padded : c->Mmax) : 0;
    a.ws = c->gemm_ws; a.ws_bytes = c->gemm_ws_bytes;
    a.concurrent = c->concurrency;
    if (c->prof) c->gemm_flops_all += 2.0 * a.M * a.N * a.K;
    if (c->prof && big) {
        c->gemm_flops += 2.0 * a.M * a.N * a.K;
        // algorithmic bytes of the launch: both operands once, every output once, the fused epilogue inputs once
        const double mn = (double)a.M * a.N;
        const bool f32out = (epi == EPI_F32 || epi == EPI_RESID_F32 || epi == EPI_PATCH);
        c->gemm_bytes += 2.0 * ((double)a.M * a.K + (double)a.N * a.K) + mn * (f32out ? 4 : 2) +
                         (epi == EPI_RESID_F32 ? 4 * mn : 0) + (epi == EPI_GELU_BWD ? 2 * mn : 0) +
                         (epi == EPI_GELU && a.C2 ? 2 * mn : 0);
    }
    HIP_TRY(launch_gemm(epi, a, s));
    return 0;
}

enum { W_QW = 1, W_QB = 2, W_KW = 4, W_KB = 8, W_VW = 16, W_VB = 32, W_OW = 64, W_OB = 128, W_1W = 256, W_1B = 512,
       W_2W = 1024, W_2B = 2048, W_L1G = 4096, W_L1B = 8192, W_L2G = 16384, W_L2B = 32768, W_ALL = 65535 };
enum { HW_CLS = 1, HW_PATCH = 2, HW_POS = 4, HW_PREG = 8, HW_PREB = 16, HW_POSTG = 32, HW_POSTB = 64, HW_PROJ = 128, HW_ALL = 255,
       HW_TOK = 256, HW_TEXT_ALL = HW_TOK | HW_POS | HW_POSTG | HW_POSTB | HW_PROJ };

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

const char* ttl_last_error(void) { return g_err; }
#ifdef TTL_EXPERIMENTS
const char* ttl_version(void) { return "ttl_hip 0.3 (gfx950, " TTL_OPERAND_NAME " operands, EXPERIMENTS build: closed A/B switches are read from the environment)"; }
#else
const char* ttl_version(void) { return "ttl_hip 0.3 (gfx950, " TTL_OPERAND_NAME " operands)"; }
#endif

// ---- the run-time switches of the product build (csrc/common.hpp TtlSwitch): name, default, what it selects
namespace {
struct SwitchInfo { const char* name; int dflt; const char* what; };
const SwitchInfo kSwitches[SW_COUNT] = {
    {"TTL_GEMM_HUGE", 2, "256x256 four-wave GEMM tiles: 0 off, 1 q/k/v + fc1, 2 q/k/v only, 3 fc1 only (read once per process)"},
    {"TTL_GEMM_HUGE_NARROW", -1, "N = D launches on the 256x256 tiles: -1 when the context runs beside other episodes, 0 never, 1 always (once per process)"},
    {"TTL_GEMM_HUGE_MIN_FILL", 85, "percent of one round of CUs a launch must fill with 256x256 tiles (once per process)"},
    {"TTL_BWD_COMPACT", 1, "top-k selections: backward on the selected views' packed activations (read when a context is created)"},
    {"TTL_CONCURRENCY", 0, "default of ttl_ctx_set_concurrency for new contexts; 0 = 1 (read when a context is created)"},
};
}  // namespace

extern "C++" int ttl_switch(TtlSwitch id) { return ttl_env_int(kSwitches[id].name, kSwitches[id].dflt); }

const char* ttl_runtime_switches(void) {
    static thread_local char buf[1024];
    size_t off = 0;
    for (int i = 0; i < SW_COUNT && off < sizeof buf; ++i)
        off += (size_t)snprintf(buf + off, sizeof buf - off, "%s=%d default=%d # %s\n", kSwitches[i].name, ttl_switch((TtlSwitch)i),
                                kSwitches[i].dflt, kSwitches[i].what);
    return buf;
}
const char* ttl_operand_dtype(void) { return TTL_OPERAND_NAME; }

static int ctx_create_impl(const ttl_config* k, ttl_ctx* parent, ttl_ctx** out, bool dry = false);
static void ctx_release(ttl_ctx* c);

// The footprint of an OWNING context, exactly: the allocation walk of ttl_ctx_create with nothing allocated (no HIP call: works without
// a GPU).  ttl_ctx_allocated_bytes() of a context created with the same configuration (and the same TTL_BWD_COMPACT) returns this
// figure until a PLPD stage adds its lazily allocated buffers (documented in include/ttl_hip.h).
size_t ttl_workspace_bytes(const ttl_config* k) {
    ttl_ctx* c = nullptr;
    if (ctx_create_impl(k, nullptr, &c, true) || !c) return 0;
    const size_t b = c->bytes;
    ctx_release(c);
    return b;
}

size_t ttl_ctx_allocated_bytes(const ttl_ctx* c) { return c ? c->bytes : 0; }

// The weight-side configuration two contexts must agree on to share frozen images (capacities may differ)
static bool same_model(const ttl_config& a, const ttl_config& b) {
    return a.image_size == b.image_size && a.patch_size == b.patch_size && a.width == b.width && a.heads == b.heads && a.mlp == b.mlp &&
           a.layers == b.layers && a.embed == b.embed && a.rank == b.rank && a.layer_lo == b.layer_lo && a.layer_hi == b.layer_hi &&
           a.tower == b.tower && a.context_length == b.context_length && a.vocab_size == b.vocab_size &&
           (a.lora_targets ? a.lora_targets : (TTL_LORA_Q | TTL_LORA_V)) == (b.lora_targets ? b.lora_targets : (TTL_LORA_Q | TTL_LORA_V));
}

static int ctx_create_impl(const ttl_config* k, ttl_ctx* parent, ttl_ctx** out, bool dry) {
    if (!out) return fail(TTL_EINVAL, "null out");
    *out = nullptr;
    int rc = check_config(k);
    if (rc) return rc;
    if (parent) {
        if (parent->parent) return fail(TTL_EINVAL, "the parent of a shared context must own its weights");
        if (!same_model(*k, parent->c)) return fail(TTL_EINVAL, "a shared context needs the parent's model configuration (capacities may differ)");
        if ((rc = ttl_weights_ready(parent))) return rc;
    }
    if (!dry) {
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) return fail(e != hipSuccess ? (int)e : TTL_ESTATE, "no HIP device available (%s)", hipGetErrorString(e));
    }
    ttl_ctx* c = new ttl_ctx();
    c->dry = dry;
    set_geometry(c, k);
    c->parent = parent;
    if (parent) parent->refs.fetch_add(1);
    const size_t D = c->D, F = c->F, M = c->Mmax, E = c->E, N = k->max_views, T = c->T, H = c->H, r = c->r;
    // frozen tensor: the parent's image (never written after loading) or an allocation of this context's own
#define WSHARE(dst, src, count, zero) do { if (parent) (dst) = (src); else ALLOC(dst, count, zero); } while (0)
    struct Guard { ttl_ctx* c; bool ok = false; ~Guard() { if (!ok) { if (c->dry) ctx_release(c); else ttl_ctx_destroy(c); } } } guard{c};
    {   // head-major q/k/v: needs row / T by multiply-high for every row a big-M launch can store
        c->hm_magic = qkv_hm_magic((int)T, (int)M + 320);
        c->use_hm = TTL_EXPERIMENT("TTL_QKV_HEAD_MAJOR", 1) != 0 && c->hm_magic != 0;
    }
    {   // TTL_CONCURRENCY: the default of ttl_ctx_set_concurrency for every context of the process (profiling a single stream with the
        // kernels of the three-stream run)
        const int v = ttl_switch(SW_CONCURRENCY);
        if (v >= 1) c->concurrency = v;
    }
    c->layers.resize(c->L);
    for (int i = 0; i < c->L; ++i) {
        Layer& l = c->layers[i];
        memset(&l, 0, sizeof l);
        l.trained = (i >= k->layer_lo);
        l.lora = (i >= k->layer_lo && i <= k->layer_hi);
        l.ldwo = l.ldat = (int)D + ((l.lora && c->has_o) ? 64 : 0);
        const Layer* pl = parent ? &parent->layers[i] : nullptr;
        // the projection images of a layer WITH adapters carry the LoRA K-extension columns, which every context refreshes from
        // its own adapters: private, initialised from the parent's below.  Everything else of a layer is frozen for good.
        const bool own_proj = l.lora || !parent;
        if (own_proj) { ALLOC(l.wqkv, 3 * D * c->ldw, true); ALLOC(l.wo, D * l.ldwo, true); }
        else { l.wqkv = pl->wqkv; l.wo = pl->wo; }
        WSHARE(l.bqkv, pl->bqkv, 3 * D, true); WSHARE(l.bo, pl->bo, D, true);
        WSHARE(l.w1, pl->w1, F * D, false); WSHARE(l.b1, pl->b1, F, true);
        WSHARE(l.w2, pl->w2, D * F, false); WSHARE(l.b2, pl->b2, D, true);
        WSHARE(l.ln1g, pl->ln1g, D, true); WSHARE(l.ln1b, pl->ln1b, D, true); WSHARE(l.ln2g, pl->ln2g, D, true); WSHARE(l.ln2b, pl->ln2b, D, true);
        if (l.trained) {
            if (own_proj) { ALLOC(l.wqkvT, D * c->ldwt, true); ALLOC(l.woT, D * l.ldwo, true); }
            else { l.wqkvT = pl->wqkvT; l.woT = pl->woT; }
            WSHARE(l.w1T, pl->w1T, D * F, false); WSHARE(l.w2T, pl->w2T, F * D, false);
            if (parent && l.lora) {
                HIP_TRY(hipMemcpy(l.wqkv, pl->wqkv, 3 * D * c->ldw * sizeof(op_t), hipMemcpyDeviceToDevice));
                HIP_TRY(hipMemcpy(l.wo, pl->wo, D * l.ldwo * sizeof(op_t), hipMemcpyDeviceToDevice));
                HIP_TRY(hipMemcpy(l.wqkvT, pl->wqkvT, D * c->ldwt * sizeof(op_t), hipMemcpyDeviceToDevice));
                HIP_TRY(hipMemcpy(l.woT, pl->woT, D * l.ldwo * sizeof(op_t), hipMemcpyDeviceToDevice));
            }
            ALLOC(l.acat, 3 * r * D, true); ALLOC(l.btcat, 3 * r * D, true); ALLOC(l.acat_o, r * D, true); ALLOC(l.btcat_o, r * D, true);
            ALLOC(l.h_mid, M * D, false);  // h_in is a pointer into the stream buffers
            ALLOC(l.x1ext, M * c->ldx, true); ALLOC(l.qkv, (M + T) * 3 * D, false); ALLOC(l.attn, M * l.ldat, true); ALLOC(l.u, M * F, false);
            ALLOC(l.lse, N * H * T, false);
            ALLOC(l.mu1, M, false); ALLOC(l.rs1, M, false); ALLOC(l.mu2, M, false); ALLOC(l.rs2, M, false);
        }
    }
    if (parent) for (int i = 0; i < c->L; ++i) c->layers[i].loaded = W_ALL;
    WSHARE(c->wpatch, parent->wpatch, D * c->Kp, true);
    if (c->text) {
        WSHARE(c->tok, parent->tok, (size_t)k->vocab_size * D, false);
        ALLOC(c->ids, N * T, true); ALLOC(c->pool, N, true); ALLOC(c->poolrows, N, true);
        ALLOC(c->hpool, N * D, true); ALLOC(c->hmid_g, N * D, false); ALLOC(c->mu2_g, N, false); ALLOC(c->rs2_g, N, false);
        ALLOC(c->u_g, N * F, false);
        ALLOC(c->logits_nk, N * k->max_classes, false); ALLOC(c->dlogits_nk, N * k->max_classes, false);
        ALLOC(c->dlogits_kn, N * k->max_classes, false);
    }
    WSHARE(c->cls, parent->cls, D, true); WSHARE(c->pos, parent->pos, T * D, true);
    WSHARE(c->preg, parent->preg, D, true); WSHARE(c->preb, parent->preb, D, true); WSHARE(c->postg, parent->postg, D, true); WSHARE(c->postb, parent->postb, D, true);
    WSHARE(c->wp, parent->wp, E * D, true); WSHARE(c->wpT, parent->wpT, D * E, true);
    if (parent) c->head_loaded = parent->head_loaded;
#undef WSHARE
    ALLOC(c->tfeat, (size_t)k->max_classes * E, true); ALLOC(c->tfeatT, (size_t)k->max_classes * E, true);
    ALLOC(c->patches, N * c->G2 * c->Kp, true);
    ALLOC(c->h, M * D, false);
    c->h_out.assign(c->nS, nullptr);
    for (int i = 0; i < c->nS; ++i) ALLOC(c->h_out[i], M * D, false);
    // (q/k/v buffers: + T rows, the head-major image of the padding rows of the last row tile spills into one more view's block)
    ALLOC(c->x1, M * D, false); ALLOC(c->qkv, (M + T) * 3 * D, false); ALLOC(c->attn, M * D, false);
    ALLOC(c->x2, M * D, false); ALLOC(c->g, M * F, false);
    ALLOC(c->cls_mean, N, false); ALLOC(c->cls_rstd, N, false); ALLOC(c->ycls, N * D, false); ALLOC(c->feat, N * E, false);
    ALLOC(c->logits, N * k->max_classes, false); ALLOC(c->dlogits, N * k->max_classes, false);
    ALLOC(c->head_te, N * E, false); ALLOC(c->head_td, N * D, false);
    ALLOC(c->dh, M * D, false); ALLOC(c->dh2, M * D, false); ALLOC(c->dx, M * D, false);
    ALLOC(c->dh16, M * c->ldh, true); ALLOC(c->dbig, M * F, false); ALLOC(c->dattn, M * (D + 64), false);   // (dattn shares the saved attention output's pitch)
    ALLOC(c->dqkv, M * c->ldwt, true);
    ALLOC(c->dcls, N * D, false); ALLOC(c->dxc, N * D, false); ALLOC(c->dhmc, N * D, false);
    ALLOC(c->dcls16, N * D, false); ALLOC(c->dgc, N * F, false); ALLOC(c->dhmc16, N * c->ldh, true); ALLOC(c->doc, N * D, false);
    ALLOC(c->att_g, N * (D + 64), true);
    {
        const int cap = (int)N / 4;          // top-k selections keep 10 % of the views (ttl.py:376); a quarter of them fit
        if (ttl_switch(SW_BWD_COMPACT) != 0 && !c->text && cap >= 1 && c->nS > 0) {
            c->sel_cap = cap;
            c->sel_rows = round_up(cap * (int)T, 1280);
            const size_t Ms = (size_t)c->sel_rows;
            c->selb.resize(c->nS);
            for (int i = 0; i < c->nS; ++i) {
                ttl_ctx::SelBuf& b = c->selb[i];
                const Layer& l = c->layers[k->layer_lo + i];
                ALLOC(b.h_in, Ms * D, true); ALLOC(b.h_mid, Ms * D, true); ALLOC(b.x1ext, Ms * c->ldx, true);
                ALLOC(b.qkv, (Ms + T) * 3 * D, true); ALLOC(b.attn, Ms * l.ldat, true); ALLOC(b.u, Ms * F, true);
                ALLOC(b.lse, (size_t)cap * H * T, true);
                ALLOC(b.mu1, Ms, true); ALLOC(b.rs1, Ms, true); ALLOC(b.mu2, Ms, true); ALLOC(b.rs2, Ms, true);
            }
            ALLOC(c->sel_mean, cap, true); ALLOC(c->sel_rstd, cap, true); ALLOC(c->sel_y, (size_t)cap * D, true);
            ALLOC(c->sel_f, (size_t)cap * E, true); ALLOC(c->sel_h, (size_t)cap * D, true);
            ALLOC(c->sel_dz, (size_t)cap * k->max_classes, true);
        }
    }
    ALLOC(c->wg_partial, (size_t)lora_wgrad_chunks((int)M) * 2 * c->ntg * r * D, false);
    c->gemm_ws_bytes = (size_t)8 << 20;
    ALLOC(c->gemm_ws, c->gemm_ws_bytes / sizeof(float), false);
    // (the text tower runs the loss on [views, prompts] logits: either count can be the larger one)
    const size_t nmax = N > (size_t)k->max_classes ? N : (size_t)k->max_classes;
    ALLOC(c->loss_scratch, 7 * nmax + 16, true);
    ALLOC(c->idx_buf, nmax, true); ALLOC(c->n_buf, 4, true); ALLOC(c->loss_buf, 4, true); ALLOC(c->H_buf, nmax, true);
    ALLOC(c->keep_buf, nmax, true); ALLOC(c->plpd_val, nmax, true);
    ALLOC(c->sc.f, SC_NF, true); ALLOC(c->sc.i, SC_NI, true); ALLOC(c->sc_unit, SC_NF, true);
    {   // fp16-operand build: dynamic loss scaling from 2^10 like the reference's GradScaler(init_scale=1000) (ttl.py:222);
        // bf16 needs no loss scale (scale 1, fixed) but keeps the whole-step skip on non-finite gradients
        const float init[SC_NF] = {TTL_GRAD_SCALE, 1.0f / TTL_GRAD_SCALE, 1.f, 1.f};
        const float unit[SC_NF] = {1.f, 1.f, 1.f, 1.f};
        if (!dry) HIP_TRY(hipMemcpy(c->sc.f, init, sizeof init, hipMemcpyHostToDevice));
        if (!dry) HIP_TRY(hipMemcpy(c->sc_unit, unit, sizeof unit, hipMemcpyHostToDevice));
        c->sc_dynamic = (TTL_GRAD_SCALE != 1.0f);
    }
    guard.ok = true;
    *out = c;
    return 0;
}

int ttl_ctx_create(const ttl_config* k, ttl_ctx** out) { return ctx_create_impl(k, nullptr, out); }

int ttl_ctx_create_shared(const ttl_config* k, ttl_ctx* parent, ttl_ctx** out) {
    if (!parent) return fail(TTL_EINVAL, "null parent");
    return ctx_create_impl(k, parent, out);
}

static void ctx_release(ttl_ctx* c) {
    if (c->refs.fetch_sub(1) != 1) return;          // contexts sharing its weight images are still alive
    ttl_ctx* parent = c->parent;
    for (auto& pe : c->prof_events) { (void)hipEventDestroy(pe.second.first); (void)hipEventDestroy(pe.second.second); }
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
    if (parent) ctx_release(parent);
}

void ttl_ctx_destroy(ttl_ctx* c) {
    if (!c) return;
    (void)hipDeviceSynchronize();
    ctx_release(c);
}

// ------------------------------------------------------------------------------ weights
static int upload(ttl_ctx* c, const float* data, size_t count, float** tmp) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(float));
    if (e != hipSuccess) return fail(TTL_ENOMEM, "staging hipMalloc failed: %s", hipGetErrorString(e));
    e = hipMemcpy(q, data, count * sizeof(float), hipMemcpyDefault);
    if (e != hipSuccess) { (void)hipFree(q); return fail((int)e, "weight copy failed: %s", hipGetErrorString(e)); }
    *tmp = (float*)q;
    return 0;
}

// IEEE half -> float, exact (subnormals, inf, nan)
static float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t ex = (h >> 10) & 31u, man = h & 1023u, bits;
    if (ex == 0) {
        if (!man) bits = sign;
        else {
            int sh = 0;
            while (!(man & 1024u)) { man <<= 1; ++sh; }
            bits = sign | ((uint32_t)(113 - sh) << 23) | ((man & 1023u) << 13);
        }
    } else if (ex == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((ex + 112u) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

int ttl_load_weight_typed(ttl_ctx* c, const char* name, const void* data, size_t count, int dtype) {
    if (dtype == TTL_DTYPE_F32) return ttl_load_weight(c, name, (const float*)data, count);
    if (dtype != TTL_DTYPE_F16 && dtype != TTL_DTYPE_BF16) return fail(TTL_EINVAL, "%s: unknown dtype %d", name ? name : "?", dtype);
    if (!c || !name || !data) return fail(TTL_EINVAL, "null argument");
    std::vector<uint16_t> raw(count);
    HIP_TRY(hipMemcpy(raw.data(), data, count * sizeof(uint16_t), hipMemcpyDefault));     // host or device source
    std::vector<float> wide(count);
    for (size_t i = 0; i < count; ++i) {
        if (dtype == TTL_DTYPE_BF16) { const uint32_t b = (uint32_t)raw[i] << 16; memcpy(&wide[i], &b, 4); }
        else wide[i] = half_bits_to_float(raw[i]);
    }
    return ttl_load_weight(c, name, wide.data(), count);
}

int ttl_load_weight(ttl_ctx* c, const char* name, const float* data, size_t count) {
    if (!c || !name || !data) return fail(TTL_EINVAL, "null argument");
    if (c->parent) return fail(TTL_ESTATE, "%s: this context shares its parent's weights (ttl_ctx_create_shared); load them into the parent", name);
    // sharing freezes the owner's weights: the sharers read its frozen images in place and hold private COPIES of the projection
    // images of the adapter layers, taken at creation — a later load would leave one model with tensors of two generations
    if (c->refs.load() > 1)
        return fail(TTL_ESTATE, "%s: %d context(s) share this context's weight images (ttl_ctx_create_shared); destroy them before loading weights",
                    name, c->refs.load() - 1);
    const size_t D = c->D, F = c->F, E = c->E, T = c->T;
    float* tmp = nullptr;
    int rc = 0;
    hipStream_t s = nullptr;
#define NEED(n)                                                                                         \
    if (count != (size_t)(n)) return fail(TTL_EINVAL, "%s: expected %zu elements, got %zu", name, (size_t)(n), count)
#define F32COPY(dst) HIP_TRY(hipMemcpy((dst), data, count * sizeof(float), hipMemcpyDefault))
    std::string nm(name);
    int li = -1;
    const char* lp = strstr(name, "encoder.layers.");
    if (lp) {
        li = atoi(lp + 15);
        if (li < 0 || li >= c->L) return fail(TTL_EINVAL, "%s: layer out of range", name);
        Layer& l = c->layers[li];
        const char* tail = strchr(lp + 15, '.');
        if (!tail) return fail(TTL_EINVAL, "bad name %s", name);
        std::string t(tail + 1);
        auto proj_w = [&](int which, unsigned bit) -> int {  // q/k/v weight [D][D] -> rows which*D.. of wqkv (+ transposed image)
            NEED(D * D);
            if ((rc = upload(c, data, count, &tmp))) return rc;
            hipError_t e = launch_cast_rows_f32_op(tmp, (int)D, (int)D, l.wqkv + (size_t)which * D * c->ldw, c->ldw, s);
            if (e == hipSuccess && l.trained) e = launch_transpose_f32_op(tmp, (int)D, (int)D, l.wqkvT + (size_t)which * D, c->ldwt, s);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            (void)hipFree(tmp);
            if (e != hipSuccess) return fail((int)e, "%s: %s", name, hipGetErrorString(e));
            l.loaded |= bit;
            return 0;
        };
        auto plain_w = [&](op_t* dst, op_t* dstT, size_t rows, size_t cols, unsigned bit, int ld = 0, int ldT = 0) -> int {
            NEED(rows * cols);
            if ((rc = upload(c, data, count, &tmp))) return rc;
            hipError_t e = ld ? launch_cast_rows_f32_op(tmp, (int)rows, (int)cols, dst, ld, s) : launch_cast_f32_op(tmp, dst, count, s);
            if (e == hipSuccess && dstT) e = launch_transpose_f32_op(tmp, (int)rows, (int)cols, dstT, ldT ? ldT : (int)rows, s);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            (void)hipFree(tmp);
            if (e != hipSuccess) return fail((int)e, "%s: %s", name, hipGetErrorString(e));
            l.loaded |= bit;
            return 0;
        };
        if (t == "self_attn.q_proj.weight") return proj_w(0, W_QW);
        if (t == "self_attn.k_proj.weight") return proj_w(1, W_KW);
        if (t == "self_attn.v_proj.weight") return proj_w(2, W_VW);
        if (t == "self_attn.q_proj.bias") { NEED(D); F32COPY(l.bqkv); l.loaded |= W_QB; return 0; }
        if (t == "self_attn.k_proj.bias") { NEED(D); F32COPY(l.bqkv + D); l.loaded |= W_KB; return 0; }
        if (t == "self_attn.v_proj.bias") { NEED(D); F32COPY(l.bqkv + 2 * D); l.loaded |= W_VB; return 0; }
        if (t == "self_attn.out_proj.weight") return plain_w(l.wo, l.trained ? l.woT : nullptr, D, D, W_OW, l.ldwo, l.ldwo);
        if (t == "self_attn.out_proj.bias") { NEED(D); F32COPY(l.bo); l.loaded |= W_OB; return 0; }
        if (t == "mlp.fc1.weight") return plain_w(l.w1, l.trained ? l.w1T : nullptr, F, D, W_1W);
        if (t == "mlp.fc1.bias") { NEED(F); F32COPY(l.b1); l.loaded |= W_1B; return 0; }
        if (t == "mlp.fc2.weight") return plain_w(l.w2, l.trained ? l.w2T : nullptr, D, F, W_2W);
        if (t == "mlp.fc2.bias") { NEED(D); F32COPY(l.b2); l.loaded |= W_2B; return 0; }
        if (t == "layer_norm1.weight") { NEED(D); F32COPY(l.ln1g); l.loaded |= W_L1G; return 0; }
        if (t == "layer_norm1.bias") { NEED(D); F32COPY(l.ln1b); l.loaded |= W_L1B; return 0; }
        if (t == "layer_norm2.weight") { NEED(D); F32COPY(l.ln2g); l.loaded |= W_L2G; return 0; }
        if (t == "layer_norm2.bias") { NEED(D); F32COPY(l.ln2b); l.loaded |= W_L2B; return 0; }
        return fail(TTL_EINVAL, "unknown layer tensor %s", name);
    }
    if (c->text) {
        if (nm == "text_model.embeddings.token_embedding.weight") { NEED((size_t)c->c.vocab_size * D); F32COPY(c->tok); c->head_loaded |= HW_TOK; return 0; }
        if (nm == "text_model.embeddings.position_embedding.weight") { NEED(T * D); F32COPY(c->pos); c->head_loaded |= HW_POS; return 0; }
        if (nm == "text_model.final_layer_norm.weight") { NEED(D); F32COPY(c->postg); c->head_loaded |= HW_POSTG; return 0; }
        if (nm == "text_model.final_layer_norm.bias") { NEED(D); F32COPY(c->postb); c->head_loaded |= HW_POSTB; return 0; }
        if (nm != "text_projection.weight") return fail(TTL_EINVAL, "unknown text-tower tensor %s", name);
    }
    if (nm == "vision_model.embeddings.class_embedding") { NEED(D); F32COPY(c->cls); c->head_loaded |= HW_CLS; return 0; }
    if (nm == "vision_model.embeddings.position_embedding.weight") { NEED(T * D); F32COPY(c->pos); c->head_loaded |= HW_POS; return 0; }
    if (nm == "vision_model.embeddings.patch_embedding.weight") {
        size_t kk = 3 * (size_t)c->P * c->P;
        NEED(D * kk);
        if ((rc = upload(c, data, count, &tmp))) return rc;
        hipError_t e = launch_cast_rows_f32_op(tmp, (int)D, (int)kk, c->wpatch, c->Kp, s);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        (void)hipFree(tmp);
        if (e != hipSuccess) return fail((int)e, "%s: %s", name, hipGetErrorString(e));
        c->head_loaded |= HW_PATCH;
        return 0;
    }
    if (nm == "vision_model.pre_layrnorm.weight") { NEED(D); F32COPY(c->preg); c->head_loaded |= HW_PREG; return 0; }
    if (nm == "vision_model.pre_layrnorm.bias") { NEED(D); F32COPY(c->preb); c->head_loaded |= HW_PREB; return 0; }
    if (nm == "vision_model.post_layernorm.weight") { NEED(D); F32COPY(c->postg); c->head_loaded |= HW_POSTG; return 0; }
    if (nm == "vision_model.post_layernorm.bias") { NEED(D); F32COPY(c->postb); c->head_loaded |= HW_POSTB; return 0; }
    if (nm == (c->text ? "text_projection.weight" : "visual_projection.weight")) {
        NEED(E * D);
        F32COPY(c->wp);
        std::vector<float> host(count), tr(count);
        HIP_TRY(hipMemcpy(host.data(), c->wp, count * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t e = 0; e < E; ++e)
            for (size_t d = 0; d < D; ++d) tr[d * E + e] = host[e * D + d];
        HIP_TRY(hipMemcpy(c->wpT, tr.data(), count * sizeof(float), hipMemcpyHostToDevice));
        c->head_loaded |= HW_PROJ;
        return 0;
    }
    return fail(TTL_EINVAL, "unknown tensor %s", name);
#undef NEED
#undef F32COPY
}

int ttl_weights_ready(ttl_ctx* c) {
    if (!c) return fail(TTL_EINVAL, "null ctx");
    const unsigned want = c->text ? (unsigned)HW_TEXT_ALL : (unsigned)HW_ALL;
    if (c->head_loaded != want) return fail(TTL_ESTATE, "embedding/head tensors missing (mask 0x%x of 0x%x)", c->head_loaded, want);
    for (int i = 0; i < c->L; ++i)
        if (c->layers[i].loaded != W_ALL) return fail(TTL_ESTATE, "layer %d tensors missing (mask 0x%x)", i, c->layers[i].loaded);
    return 0;
}

// peer features [K,E] (+ transposed copy [E,K] for coalesced logits), optionally L2-normalised on the way in
static int set_peer_features(ttl_ctx* c, const float* feats, int K, int normalize, float scale, hipStream_t s) {
    if (K < 1 || K > c->c.max_classes) return fail(TTL_EINVAL, "feature rows %d outside [1,%d]", K, c->c.max_classes);
    // tfeat doubles as the staging buffer: copy in, then normalise / transpose out of it in place (row-wise safe)
    HIP_TRY(hipMemcpyAsync(c->tfeat, feats, (size_t)K * c->E * sizeof(float), hipMemcpyDefault, s));
    HIP_TRY(launch_unit_rows(c->tfeat, K, c->E, normalize, c->tfeat, c->tfeatT, s));
    c->K = K;
    c->scale = scale;
    return 0;
}

int ttl_set_text_features(ttl_ctx* c, const float* tfeat, int K, float scale, void* stream) {
    if (!c || !tfeat) return fail(TTL_EINVAL, "null argument");
    if (c->text) return fail(TTL_ESTATE, "ttl_set_text_features on a text-tower context (use ttl_set_image_features)");
    return set_peer_features(c, tfeat, K, 0, scale, (hipStream_t)stream);
}

int ttl_set_image_features(ttl_ctx* c, const float* feats, int n_views, int normalize, float scale, void* stream) {
    if (!c || !feats) return fail(TTL_EINVAL, "null argument");
    if (!c->text) return fail(TTL_ESTATE, "ttl_set_image_features needs a text-tower context");
    return set_peer_features(c, feats, n_views, normalize, scale, (hipStream_t)stream);
}

int ttl_set_logit_scale(ttl_ctx* c, float logit_scale_exp) {
    if (!c) return fail(TTL_EINVAL, "null ctx");
    c->scale = logit_scale_exp;
    return 0;
}

int ttl_set_prompts(ttl_ctx* c, const int* ids, int n_prompts, void* stream) {
    if (!c || !ids) return fail(TTL_EINVAL, "null argument");
    if (!c->text) return fail(TTL_ESTATE, "ttl_set_prompts needs a text-tower context");
    if (n_prompts < 1 || n_prompts > c->c.max_views) return fail(TTL_EINVAL, "n_prompts %d outside [1,%d]", n_prompts, c->c.max_views);
    hipStream_t s = (hipStream_t)stream;
    const size_t cnt = (size_t)n_prompts * c->T;
    std::vector<int> host(cnt), pool(n_prompts), rows(n_prompts);
    HIP_TRY(hipMemcpy(host.data(), ids, cnt * sizeof(int), hipMemcpyDefault));
    for (int p = 0; p < n_prompts; ++p) {
        // pooled position = argmax of the ids (first maximum), HF CLIPTextTransformer with eos_token_id == 2
        int best = 0;
        for (int t = 0; t < c->T; ++t) {
            int v = host[(size_t)p * c->T + t];
            if (v < 0 || v >= c->c.vocab_size) return fail(TTL_EINVAL, "token id %d outside the vocabulary (prompt %d, position %d)", v, p, t);
            if (v > host[(size_t)p * c->T + best]) best = t;
        }
        pool[p] = best;
        rows[p] = p * c->T + best;   // row of the pooled token in the [n*T, .] activation buffers
    }
    HIP_TRY(hipMemcpyAsync(c->ids, host.data(), cnt * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->pool, pool.data(), n_prompts * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(c->poolrows, rows.data(), n_prompts * sizeof(int), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));   // the host vectors go out of scope
    c->n_prompts = n_prompts;
    return 0;
}

int ttl_bind_lora(ttl_ctx* c, float* params, float* grads, size_t n) {
    if (!c || !params || !grads) return fail(TTL_EINVAL, "null argument");
    size_t want = (size_t)c->nT * c->ntg * 2 * c->r * c->D;
    if (n != want) return fail(TTL_EINVAL, "lora buffer has %zu elements, geometry needs %zu", n, want);
    c->lora_p = params; c->lora_g = grads; c->lora_n = n;
    return 0;
}

// ------------------------------------------------------------------------------ forward
static HeadArgs head_args(ttl_ctx* c, const float* h, float* feats_out, float* logits) {
    HeadArgs a;
    a.h = c->text ? c->hpool : h; a.T = c->text ? 1 : c->T; a.D = c->D; a.E = c->E; a.K = c->K;
    a.ln_g = c->postg; a.ln_b = c->postb; a.eps = c->c.ln_eps;
    a.WpT = c->wpT; a.Wp = c->wp; a.tfeat = c->tfeat; a.tfeatT = c->tfeatT; a.scale = c->scale;
    a.cls_mean = c->cls_mean; a.cls_rstd = c->cls_rstd; a.y = c->ycls; a.f = c->feat; a.logits = logits; a.feats_out = feats_out;
    a.tmp_e = c->head_te; a.tmp_d = c->head_td;
    a.gscale = c->prescaled ? c->sc_unit : c->sc.f;
    return a;
}

// where the fused reset / optimizer launches write the operand-dtype images of the bound LoRA parameters (lora_refresh's outputs)
static bool lora_images(ttl_ctx* c, LoraImages* im) {
    memset(im, 0, sizeof *im);
    if (!c->lora_p || c->nT > LORA_IMG_MAX_LAYERS) return false;
    im->layers = c->nT; im->ntg = c->ntg; im->D = c->D; im->r = c->r; im->ldw = c->ldw; im->ldwt = c->ldwt;
    int k = 0;
    for (int t = 0; t < 4; ++t)
        if (c->tg & (1 << t)) im->proj[k++] = t;
    for (int i = 0; i < c->nT; ++i) {
        Layer& l = c->layers[c->c.layer_lo + i];
        im->ldwo = l.ldwo;
        im->L[i] = {l.wqkv, l.wqkvT, l.acat, l.btcat, l.wo, l.woT, l.acat_o, l.btcat_o};
    }
    return true;
}

static int lora_refresh(ttl_ctx* c, hipStream_t s) {
    if (!c->lora_p) return 0;
    Prof p(c, 4, s);
    const size_t per = (size_t)c->r * c->D;
    for (int i = 0; i < c->nT; ++i) {
        Layer& l = c->layers[c->c.layer_lo + i];
        const float* base = c->lora_p + (size_t)i * c->ntg * 2 * per;
        LoraPtrs P = {};
        int k = 0;
        for (int t = 0; t < 4; ++t)
            if (c->tg & (1 << t)) { P.A[t] = base + (size_t)k * 2 * per; P.B[t] = P.A[t] + per; ++k; }
        HIP_TRY(launch_lora_refresh(P, c->D, c->r, l.wqkv, c->ldw, l.wqkvT, c->ldwt, l.acat, l.btcat, l.wo, l.woT, l.ldwo, l.acat_o,
                                    l.btcat_o, s));
    }
    return 0;
}

// from_layer == 0: the whole tower.  from_layer == layer_lo: resume from the residual stream the
// last full forward left at the input of the first trained layer (c->h, untouched since): the
// frozen layers below it do not depend on the LoRA parameters, so for the same views the result
// is identical — used for the later updates of a multi-step episode and for the adapted
// inference on view 0 (rows 0..T-1 of that buffer).
// images_fresh: the caller's last launch on this stream (the fused episode's reset / optimizer) has already written the LoRA images
static int forward_impl(ttl_ctx* c, const float* x, int n, int save, int from_layer, float* logits_out, float* feats_out,
                        void* stream, bool images_fresh = false) {
    if (!c || (!x && from_layer == 0 && !c->text)) return fail(TTL_EINVAL, "null argument");
    if (n < 1 || n > c->c.max_views) return fail(TTL_EINVAL, "n_views %d outside [1,%d]", n, c->c.max_views);
    if (c->K < 1 && (logits_out || save))   // a features-only forward needs no peer features
        return fail(TTL_ESTATE, c->text ? "ttl_set_image_features has not been called" : "ttl_set_text_features has not been called");
    if (!c->lora_p && save) return fail(TTL_ESTATE, "ttl_bind_lora has not been called");
    if (c->text && n != c->n_prompts) return fail(TTL_ESTATE, "text forward over %d prompts, ttl_set_prompts gave %d", n, c->n_prompts);
    const int causal = c->text;
    int rc = ttl_weights_ready(c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    const int D = c->D, F = c->F, T = c->T, M = n * T, H = c->H;
    if (!images_fresh && (rc = lora_refresh(c, s))) return rc;
    float* h = c->h;
    bool pooled_now = false;
    bool ln1_of_layer0_done = false;     // image tower, from_layer == 0: LayerNorm 1 of layer 0 comes out of the embedding pass
    if (from_layer == 0 && c->text) {
        Prof p(c, 3, s);
        HIP_TRY(launch_text_embed(c->ids, c->tok, c->pos, c->h, M, T, D, s));   // no pre-LN in the text tower
        c->stream_views = n;
    } else if (from_layer == 0) {
        // patch embedding: im2col -> GEMM (+pos); the CLS rows, the pre-LayerNorm and LayerNorm 1 of layer 0 are ONE pass behind it
        // (TTL_EMBED_FUSED=0, experiments build only: the three launches of rounds 1-5 — the bit-identity test's other arm)
        static const int embed_fused = TTL_EXPERIMENT("TTL_EMBED_FUSED", 1);
        {
            Prof p(c, 3, s);
            HIP_TRY(launch_im2col(x, c->patches, n, c->S, c->P, c->Kp, s));
            if (!embed_fused) HIP_TRY(launch_cls_rows(c->h, c->cls, c->pos, n, T, D, s));
        }
        {
            GemmArgs a = {};
            a.A = c->patches; a.lda = c->Kp; a.B = c->wpatch; a.ldb = c->Kp; a.M = n * c->G2; a.N = D; a.K = c->Kp;
            a.C = c->h; a.ldc = D; a.pos = c->pos; a.G2 = c->G2; a.T = T;
            if ((rc = gemm(c, EPI_PATCH, a, s))) return rc;
        }
        {   // (round 6: was launch_cls_rows + launch_layernorm in place + layer 0's launch_layernorm — bit-identical, two launches and two passes
            //  over the embedded rows fewer; the layer loop below skips layer 0's LayerNorm 1)
            Prof p(c, 3, s);
            Layer& l0 = c->layers[0];
            const bool tr0 = l0.trained && c->lora_p, sv0 = tr0 && save;
            if (embed_fused) {
                HIP_TRY(launch_embed_layernorms(h, c->cls, c->pos, T, c->preg, c->preb, l0.ln1g, l0.ln1b, tr0 ? l0.x1ext : c->x1, tr0 ? c->ldx : D,
                                                sv0 ? l0.mu1 : nullptr, sv0 ? l0.rs1 : nullptr, M, D, c->c.ln_eps, s));
                ln1_of_layer0_done = true;
            } else {
                HIP_TRY(launch_layernorm(h, D, c->preg, c->preb, h, nullptr, 0, nullptr, nullptr, M, D, c->c.ln_eps, s));
            }
        }
        c->stream_views = n;
    } else if (from_layer != c->c.layer_lo || n > c->stream_views) {
        return fail(TTL_ESTATE, "cannot resume at layer %d for %d views (stream holds %d views at layer %d)", from_layer, n,
                    c->stream_views, c->c.layer_lo);
    }
    for (int i = from_layer; i < c->L; ++i) {
        Layer& l = c->layers[i];
        const bool tr = l.trained && c->lora_p;   // on the gradient's path: activations kept in their own buffers (no
                                                  // adapters at all on the image tower when the text tower is the one being tuned)
        const bool lo = tr && l.lora;             // LoRA path active (B == 0 forever in the other layers, Q10)
        const bool sv = tr && save;
        op_t* x1 = tr ? l.x1ext : c->x1;
        const int ldx1 = tr ? c->ldx : D;
        op_t* qkv = tr ? l.qkv : c->qkv;
        op_t* att = tr ? l.attn : c->attn;
        const int ldat = tr ? l.ldat : D;              // (row pitch of att: K-extension columns for an out_proj adapter)
        const bool lo_qkv = lo && c->nqkv > 0, lo_o = lo && c->has_o;
        float* h_in = h;  // trained layers write h_mid / h_out to fresh buffers, so h_in survives for LN1 backward
        if (!(i == 0 && ln1_of_layer0_done)) {
            Prof p(c, 3, s);
            HIP_TRY(launch_layernorm(h_in, D, l.ln1g, l.ln1b, nullptr, x1, ldx1, sv ? l.mu1 : nullptr, sv ? l.rs1 : nullptr, M, D, c->c.ln_eps, s));
        }
        if (lo_qkv) {
            Prof p(c, 4, s);
            const int xoff[3] = {0, 0, 0};
            HIP_TRY(launch_lora_skinny(x1, ldx1, xoff, c->nqkv, l.acat, D, c->r, c->scaling, x1 + D, ldx1, M, s));
        }
        // TTL_POOLED_LAST_LAYER=0: run the last layer densely (A/B and the equivalence test)
        static const int pooled_last = TTL_EXPERIMENT("TTL_POOLED_LAST_LAYER", 1);
        if (pooled_last && i == c->L - 1 && n < 1024) {   // (row maps / compact [n, .] buffers: guarded small-M GEMM kernels only)
            pooled_now = true;     // -> ttl_ctx::saved_pooled: the backward and the view packing read the LN2 statistics where THIS forward left them
            // ---- last layer: the head reads ONE row per sequence (image tower: CLS, HF modeling_clip.py pooled =
            // last_hidden_state[:, 0]; text tower: the end-of-text token), so beyond K and V of every token everything
            // runs on those n rows: the query projection, attention for that query, out_proj, LN2 and the MLP — in place,
            // so the saved activations sit where the pooled-row backward reads them.  Image tower: the CLS rows have the
            // fixed pitch T*D (T*F, T*ldx1); text tower: row maps (poolrows[v] = v*T + eot position).  Identical logits;
            // 155 of the 2263 GFLOP of a 64-view image forward disappear.
            const int* map = c->text ? c->poolrows : nullptr;
            const long long pitch = map ? 1 : T;                  // row pitch multiplier when addressing by stride
            const int Kq = lo_qkv ? D + c->ext : D;
            {
                GemmArgs a = {};   // K and V for all tokens: rows D..3D of the [3D][ldw] weight image
                a.A = x1; a.lda = ldx1; a.B = l.wqkv + (size_t)D * c->ldw; a.ldb = c->ldw; a.M = M; a.N = 2 * D; a.K = Kq;
                a.C = qkv + D; a.ldc = 3 * D; a.bias = l.bqkv + D;
                if ((rc = gemm(c, EPI_OP, a, s))) return rc;
            }
            {
                GemmArgs a = {};   // Q for the pooled rows
                a.A = x1; a.lda = (int)(pitch * ldx1); a.B = l.wqkv; a.ldb = c->ldw; a.M = n; a.N = D; a.K = Kq;
                a.C = qkv; a.ldc = (int)(pitch * 3 * D); a.bias = l.bqkv; a.amap = map; a.cmap = map;
                if ((rc = gemm(c, EPI_OP, a, s))) return rc;
            }
            {
                Prof p(c, 1, s);
                HIP_TRY(launch_attention_fwd_cls(qkv, qkv_row_major(T, D, 3 * D), att, ldat, sv ? l.lse : nullptr, n, T, H, s, c->text ? c->pool : nullptr, causal));
            }
            l.qkv_hm = false;   // (K/V by a big-M launch into columns D.., Q of the pooled rows by a small-M one: row-major)
            if (lo_o) {   // out_proj adapter: U_o = s * attn * A_o^T into the extension columns of the pooled rows
                Prof p(c, 4, s);
                const int xoff[3] = {0, 0, 0};
                HIP_TRY(launch_lora_skinny(att, pitch * ldat, xoff, 1, l.acat_o, D, c->r, c->scaling, att + D, pitch * ldat, n, s, map));
            }
            float* h_mid = tr ? l.h_mid : h_in;
            {
                GemmArgs a = {};
                a.A = att; a.lda = (int)(pitch * ldat); a.B = l.wo; a.ldb = l.ldwo; a.M = n; a.N = D; a.K = lo_o ? D + 64 : D;
                a.C = h_mid; a.ldc = (int)(pitch * D); a.bias = l.bo; a.resid = h_in; a.ldr = (int)(pitch * D); a.amap = map; a.cmap = map;
                if ((rc = gemm(c, EPI_RESID_F32, a, s))) return rc;
            }
            {
                Prof p(c, 3, s);   // statistics of the pooled rows are stored compactly: [n]
                HIP_TRY(launch_layernorm(h_mid, pitch * D, l.ln2g, l.ln2b, nullptr, c->x2, D, sv ? l.mu2 : nullptr,
                                         sv ? l.rs2 : nullptr, n, D, c->c.ln_eps, s, map));
            }
            {
                GemmArgs a = {};
                a.A = c->x2; a.lda = D; a.B = l.w1; a.ldb = D; a.M = n; a.N = F; a.K = D;
                a.C = c->g; a.ldc = F; a.bias = l.b1; a.C2 = sv ? l.u : nullptr; a.ldc2 = (int)(pitch * F); a.c2map = map;
                if ((rc = gemm(c, EPI_GELU, a, s))) return rc;
            }
            float* h_next = tr ? c->h_out[i - c->c.layer_lo] : h_mid;
            {
                GemmArgs a = {};
                a.A = c->g; a.lda = F; a.B = l.w2; a.ldb = F; a.M = n; a.N = D; a.K = F;
                a.C = h_next; a.ldc = (int)(pitch * D); a.bias = l.b2; a.resid = h_mid; a.ldr = (int)(pitch * D); a.cmap = map;
                if ((rc = gemm(c, EPI_RESID_F32, a, s))) return rc;
            }
            if (tr) l.h_in = h_in;
            h = h_next;
            continue;
        }
        bool hm = false;
        {
            GemmArgs a = {};
            a.A = x1; a.lda = ldx1; a.B = l.wqkv; a.ldb = c->ldw; a.M = M; a.N = 3 * D; a.K = lo_qkv ? D + c->ext : D;
            a.C = qkv; a.ldc = 3 * D; a.bias = l.bqkv;
            hm = qkv_goes_head_major(c, a);
            if ((rc = gemm(c, EPI_OP, a, s))) return rc;
        }
        if (tr) l.qkv_hm = hm;
        {
            Prof p(c, 1, s);
            HIP_TRY(launch_attention_fwd(qkv, hm ? qkv_head_major(T, H) : qkv_row_major(T, D, 3 * D), att, ldat, sv ? l.lse : nullptr, n, T, H, s, causal));
        }
        if (lo_o) {
            Prof p(c, 4, s);
            const int xoff[3] = {0, 0, 0};
            HIP_TRY(launch_lora_skinny(att, ldat, xoff, 1, l.acat_o, D, c->r, c->scaling, att + D, ldat, M, s));
        }
        float* h_mid = tr ? l.h_mid : h_in;
        {
            GemmArgs a = {};
            a.A = att; a.lda = ldat; a.B = l.wo; a.ldb = l.ldwo; a.M = M; a.N = D; a.K = lo_o ? D + 64 : D;
            a.C = h_mid; a.ldc = D; a.bias = l.bo; a.resid = h_in; a.ldr = D;
            if ((rc = gemm(c, EPI_RESID_F32, a, s))) return rc;
        }
        {
            Prof p(c, 3, s);
            HIP_TRY(launch_layernorm(h_mid, D, l.ln2g, l.ln2b, nullptr, c->x2, D, sv ? l.mu2 : nullptr, sv ? l.rs2 : nullptr, M, D, c->c.ln_eps, s));
        }
        {
            GemmArgs a = {};
            a.A = c->x2; a.lda = D; a.B = l.w1; a.ldb = D; a.M = M; a.N = F; a.K = D;
            a.C = c->g; a.ldc = F; a.bias = l.b1; a.C2 = sv ? l.u : nullptr; a.ldc2 = F;
            if ((rc = gemm(c, EPI_GELU, a, s))) return rc;
        }
        float* h_next = tr ? c->h_out[i - c->c.layer_lo] : h_mid;
        {
            GemmArgs a = {};
            a.A = c->g; a.lda = F; a.B = l.w2; a.ldb = F; a.M = M; a.N = D; a.K = F;
            a.C = h_next; a.ldc = D; a.bias = l.b2; a.resid = h_mid; a.ldr = D;
            if ((rc = gemm(c, EPI_RESID_F32, a, s))) return rc;
        }
        if (tr) l.h_in = h_in;  // pointer only: h_in stays untouched from here on (the stream moved on)
        h = h_next;
    }
    {
        Prof p(c, 5, s);
        if (c->text) HIP_TRY(launch_gather_rows_f32(h, D, c->pool, T, c->hpool, n, D, s));   // end-of-text rows -> [n, D]
        // image tower, forward that is not saved for a backward (the adapted 1-view prediction, plain model(x) calls): the logit
        // launch writes straight into the caller's buffer; a saved forward keeps c->logits, which the fused loss reads
        const bool direct = logits_out && !save && !c->text;
        HeadArgs a = head_args(c, h, feats_out, direct ? logits_out : c->logits);
        HIP_TRY(launch_head_fwd(a, n, s));
        if (c->text && c->K > 0) {   // the head produced [prompts, views]; the loss and the caller want [views, prompts]
            HIP_TRY(launch_transpose_f32(c->logits, n, c->K, c->logits_nk, s));
            if (logits_out) HIP_TRY(hipMemcpyAsync(logits_out, c->logits_nk, (size_t)n * c->K * sizeof(float), hipMemcpyDeviceToDevice, s));
        } else if (logits_out && !direct)
            HIP_TRY(hipMemcpyAsync(logits_out, c->logits, (size_t)n * c->K * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    c->saved = save != 0;
    c->saved_n = n;
    if (save) c->saved_pooled = pooled_now;
    return 0;
}

int ttl_head_logits(ttl_ctx* c, const float* feats, int n, float* logits_out, void* stream) {
    if (!c || !feats || !logits_out) return fail(TTL_EINVAL, "null argument");
    if (c->text) return fail(TTL_ESTATE, "ttl_head_logits on a text-tower context");
    if (n < 1 || n > c->c.max_views) return fail(TTL_EINVAL, "n_views %d outside [1,%d]", n, c->c.max_views);
    if (c->K < 1) return fail(TTL_ESTATE, "ttl_set_text_features first");
    HeadArgs a = head_args(c, nullptr, nullptr, logits_out);
    a.f = const_cast<float*>(feats);      // read only by the logit stage
    HIP_TRY(launch_head_logits(a, n, (hipStream_t)stream));
    return 0;
}

int ttl_vit_forward(ttl_ctx* c, const float* x, int n, int save, float* logits_out, float* feats_out, void* stream) {
    if (c && c->text) return fail(TTL_ESTATE, "ttl_vit_forward on a text-tower context (use ttl_text_forward)");
    return forward_impl(c, x, n, save, 0, logits_out, feats_out, stream);
}

int ttl_text_forward(ttl_ctx* c, int save, float* logits_out, float* feats_out, void* stream) {
    if (!c) return fail(TTL_EINVAL, "null ctx");
    if (!c->text) return fail(TTL_ESTATE, "ttl_text_forward needs a text-tower context");
    if (c->n_prompts < 1) return fail(TTL_ESTATE, "ttl_set_prompts has not been called");
    return forward_impl(c, nullptr, c->n_prompts, save, 0, logits_out, feats_out, stream);
}

int ttl_entropy_select_loss(const float* logits, int N, int K, int mode, double rho, float thresh, float margin, float reweight,
                            const unsigned char* keep, float* H_out, int64_t* idx_out, int* n_out, float* loss_out,
                            float* dlogits_out, void* stream) {
    if (!logits || !dlogits_out || !n_out) return fail(TTL_EINVAL, "null argument");
    if (N < 1 || K < 1) return fail(TTL_EINVAL, "bad shape");
    hipStream_t s = (hipStream_t)stream;
    float* scratch = nullptr;
    size_t cnt = 4 * (size_t)N + 3 * (size_t)K + 16;
    HIP_TRY(hipMallocAsync((void**)&scratch, cnt * sizeof(float), s));
    hipError_t e = launch_entropy_loss(logits, N, K, 0, mode, rho, thresh, margin, reweight, 0, H_out, (long long*)idx_out, n_out,
                                       loss_out, dlogits_out, scratch, s, keep);
    (void)hipFreeAsync(scratch, s);
    HIP_TRY(e);
    return 0;
}

int ttl_tpt_select_loss(const float* logits, int N, int K, double rho, int reuse_idx, float* H_out, int64_t* idx_io, int* n_io,
                        float* loss_out, float* dlogits_out, void* stream) {
    if (!logits || !dlogits_out || !n_io || !idx_io) return fail(TTL_EINVAL, "null argument");
    if (N < 1 || K < 1) return fail(TTL_EINVAL, "bad shape");
    hipStream_t s = (hipStream_t)stream;
    float* scratch = nullptr;
    size_t cnt = 4 * (size_t)N + 3 * (size_t)K + 16;
    HIP_TRY(hipMallocAsync((void**)&scratch, cnt * sizeof(float), s));
    hipError_t e = launch_entropy_loss(logits, N, K, 1, TTL_SEL_TOPK, rho, 0.f, 0.f, 0.f, reuse_idx, H_out, (long long*)idx_io, n_io,
                                       loss_out, dlogits_out, scratch, s);
    (void)hipFreeAsync(scratch, s);
    HIP_TRY(e);
    return 0;
}

// the same two entries on a context's own scratch (no allocation on the step-wise path; a context serves one stream at a time)
int ttl_ctx_entropy_select_loss(ttl_ctx* c, const float* logits, int N, int K, int mode, double rho, float thresh, float margin,
                                float reweight, const unsigned char* keep, float* H_out, int64_t* idx_out, int* n_out, float* loss_out,
                                float* dlogits_out, void* stream) {
    if (!c || !logits || !dlogits_out || !n_out) return fail(TTL_EINVAL, "null argument");
    const int nmax = c->c.max_views > c->c.max_classes ? c->c.max_views : c->c.max_classes;
    if (N < 1 || K < 1 || N > nmax || K > nmax) return fail(TTL_EINVAL, "logits [%d,%d] exceed the context's capacity %d", N, K, nmax);
    HIP_TRY(launch_entropy_loss(logits, N, K, 0, mode, rho, thresh, margin, reweight, 0, H_out, (long long*)idx_out, n_out, loss_out,
                                dlogits_out, c->loss_scratch, (hipStream_t)stream, keep));
    return 0;
}

int ttl_ctx_tpt_select_loss(ttl_ctx* c, const float* logits, int N, int K, double rho, int reuse_idx, float* H_out, int64_t* idx_io,
                            int* n_io, float* loss_out, float* dlogits_out, void* stream) {
    if (!c || !logits || !dlogits_out || !n_io || !idx_io) return fail(TTL_EINVAL, "null argument");
    const int nmax = c->c.max_views > c->c.max_classes ? c->c.max_views : c->c.max_classes;
    if (N < 1 || K < 1 || N > nmax || K > nmax) return fail(TTL_EINVAL, "logits [%d,%d] exceed the context's capacity %d", N, K, nmax);
    HIP_TRY(launch_entropy_loss(logits, N, K, 1, TTL_SEL_TOPK, rho, 0.f, 0.f, 0.f, reuse_idx, H_out, (long long*)idx_io, n_io, loss_out,
                                dlogits_out, c->loss_scratch, (hipStream_t)stream));
    return 0;
}

// ------------------------------------------------------------------------------ backward
// inf_cleared: the caller's loss launch has already zeroed found_inf on this stream (the fused episode: deyo_select_grad_kernel)
// How many views a top-k selection of n keeps, if the backward may restrict itself to them (0: run it on all n views).
static int selected_views(const ttl_ctx* c, int n, int mode, double rho) {
    if (!c->sel_cap || c->text || mode != TTL_SEL_TOPK) return 0;
    const int k = (int)((double)n * rho);     // Python: int(batch_entropy.size()[0] * top), ttl.py:52 / deyo.py:105 (launch_entropy_loss)
    return (k >= 1 && k < n && k <= c->sel_cap) ? k : 0;
}

// Pack what the backward reads of views sel_idx[0 .. n_sel) (device list) into the context's SelBuf set: one launch.
static int pack_selected_views(ttl_ctx* c, const float* dlogits, int n, const long long* sel_idx, int n_sel, hipStream_t s) {
    const size_t D = c->D, F = c->F, T = c->T, H = c->H, E = c->E, K = c->K;
    GatherTable t = {};
    hipError_t err = hipSuccess;
    Prof p(c, 3, s);
    auto flush = [&]() { if (t.n && err == hipSuccess) err = launch_gather_view_blocks(t, sel_idx, n_sel, s); t.n = 0; };
    auto add = [&](const void* src, void* dst, size_t stride, size_t block) {
        if (t.n == GATHER_MAX) flush();          // (more saved layers than one table holds: --layer_range from 0)
        t.e[t.n++] = {src, dst, stride, block};
    };
    for (int i = 0; i < c->nS; ++i) {
        const Layer& l = c->layers[c->c.layer_lo + i];
        const ttl_ctx::SelBuf& b = c->selb[i];
        const bool top = (c->c.layer_lo + i == c->L - 1);
        const bool first = (i == 0);
        const size_t qkvb = 3 * D * T * sizeof(op_t);                          // both layouts keep a view's q/k/v in one block
        add(l.qkv, b.qkv, qkvb, qkvb);
        add(l.attn, b.attn, T * l.ldat * sizeof(op_t), T * l.ldat * sizeof(op_t));
        add(l.lse, b.lse, H * T * 4, H * T * 4);
        add(l.u, b.u, T * F * sizeof(op_t), T * F * sizeof(op_t));
        add(l.h_mid, b.h_mid, T * D * 4, T * D * 4);
        const size_t st2 = (top && c->saved_pooled) ? 4 : T * 4;              // pooled last layer: one LN2 statistic per view
        add(l.mu2, b.mu2, st2, st2); add(l.rs2, b.rs2, st2, st2);
        if (l.lora) add(l.x1ext, b.x1ext, T * c->ldx * sizeof(op_t), T * c->ldx * sizeof(op_t));
        if (!first) {                                                          // LN1 backward: not reached in the first trained layer
            add(l.h_in, b.h_in, T * D * 4, T * D * 4);
            add(l.mu1, b.mu1, T * 4, T * 4); add(l.rs1, b.rs1, T * 4, T * 4);
        }
    }
    add(c->cls_mean, c->sel_mean, 4, 4); add(c->cls_rstd, c->sel_rstd, 4, 4);
    add(c->ycls, c->sel_y, D * 4, D * 4); add(c->feat, c->sel_f, E * 4, E * 4);
    add(c->h_out[c->nS - 1], c->sel_h, T * D * 4, D * 4);                      // the CLS rows of the last layer's output
    add(dlogits, c->sel_dz, K * 4, K * 4);
    flush();
    HIP_TRY(err);
    return 0;
}

// sel_idx != null: dlogits is zero outside the n_sel views sel_idx lists (device, distinct view numbers < n; a top-k selection's
// list: selected_views) — the backward then runs on those views' packed activations only.  Per row the same arithmetic as the full
// backward; only the LoRA-gradient sums over rows leave out the zero rows.
static int backward_impl(ttl_ctx* c, const float* dlogits, int n, void* stream, bool inf_cleared = false, const long long* sel_idx = nullptr,
                         int n_sel = 0) {
    if (!c || !dlogits) return fail(TTL_EINVAL, "null argument");
    if (!c->saved || c->saved_n != n) return fail(TTL_ESTATE, "no saved forward for %d sequences (run the forward with save_for_backward)", n);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const bool packed = sel_idx != nullptr;
    if (packed) {
        if (c->text || n_sel < 1 || n_sel >= n || n_sel > c->sel_cap) return fail(TTL_EINVAL, "cannot restrict the backward to %d of %d views (capacity %d)", n_sel, n, c->sel_cap);
        if ((rc = pack_selected_views(c, dlogits, n, sel_idx, n_sel, s))) return rc;
        dlogits = c->sel_dz; n = n_sel;
    }
    struct PaddedRows { ttl_ctx* c; ~PaddedRows() { c->cur_padded = 0; } } padded_guard{c};
    c->cur_padded = packed ? c->sel_rows : 0;
    const int D = c->D, F = c->F, T = c->T, M = n * T, H = c->H, r = c->r;
    const int causal = c->text;
    const int* pool = c->text ? c->pool : nullptr;
    if (!inf_cleared) HIP_TRY(hipMemsetAsync(c->sc.i + SC_FOUND_INF, 0, sizeof(int), s));   // found_inf describes THIS backward's gradients
    float* dh = c->dh;      // gradient w.r.t. the residual stream at the current depth
    float* dh_alt = c->dh2;
    {
        Prof p(c, 5, s);
        HeadArgs a = head_args(c, c->h_out[c->nS - 1], nullptr, c->logits);
        if (packed) { a.h = c->sel_h; a.T = 1; a.cls_mean = c->sel_mean; a.cls_rstd = c->sel_rstd; a.y = c->sel_y; a.f = c->sel_f; }
        HIP_TRY(launch_head_bwd(a, dlogits, c->dcls, c->dcls16, n, s));
    }
    const size_t per = (size_t)r * D;
    const int ldh = c->ldh;
    const int zoff[3] = {0, 0, 0};
    const float* dres_cls = nullptr;   // != null: d/d h_mid of the layer above is compact (CLS rows only)
    for (int i = c->L - 1; i >= c->c.layer_lo; --i) {
        Layer lv = c->layers[i];        // (a copy: the packed run reads the saved activations from the SelBuf set)
        if (packed) {
            const ttl_ctx::SelBuf& b = c->selb[i - c->c.layer_lo];
            lv.h_in = b.h_in; lv.h_mid = b.h_mid; lv.x1ext = b.x1ext; lv.qkv = b.qkv; lv.attn = b.attn; lv.u = b.u;
            lv.lse = b.lse; lv.mu1 = b.mu1; lv.rs1 = b.rs1; lv.mu2 = b.mu2; lv.rs2 = b.rs2;
        }
        const Layer& l = lv;
        const bool first = (i == c->c.layer_lo);
        const bool lo_qkv = l.lora && c->nqkv > 0, lo_o = l.lora && c->has_o;
        const int need_dk = (!first || (l.lora && c->sk >= 0)) ? 1 : 0;   // the first trained layer needs dK only for a k_proj adapter
        float* gl = c->lora_g + (size_t)(i - c->c.layer_lo) * c->ntg * 2 * per;   // this layer's gradients: per target A [r,D], B [D,r]
        WgradList Lo = {};   // out_proj adapter products (their row count differs in the top layer)
        int Mo = M;
        if (i == c->L - 1) {
            // ---- top layer: the loss reads the CLS token only, so d/d h_out is non-zero on the n CLS
            // rows: MLP, LN2 and out_proj backward run on a compact [n, .] problem (row pitch T*D / T*F
            // picks the CLS rows of the saved activations) and attention backward is rank-1 per head.
            // pooled rows of the saved activations: token 0 of every view sits at a fixed pitch (image tower);
            // the end-of-text position differs per prompt, so the text tower gathers them into compact copies
            const op_t* u_rows = l.u; int ld_u = T * F;
            const float* hmid_rows = l.h_mid; long long hmid_pitch = (long long)T * D;
            const float *mu2 = l.mu2, *rs2 = l.rs2;
            // the pooled-row forward stores them as [n]: decided by the forward that SAVED them (n is n_sel in a packed run: a dense
            // forward of >= 1024 views leaves them at pitch T, and pack_selected_views copies that layout)
            const bool compact_stats = c->saved_pooled;
            int stat_pitch = compact_stats ? 1 : T;
            if (pool) {
                Prof p(c, 3, s);
                HIP_TRY(launch_gather_rows_op(l.u, F, pool, T, c->u_g, n, F, s));
                HIP_TRY(launch_gather_rows_f32(l.h_mid, D, pool, T, c->hmid_g, n, D, s));
                u_rows = c->u_g; ld_u = F; hmid_rows = c->hmid_g; hmid_pitch = D;
                if (!compact_stats) {   // dense last layer: statistics sit at every row
                    HIP_TRY(launch_gather_rows_f32(l.mu2, 1, pool, T, c->mu2_g, n, 1, s));
                    HIP_TRY(launch_gather_rows_f32(l.rs2, 1, pool, T, c->rs2_g, n, 1, s));
                    mu2 = c->mu2_g; rs2 = c->rs2_g; stat_pitch = 1;
                }
            }
            {
                GemmArgs a = {};
                a.A = c->dcls16; a.lda = D; a.B = l.w2T; a.ldb = D; a.M = n; a.N = F; a.K = D;
                a.C = c->dgc; a.ldc = F; a.aux = u_rows; a.ldaux = ld_u;
                if ((rc = gemm(c, EPI_GELU_BWD, a, s))) return rc;
            }
            {
                GemmArgs a = {};
                a.A = c->dgc; a.lda = F; a.B = l.w1T; a.ldb = F; a.M = n; a.N = D; a.K = F;
                a.C = c->dxc; a.ldc = D;
                if ((rc = gemm(c, EPI_F32, a, s))) return rc;
            }
            {
                Prof p(c, 3, s);
                HIP_TRY(launch_layernorm_bwd(c->dxc, hmid_rows, mu2, rs2, l.ln2g, c->dcls, c->dhmc, c->dhmc16, n, D, s,
                                             hmid_pitch, (long long)D, stat_pitch, 0, nullptr, ldh));
            }
            if (lo_o) {   // out_proj adapter on the pooled rows: dU_o = s * d(h_mid) * B_o into the extension columns
                Prof p(c, 4, s);
                HIP_TRY(launch_lora_skinny(c->dhmc16, ldh, zoff, 1, l.btcat_o, D, r, c->scaling, c->dhmc16 + D, ldh, n, s));
                const op_t* att_rows = l.attn; long long att_pitch = (long long)T * l.ldat;
                if (pool) {   // text tower: the pooled position differs per prompt -> compact copy of those rows
                    HIP_TRY(launch_gather_rows_op(l.attn, l.ldat, pool, T, c->att_g, n, l.ldat, s));
                    att_rows = c->att_g; att_pitch = l.ldat;
                }
                float* go = gl + (size_t)c->nqkv * 2 * per;
                Lo.p[0] = {att_rows + D, att_pitch, c->dhmc16, ldh, go + per, 1};        // dB_o = d(h_mid)^T U_o
                Lo.p[1] = {c->dhmc16 + D, ldh, att_rows, att_pitch, go, 0};              // dA_o = dU_o^T attn
                Lo.n = 2; Mo = n;
            }
            {
                GemmArgs a = {};
                a.A = c->dhmc16; a.lda = ldh; a.B = l.woT; a.ldb = l.ldwo; a.M = n; a.N = D; a.K = lo_o ? D + 64 : D;
                a.C = c->doc; a.ldc = D;
                if ((rc = gemm(c, EPI_OP, a, s))) return rc;
            }
            {
                Prof p(c, 2, s);
                HIP_TRY(launch_attention_bwd_cls(l.qkv, l.qkv_hm ? qkv_head_major(T, H) : qkv_row_major(T, D, 3 * D), l.attn, l.ldat, c->doc, l.lse,
                                                 c->dqkv, c->ldwt, n, T, H, need_dk, s, pool, causal));
            }
            dres_cls = c->dhmc;
        } else {
            // ---- MLP: dg = dh·W2 (∘ gelu'(u)) ; dx2 = du·W1 ; dh_mid = dh + LN2^T(dx2)
            {
                GemmArgs a = {};
                a.A = c->dh16; a.lda = ldh; a.B = l.w2T; a.ldb = D; a.M = M; a.N = F; a.K = D;
                a.C = c->dbig; a.ldc = F; a.aux = l.u; a.ldaux = F;
                if ((rc = gemm(c, EPI_GELU_BWD, a, s))) return rc;
            }
            {
                GemmArgs a = {};
                a.A = c->dbig; a.lda = F; a.B = l.w1T; a.ldb = F; a.M = M; a.N = D; a.K = F;
                a.C = c->dx; a.ldc = D;
                if ((rc = gemm(c, EPI_F32, a, s))) return rc;
            }
            {
                Prof p(c, 3, s);
                HIP_TRY(launch_layernorm_bwd(c->dx, l.h_mid, l.mu2, l.rs2, l.ln2g, dh, dh_alt, c->dh16, M, D, s, 0, 0, 1, 0, nullptr, ldh));
            }
            if (lo_o) {   // out_proj adapter: dU_o = s * d(h_mid) * B_o into the extension columns of dh16
                Prof p(c, 4, s);
                HIP_TRY(launch_lora_skinny(c->dh16, ldh, zoff, 1, l.btcat_o, D, r, c->scaling, c->dh16 + D, ldh, M, s));
                float* go = gl + (size_t)c->nqkv * 2 * per;
                Lo.p[0] = {l.attn + D, (long long)l.ldat, c->dh16, (long long)ldh, go + per, 1};   // dB_o = d(h_mid)^T U_o
                Lo.p[1] = {c->dh16 + D, (long long)ldh, l.attn, (long long)l.ldat, go, 0};         // dA_o = dU_o^T attn
                Lo.n = 2; Mo = M;
            }
            // ---- attention output projection: do = [dh_mid | dU_o]·[Wo | A_o]
            {
                GemmArgs a = {};
                a.A = c->dh16; a.lda = ldh; a.B = l.woT; a.ldb = l.ldwo; a.M = M; a.N = D; a.K = lo_o ? D + 64 : D;
                a.C = c->dattn; a.ldc = l.ldat;   // attention backward reads out and d(out) with one pitch
                if ((rc = gemm(c, EPI_OP, a, s))) return rc;
            }
            {
                Prof p(c, 2, s);
                HIP_TRY(launch_attention_bwd(l.qkv, l.qkv_hm ? qkv_head_major(T, H) : qkv_row_major(T, D, 3 * D), l.attn, c->dattn, l.ldat, l.lse,
                                             c->dqkv, c->ldwt, n, T, H, need_dk, s, causal));
            }
            dres_cls = nullptr;
        }
        float* dhm = dh_alt;  // d/d h_mid (dense layers)
        // ---- LoRA: dU_t = s·d_t·B_t ; dA_t = dU_t^T x, dB_t = d_t^T U_t   (layers above layer_hi carry no trainable adapters)
        if (l.lora) {
            Prof p(c, 4, s);
            WgradList Lq = {};
            if (lo_qkv) {
                int xoff[3], k = 0;
                if (c->sq >= 0) xoff[k++] = 0;
                if (c->sk >= 0) xoff[k++] = D;
                if (c->sv >= 0) xoff[k++] = 2 * D;
                HIP_TRY(launch_lora_skinny(c->dqkv, c->ldwt, xoff, c->nqkv, l.btcat, D, r, c->scaling, c->dqkv + 3 * D, c->ldwt, M, s));
                for (int j = 0; j < c->nqkv; ++j) {
                    float* gA = gl + (size_t)j * 2 * per;
                    Lq.p[Lq.n++] = {l.x1ext + D + j * r, (long long)c->ldx, c->dqkv + xoff[j], (long long)c->ldwt, gA + per, 1};   // dB_t
                    Lq.p[Lq.n++] = {c->dqkv + 3 * D + j * r, (long long)c->ldwt, l.x1ext, (long long)c->ldx, gA, 0};               // dA_t
                }
            }
            if (Lo.n && Mo == M) { Lq.p[Lq.n++] = Lo.p[0]; Lq.p[Lq.n++] = Lo.p[1]; Lo.n = 0; }   // same row count: one launch
            const float* scf = c->prescaled ? c->sc_unit : c->sc.f;
            if (Lq.n) HIP_TRY(launch_lora_wgrad(Lq, M, D, r, c->wg_partial, s, scf, c->sc.i));
            if (Lo.n) HIP_TRY(launch_lora_wgrad(Lo, Mo, D, r, c->wg_partial + (size_t)Lq.n * lora_wgrad_chunks(M) * r * D, s, scf, c->sc.i));
        }
        if (first) break;
        // ---- dx1 = [dq dk dv | dU]·[Wqkv | A]  ; dh_in = dh_mid + LN1^T(dx1)
        {
            GemmArgs a = {};
            a.A = c->dqkv; a.lda = c->ldwt; a.B = l.wqkvT; a.ldb = c->ldwt; a.M = M; a.N = D; a.K = lo_qkv ? c->ldwt : 3 * D;
            a.C = c->dx; a.ldc = D;
            if ((rc = gemm(c, EPI_F32, a, s))) return rc;
        }
        {
            Prof p(c, 3, s);
            if (dres_cls)
                HIP_TRY(launch_layernorm_bwd(c->dx, l.h_in, l.mu1, l.rs1, l.ln1g, dres_cls, dh, c->dh16, M, D, s, 0, 0, 1, T, pool, ldh));
            else
                HIP_TRY(launch_layernorm_bwd(c->dx, l.h_in, l.mu1, l.rs1, l.ln1g, dhm, dh, c->dh16, M, D, s, 0, 0, 1, 0, nullptr, ldh));
        }
        // dh now holds d/d h_in of layer i == d/d h_out of layer i-1
    }
    return 0;
}

int ttl_vit_backward_lora(ttl_ctx* c, const float* dlogits, int n, void* stream) {
    if (c && c->text) return fail(TTL_ESTATE, "ttl_vit_backward_lora on a text-tower context (use ttl_text_backward_lora)");
    return backward_impl(c, dlogits, n, stream);
}

int ttl_ctx_backward_prescaled(ttl_ctx* c, int on) {
    if (!c) return fail(TTL_EINVAL, "null ctx");
    c->prescaled = on != 0;
    return 0;
}

int ttl_ctx_set_concurrency(ttl_ctx* c, int episodes_in_flight) {
    if (!c) return fail(TTL_EINVAL, "null ctx");
    if (episodes_in_flight < 1) return fail(TTL_EINVAL, "episodes_in_flight must be >= 1");
    c->concurrency = episodes_in_flight;
    return 0;
}

int ttl_vit_backward_lora_selected(ttl_ctx* c, const float* dlogits, int n, const int64_t* idx, int n_selected, void* stream) {
    if (c && c->text) return fail(TTL_ESTATE, "ttl_vit_backward_lora_selected on a text-tower context");
    if (!c || !idx) return fail(TTL_EINVAL, "null argument");
    // the same rule as the fused episode, so that both paths sum the LoRA gradients over the same rows
    const bool pack = c->sel_cap && n_selected >= 1 && n_selected < n && n_selected <= c->sel_cap;
    return backward_impl(c, dlogits, n, stream, false, pack ? (const long long*)idx : nullptr, pack ? n_selected : 0);
}

int ttl_text_backward_lora(ttl_ctx* c, const float* dlogits, void* stream) {
    if (!c || !dlogits) return fail(TTL_EINVAL, "null argument");
    if (!c->text) return fail(TTL_ESTATE, "ttl_text_backward_lora needs a text-tower context");
    // the caller's gradient is [views, prompts]; the head backward walks prompts
    HIP_TRY(launch_transpose_f32(dlogits, c->K, c->n_prompts, c->dlogits_kn, (hipStream_t)stream));
    return backward_impl(c, c->dlogits_kn, c->n_prompts, stream);
}

int ttl_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, float wd, int step,
                   const int* nsel, void* stream) {
    if (!p || !g || !m || !v) return fail(TTL_EINVAL, "null argument");
    if (step < 1) return fail(TTL_EINVAL, "step must be >= 1");
    HIP_TRY(launch_adamw(p, g, m, v, n, lr, b1, b2, eps, wd, step, nsel, (hipStream_t)stream));
    return 0;
}

int ttl_scaler_config(ttl_ctx* c, int dynamic, float init_scale, float growth_factor, float backoff_factor, int growth_interval) {
    if (!c) return fail(TTL_EINVAL, "null ctx");
    if (!(init_scale > 0.f) || !(growth_factor >= 1.f) || !(backoff_factor > 0.f && backoff_factor <= 1.f) || growth_interval < 1)
        return fail(TTL_EINVAL, "bad GradScaler parameters");
    HIP_TRY(hipDeviceSynchronize());
    const float f[SC_NF] = {init_scale, 1.0f / init_scale, 1.f, 1.f};
    const int z[SC_NI] = {0};
    HIP_TRY(hipMemcpy(c->sc.f, f, sizeof f, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->sc.i, z, sizeof z, hipMemcpyHostToDevice));
    c->sc_dynamic = dynamic != 0; c->sc_growth = growth_factor; c->sc_backoff = backoff_factor; c->sc_interval = growth_interval;
    return 0;
}

int ttl_scaler_state(ttl_ctx* c, float* scale, int* growth_tracker, int* skipped_steps, int* optimizer_steps) {
    if (!c) return fail(TTL_EINVAL, "null ctx");
    HIP_TRY(hipDeviceSynchronize());
    float f[SC_NF]; int i[SC_NI];
    HIP_TRY(hipMemcpy(f, c->sc.f, sizeof f, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(i, c->sc.i, sizeof i, hipMemcpyDeviceToHost));
    if (scale) *scale = f[SC_SCALE];
    if (growth_tracker) *growth_tracker = i[SC_TRACKER];
    if (skipped_steps) *skipped_steps = i[SC_SKIPPED];
    if (optimizer_steps) *optimizer_steps = i[SC_STEP];
    return 0;
}

int ttl_scaler_unscale(ttl_ctx* c, float* grads, size_t n, void* stream) {
    if (!c || !grads) return fail(TTL_EINVAL, "null argument");
    HIP_TRY(launch_scaler_unscale(grads, n, c->sc, (hipStream_t)stream));
    return 0;
}

int ttl_optimizer_step(ttl_ctx* c, float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                       float wd, int step, const int* nsel, void* stream) {
    if (!c || !p || !g || !m || !v) return fail(TTL_EINVAL, "null argument");
    if (step < 0) return fail(TTL_EINVAL, "step must be >= 1 (or 0: count on the device)");
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(launch_scaler_pre_step(c->sc, nsel, step, b1, b2, c->sc_dynamic, c->sc_growth, c->sc_backoff, c->sc_interval, s));
    HIP_TRY(launch_adamw_dev(p, g, m, v, n, lr, b1, b2, eps, wd, c->sc, s));
    return 0;
}

int ttl_lora_reset(float* p, const float* snap, float* m, float* v, size_t n, void* stream) {
    if (!p || !snap) return fail(TTL_EINVAL, "null argument");
    HIP_TRY(launch_lora_reset(p, snap, m, v, n, (hipStream_t)stream));
    return 0;
}

// ------------------------------------------------------------------------------ PLPD filter (deyo.py:115-151)
static int plpd_check(const ttl_plpd_args* p, int n_views, int S) {
    if (p->aug_type != TTL_PLPD_OCC && p->aug_type != TTL_PLPD_PATCH && p->aug_type != TTL_PLPD_PIXEL)
        return fail(TTL_EINVAL, "unknown aug_type %d", p->aug_type);
    if (p->n_candidates < 1 || p->n_candidates > n_views) return fail(TTL_EINVAL, "n_candidates %d outside [1,%d]", p->n_candidates, n_views);
    if (p->aug_type != TTL_PLPD_OCC && !p->perm) return fail(TTL_EINVAL, "aug_type 'patch' / 'pixel' needs the host-drawn permutations");
    if (p->aug_type == TTL_PLPD_PATCH && (p->patch_len < 1 || p->patch_len > S)) return fail(TTL_EINVAL, "patch_len %d outside [1,%d]", p->patch_len, S);
    if (p->aug_type == TTL_PLPD_OCC && (p->occlusion_size < 1 || p->row_start < 0 || p->column_start < 0 ||
                                        p->row_start + p->occlusion_size > S || p->column_start + p->occlusion_size > S))
        return fail(TTL_EINVAL, "occlusion window [%d+%d, %d+%d] outside the %d x %d view", p->row_start, p->occlusion_size, p->column_start,
                    p->occlusion_size, S, S);
    return 0;
}
static PlpdArgs plpd_launch_args(const ttl_plpd_args* p, int update, int S) {
    PlpdArgs a = {p->aug_type, p->patch_len, p->occlusion_size, p->row_start, p->column_start, p->perm};
    if (a.perm) a.perm += (size_t)update * (p->aug_type == TTL_PLPD_PATCH ? (size_t)p->n_candidates * p->patch_len * p->patch_len : (size_t)S * S);
    return a;
}
// the context `f` that runs the second forward owns the destroyed views (allocated on first use: never inside a stream capture,
// ttl_episode_capture runs the episode eagerly first)
static int plpd_buffers(ttl_ctx* f, const ttl_plpd_args* p) {
    ttl_ctx* c = f;
    if (!c->plpd_x) ALLOC(c->plpd_x, (size_t)c->c.max_views * 3 * c->S * c->S, false);
    const size_t need = plpd_views_workspace_floats(c->c.max_views, c->S, p->aug_type, p->patch_len);
    if (need > c->plpd_ws_floats) { ALLOC(c->plpd_ws, need, false); c->plpd_ws_floats = need; }
    return 0;
}

size_t ttl_plpd_views_workspace_bytes(int n_max, int size, const ttl_plpd_args* p) {
    if (!p || n_max < 1 || size < 1) return 0;
    return plpd_views_workspace_floats(n_max, size, p->aug_type, p->patch_len) * sizeof(float);
}

int ttl_plpd_views(const float* x, int size, const int64_t* idx, const int* n_sel, int n_max, const ttl_plpd_args* p, float* out,
                   void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !idx || !n_sel || !p || !out) return fail(TTL_EINVAL, "null argument");
    if (n_max < 1 || size < 1) return fail(TTL_EINVAL, "bad shape");
    ttl_plpd_args q = *p;
    q.n_candidates = n_max;
    int rc = plpd_check(&q, n_max, size);
    if (rc) return rc;
    const size_t need = ttl_plpd_views_workspace_bytes(n_max, size, p);
    if (need && (!workspace || workspace_bytes < need)) return fail(TTL_EINVAL, "workspace smaller than ttl_plpd_views_workspace_bytes()");
    HIP_TRY(launch_plpd_views(x, (const long long*)idx, n_sel, n_max, size, plpd_launch_args(&q, 0, size), out, (float*)workspace, (hipStream_t)stream));
    return 0;
}

int ttl_plpd_keep(const float* logits, const float* logits_prime, const int64_t* idx, const int* n_sel, int n_max, int N, int K,
                  float threshold, unsigned char* keep_out, float* plpd_out, void* stream) {
    if (!logits || !logits_prime || !idx || !n_sel || !keep_out) return fail(TTL_EINVAL, "null argument");
    if (n_max < 1 || n_max > N || K < 1) return fail(TTL_EINVAL, "bad shape");
    HIP_TRY(launch_plpd_keep(logits, logits_prime, (const long long*)idx, n_sel, n_max, N, K, threshold, keep_out, plpd_out, (hipStream_t)stream));
    return 0;
}

// Launches of an episode outside the towers (round 4): 1 reset (LoRA, Adam moments, scaler step counters),
// per update 2 head forward (LayerNorm of the pooled rows + projection; norm + logits) + 2 loss (row statistics; selection + loss
// + dZ, which also clears found_inf) + 3 head backward + 1 optimizer (GradScaler decision + AdamW), then 2 head launches of the
// adapted prediction writing into the caller's buffer and, with a target, 1 hit-count launch.
static int episode_tail(ttl_ctx* c, const ttl_episode_args* a, const float* logits1, hipStream_t s) {
    // found_inf describes the gradients between a backward and its optimizer step; the fused optimizer launch cannot clear it (every
    // block of that launch reads it), so an episode whose LAST update overflowed would leave it set for a step-wise
    // ttl_scaler_unscale + ttl_optimizer_step that follows without a backward in between (round-4 advisor)
    if (a->n_updates > 0) HIP_TRY(hipMemsetAsync(c->sc.i + SC_FOUND_INF, 0, sizeof(int), s));
    if (!a->target && !a->hits_out) return 0;
    if (!a->target || !a->hits_out) return fail(TTL_EINVAL, "target and hits_out go together");
    Prof p(c, 5, s);
    HIP_TRY(launch_topk_hits(logits1, c->text ? c->n_prompts : c->K, (const long long*)a->target, (long long*)a->hits_out, s));
    return 0;
}

int ttl_episode(ttl_ctx* c, const ttl_episode_args* a, void* stream) {
    if (!c || !a || !a->x || !a->snapshot || !a->exp_avg || !a->exp_avg_sq || !a->logits1_out) return fail(TTL_EINVAL, "null argument");
    if (c->text) return fail(TTL_ESTATE, "ttl_episode on a text-tower context (use ttl_episode_text)");
    if ((a->target == nullptr) != (a->hits_out == nullptr)) return fail(TTL_EINVAL, "target and hits_out go together");   // before the first launch
    if (!c->lora_p) return fail(TTL_ESTATE, "ttl_bind_lora has not been called");
    if (c->prescaled) return fail(TTL_ESTATE, "ttl_ctx_backward_prescaled is on: the fused episode scales its own backward (switch it off first)");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const ttl_plpd_args* pl = a->plpd;
    if (pl) {      // --filter_plpd 1: everything is checked before the first launch
        if (a->objective != 0) return fail(TTL_EINVAL, "the PLPD filter belongs to the DeYO objective (deyo.py:115)");
        if (a->n_views < 1 || a->n_views > c->c.max_views) return fail(TTL_EINVAL, "n_views %d outside [1,%d]", a->n_views, c->c.max_views);
        if ((rc = plpd_check(pl, a->n_views, c->S))) return rc;
        ttl_ctx* x2 = pl->aux;
        if (!x2 || x2 == c || x2->text) return fail(TTL_EINVAL, "plpd.aux must be a second image-tower context");
        if (!same_model(x2->c, c->c) || x2->c.max_views < pl->n_candidates) return fail(TTL_EINVAL, "plpd.aux: another model or too few views");
        if (x2->lora_p != c->lora_p) return fail(TTL_ESTATE, "plpd.aux must be bound (ttl_bind_lora) to the same parameter buffer");
        if (x2->K != c->K) return fail(TTL_ESTATE, "plpd.aux holds %d class embeddings, the context %d", x2->K, c->K);
        if ((rc = plpd_buffers(x2, pl))) return rc;
    }
    LoraImages im;
    const bool fresh = lora_images(c, &im);     // reset and optimizer launches keep the operand-dtype LoRA images current themselves
    {
        Prof p(c, 5, s);   // LoRA_AB.reset + optimizer.load_state_dict(empty): step count 0; the loss scale PERSISTS (Q14)
        HIP_TRY(launch_episode_reset(c->lora_p, a->snapshot, a->exp_avg, a->exp_avg_sq, c->lora_n, c->sc, s, fresh ? &im : nullptr));
    }
    for (int u = 0; u < a->n_updates; ++u) {
        if ((rc = forward_impl(c, a->x, a->n_views, 1, (u == 0) ? 0 : c->c.layer_lo, (u == 0) ? a->logits0_out : nullptr, nullptr,
                               stream, fresh)))
            return rc;
        const unsigned char* keep = nullptr;
        if (pl) {
            // deyo.py:102-108 first-stage selection -> :116-134 destroyed views -> :135 second forward (on the auxiliary context: the
            // activations saved above belong to the pending backward) -> :137-146 keep mask
            ttl_ctx* x2 = pl->aux;
            {
                Prof p(c, 5, s);
                HIP_TRY(launch_entropy_loss(c->logits, a->n_views, c->K, 0, a->mode, a->rho, a->thresh, a->margin, a->reweight, 0, c->H_buf,
                                            c->idx_buf, c->n_buf, c->loss_buf, c->dlogits, c->loss_scratch, s));
                HIP_TRY(launch_plpd_views(a->x, c->idx_buf, c->n_buf, pl->n_candidates, c->S, plpd_launch_args(pl, u, c->S), x2->plpd_x,
                                          x2->plpd_ws, s));
            }
            if ((rc = forward_impl(x2, x2->plpd_x, pl->n_candidates, 0, 0, nullptr, nullptr, stream))) return rc;
            {
                Prof p(c, 5, s);
                HIP_TRY(launch_plpd_keep(c->logits, x2->logits, c->idx_buf, c->n_buf, pl->n_candidates, a->n_views, c->K, pl->threshold,
                                         c->keep_buf, c->plpd_val, s));
            }
            keep = c->keep_buf;
        }
        {
            Prof p(c, 5, s);
            HIP_TRY(launch_entropy_loss(c->logits, a->n_views, c->K, a->objective, a->mode, a->rho, a->thresh, a->margin, a->reweight,
                                        (a->objective == 1 && u > 0) ? 1 : 0, c->H_buf, c->idx_buf, c->n_buf, c->loss_buf, c->dlogits,
                                        c->loss_scratch, s, keep, c->sc.i + SC_FOUND_INF));
        }
        {   // top-k selections: the gradient is zero outside the listed views -> backward on those views only
            const int nsel = selected_views(c, a->n_views, a->mode, a->rho);
            if ((rc = backward_impl(c, c->dlogits, a->n_views, stream, true, nsel ? c->idx_buf : nullptr, nsel))) return rc;
        }
        {
            Prof p(c, 5, s);   // scaler.step(optimizer); scaler.update()  (deyo.py:186-188): whole step or nothing
            HIP_TRY(launch_adamw_fused(c->lora_p, c->lora_g, a->exp_avg, a->exp_avg_sq, c->lora_n, a->lr, a->beta1, a->beta2, a->eps,
                                       a->weight_decay, c->sc, c->n_buf, u, c->sc_dynamic, c->sc_growth, c->sc_backoff, c->sc_interval, s,
                                       fresh ? &im : nullptr));
        }
    }
    // adapted prediction on view 0 (ttl.py:350-352): layers below layer_lo are unchanged by the
    // update, so resume from the stream row block of view 0
    if (a->n_updates < 1) rc = forward_impl(c, a->x, 1, 0, 0, a->logits1_out, nullptr, stream, fresh);
    else rc = forward_impl(c, a->x, 1, 0, c->c.layer_lo, a->logits1_out, nullptr, stream, fresh);
    if (rc) return rc;
    return episode_tail(c, a, a->logits1_out, s);
}

// ---- the episode as a HIP graph: ~130 launches replayed with one hipGraphLaunch (the enqueue costs the host ~2.7 ms
// per image otherwise, which bounds small-view-count runs: 8 views take < 1 ms of GPU time)
struct ttl_graph { hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };

int ttl_episode_capture(ttl_ctx* c, const ttl_episode_args* a, void* stream, ttl_graph** out) {
    if (!c || !a || !out) return fail(TTL_EINVAL, "null argument");
    *out = nullptr;
    if (c->prof) return fail(TTL_ESTATE, "cannot capture while profiling is enabled (events are recorded per launch)");
    hipStream_t s = (hipStream_t)stream;
    if (!s) return fail(TTL_EINVAL, "capture needs an explicit (non-default) stream");
    // first run outside the capture: one-time hipFuncSetAttribute calls of the launchers must not fall inside it
    int rc = ttl_episode(c, a, stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    rc = ttl_episode(c, a, stream);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail((int)e, "hipStreamEndCapture: %s", hipGetErrorString(e));
    ttl_graph* tg = new ttl_graph();
    tg->graph = g;
    e = hipGraphInstantiate(&tg->exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) { (void)hipGraphDestroy(g); delete tg; return fail((int)e, "hipGraphInstantiate: %s", hipGetErrorString(e)); }
    *out = tg;
    return 0;
}

int ttl_graph_launch(ttl_graph* g, void* stream) {
    if (!g || !g->exec) return fail(TTL_EINVAL, "null graph");
    HIP_TRY(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return 0;
}

void ttl_graph_destroy(ttl_graph* g) {
    if (!g) return;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

// --lora_encoder text (clip/custom_clip.py:672-678): image features of the views without grad on the image-tower
// context `v` (no adapters bound), then the same update loop on the text-tower context `c`.
int ttl_episode_text(ttl_ctx* c, ttl_ctx* v, const ttl_episode_args* a, void* stream) {
    if (!c || !v || !a || !a->x || !a->snapshot || !a->exp_avg || !a->exp_avg_sq || !a->logits1_out) return fail(TTL_EINVAL, "null argument");
    if (!c->text || v->text) return fail(TTL_ESTATE, "ttl_episode_text(text_ctx, image_ctx, ...): wrong tower kinds");
    if ((a->target == nullptr) != (a->hits_out == nullptr)) return fail(TTL_EINVAL, "target and hits_out go together");   // before the first launch
    if (!c->lora_p) return fail(TTL_ESTATE, "ttl_bind_lora has not been called on the text context");
    if (c->prescaled) return fail(TTL_ESTATE, "ttl_ctx_backward_prescaled is on: the fused episode scales its own backward (switch it off first)");
    if (c->n_prompts < 1) return fail(TTL_ESTATE, "ttl_set_prompts has not been called");
    if (a->n_views > c->c.max_classes) return fail(TTL_EINVAL, "n_views %d exceeds the text context's capacity %d", a->n_views, c->c.max_classes);
    if (v->E != c->E) return fail(TTL_EINVAL, "embed dims differ (%d vs %d)", v->E, c->E);
    hipStream_t s = (hipStream_t)stream;
    int rc;
    const ttl_plpd_args* pl = a->plpd;
    if (pl) {      // --filter_plpd 1 with --lora_encoder text: the destroyed views need image FEATURES only, the image context serves
        if (a->objective != 0) return fail(TTL_EINVAL, "the PLPD filter belongs to the DeYO objective (deyo.py:115)");
        if (a->n_views < 1 || a->n_views > v->c.max_views) return fail(TTL_EINVAL, "n_views %d outside [1,%d]", a->n_views, v->c.max_views);
        if ((rc = plpd_check(pl, a->n_views, v->S))) return rc;
        if ((rc = plpd_buffers(v, pl))) return rc;
        if (!c->plpd_tf) {
            const size_t nv = c->c.max_classes, np = c->c.max_views;      // (text context: "views" = prompts, "classes" = image views)
            ALLOC(c->plpd_tf, nv * c->E, false); ALLOC(c->plpd_tfT, nv * c->E, false);
            ALLOC(c->plpd_lkn, np * nv, false); ALLOC(c->plpd_lnk, np * nv, false);
        }
    }
    LoraImages im;
    const bool fresh = lora_images(c, &im);
    HIP_TRY(launch_episode_reset(c->lora_p, a->snapshot, a->exp_avg, a->exp_avg_sq, c->lora_n, c->sc, s, fresh ? &im : nullptr));
    // image side: forward only; its own logits (against whatever peer features it holds) are not used
    float* feats = v->head_te;   // [max_views, E] scratch of the image context
    if ((rc = forward_impl(v, a->x, a->n_views, 0, 0, nullptr, feats, stream))) return rc;
    if ((rc = set_peer_features(c, feats, a->n_views, 1, c->scale, s))) return rc;
    const int N = a->n_views, K = c->n_prompts;
    for (int u = 0; u < a->n_updates; ++u) {
        if ((rc = forward_impl(c, nullptr, K, 1, (u == 0) ? 0 : c->c.layer_lo, (u == 0) ? a->logits0_out : nullptr, nullptr, stream, fresh)))
            return rc;
        const unsigned char* keep = nullptr;
        if (pl) {
            // the destroyed views are scored against the text features of the PENDING forward (c->feat: the adapters have not moved
            // since): image features of x' on the image context, normalised, times the normalised text features
            const int nc = pl->n_candidates;
            HIP_TRY(launch_entropy_loss(c->logits_nk, N, K, 0, a->mode, a->rho, a->thresh, a->margin, a->reweight, 0, c->H_buf, c->idx_buf,
                                        c->n_buf, c->loss_buf, c->dlogits_nk, c->loss_scratch, s));
            HIP_TRY(launch_plpd_views(a->x, c->idx_buf, c->n_buf, nc, v->S, plpd_launch_args(pl, u, v->S), v->plpd_x, v->plpd_ws, s));
            if ((rc = forward_impl(v, v->plpd_x, nc, 0, 0, nullptr, feats, stream))) return rc;
            HIP_TRY(launch_unit_rows(feats, nc, c->E, 1, c->plpd_tf, c->plpd_tfT, s));
            HeadArgs h = head_args(c, nullptr, nullptr, c->plpd_lkn);
            h.tfeat = c->plpd_tf; h.tfeatT = c->plpd_tfT; h.K = nc;
            HIP_TRY(launch_head_logits(h, K, s));                                      // [prompts, candidates]
            HIP_TRY(launch_transpose_f32(c->plpd_lkn, K, nc, c->plpd_lnk, s));         // -> [candidates, prompts]
            HIP_TRY(launch_plpd_keep(c->logits_nk, c->plpd_lnk, c->idx_buf, c->n_buf, nc, N, K, pl->threshold, c->keep_buf, c->plpd_val, s));
            keep = c->keep_buf;
        }
        {
            Prof p(c, 5, s);
            HIP_TRY(launch_entropy_loss(c->logits_nk, N, K, a->objective, a->mode, a->rho, a->thresh, a->margin, a->reweight,
                                        (a->objective == 1 && u > 0) ? 1 : 0, c->H_buf, c->idx_buf, c->n_buf, c->loss_buf,
                                        c->dlogits_nk, c->loss_scratch, s, keep, c->sc.i + SC_FOUND_INF));
            HIP_TRY(launch_transpose_f32(c->dlogits_nk, N, K, c->dlogits_kn, s));
        }
        if ((rc = backward_impl(c, c->dlogits_kn, K, stream, true))) return rc;
        {
            Prof p(c, 5, s);
            HIP_TRY(launch_adamw_fused(c->lora_p, c->lora_g, a->exp_avg, a->exp_avg_sq, c->lora_n, a->lr, a->beta1, a->beta2, a->eps,
                                       a->weight_decay, c->sc, c->n_buf, u, c->sc_dynamic, c->sc_growth, c->sc_backoff, c->sc_interval, s,
                                       fresh ? &im : nullptr));
        }
    }
    // adapted prediction on view 0: new text features (layers below layer_lo unchanged -> resume), row 0 of the logits
    if ((rc = forward_impl(c, nullptr, K, 0, a->n_updates < 1 ? 0 : c->c.layer_lo, nullptr, nullptr, stream, fresh))) return rc;
    HIP_TRY(hipMemcpyAsync(a->logits1_out, c->logits_nk, (size_t)K * sizeof(float), hipMemcpyDeviceToDevice, s));
    return episode_tail(c, a, a->logits1_out, s);
}

// ------------------------------------------------------------------------------ kernel-level entry points
int ttl_gemm_nt(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int K, void* stream) {
    GemmArgs a = {};
    a.A = (const op_t*)A; a.lda = lda; a.B = (const op_t*)B; a.ldb = ldb; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc;
    // TTL_GEMM_PADDED=1: the caller's C has round_up(M,1280) rows -> the unguarded product kernels (bench tools)
    static const int padded = TTL_EXPERIMENT("TTL_GEMM_PADDED", 0);
    a.padded = padded ? round_up(M, 1280) : 0;
    hipError_t e = launch_gemm(EPI_F32, a, (hipStream_t)stream);
    if (e != hipSuccess) return fail((int)e, "gemm: %s (need K%%64==0, N%%128==0)", hipGetErrorString(e));
    return 0;
}

int ttl_gemm_nt_epi(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int epi, const float* bias,
                    const float* resid, int ldr, int rows_allocated, void* stream) {
    if (!A || !B || !C) return fail(TTL_EINVAL, "null argument");
    if (epi < 0 || epi > 3 || (epi == 2 && !resid)) return fail(TTL_EINVAL, "epi must be 0..3 (2 needs resid)");
    GemmArgs a = {};
    a.A = (const op_t*)A; a.lda = lda; a.B = (const op_t*)B; a.ldb = ldb; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc;
    a.bias = bias; a.resid = resid; a.ldr = ldr;
    a.padded = rows_allocated >= round_up(M, 1280) ? rows_allocated : 0;
    hipError_t e = launch_gemm((GemmEpi)epi, a, (hipStream_t)stream);
    if (e != hipSuccess) return fail((int)e, "gemm: %s (need K%%64==0, N%%128==0)", hipGetErrorString(e));
    return 0;
}

int ttl_gemm_nt_fused(const void* A, int lda, const void* B, int ldb, void* C, int ldc, void* C2, int ldc2, int M, int N, int K,
                      const float* bias, int hm_T, int rows_allocated, void* stream) {
    if (!A || !B || !C) return fail(TTL_EINVAL, "null argument");
    if (hm_T < -1 || (hm_T > 0 && (C2 || N % 192))) return fail(TTL_EINVAL, "head-major output: N = 3 * heads * 64, no second output");
    if (hm_T == -1 && (!C2 || bias)) return fail(TTL_EINVAL, "MLP dgrad form: C2 = the saved pre-activation, no bias");
    GemmArgs a = {};
    a.A = (const op_t*)A; a.lda = lda; a.B = (const op_t*)B; a.ldb = ldb; a.M = M; a.N = N; a.K = K; a.C = C; a.ldc = ldc;
    a.bias = bias;
    if (hm_T == -1) { a.aux = (const op_t*)C2; a.ldaux = ldc2; } else { a.C2 = (op_t*)C2; a.ldc2 = ldc2; }
    a.padded = rows_allocated >= round_up(M, 1280) ? rows_allocated : 0;
    if (hm_T > 0) { a.hm_T = hm_T; a.hm_magic = qkv_hm_magic(hm_T, M + 320); }
    const GemmEpi epi = hm_T > 0 ? EPI_OP : hm_T == 0 ? EPI_GELU : EPI_GELU_BWD;
    if ((hm_T > 0 && !a.hm_magic) || !gemm_takes_big(epi, a)) return fail(TTL_EINVAL, "not a big-M launch (M >= 1024, N %% 256 == 0, K %% 64 == 0, padded rows)");
    hipError_t e = launch_gemm(epi, a, (hipStream_t)stream);
    if (e != hipSuccess) return fail((int)e, "gemm: %s", hipGetErrorString(e));
    return 0;
}

int ttl_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int rows, int dim,
                      float eps, void* stream) {
    HIP_TRY(launch_layernorm(x, dim, gamma, beta, y, nullptr, 0, mean, rstd, rows, dim, eps, (hipStream_t)stream));
    return 0;
}

int ttl_cast_f32_operand(const float* src, void* dst, size_t n, void* stream) {
    HIP_TRY(launch_cast_f32_op(src, (op_t*)dst, n, (hipStream_t)stream));
    return 0;
}

int ttl_attention_fwd(const void* qkv, void* out, float* lse, int n, int T, int H, int causal, void* stream) {
    HIP_TRY(launch_attention_fwd((const op_t*)qkv, qkv_row_major(T, H * 64, 3 * H * 64), (op_t*)out, H * 64, lse, n, T, H, (hipStream_t)stream, causal));
    return 0;
}

int ttl_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, int ld_dqkv, int n, int T,
                      int H, int need_dk, int causal, void* stream) {
    HIP_TRY(launch_attention_bwd((const op_t*)qkv, qkv_row_major(T, H * 64, 3 * H * 64), (const op_t*)out, (const op_t*)dout, H * 64, lse, (op_t*)dqkv,
                                 ld_dqkv, n, T, H, need_dk, (hipStream_t)stream, causal));
    return 0;
}

size_t ttl_make_views_workspace_bytes(int height, int width, int n_views, int size) {
    if (height < 1 || width < 1 || n_views < 1 || size < 1) return 0;
    return (size_t)n_views * 2 * size * views_kstride(height, width, size) * sizeof(int);
}

int ttl_make_views(const unsigned char* image_hwc, int height, int width, const int* boxes, int n_views, int size,
                   const float mean[3], const float stdv[3], float* out, void* workspace, size_t workspace_bytes,
                   void* stream) {
    if (!image_hwc || !boxes || !out || !mean || !stdv || !workspace) return fail(TTL_EINVAL, "null argument");
    if (height < 1 || width < 1 || n_views < 1 || size < 1) return fail(TTL_EINVAL, "bad shape");
    if (workspace_bytes < ttl_make_views_workspace_bytes(height, width, n_views, size))
        return fail(TTL_EINVAL, "workspace smaller than ttl_make_views_workspace_bytes()");
    HIP_TRY(launch_make_views(image_hwc, height, width, boxes, n_views, size, mean, stdv, out, (int*)workspace,
                              views_kstride(height, width, size), (hipStream_t)stream));
    return 0;
}

int ttl_debug_copy(ttl_ctx* c, const char* name, int layer, void* dst, size_t bytes) {
    if (!c || !name || !dst) return fail(TTL_EINVAL, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    const void* src = nullptr;
    size_t have = 0;
    const size_t M = (size_t)c->saved_n * c->T, D = c->D, F = c->F;
    std::string nm(name);
    bool tr = layer >= c->c.layer_lo && layer <= c->c.layer_hi;
    if (nm == "features") { src = c->feat; have = (size_t)c->saved_n * c->E * 4; }
    // what the last ttl_episode* update really used: the confidence-selection list (int64, the reference's order:
    // deyo.py:103-108 / ttl.py:50-54), its length (int32) and the per-view entropies (fp32)
    // (capacity of the buffers: the adapted 1-view inference that ends an episode has reset saved_n to 1 by now)
    else if (nm == "idx") { src = c->idx_buf; have = (size_t)(c->c.max_views > c->c.max_classes ? c->c.max_views : c->c.max_classes) * 8; }
    else if (nm == "n_selected") { src = c->n_buf; have = 4; }
    else if (nm == "entropy") { src = c->H_buf; have = (size_t)(c->c.max_views > c->c.max_classes ? c->c.max_views : c->c.max_classes) * 4; }
    else if (nm == "plpd") { src = c->plpd_val; have = (size_t)(c->c.max_views > c->c.max_classes ? c->c.max_views : c->c.max_classes) * 4; }
    else if (nm == "keep") { src = c->keep_buf; have = (size_t)(c->c.max_views > c->c.max_classes ? c->c.max_views : c->c.max_classes); }
    else if (nm == "dh") { src = c->dh; have = M * D * 4; }
    else if (nm == "dqkv") { src = c->dqkv; have = M * c->ldwt * sizeof(op_t); }
    else if (!tr) return fail(TTL_EINVAL, "%s: layer %d is not a trained (saved) layer", name, layer);
    else {
        Layer& l = c->layers[layer];
        if (nm == "h_in") { src = l.h_in; have = M * D * 4; }
        else if (nm == "h_mid") { src = l.h_mid; have = M * D * 4; }
        else if (nm == "h_out") { src = c->h_out[layer - c->c.layer_lo]; have = M * D * 4; }
        else if (nm == "qkv") { src = l.qkv; have = M * 3 * D * sizeof(op_t); }
        else if (nm == "attn_out") { src = l.attn; have = M * l.ldat * sizeof(op_t); }
        else if (nm == "x1") { src = l.x1ext; have = M * c->ldx * sizeof(op_t); }
        else if (nm == "u") { src = l.u; have = M * F * sizeof(op_t); }
        else if (nm == "lse") { src = l.lse; have = (size_t)c->saved_n * c->H * c->T * 4; }
        else return fail(TTL_EINVAL, "unknown buffer %s", name);
    }
    if (bytes > have) return fail(TTL_EINVAL, "%s holds %zu bytes, %zu requested", name, have, bytes);
    if (nm == "qkv" && c->layers[layer].qkv_hm) {
        // the caller sees q/k/v row-major [n*T][3D] whatever layout the forward used: un-permute the head-major image on the host
        std::vector<op_t> raw(M * 3 * D);
        HIP_TRY(hipMemcpy(raw.data(), src, raw.size() * sizeof(op_t), hipMemcpyDeviceToHost));
        const size_t T = c->T, H = c->H, nel = bytes / sizeof(op_t);
        op_t* out = (op_t*)dst;
        for (size_t e = 0; e < nel; ++e) {
            const size_t m = e / (3 * D), col = e % (3 * D), v = m / T, t = m % T, pl = col / D, hd = (col % D) / 64, d = col % 64;
            out[e] = raw[((v * 3 + pl) * H + hd) * T * 64 + t * 64 + d];
        }
        return 0;
    }
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int ttl_profile_enable(ttl_ctx* c, int on) {
    if (!c) return fail(TTL_EINVAL, "null ctx");
    c->prof = on != 0;
    return 0;
}

int ttl_profile_read(ttl_ctx* c, double ms[TTL_NCLASS], long long launches[TTL_NCLASS], double* gemm_flops) {
    if (!c) return fail(TTL_EINVAL, "null ctx");
    HIP_TRY(hipDeviceSynchronize());
    for (auto& pe : c->prof_events) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, pe.second.first, pe.second.second) == hipSuccess) {
            c->prof_ms[pe.first] += t;
            c->prof_n[pe.first] += 1;
        }
        (void)hipEventDestroy(pe.second.first);
        (void)hipEventDestroy(pe.second.second);
    }
    c->prof_events.clear();
    for (int i = 0; i < TTL_NCLASS; ++i) {
        if (ms) ms[i] = c->prof_ms[i];
        if (launches) launches[i] = c->prof_n[i];
        c->prof_ms[i] = 0; c->prof_n[i] = 0;
    }
    if (gemm_flops) *gemm_flops = c->gemm_flops;
    c->gemm_flops = 0;
    c->gemm_bytes_last = c->gemm_bytes;
    c->gemm_bytes = 0;
    c->gemm_flops_all_last = c->gemm_flops_all;
    c->gemm_flops_all = 0;
    return 0;
}

int ttl_profile_gemm_flops_all(ttl_ctx* c, double* flops) {
    if (!c || !flops) return fail(TTL_EINVAL, "null argument");
    *flops = c->gemm_flops_all_last;
    return 0;
}

int ttl_profile_gemm_bytes(ttl_ctx* c, double* bytes) {
    if (!c || !bytes) return fail(TTL_EINVAL, "null argument");
    *bytes = c->gemm_bytes_last;
    return 0;
}

}  // extern "C"
#pragma GCC visibility pop
