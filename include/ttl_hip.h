/*
 * ttl_hip.h — C ABI of libttl_hip_fp16.so (fp16 operands: the build the Python surface loads by default) and of its sibling builds of the
 * same sources (libttl_hip.so: bf16 operands; libttl_hip_strict.so, libttl_hip_fp16_exp.so: tests / tools): TTL's per-sample hot path on
 * MI355X (gfx950).
 *
 * The reference is pure Python on torch/transformers/peft; the "FFI" a maintainer binds is
 * ctypes.  Each entry point below names the reference code it replaces
 * (paths relative to the reference repo root).  Conventions:
 *   - every pointer is a raw DEVICE pointer unless the comment says host;
 *   - tensors are dense row-major, fp32 unless stated;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); nothing here
 *     synchronises the host except ttl_load_weight, ttl_ctx_create/destroy, ttl_debug_copy;
 *   - return value 0 = ok, otherwise a negative TTL_E* code or a positive hipError_t;
 *     ttl_last_error() returns a thread-local message.  Nothing throws across the ABI.
 *   - a context is not re-entrant (one shared workspace arena).
 */
#ifndef TTL_HIP_H
#define TTL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTL_OK 0
#define TTL_EINVAL (-1)   /* bad argument / unsupported geometry */
#define TTL_ESTATE (-2)   /* call order violated (weights missing, lora not bound, ...) */
#define TTL_ENOMEM (-3)

/* selection modes of the confidence filter */
#define TTL_SEL_LE_THRESH 0 /* torch.where(H <= thresh)                deyo.py:107 (default)   */
#define TTL_SEL_TOPK 1      /* argsort(H)[:int(N*rho)]                 deyo.py:105, ttl.py:52  */

/* Geometry of the image tower (HF CLIPVisionConfig as built at clip/custom_clip.py:581) plus
 * the peft LoraConfig of clip/custom_clip.py:583-590 and --layer_range (ttl.py:159-161). */
typedef struct ttl_config {
    int image_size;   /* 224 */
    int patch_size;   /* 16 (B/16) or 14 (L/14) */
    int width;        /* D: 768 / 1024; head dim is fixed at 64 */
    int heads;        /* H: 12 / 16 */
    int mlp;          /* F: 3072 / 4096 */
    int layers;       /* L: 12 / 24 */
    int embed;        /* E: 512 / 768 */
    int rank;         /* r: 16 or 32 */
    float lora_alpha; /* 32 */
    int layer_lo;     /* first encoder layer whose q/v adapters train */
    int layer_hi;     /* last one (inclusive); may be below the top layer: the layers above it then carry no
                       * trainable adapters but still pass the gradient down (their activations are saved too) */
    float ln_eps;     /* 1e-5 */
    int max_views;    /* capacity N of one ttl_vit_forward call (text tower: capacity in prompts) */
    int max_classes;  /* capacity K (text tower: capacity in image views) */
    int tower;          /* TTL_TOWER_IMAGE (0) or TTL_TOWER_TEXT: the text tower of --lora_encoder text,
                         * clip/custom_clip.py:602-607 — same encoder layers, causal attention over
                         * context_length tokens, pooled at the end-of-text token; image_size / patch_size unused */
    int context_length; /* text tower: 77 */
    int vocab_size;     /* text tower: 49408 */
    int lora_targets;   /* attention projections that carry an adapter (peft LoraConfig.target_modules, clip/custom_clip.py:586):
                         * bit mask of TTL_LORA_*; 0 = the reference's q_proj + v_proj.  The bound LoRA buffer holds, per trained
                         * layer, A [r,D] then B [D,r] of every enabled projection in the order q, k, v, out. */
} ttl_config;
enum { TTL_TOWER_IMAGE = 0, TTL_TOWER_TEXT = 1 };
enum { TTL_LORA_Q = 1, TTL_LORA_K = 2, TTL_LORA_V = 4, TTL_LORA_O = 8 };

typedef struct ttl_ctx ttl_ctx;

const char* ttl_last_error(void);
const char* ttl_version(void);
/* "fp16" (libttl_hip_fp16.so: the reference's autocast dtype, ttl.py:79 — the build the Python surface loads by default, inside
 * the 1e-3 logit tolerance), "bf16" (libttl_hip.so, opt-in) or "fp32" (libttl_hip_strict.so, test build).
 * Every "operand" buffer below (qkv, attention out, ...) holds values of that type. */
const char* ttl_operand_dtype(void);
/* The environment variables this library reads, one per line: "NAME=<value in effect> default=<default> # what it selects".
 * These five are ALL a product build reads (every other TTL_* knob of the sources is a closed experiment, compiled to its default
 * unless the library was built with -DTTL_EXPERIMENTS: libttl_hip_fp16_exp.so, tests / tools only).  bench.py prints them under
 * protocol.kernel_env and refuses to time a run in which one of them is not at its default (--variant-env overrides).
 * Thread-local buffer, valid until the next call on the thread. */
const char* ttl_runtime_switches(void);

/* Device memory ttl_ctx_create will allocate for `cfg` (weights + activation arena + the packed-backward buffers of top-k
 * selections), bytes: the allocation walk of ttl_ctx_create itself with nothing allocated, so the figure is exact (and needs no
 * GPU).  NOT included: what a PLPD stage allocates on its first use in a context — the destroyed views
 * (max_views*3*image_size^2*4 bytes) plus ttl_plpd_views_workspace_bytes(); text tower: 2*max_views*E*4 + 2*prompts*max_views*4. */
size_t ttl_workspace_bytes(const ttl_config* cfg);
/* Device bytes `ctx` has allocated so far (== ttl_workspace_bytes(cfg) for an owning context until a PLPD stage has run). */
size_t ttl_ctx_allocated_bytes(const ttl_ctx* ctx);

/* Replaces model construction + .cuda(): clip/custom_clip.py:570-623, ttl.py:178-179. */
int ttl_ctx_create(const ttl_config* cfg, ttl_ctx** out);
void ttl_ctx_destroy(ttl_ctx* ctx);
/* A further context on the SAME model object: the reference has one `model` per process (ttl.py:178-179) that every test
 * image runs through; here several episodes are in flight per GPU, each in a context of its own (activation arena, LoRA
 * images), and the frozen weight images — everything but the projection images of the layers that carry adapters, whose
 * LoRA K-extension columns each context refreshes from its own adapters — are the parent's, read-only.  `parent` must own its
 * weights (not itself shared), have them all loaded (ttl_weights_ready), have the same model configuration as `cfg`
 * (max_views / max_classes may differ).  The parent's memory is reference-counted: it is released when the parent AND every context
 * sharing its images have been destroyed, in whatever order (a destroyed parent must still not be USED).  ttl_load_weight on the new
 * context is an error, and sharing FREEZES the parent's weights: ttl_load_weight on the parent fails with TTL_ESTATE while any
 * sharer is alive.  (ttl_workspace_bytes(cfg) is the footprint of an OWNING context; a sharing one allocates less.) */
int ttl_ctx_create_shared(const ttl_config* cfg, ttl_ctx* parent, ttl_ctx** out);
/* How many episodes the caller keeps in flight on this GPU (driver.EpisodePipeline: one context per HIP stream, default 1 = this
 * context has the GPU to itself).  Results agree within operand rounding whatever the value — bit-identity across values is NOT
 * promised (the N = D projections move to a kernel with another MFMA shape that folds bias and residual into the accumulator's
 * initial value; observed bit-identical on gfx950, not guaranteed): compare or resume runs at ONE value (ttl_amd.eval's resume tag
 * carries --streams, bench.py prints streams_per_gpu).  Kernel tile choices do depend on it: with other episodes in flight the idle CUs of
 * a partial round run their kernels, so a launch is chosen by its CU-time instead of its makespan (csrc/gemm_huge.hip: the N = D
 * projections on 256 x 256 tiles, +2.5 % images/s at three episodes in flight, slower alone).  Takes effect for the launches (and
 * graph captures) that follow. */
int ttl_ctx_set_concurrency(ttl_ctx* ctx, int episodes_in_flight);

/* Load one fp32 tensor of the HF vision tower by its state-dict name,
 * e.g. "vision_model.encoder.layers.3.mlp.fc1.weight", "visual_projection.weight".
 * `data` may be a host or a device pointer; the call converts to the internal bf16 / fp32
 * layouts and synchronises.  Replaces CLIPModel.from_pretrained(...).to(device),
 * clip/custom_clip.py:581. */
int ttl_load_weight(ttl_ctx* ctx, const char* name, const float* data, size_t count);
/* The same for a checkpoint kept in another element type (SURVEY.md §8b `ttl_load_weights(ctx, name, ptr, dtype)`): fp16 and bf16
 * tensors are widened to fp32 exactly and then take the path above.  Replaces CLIPModel.from_pretrained(..., torch_dtype=...),
 * clip/custom_clip.py:581. */
enum { TTL_DTYPE_F32 = 0, TTL_DTYPE_F16 = 1, TTL_DTYPE_BF16 = 2 };
int ttl_load_weight_typed(ttl_ctx* ctx, const char* name, const void* data, size_t count, int dtype);
/* 0 once every tensor of the geometry has been loaded, else TTL_ESTATE (message lists one). */
int ttl_weights_ready(ttl_ctx* ctx);

/* Cached, L2-normalised class embeddings t̂ [K,E] and exp(logit_scale).  Replaces the text
 * features + logit head inputs of clip/custom_clip.py:651-663,686 (constant per dataset, Q12). */
int ttl_set_text_features(ttl_ctx* ctx, const float* tfeat, int n_classes, float logit_scale_exp,
                          void* stream);
/* logits[v][k] = exp(logit_scale) * <f_v / ||f_v||, t̂_k> for image features f [n_views, E] (fp32, device, not normalised)
 * against the context's cached class embeddings: the logit head of clip/custom_clip.py:679-686 on its own (SURVEY.md §8b
 * `ttl_head_logits`), e.g. to re-score cached image features after ttl_set_text_features switched the label set. */
int ttl_head_logits(ttl_ctx* ctx, const float* feats, int n_views, float* logits_out, void* stream);

/* Bind the trainable LoRA parameters and their gradients: two flat fp32 device buffers laid out
 * in the order of the 12 param groups of ttl.py:195-213 — for layer = layer_lo..layer_hi:
 *   q_proj.lora_A [r,D], q_proj.lora_B [D,r], v_proj.lora_A [r,D], v_proj.lora_B [D,r].
 * n must equal (layer_hi-layer_lo+1) * 4 * r * D.  The caller (torch) owns both buffers. */
int ttl_bind_lora(ttl_ctx* ctx, float* params, float* grads, size_t n);

/* model(images) — ClipTestTimeTuning.forward/inference, clip/custom_clip.py:665-703, through
 * VisionEncoder.forward (:62-71) and HF CLIPModel.get_image_features.
 *   x [N,3,S,S] fp32 NCHW;  logits_out [N,K] fp32;  feats_out [N,E] fp32 or NULL (un-normalised
 *   image features).  save_for_backward != 0 keeps the activations of the trained layers for
 *   ttl_vit_backward_lora (the autograd graph of deyo.py:97/186). */
int ttl_vit_forward(ttl_ctx* ctx, const float* x, int n_views, int save_for_backward,
                    float* logits_out, float* feats_out, void* stream);

/* softmax_entropy + filter + weighting + mean loss and its gradient w.r.t. the logits:
 * deyo.py:85-90 (entropy), :103-108 (filter), :175-181 (coeff, loss); autograd of those.
 *   mode TTL_SEL_LE_THRESH: S = {i : H_i <= thresh}; mode TTL_SEL_TOPK: first int(N*rho) of the
 *   ascending (stable) argsort of H.   coeff_i = reweight*exp(-(H_i-margin)) (reweight==0: 1).
 *   H_out [N]; idx_out [N] int64 (first *n_out valid; index order for LE_THRESH, entropy order
 *   for TOPK — the order torch returns); n_out [1] int32; loss_out [1]; dlogits_out [N,K]
 *   (zero rows for unselected views; all zero and loss 0 when n == 0, deyo.py:110-113).
 *   keep (device uint8 [N] or NULL): second-stage filter — a selected view with keep[i] == 0 is dropped
 *   before the loss (the PLPD filter of deyo.py:144-151); idx_out still lists the first-stage set,
 *   *n_out is the surviving count.
 *   Any output pointer except dlogits_out/n_out may be NULL. */
int ttl_entropy_select_loss(const float* logits, int n_views, int n_classes, int mode, double rho,
                            float thresh, float margin, float reweight, const unsigned char* keep,
                            float* H_out, int64_t* idx_out, int* n_out, float* loss_out,
                            float* dlogits_out, void* stream);

/* TPT objective: select_confident_samples + avg_entropy and its gradient, ttl.py:50-61,87-108.
 *   reuse_idx != 0: use idx_io[0..*n_io) chosen by an earlier step (ttl.py:97-98). */
int ttl_tpt_select_loss(const float* logits, int n_views, int n_classes, double rho, int reuse_idx,
                        float* H_out, int64_t* idx_io, int* n_io, float* loss_out,
                        float* dlogits_out, void* stream);
/* Both entries on the scratch memory of a context (no allocation per call: the step-wise host loop uses these; the
 * context-free forms above take their scratch from the stream-ordered pool).  N, K <= max(max_views, max_classes). */
int ttl_ctx_entropy_select_loss(ttl_ctx* ctx, const float* logits, int n_views, int n_classes, int mode, double rho,
                                float thresh, float margin, float reweight, const unsigned char* keep,
                                float* H_out, int64_t* idx_out, int* n_out, float* loss_out,
                                float* dlogits_out, void* stream);
int ttl_ctx_tpt_select_loss(ttl_ctx* ctx, const float* logits, int n_views, int n_classes, double rho, int reuse_idx,
                            float* H_out, int64_t* idx_io, int* n_io, float* loss_out, float* dlogits_out, void* stream);

/* loss.backward() restricted to what the reference's graph contains: head ->
 * post-LN -> layers layer_hi..layer_lo, writing dA/dB of q_proj and v_proj into the bound grads
 * buffer (overwritten, i.e. optimizer.zero_grad() + backward, deyo.py:185-186). */
int ttl_vit_backward_lora(ttl_ctx* ctx, const float* dlogits, int n_views, void* stream);
/* on != 0: the dlogits handed to the backward entries of this context ALREADY carry the caller's loss scale — the reference's own
 * `scaler.scale(loss).backward()` (deyo.py:185, ttl.py:222 GradScaler(init_scale=1000)) running through the autograd node of
 * ClipTestTimeTuning.forward: the context's own loss scale is then NOT applied on top (2^10 x 1000 overflows the fp16 backward),
 * the gradients come back scaled like dlogits — torch's `scaler.step(optimizer)` unscales them and looks for inf/nan, exactly as in
 * the reference.  Off (default): the context scales and unscales by itself (fused episode, step-wise host loop).  ttl_episode*
 * refuse to run while it is on. */
int ttl_ctx_backward_prescaled(ttl_ctx* ctx, int on);
/* The same backward when the caller knows that dlogits is zero outside the n_selected views idx lists (device memory, distinct view
 * numbers < n_views) — the list a top-k selection produced: deyo.py:105 `argsort(entropys)[:int(N * selection_p)]` (--filter_ent 1),
 * ttl.py:52 select_confident_samples (TPT).  Rows of the other views contribute exactly nothing to any LoRA gradient, so the backward
 * runs on the listed views only (their saved activations are packed first; 6 of 64 views at the reference's selection_p = 0.1).
 * Falls back to the full backward when n_selected is 0, equals n_views or exceeds max_views / 4.  ttl_episode applies the same rule. */
int ttl_vit_backward_lora_selected(ttl_ctx* ctx, const float* dlogits, int n_views, const int64_t* idx, int n_selected, void* stream);

/* torch.optim.AdamW.step over one flat buffer (ttl.py:218; deyo.py:187).  `step` is the 1-based
 * step count of this update.  If `n_selected` (device int32, may be NULL) is 0 the update is
 * skipped (deyo.py:183: `if final_backward != 0`); a non-finite gradient element also skips
 * that element's update.  Context-free primitive; the reference's GradScaler contract (whole step or
 * nothing, dynamic scale) is ttl_optimizer_step below, which ttl_episode and the host surface use. */
int ttl_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                   const int* n_selected, void* stream);

/* ---- torch.amp.GradScaler semantics (ttl.py:222 `GradScaler(init_scale=1000)`, deyo.py:186-188 / ttl.py:106-108
 * `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`), state on the device per context:
 * the backward runs on scale * loss, the gradient reduction unscales and records any inf/nan, and then the WHOLE
 * optimizer step is skipped and the scale multiplied by backoff_factor, or the step is taken and the scale multiplied
 * by growth_factor after growth_interval consecutive clean steps.  The scale is the only state that survives from one
 * test image to the next.  Defaults: fp16-operand build dynamic from 2^10; bf16 build scale 1, not
 * dynamic (bf16 has fp32's exponent range) — the whole-step skip on non-finite gradients applies to both. */
int ttl_scaler_config(ttl_ctx* ctx, int dynamic, float init_scale, float growth_factor, float backoff_factor, int growth_interval);
/* Synchronises.  Any output may be NULL.  optimizer_steps = steps really taken since the last episodic reset. */
int ttl_scaler_state(ttl_ctx* ctx, float* scale, int* growth_tracker, int* skipped_steps, int* optimizer_steps);
/* scaler.unscale_(optimizer) for gradients produced outside ttl_vit_backward_lora (which unscales by itself). */
int ttl_scaler_unscale(ttl_ctx* ctx, float* grads, size_t n, void* stream);
/* scaler.step(optimizer) + scaler.update(): ttl_adamw_step with the context's GradScaler decision — all of the step
 * or none of it.  step >= 1: the 1-based count to use for the bias corrections if the step is taken; 0: the context
 * counts taken steps itself (reset by ttl_episode / ttl_lora_reset is NOT implied: see ttl_scaler_state). */
int ttl_optimizer_step(ttl_ctx* ctx, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                       const int* n_selected, void* stream);

/* Episodic reset: LoRA_AB.reset() (clip/custom_clip.py:202-215) + optimizer.load_state_dict of the
 * empty state (ttl.py:344): params <- snapshot, exp_avg = exp_avg_sq = 0.  m/v may be NULL. */
int ttl_lora_reset(float* params, const float* snapshot, float* exp_avg, float* exp_avg_sq,
                   size_t n, void* stream);

/* One whole test image without host round trips: the body of the loop at ttl.py:338-352.
 *   reset -> n_updates x [forward(x), loss (objective 0 = DeYO weighted entropy, 1 = TPT),
 *   LoRA backward, AdamW] -> logits of view 0 with the adapted weights.
 *   snapshot/exp_avg/exp_avg_sq: flat fp32 buffers shaped like the bound params.
 *   logits0_out [N,K] (first forward) may be NULL; logits1_out [1,K]. */
/* ---- PLPD filter of DeYO (--filter_plpd 1, deyo.py:115-151) on the device ----
 * x' = destroy(x[filter_ids_1]) (deyo.py:116-134), a second forward of x' (deyo.py:135), and
 * keep = softmax(z[ids1])[argmax] - softmax(z')[same class] > plpd_threshold (deyo.py:137-146) folded into the loss launch.
 * The permutations are drawn by the HOST exactly as the reference draws them — torch.argsort(torch.rand(B, patch_len^2), dim=-1)
 * per update for 'patch' (B = n_candidates rows), torch.randperm(S*S) per update for 'pixel', on torch's CPU generator — and
 * handed over as device int32 arrays; 'occ' needs none. */
enum { TTL_PLPD_OCC = 0, TTL_PLPD_PATCH = 1, TTL_PLPD_PIXEL = 2 };
typedef struct ttl_plpd_args {
    int aug_type;                                  /* TTL_PLPD_* (--aug_type) */
    float threshold;                               /* --plpd_threshold */
    int patch_len;                                 /* --patch_len ('patch') */
    int occlusion_size, row_start, column_start;   /* --occlusion_size / --row_start / --column_start ('occ') */
    const int* perm;     /* device int32; 'patch': [n_updates][n_candidates][patch_len^2], 'pixel': [n_updates][size*size], 'occ': NULL */
    int n_candidates;    /* views the FIRST selection stage yields — the row count of the reference's torch.rand(B, P): int(N*rho) in
                          * top-rho mode, N in threshold mode (H <= ln 1000 always holds for K <= 1000 classes).  The device-side count
                          * guards every kernel; the second forward runs on n_candidates rows. */
    ttl_ctx* aux;        /* image mode: a SECOND context of the same model (ttl_ctx_create_shared is fine) holding the same class
                          * embeddings and bound (ttl_bind_lora) to the SAME parameter buffer: the PLPD forward must not overwrite the
                          * activations the pending backward needs.  Text mode (ttl_episode_text): NULL, the image context serves. */
} ttl_plpd_args;
/* The destroyed views on their own (step-wise path, tests): x device fp32 [n_views,3,size,size]; idx device int64 (the first-stage
 * selection list), n_sel device int32 (its length; rows b >= *n_sel of `out` are left untouched), n_max = rows the launch is sized
 * for; p->perm = the permutations of ONE update; out device fp32 [n_max,3,size,size]; workspace: device scratch of at least
 * ttl_plpd_views_workspace_bytes() bytes (may be NULL when that is 0). */
size_t ttl_plpd_views_workspace_bytes(int n_max, int size, const ttl_plpd_args* p);
int ttl_plpd_views(const float* x, int size, const int64_t* idx, const int* n_sel, int n_max, const ttl_plpd_args* p, float* out,
                   void* workspace, size_t workspace_bytes, void* stream);
/* keep_out device uint8 [N]: 1 for the first-stage views whose PLPD exceeds the threshold, 0 elsewhere (the `keep` argument of
 * ttl_entropy_select_loss); logits [N,K], logits_prime [n_max,K] (row b belongs to view idx[b]); plpd_out fp32 [n_max] or NULL. */
int ttl_plpd_keep(const float* logits, const float* logits_prime, const int64_t* idx, const int* n_sel, int n_max, int N, int K,
                  float threshold, unsigned char* keep_out, float* plpd_out, void* stream);

typedef struct ttl_episode_args {
    const float* x;      /* [N,3,S,S] */
    int n_views;
    int n_updates;       /* effective optimizer steps: tta_steps^2 on the DeYO branch (Q6) */
    int objective;       /* 0 = deyo (deyo.py:92-196), 1 = tpt (ttl.py:87-108) */
    int mode;            /* TTL_SEL_* (deyo objective) */
    double rho;          /* selection_p; int(N*rho) is evaluated in double like Python */
    float thresh, margin, reweight;
    float lr, beta1, beta2, eps, weight_decay;
    const float* snapshot;
    float* exp_avg;
    float* exp_avg_sq;
    float* logits0_out;
    float* logits1_out;
    /* accuracy(output, target, topk=(1, 5)) of utils/tools.py:88-102 on the adapted prediction, counted on the device (ttl.py:354-356
     * keeps AverageMeters on the host): target = device int64 [1] (the label), hits_out = device int64 [3] += {top-1 hit, top-5 hit
     * (top-min(5,K)), 1}.  Both NULL: no counting.  Ties rank the lower class index first; a label outside [0, K) never hits. */
    const int64_t* target;
    int64_t* hits_out;
    /* --filter_plpd 1 inside the fused episode (DeYO objective only); NULL = no PLPD stage.  Per update: first-stage selection ->
     * destroyed views -> second forward -> keep mask -> loss over the survivors.  ttl_debug_copy afterwards: "idx" = the FIRST-stage
     * list (n_candidates entries, the reference's order), "keep" = uint8 mask over the views, "plpd" = fp32 PLPD value per candidate,
     * "n_selected" = number of survivors (len(filter_ids_2)). */
    const ttl_plpd_args* plpd;
} ttl_episode_args;
int ttl_episode(ttl_ctx* ctx, const ttl_episode_args* args, void* stream);

/* The same episode captured into a HIP graph (stream capture of the launch sequence above) and replayed with one
 * hipGraphLaunch: the per-image host cost drops from ~130 kernel enqueues to one call.  Every pointer in `args`
 * (x, snapshot, exp_avg, exp_avg_sq, logits0/1_out) is baked into the graph: keep those buffers alive and refill x
 * in place.  `stream` must be an explicit stream; ttl_episode_capture also RUNS the episode once (warm-up). */
typedef struct ttl_graph ttl_graph;
int ttl_episode_capture(ttl_ctx* ctx, const ttl_episode_args* args, void* stream, ttl_graph** out);
int ttl_graph_launch(ttl_graph* graph, void* stream);
void ttl_graph_destroy(ttl_graph* graph);

/* ---- --lora_encoder text (clip/custom_clip.py:602-607,615-616,672-678; ttl.py:143-147,190-192) ----
 * A context created with tower = TTL_TOWER_TEXT holds the HF text tower ("text_model.embeddings.token_embedding.weight",
 * "text_model.embeddings.position_embedding.weight", "text_model.encoder.layers.{i}.*", "text_model.final_layer_norm.*",
 * "text_projection.weight" through ttl_load_weight) and its q/v LoRA (ttl_bind_lora, same layout).  The image tower runs
 * on a second, ordinary context WITHOUT ttl_bind_lora (no adapters exist on it in this mode) and only supplies features. */
/* exp(logit_scale) used by the text context's logits. */
int ttl_set_logit_scale(ttl_ctx* text_ctx, float logit_scale_exp);
/* prompt_learner.tokenized_prompts (clip/custom_clip.py:655): ids int32 [n_prompts][context_length], host or device.
 * The pooled position of a prompt is the argmax of its ids (end-of-text), like HF CLIPTextTransformer.  Synchronises. */
int ttl_set_prompts(ttl_ctx* text_ctx, const int* ids, int n_prompts, void* stream);
/* Image features of the current views [n_views,E] (device), L2-normalised on the way in when normalize != 0
 * (clip/custom_clip.py:680); asynchronous. */
int ttl_set_image_features(ttl_ctx* text_ctx, const float* feats, int n_views, int normalize, float logit_scale_exp, void* stream);
/* get_text_features with grad (clip/custom_clip.py:651-663,677-678) + logits: logits_out [n_views,n_prompts] or NULL,
 * feats_out un-normalised text features [n_prompts,E] or NULL. */
int ttl_text_forward(ttl_ctx* text_ctx, int save_for_backward, float* logits_out, float* feats_out, void* stream);
/* dlogits [n_views,n_prompts] -> gradients of the bound text LoRA buffer. */
int ttl_text_backward_lora(ttl_ctx* text_ctx, const float* dlogits, void* stream);
/* Whole per-image episode in text mode: image features of args->x on image_ctx (no grad), then
 * n_updates x [text forward, loss, text LoRA backward, AdamW], then the adapted logits of view 0. */
int ttl_episode_text(ttl_ctx* text_ctx, ttl_ctx* image_ctx, const ttl_episode_args* args, void* stream);

/* ---- kernel-level entry points (used by the unit parity tests; same kernels as above) ---- */
/* C[M,N] = A[M,K](operand dtype, lda) * B[N,K]^T(operand dtype, ldb) -> fp32 C (ldc).  K % 64 == 0, N % 128 == 0. */
int ttl_gemm_nt(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N,
                     int K, void* stream);
/* Same product with the fused epilogues of the path (modeling_clip.py:309-311,333,346-350): epi 0 = fp32 C; 1 = operand-dtype C
 * (+ bias); 2 = fp32 C = resid + product + bias; 3 = operand-dtype C = quick_gelu(product + bias).  rows_allocated = rows every
 * output / resid buffer really has: >= round_up(M, 1280) selects the unguarded big-M kernels the episode uses, 0 the guarded ones. */
int ttl_gemm_nt_epi(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, int epi,
                    const float* bias, const float* resid, int ldr, int rows_allocated, void* stream);
/* The two fused forms the episode's wide projections use (csrc/gemm_huge.hip, csrc/gemm_big.hip), operand-dtype outputs:
 *   hm_T > 0 (q/k/v, modeling_clip.py:309-311): C is written head-major — row m = view * hm_T + t, column plane * D + head * 64 + d
 *            (D = N / 3) goes to C[((view * 3 + plane) * D + head * 64) * hm_T + t * 64 + d] (C holds ceil(M / hm_T) whole views);
 *            C2 must be NULL;
 *   hm_T == 0 (fc1, modeling_clip.py:346-348): C = quick_gelu(product + bias) and, if C2 != NULL, C2 = product + bias;
 *   hm_T == -1 (its backward: d/d u of g = quick_gelu(u) folded into the fc2 dgrad): C = product * quick_gelu'(C2), C2 = the saved
 *            pre-activation u (read only, ldc2), bias must be NULL.
 * rows_allocated as ttl_gemm_nt_epi.  TTL_EINVAL when the shape does not run on a big-M kernel (M < 1024, N % 256, ...). */
int ttl_gemm_nt_fused(const void* A, int lda, const void* B, int ldb, void* C, int ldc, void* C2, int ldc2, int M, int N, int K,
                      const float* bias, int hm_T, int rows_allocated, void* stream);
/* y = LayerNorm(x) over the last dim (fp32 in, fp32 out, optional mean/rstd [rows]). */
int ttl_layernorm_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                      float* rstd, int rows, int dim, float eps, void* stream);
/* fp32 -> operand dtype (round to nearest even). */
int ttl_cast_f32_operand(const float* src, void* dst, size_t n, void* stream);
/* softmax(q k^T / 8) v for head dim 64: qkv (operand dtype) [n*T, 3*H*64] (q | k | v), out [n*T, H*64],
 * lse fp32 [n,H,T] or NULL; causal != 0 masks keys after the query (text tower). */
int ttl_attention_fwd(const void* qkv, void* out, float* lse, int n_views, int tokens, int heads,
                      int causal, void* stream);
/* dq,dk,dv of the above: dqkv (operand dtype) [n*T, ld_dqkv] (dq | dk | dv); need_dk == 0 skips dk. */
int ttl_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                      int ld_dqkv, int n_views, int tokens, int heads, int need_dk, int causal, void* stream);

/* GPU-side view generator: replaces the host pipeline data/datautils.py:98-157 with aug_list = [] (Q13)
 * and ToTensor + Normalize of ttl.py:225-241; bit-exact with that pipeline's Pillow resampling.
 *   image_hwc device uint8 [height][width][3]; boxes device int32 [n_views][5] = top, left, h, w, flags
 *   (crop boxes in source pixels; ttl_amd/views.py draws them like torchvision's RandomResizedCrop, the
 *   boxes must lie inside the image); flags bit0 = mirror horizontally, bit1 = base view:
 *   Resize(shorter side -> size, bicubic) over the whole image + CenterCrop(size), box ignored.
 *   Other views: crop, then bilinear resize to size x size.  out device fp32 [n_views,3,size,size];
 *   mean/std are HOST pointers to 3 floats; workspace is device scratch of at least
 *   ttl_make_views_workspace_bytes() bytes (tap tables). */
size_t ttl_make_views_workspace_bytes(int height, int width, int n_views, int size);
int ttl_make_views(const unsigned char* image_hwc, int height, int width, const int* boxes, int n_views, int size,
                   const float mean[3], const float stdv[3], float* out, void* workspace, size_t workspace_bytes,
                   void* stream);

/* Copy an internal buffer to the host (synchronises): name in {"h_in","h_mid","h_out","qkv",
 * "attn_out","x1","u","lse","dh","features","lora_grads"}; layer indexes encoder layers where it applies.
 * Also what the last fused update really used: "idx" (int64 selection list, the reference's order: deyo.py:103-108 /
 * ttl.py:50-54), "n_selected" (int32), "entropy" (fp32 per view). */
int ttl_debug_copy(ttl_ctx* ctx, const char* name, int layer, void* host_dst, size_t bytes);

/* Per-kernel-class device time of the calls made while profiling is on (HIP events on `stream`).
 * classes: 0 gemm (the big-M kernels, M >= 1024: 160x256 tiles of gemm_big.hip, 160x128 of gemm.hip), 1 attention fwd, 2 attention bwd, 3 layernorm/elementwise,
 * 4 lora, 5 head/loss/opt, 6 small-M gemm (1-view inference, CLS-row GEMMs: latency-bound 128x128 launches).
 * gemm_flops / ttl_profile_gemm_bytes cover class 0 only.
 * ttl_profile_read synchronises and returns accumulated milliseconds and launch counts. */
#define TTL_NCLASS 7
int ttl_profile_enable(ttl_ctx* ctx, int on);
int ttl_profile_read(ttl_ctx* ctx, double ms[TTL_NCLASS], long long launches[TTL_NCLASS],
                     double* gemm_flops);
/* Algorithmic bytes (operands + outputs + fused epilogue inputs, each once) of the GEMM launches covered by the
 * last ttl_profile_read: the denominator the measured HBM traffic of those launches is compared with. */
int ttl_profile_gemm_bytes(ttl_ctx* ctx, double* bytes);
/* 2*M*N*K summed over ALL GEMM launches (classes 0 and 6) covered by the last ttl_profile_read: the executed matrix FLOPs. */
int ttl_profile_gemm_flops_all(ttl_ctx* ctx, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* TTL_HIP_H */
