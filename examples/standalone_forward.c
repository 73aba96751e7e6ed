/* Plain-C host for libttl_hip.so: no Python, no torch — only the C ABI of include/ttl_hip.h and the HIP runtime
 * for device buffers.  Reads a small binary bundle (written by tests/test_gpu_standalone_c.py), runs
 * model(images) and one fused episode, and writes the logits back.
 *
 *   gcc -O2 -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ examples/standalone_forward.c \
 *       -L /opt/rocm/lib -lamdhip64 -ldl -o standalone_forward
 *   ./standalone_forward path/to/libttl_hip.so bundle.bin out.bin
 *
 * Bundle layout (little endian): int32 cfg[14] in ttl_config order (floats bit-cast), int32 n_views, n_classes,
 * n_tensors; then n_tensors x { int32 name_len, name bytes, int64 count, float data[count] }; then
 * text features [K,E], logit scale (float), lora [n_lora], views [N,3,S,S].
 */
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ttl_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define LOAD(sym) *(void**)(&p_##sym) = dlsym(h, #sym); if (!p_##sym) { fprintf(stderr, "missing %s\n", #sym); return 2; }
#define CHECK_TTL(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, p_ttl_last_error()); return 3; } } while (0)

static const char* (*p_ttl_last_error)(void);
static int (*p_ttl_ctx_create)(const ttl_config*, ttl_ctx**);
static void (*p_ttl_ctx_destroy)(ttl_ctx*);
static int (*p_ttl_load_weight)(ttl_ctx*, const char*, const float*, size_t);
static int (*p_ttl_weights_ready)(ttl_ctx*);
static int (*p_ttl_set_text_features)(ttl_ctx*, const float*, int, float, void*);
static int (*p_ttl_bind_lora)(ttl_ctx*, float*, float*, size_t);
static int (*p_ttl_vit_forward)(ttl_ctx*, const float*, int, int, float*, float*, void*);
static int (*p_ttl_episode)(ttl_ctx*, const ttl_episode_args*, void*);

static int rd(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes ? 0 : -1; }

int main(int argc, char** argv) {
    if (argc != 4) { fprintf(stderr, "usage: %s libttl_hip.so bundle.bin out.bin\n", argv[0]); return 1; }
    void* h = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(ttl_last_error) LOAD(ttl_ctx_create) LOAD(ttl_ctx_destroy) LOAD(ttl_load_weight) LOAD(ttl_weights_ready)
    LOAD(ttl_set_text_features) LOAD(ttl_bind_lora) LOAD(ttl_vit_forward) LOAD(ttl_episode)

    FILE* f = fopen(argv[2], "rb");
    if (!f) { perror("bundle"); return 1; }
    ttl_config cfg;
    memset(&cfg, 0, sizeof cfg);
    int32_t raw[14], nv, nk, nt;
    if (rd(f, raw, sizeof raw) || rd(f, &nv, 4) || rd(f, &nk, 4) || rd(f, &nt, 4)) return 1;
    cfg.image_size = raw[0]; cfg.patch_size = raw[1]; cfg.width = raw[2]; cfg.heads = raw[3]; cfg.mlp = raw[4];
    cfg.layers = raw[5]; cfg.embed = raw[6]; cfg.rank = raw[7]; memcpy(&cfg.lora_alpha, &raw[8], 4);
    cfg.layer_lo = raw[9]; cfg.layer_hi = raw[10]; memcpy(&cfg.ln_eps, &raw[11], 4); cfg.max_views = raw[12]; cfg.max_classes = raw[13];
    cfg.tower = TTL_TOWER_IMAGE;

    ttl_ctx* ctx = NULL;
    CHECK_TTL(p_ttl_ctx_create(&cfg, &ctx));
    for (int i = 0; i < nt; ++i) {
        int32_t nl; int64_t cnt; char name[256];
        if (rd(f, &nl, 4) || nl <= 0 || nl >= 255 || rd(f, name, nl) || rd(f, &cnt, 8)) return 1;
        name[nl] = 0;
        float* buf = (float*)malloc((size_t)cnt * 4);
        if (!buf || rd(f, buf, (size_t)cnt * 4)) return 1;
        CHECK_TTL(p_ttl_load_weight(ctx, name, buf, (size_t)cnt));      /* host pointer: the library stages it */
        free(buf);
    }
    CHECK_TTL(p_ttl_weights_ready(ctx));

    const size_t E = cfg.embed, S = cfg.image_size;
    const size_t n_lora = (size_t)(cfg.layer_hi - cfg.layer_lo + 1) * 4 * cfg.rank * cfg.width;
    const size_t n_x = (size_t)nv * 3 * S * S;
    float* tf = (float*)malloc(nk * E * 4); float scale; float* lora = (float*)malloc(n_lora * 4); float* x = (float*)malloc(n_x * 4);
    if (rd(f, tf, nk * E * 4) || rd(f, &scale, 4) || rd(f, lora, n_lora * 4) || rd(f, x, n_x * 4)) return 1;
    fclose(f);

    float *d_x, *d_lora, *d_grad, *d_snap, *d_m, *d_v, *d_logits, *d_l1;
    CHECK_HIP(hipMalloc((void**)&d_x, n_x * 4)); CHECK_HIP(hipMalloc((void**)&d_lora, n_lora * 4)); CHECK_HIP(hipMalloc((void**)&d_grad, n_lora * 4));
    CHECK_HIP(hipMalloc((void**)&d_snap, n_lora * 4)); CHECK_HIP(hipMalloc((void**)&d_m, n_lora * 4)); CHECK_HIP(hipMalloc((void**)&d_v, n_lora * 4));
    CHECK_HIP(hipMalloc((void**)&d_logits, (size_t)nv * nk * 4)); CHECK_HIP(hipMalloc((void**)&d_l1, (size_t)nk * 4));
    CHECK_HIP(hipMemcpy(d_x, x, n_x * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_lora, lora, n_lora * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_snap, lora, n_lora * 4, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(d_grad, 0, n_lora * 4)); CHECK_HIP(hipMemset(d_m, 0, n_lora * 4)); CHECK_HIP(hipMemset(d_v, 0, n_lora * 4));

    hipStream_t s;
    CHECK_HIP(hipStreamCreate(&s));
    CHECK_TTL(p_ttl_set_text_features(ctx, tf, nk, scale, s));
    CHECK_TTL(p_ttl_bind_lora(ctx, d_lora, d_grad, n_lora));
    CHECK_TTL(p_ttl_vit_forward(ctx, d_x, nv, 0, d_logits, NULL, s));                      /* model(images) */
    ttl_episode_args a;
    memset(&a, 0, sizeof a);
    a.x = d_x; a.n_views = nv; a.n_updates = 1; a.objective = 0; a.mode = TTL_SEL_LE_THRESH; a.rho = 0.1;
    a.thresh = 6.907755f; a.margin = 0.4f; a.reweight = 1.0f;
    a.lr = 5e-3f; a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f; a.weight_decay = 1e-2f;
    a.snapshot = d_snap; a.exp_avg = d_m; a.exp_avg_sq = d_v; a.logits0_out = NULL; a.logits1_out = d_l1;
    CHECK_TTL(p_ttl_episode(ctx, &a, s));                                                   /* ttl.py:338-352 */
    CHECK_HIP(hipStreamSynchronize(s));

    float* out = (float*)malloc(((size_t)nv * nk + nk) * 4);
    CHECK_HIP(hipMemcpy(out, d_logits, (size_t)nv * nk * 4, hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(out + (size_t)nv * nk, d_l1, (size_t)nk * 4, hipMemcpyDeviceToHost));
    FILE* o = fopen(argv[3], "wb");
    if (!o || fwrite(out, 4, (size_t)nv * nk + nk, o) != (size_t)nv * nk + nk) return 1;
    fclose(o);
    p_ttl_ctx_destroy(ctx);
    CHECK_HIP(hipStreamDestroy(s));
    CHECK_HIP(hipFree(d_x)); CHECK_HIP(hipFree(d_lora)); CHECK_HIP(hipFree(d_grad)); CHECK_HIP(hipFree(d_snap));
    CHECK_HIP(hipFree(d_m)); CHECK_HIP(hipFree(d_v)); CHECK_HIP(hipFree(d_logits)); CHECK_HIP(hipFree(d_l1));
    free(out); free(tf); free(lora); free(x);
    printf("standalone ok: %d views, %d classes\n", nv, nk);
    return 0;
}
