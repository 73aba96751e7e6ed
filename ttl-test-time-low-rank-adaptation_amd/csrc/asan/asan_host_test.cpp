// `make asan` driver (CPU only; see asan/hip/hip_runtime.h): walks the host glue of libttl_hip — config validation, weight
// loading by name (every tensor, wrong sizes, unknown names, half-precision sources), shared contexts, the forward / backward /
// optimizer / fused-episode launch sequences for several geometries and adapter sets (the stub launch layer touches the extents
// of every operand), debug copies, graphs, the text tower, and the error paths — under AddressSanitizer.  Prints ASAN_HOST_OK.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../../include/ttl_hip.h"

static int g_fail = 0;
#define EXPECT(cond)                                                                  \
    do {                                                                              \
        if (!(cond)) { fprintf(stderr, "%s:%d: EXPECT(%s) failed: %s\n", __FILE__, __LINE__, #cond, ttl_last_error()); ++g_fail; } \
    } while (0)

static std::vector<float> rnd(size_t n, unsigned seed) {
    std::vector<float> v(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; v[i] = ((int)(s >> 9) % 2001 - 1000) * 1e-4f; }
    return v;
}

struct Geo { int S, P, D, H, F, L, E, r, lo, hi, N, K, targets; };

static ttl_config make_cfg(const Geo& g, int tower) {
    ttl_config c;
    memset(&c, 0, sizeof c);
    c.image_size = g.S; c.patch_size = g.P; c.width = g.D; c.heads = g.H; c.mlp = g.F; c.layers = g.L; c.embed = g.E; c.rank = g.r;
    c.lora_alpha = 32.f; c.layer_lo = g.lo; c.layer_hi = g.hi; c.ln_eps = 1e-5f; c.max_views = g.N; c.max_classes = g.K;
    c.tower = tower; c.context_length = 16; c.vocab_size = 64; c.lora_targets = g.targets;
    return c;
}

static int load_all(ttl_ctx* c, const Geo& g, int text) {
    const size_t D = g.D, F = g.F, E = g.E;
    const size_t T = text ? 16 : (size_t)(g.S / g.P) * (g.S / g.P) + 1;
    const std::string tw = text ? "text_model" : "vision_model";
    int rc = 0;
    unsigned seed = 1;
    auto ld = [&](const std::string& name, size_t n) { auto v = rnd(n, seed++); int r = ttl_load_weight(c, name.c_str(), v.data(), n); if (r) rc = r; };
    for (int i = 0; i < g.L; ++i) {
        const std::string b = tw + ".encoder.layers." + std::to_string(i) + ".";
        for (const char* pj : {"q_proj", "k_proj", "v_proj", "out_proj"}) { ld(b + "self_attn." + pj + ".weight", D * D); ld(b + "self_attn." + pj + ".bias", D); }
        ld(b + "mlp.fc1.weight", F * D); ld(b + "mlp.fc1.bias", F); ld(b + "mlp.fc2.weight", D * F); ld(b + "mlp.fc2.bias", D);
        for (const char* ln : {"layer_norm1", "layer_norm2"}) { ld(b + ln + ".weight", D); ld(b + ln + ".bias", D); }
    }
    if (text) {
        ld("text_model.embeddings.token_embedding.weight", 64 * D);
        ld("text_model.embeddings.position_embedding.weight", T * D);
        ld("text_model.final_layer_norm.weight", D); ld("text_model.final_layer_norm.bias", D);
        ld("text_projection.weight", E * D);
    } else {
        ld("vision_model.embeddings.class_embedding", D);
        ld("vision_model.embeddings.position_embedding.weight", T * D);
        ld("vision_model.embeddings.patch_embedding.weight", D * 3 * g.P * g.P);
        ld("vision_model.pre_layrnorm.weight", D); ld("vision_model.pre_layrnorm.bias", D);
        ld("vision_model.post_layernorm.weight", D); ld("vision_model.post_layernorm.bias", D);
        ld("visual_projection.weight", E * D);
    }
    return rc;
}

static int popcount4(int m) { return (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1) + ((m >> 3) & 1); }

static void image_tower(const Geo& g) {
    ttl_config cfg = make_cfg(g, TTL_TOWER_IMAGE);
    EXPECT(ttl_workspace_bytes(&cfg) > 0);
    ttl_ctx* c = nullptr;
    EXPECT(ttl_ctx_create(&cfg, &c) == 0 && c);
    if (!c) return;
    EXPECT(ttl_weights_ready(c) != 0);                                    // nothing loaded yet
    std::vector<float> x((size_t)g.N * 3 * g.S * g.S, 0.1f), logits((size_t)g.N * g.K), feats((size_t)g.N * g.E);
    EXPECT(ttl_vit_forward(c, x.data(), g.N, 0, logits.data(), nullptr, nullptr) != 0);     // weights missing
    EXPECT(load_all(c, g, 0) == 0);
    EXPECT(ttl_weights_ready(c) == 0);
    // ---- error paths of the loader
    auto w = rnd((size_t)g.D * g.D, 9);
    EXPECT(ttl_load_weight(c, "vision_model.encoder.layers.0.self_attn.q_proj.weight", w.data(), 7) != 0);          // wrong size
    EXPECT(ttl_load_weight(c, "vision_model.encoder.layers.99.mlp.fc1.bias", w.data(), g.F) != 0);                   // layer out of range
    EXPECT(ttl_load_weight(c, "vision_model.encoder.layers.0.mlp.fc3.weight", w.data(), 4) != 0);                    // unknown tensor
    EXPECT(ttl_load_weight(c, "no_such_tensor", w.data(), 4) != 0);
    EXPECT(ttl_load_weight(c, nullptr, w.data(), 4) != 0);
    std::vector<unsigned short> half((size_t)g.D * g.D, 0x3c00);                                                      // 1.0 in fp16
    EXPECT(ttl_load_weight_typed(c, "vision_model.encoder.layers.0.self_attn.q_proj.weight", half.data(), half.size(), TTL_DTYPE_F16) == 0);
    EXPECT(ttl_load_weight_typed(c, "vision_model.encoder.layers.0.self_attn.q_proj.weight", half.data(), half.size(), TTL_DTYPE_BF16) == 0);
    EXPECT(ttl_load_weight_typed(c, "vision_model.encoder.layers.0.self_attn.q_proj.weight", half.data(), half.size(), 77) != 0);
    // ---- peer features, LoRA binding
    auto tf = rnd((size_t)g.K * g.E, 3);
    EXPECT(ttl_vit_forward(c, x.data(), g.N, 0, logits.data(), nullptr, nullptr) != 0);     // no text features yet
    EXPECT(ttl_set_text_features(c, tf.data(), g.K + 1, 100.f, nullptr) != 0);              // over capacity
    EXPECT(ttl_set_text_features(c, tf.data(), g.K, 100.f, nullptr) == 0);
    const int ntg = popcount4(g.targets ? g.targets : 5);
    const size_t nl = (size_t)(g.hi - g.lo + 1) * ntg * 2 * g.r * g.D;
    auto lora = rnd(nl, 5);
    std::vector<float> grads(nl), snap = lora, m(nl, 0.f), v(nl, 0.f);
    EXPECT(ttl_vit_forward(c, x.data(), g.N, 1, logits.data(), nullptr, nullptr) != 0);     // save without bound adapters
    EXPECT(ttl_bind_lora(c, lora.data(), grads.data(), nl - 1) != 0);
    EXPECT(ttl_bind_lora(c, lora.data(), grads.data(), nl) == 0);
    // ---- forward / backward / step through the separate entry points, ragged view counts
    EXPECT(ttl_vit_forward(c, x.data(), g.N + 1, 0, logits.data(), nullptr, nullptr) != 0);
    EXPECT(ttl_vit_forward(c, x.data(), 0, 0, logits.data(), nullptr, nullptr) != 0);
    for (int n : {g.N, 1, g.N > 2 ? g.N - 1 : 1}) {
        EXPECT(ttl_vit_forward(c, x.data(), n, 1, logits.data(), feats.data(), nullptr) == 0);
        std::vector<float> dl((size_t)n * g.K, 1e-3f);
        EXPECT(ttl_vit_backward_lora(c, dl.data(), n + 1, nullptr) != 0);                   // no saved forward for that count
        EXPECT(ttl_vit_backward_lora(c, dl.data(), n, nullptr) == 0);
        EXPECT(ttl_optimizer_step(c, lora.data(), grads.data(), m.data(), v.data(), nl, 5e-3f, 0.9f, 0.999f, 1e-8f, 1e-2f, 1, nullptr, nullptr) == 0);
    }
    EXPECT(ttl_head_logits(c, feats.data(), g.N, logits.data(), nullptr) == 0);
    EXPECT(ttl_head_logits(c, feats.data(), g.N + 1, logits.data(), nullptr) != 0);
    float scale; int tr, sk, st;
    EXPECT(ttl_scaler_config(c, 1, 1024.f, 2.f, 0.5f, 2000) == 0);
    EXPECT(ttl_scaler_state(c, &scale, &tr, &sk, &st) == 0);
    EXPECT(ttl_scaler_unscale(c, grads.data(), nl, nullptr) == 0);
    // ---- the fused episode, 1 and 3 updates, both objectives; then as a graph
    std::vector<float> l0((size_t)g.N * g.K), l1(g.K);
    ttl_episode_args a;
    memset(&a, 0, sizeof a);
    a.x = x.data(); a.n_views = g.N; a.n_updates = 1; a.objective = 0; a.mode = TTL_SEL_LE_THRESH; a.rho = 0.1; a.thresh = 6.9f; a.margin = 0.4f;
    a.reweight = 1.f; a.lr = 5e-3f; a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f; a.weight_decay = 1e-2f;
    a.snapshot = snap.data(); a.exp_avg = m.data(); a.exp_avg_sq = v.data(); a.logits0_out = l0.data(); a.logits1_out = l1.data();
    EXPECT(ttl_episode(c, &a, nullptr) == 0);
    a.n_updates = 3; a.objective = 1; a.mode = TTL_SEL_TOPK;
    EXPECT(ttl_episode(c, &a, nullptr) == 0);
    a.n_views = g.N + 1;
    EXPECT(ttl_episode(c, &a, nullptr) != 0);
    a.n_views = g.N; a.n_updates = 1;
    ttl_graph* gr = nullptr;
    EXPECT(ttl_episode_capture(c, &a, (void*)0x1, &gr) == 0 && gr);
    if (gr) { EXPECT(ttl_graph_launch(gr, (void*)0x1) == 0); ttl_graph_destroy(gr); }
    // ---- debug copies of a saved forward
    EXPECT(ttl_vit_forward(c, x.data(), g.N, 1, logits.data(), nullptr, nullptr) == 0);
    const size_t T = (size_t)(g.S / g.P) * (g.S / g.P) + 1, M = (size_t)g.N * T;
    std::vector<unsigned short> q(M * 3 * g.D);
    EXPECT(ttl_debug_copy(c, "qkv", g.lo, q.data(), q.size() * 2) == 0);
    EXPECT(ttl_debug_copy(c, "qkv", g.lo, q.data(), q.size() * 2 + 2) != 0);                // more than the buffer holds
    EXPECT(ttl_debug_copy(c, "qkv", g.L + 3, q.data(), 16) != 0);
    EXPECT(ttl_debug_copy(c, "nonsense", g.lo, q.data(), 16) != 0);
    std::vector<float> hbuf(M * g.D);
    EXPECT(ttl_debug_copy(c, "h_in", g.lo, hbuf.data(), hbuf.size() * 4) == 0);
    EXPECT(ttl_debug_copy(c, "features", 0, feats.data(), feats.size() * 4) == 0);
    double ms[TTL_NCLASS]; long long cnt[TTL_NCLASS]; double fl;
    EXPECT(ttl_profile_enable(c, 1) == 0);
    EXPECT(ttl_episode(c, &a, nullptr) == 0);
    EXPECT(ttl_profile_read(c, ms, cnt, &fl) == 0);
    EXPECT(ttl_profile_enable(c, 0) == 0);
    // ---- a second context on the same weight images
    ttl_ctx* sh = nullptr;
    ttl_config other = cfg; other.max_views = g.N > 1 ? g.N - 1 : 1;
    EXPECT(ttl_ctx_create_shared(&other, c, &sh) == 0 && sh);
    if (sh) {
        EXPECT(ttl_weights_ready(sh) == 0);
        EXPECT(ttl_load_weight(sh, "visual_projection.weight", w.data(), (size_t)g.E * g.D) != 0);                   // shared: read-only
        ttl_ctx* sh2 = nullptr;
        EXPECT(ttl_ctx_create_shared(&other, sh, &sh2) != 0 && !sh2);                                                // parent must own
        EXPECT(ttl_set_text_features(sh, tf.data(), g.K, 100.f, nullptr) == 0);
        std::vector<float> lora2 = snap, grads2(nl);
        EXPECT(ttl_bind_lora(sh, lora2.data(), grads2.data(), nl) == 0);
        ttl_episode_args b = a; b.n_views = other.max_views;
        EXPECT(ttl_episode(sh, &b, nullptr) == 0);
        ttl_ctx_destroy(sh);
    }
    // ---- --filter_plpd 1 inside the fused episode: auxiliary context on the shared weights, bound to the SAME adapter buffer
    {
        ttl_ctx* aux = nullptr;
        EXPECT(ttl_ctx_create_shared(&cfg, c, &aux) == 0 && aux);
        std::vector<float> grads_aux(nl);
        std::vector<int> perm((size_t)3 * g.N * 16 > (size_t)3 * g.S * g.S ? (size_t)3 * g.N * 16 : (size_t)3 * g.S * g.S, 0);
        ttl_plpd_args pp;
        memset(&pp, 0, sizeof pp);
        pp.aug_type = TTL_PLPD_PATCH; pp.threshold = 0.2f; pp.patch_len = 4; pp.perm = perm.data(); pp.n_candidates = g.N; pp.aux = aux;
        ttl_episode_args e = a;
        e.objective = 0; e.mode = TTL_SEL_LE_THRESH; e.n_updates = 3; e.plpd = &pp;
        EXPECT(ttl_episode(c, &e, nullptr) != 0);                                                  // aux not bound / no class embeddings yet
        EXPECT(ttl_set_text_features(aux, tf.data(), g.K, 100.f, nullptr) == 0);
        EXPECT(ttl_bind_lora(aux, lora.data(), grads_aux.data(), nl) == 0);
        EXPECT(ttl_episode(c, &e, nullptr) == 0);
        pp.patch_len = 3;                                                                           // S % 3 != 0: the two resize stages + scratch
        EXPECT(ttl_episode(c, &e, nullptr) == 0);
        pp.aug_type = TTL_PLPD_PIXEL;
        EXPECT(ttl_episode(c, &e, nullptr) == 0);
        pp.aug_type = TTL_PLPD_OCC; pp.perm = nullptr; pp.occlusion_size = g.S / 2; pp.row_start = 1; pp.column_start = 2;
        EXPECT(ttl_episode(c, &e, nullptr) == 0);
        pp.row_start = g.S;                                                                         // window outside the view
        EXPECT(ttl_episode(c, &e, nullptr) != 0);
        pp.row_start = 1; pp.n_candidates = g.N + 1;
        EXPECT(ttl_episode(c, &e, nullptr) != 0);
        pp.n_candidates = g.N; pp.aux = c;                                                          // the saving context cannot serve itself
        EXPECT(ttl_episode(c, &e, nullptr) != 0);
        pp.aux = aux; e.objective = 1;                                                              // TPT has no PLPD stage
        EXPECT(ttl_episode(c, &e, nullptr) != 0);
        e.objective = 0; pp.aug_type = TTL_PLPD_PATCH; pp.patch_len = 4;                            // 'patch' without permutations
        EXPECT(ttl_episode(c, &e, nullptr) != 0);
        // the pieces on their own
        std::vector<long long> idx(g.N);
        for (int i = 0; i < g.N; ++i) idx[i] = g.N - 1 - i;
        int nsel = g.N;
        std::vector<float> xp((size_t)g.N * 3 * g.S * g.S);
        pp.aug_type = TTL_PLPD_OCC;
        const size_t wsb = ttl_plpd_views_workspace_bytes(g.N, g.S, &pp);
        std::vector<char> ws(wsb ? wsb : 1);
        EXPECT(ttl_plpd_views(x.data(), g.S, (const int64_t*)idx.data(), &nsel, g.N, &pp, xp.data(), ws.data(), wsb, nullptr) == 0);
        EXPECT(ttl_plpd_views(x.data(), g.S, (const int64_t*)idx.data(), &nsel, g.N, &pp, xp.data(), ws.data(), wsb ? wsb - 1 : 0, nullptr) != 0 || wsb == 0);
        std::vector<unsigned char> keep(g.N);
        std::vector<float> pv(g.N);
        EXPECT(ttl_plpd_keep(l0.data(), l0.data(), (const int64_t*)idx.data(), &nsel, g.N, g.N, g.K, 0.2f, keep.data(), pv.data(), nullptr) == 0);
        EXPECT(ttl_plpd_keep(l0.data(), l0.data(), (const int64_t*)idx.data(), &nsel, g.N + 1, g.N, g.K, 0.2f, keep.data(), pv.data(), nullptr) != 0);
        ttl_ctx_destroy(aux);
    }
    ttl_config bad = cfg; bad.rank = 8;
    EXPECT(ttl_ctx_create_shared(&bad, c, &sh) != 0);                     // invalid config
    bad = cfg; bad.layer_lo = cfg.layer_lo > 0 ? cfg.layer_lo - 1 : cfg.layer_lo + 1; bad.layer_hi = cfg.layer_hi;
    if (bad.layer_lo <= bad.layer_hi) EXPECT(ttl_ctx_create_shared(&bad, c, &sh) != 0);                              // another model
    EXPECT(ttl_ctx_create_shared(&cfg, nullptr, &sh) != 0);
    // ---- destroy order: the owner goes FIRST, the sharing context keeps computing on the (reference-counted) weight images
    ttl_ctx* late = nullptr;
    EXPECT(ttl_ctx_create_shared(&cfg, c, &late) == 0 && late);
    ttl_ctx_destroy(c);
    if (late) {
        EXPECT(ttl_set_text_features(late, tf.data(), g.K, 100.f, nullptr) == 0);
        std::vector<float> lora3 = snap, grads3(nl);
        EXPECT(ttl_bind_lora(late, lora3.data(), grads3.data(), nl) == 0);
        EXPECT(ttl_episode(late, &a, nullptr) == 0);
        ttl_ctx_destroy(late);
    }
}

static void text_tower(const Geo& g) {
    ttl_config tc = make_cfg(g, TTL_TOWER_TEXT), ic = make_cfg(g, TTL_TOWER_IMAGE);
    tc.max_views = g.K; tc.max_classes = g.N;         // text tower: capacity in prompts / in views
    ttl_ctx *t = nullptr, *im = nullptr;
    EXPECT(ttl_ctx_create(&tc, &t) == 0 && t);
    EXPECT(ttl_ctx_create(&ic, &im) == 0 && im);
    if (!t || !im) return;
    EXPECT(load_all(t, g, 1) == 0 && load_all(im, g, 0) == 0);
    EXPECT(ttl_weights_ready(t) == 0 && ttl_weights_ready(im) == 0);
    std::vector<int> ids((size_t)g.K * 16, 1);
    for (int p = 0; p < g.K; ++p) ids[(size_t)p * 16 + 3 + p % 11] = 63;       // end-of-text = arg-max, at different positions
    EXPECT(ttl_text_forward(t, 0, nullptr, nullptr, nullptr) != 0);            // no prompts yet
    EXPECT(ttl_set_prompts(t, ids.data(), g.K + 1, nullptr) != 0);
    EXPECT(ttl_set_prompts(t, ids.data(), g.K, nullptr) == 0);
    EXPECT(ttl_set_logit_scale(t, 100.f) == 0);
    EXPECT(ttl_set_text_features(t, nullptr, 1, 1.f, nullptr) != 0);
    const int ntg = popcount4(g.targets ? g.targets : 5);
    const size_t nl = (size_t)(g.hi - g.lo + 1) * ntg * 2 * g.r * g.D;
    auto lora = rnd(nl, 6);
    std::vector<float> grads(nl), snap = lora, m(nl, 0.f), v(nl, 0.f), x((size_t)g.N * 3 * g.S * g.S, 0.2f), l0((size_t)g.N * g.K), l1(g.K);
    EXPECT(ttl_bind_lora(t, lora.data(), grads.data(), nl) == 0);
    auto f = rnd((size_t)g.N * g.E, 8);
    EXPECT(ttl_set_image_features(t, f.data(), g.N + 1, 1, 100.f, nullptr) != 0);
    EXPECT(ttl_set_image_features(t, f.data(), g.N, 1, 100.f, nullptr) == 0);
    EXPECT(ttl_text_forward(t, 1, l0.data(), nullptr, nullptr) == 0);
    std::vector<float> dl((size_t)g.N * g.K, 1e-3f);
    EXPECT(ttl_text_backward_lora(t, dl.data(), nullptr) == 0);
    EXPECT(ttl_vit_forward(t, x.data(), 1, 0, l0.data(), nullptr, nullptr) != 0);           // wrong tower
    ttl_episode_args a;
    memset(&a, 0, sizeof a);
    a.x = x.data(); a.n_views = g.N; a.n_updates = 2; a.mode = TTL_SEL_LE_THRESH; a.rho = 0.1; a.thresh = 6.9f; a.margin = 0.4f; a.reweight = 1.f;
    a.lr = 5e-3f; a.beta1 = 0.9f; a.beta2 = 0.999f; a.eps = 1e-8f; a.weight_decay = 1e-2f;
    a.snapshot = snap.data(); a.exp_avg = m.data(); a.exp_avg_sq = v.data(); a.logits0_out = l0.data(); a.logits1_out = l1.data();
    EXPECT(ttl_episode_text(t, im, &a, nullptr) == 0);
    EXPECT(ttl_episode_text(im, t, &a, nullptr) != 0);                         // contexts swapped
    ttl_ctx_destroy(t);
    ttl_ctx_destroy(im);
}

int main() {
    ttl_config c;
    memset(&c, 0, sizeof c);
    ttl_ctx* ctx = nullptr;
    EXPECT(ttl_ctx_create(nullptr, &ctx) != 0);
    EXPECT(ttl_ctx_create(&c, nullptr) != 0);
    EXPECT(ttl_ctx_create(&c, &ctx) != 0 && !ctx);                              // all-zero config
    EXPECT(ttl_workspace_bytes(&c) == 0);
    const Geo tiny = {64, 16, 128, 2, 512, 4, 64, 16, 1, 3, 8, 10, 0};
    Geo g = tiny;
    for (int bad = 0; bad < 8; ++bad) {                                         // one invalid field at a time
        ttl_config k = make_cfg(tiny, TTL_TOWER_IMAGE);
        switch (bad) {
            case 0: k.width = 100; break;
            case 1: k.heads = 3; break;
            case 2: k.rank = 24; break;
            case 3: k.layer_lo = 3; k.layer_hi = 1; break;
            case 4: k.layer_hi = 4; break;
            case 5: k.max_views = 0; break;
            case 6: k.lora_targets = 16; break;
            case 7: k.tower = 5; break;
        }
        EXPECT(ttl_ctx_create(&k, &ctx) != 0 && !ctx);
    }
    image_tower(tiny);                                                          // reference adapters (q, v), small-M kernels only
    g = tiny; g.targets = 15; g.r = 32; image_tower(g);                         // q, k, v, out at rank 32: the 128-column K-extension
    g = tiny; g.lo = 1; g.hi = 2; g.targets = 10; image_tower(g);               // adapters stop below the top layer; k + out only
    g = tiny; g.lo = 0; g.hi = 3; image_tower(g);                               // every layer trained
    g = tiny; g.S = 224; g.N = 6; image_tower(g);                               // T = 197, M = 1182 >= 1024: big-M kernel, head-major q/k/v
    g = tiny; g.S = 224; g.N = 6; g.targets = 15; image_tower(g);
    g = tiny; g.K = 40; g.N = 4; text_tower(g);
    g = tiny; g.K = 80; g.N = 4; g.targets = 15; text_tower(g);                 // 80 prompts x 16 tokens = 1280 rows: big-M text launches
    ttl_ctx_destroy(nullptr);
    if (g_fail) { fprintf(stderr, "%d expectation(s) failed\n", g_fail); return 1; }
    printf("ASAN_HOST_OK %s\n", ttl_version());
    return 0;
}
