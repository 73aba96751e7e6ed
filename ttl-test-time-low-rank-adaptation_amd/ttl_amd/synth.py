"""Deterministic synthetic weights / inputs shared by the golden generator, the oracle
tests, the GPU parity tests and bench.py.

There is no network on the build or GPU boxes, so the pretrained
``openai/clip-vit-base-patch16`` checkpoint (clip/custom_clip.py:581) cannot be fetched.
Every tensor below is a pure function of (seed, tensor name) through numpy's PCG64
stream, so the golden script can load the *same* weights into the HF ``CLIPModel`` the
reference builds, and the fixtures only need to store a checksum.

Tensor names are the HF vision-tower names the reference reaches into
(SURVEY.md appendix B): ``vision_model.embeddings.*``, ``vision_model.pre_layrnorm`` (sic),
``vision_model.encoder.layers.{i}.*``, ``vision_model.post_layernorm``, ``visual_projection``.
"""
import hashlib
import zlib

import numpy as np

from .config import VitConfig

CLIP_MEAN = np.array([0.48145466, 0.4578275, 0.40821073], dtype=np.float32)  # ttl.py:225
CLIP_STD = np.array([0.26862954, 0.26130258, 0.27577711], dtype=np.float32)   # ttl.py:226


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed) & 0x7FFFFFFF, zlib.crc32(name.encode())])


def _normal(seed, name, shape, std, mean=0.0):
    return (_rng(seed, name).standard_normal(shape, dtype=np.float32) * np.float32(std)
            + np.float32(mean)).astype(np.float32)


def vision_weights(cfg: VitConfig, seed: int = 0, variant: str = None) -> dict:
    """fp32 state-dict (numpy) for the image tower + projection + logit_scale.

    Scales are chosen so that attention is peaked (not uniform), LN affine and biases are
    non-trivial and the residual stream stays O(1) through all layers: a flat synthetic
    model would hide indexing bugs that a peaked one exposes.

    variant="outliers": the same draws, then CLIP-like activation outliers (``add_activation_outliers``).
    """
    if variant not in (None, "", "outliers"):
        raise ValueError(f"unknown weights variant {variant!r}")
    w = _vision_weights_base(cfg, seed)
    if variant == "outliers":
        add_activation_outliers(cfg, w, seed)
    return w


OUTLIER_CHANNELS = 4


def outlier_layout(cfg: VitConfig, seed: int = 0):
    """(channels [4], signs [4], patch token index) of the 'outliers' variant: pure function of (cfg.width, cfg.tokens, seed)."""
    rng = _rng(seed, f"outliers{cfg.width}x{cfg.tokens}")
    ch = rng.choice(cfg.width, size=OUTLIER_CHANNELS, replace=False)
    sign = rng.choice(np.array([-1.0, 1.0], dtype=np.float32), size=OUTLIER_CHANNELS)
    tstar = 1 + int(rng.integers(0, cfg.tokens - 1))
    return ch, sign, tstar


def add_activation_outliers(cfg: VitConfig, w: dict, seed: int = 0) -> None:
    """In place: the activation statistics pretrained CLIP ViTs are known for (the checkpoint the reference loads,
    ``openai/clip-vit-base-patch16``, clip/custom_clip.py:581, cannot be fetched offline): a handful of residual channels carry
    30-100x the typical magnitude — on the CLS token and ONE patch token from the embeddings on (class / position embedding,
    amplified by the pre-LN gain), on EVERY token from an early MLP on (fc2 rows + bias of that channel), while the per-layer
    LayerNorm gains on those channels are small (the model damps them before every matmul) except one LN2 whose large gain
    feeds an outlier straight into fc1, and the post-LN gain is small (the pooled feature does not hang on them).
    Every 16-bit buffer of the HIP path (LN outputs, q/k/v, attention output, the fc1 pre-activation, the backward's
    counterparts) then holds values two orders of magnitude apart in one row."""
    ch, sign, tstar = outlier_layout(cfg, seed)
    pre = "vision_model."
    L = cfg.layers
    w[pre + "embeddings.class_embedding"][ch[:3]] += (60.0 * sign[:3]).astype(np.float32)
    w[pre + "embeddings.position_embedding.weight"][tstar, ch[1:]] += (45.0 * sign[1:]).astype(np.float32)
    w[pre + "pre_layrnorm.weight"][ch] = 4.0                # normalised outliers (~16 on the CLS row) -> ~64 in the residual stream
    w[pre + "post_layernorm.weight"][ch] = 0.02
    for i in range(L):
        lp = f"{pre}encoder.layers.{i}."
        for nm in ("layer_norm1", "layer_norm2"):
            w[lp + nm + ".weight"][ch] = 0.08
    mid = min(L - 1, max(1, L // 2))
    w[f"{pre}encoder.layers.{mid}.layer_norm2.weight"][ch[0]] = 2.5     # one LN feeds an outlier into fc1 at full size
    for i in sorted({min(2, L - 1), min(5, L - 1)}):                      # early MLPs put channel ch[3] on every token
        lp = f"{pre}encoder.layers.{i}."
        w[lp + "mlp.fc2.weight"][ch[3], :] *= 12.0
        w[lp + "mlp.fc2.bias"][ch[3]] += np.float32(25.0 * sign[3])


def _vision_weights_base(cfg: VitConfig, seed: int = 0) -> dict:
    D, F, E, P = cfg.width, cfg.mlp, cfg.embed, cfg.patch_size
    T = cfg.tokens
    w = {}
    pre = "vision_model."
    w[pre + "embeddings.class_embedding"] = _normal(seed, "cls", (D,), 0.5)
    w[pre + "embeddings.patch_embedding.weight"] = _normal(seed, "patch", (D, 3, P, P), 0.03)
    w[pre + "embeddings.position_embedding.weight"] = _normal(seed, "pos", (T, D), 0.1)
    for nm in ("pre_layrnorm", "post_layernorm"):
        w[pre + nm + ".weight"] = _normal(seed, nm + ".w", (D,), 0.1, 1.0)
        w[pre + nm + ".bias"] = _normal(seed, nm + ".b", (D,), 0.05)
    depth = float(cfg.layers)
    for i in range(cfg.layers):
        lp = f"{pre}encoder.layers.{i}."
        for nm in ("layer_norm1", "layer_norm2"):
            w[lp + nm + ".weight"] = _normal(seed, f"{i}.{nm}.w", (D,), 0.1, 1.0)
            w[lp + nm + ".bias"] = _normal(seed, f"{i}.{nm}.b", (D,), 0.05)
        qk_std = 1.2 / np.sqrt(D)        # q.k/8 has std ~1.4 -> peaked softmax
        for nm, std in (("q_proj", qk_std), ("k_proj", qk_std),
                        ("v_proj", 1.0 / np.sqrt(D)),
                        ("out_proj", 1.0 / np.sqrt(D) / np.sqrt(depth))):
            w[lp + f"self_attn.{nm}.weight"] = _normal(seed, f"{i}.{nm}.w", (D, D), std)
            w[lp + f"self_attn.{nm}.bias"] = _normal(seed, f"{i}.{nm}.b", (D,), 0.02)
        w[lp + "mlp.fc1.weight"] = _normal(seed, f"{i}.fc1.w", (F, D), 1.0 / np.sqrt(D))
        w[lp + "mlp.fc1.bias"] = _normal(seed, f"{i}.fc1.b", (F,), 0.02)
        w[lp + "mlp.fc2.weight"] = _normal(seed, f"{i}.fc2.w", (D, F),
                                           1.0 / np.sqrt(F) / np.sqrt(depth))
        w[lp + "mlp.fc2.bias"] = _normal(seed, f"{i}.fc2.b", (D,), 0.02)
    w["visual_projection.weight"] = _normal(seed, "proj", (E, D), 1.0 / np.sqrt(D))
    w["logit_scale"] = np.array(np.log(100.0), dtype=np.float32)  # pretrained value 4.6052
    return w


def text_weights(cfg, seed: int = 0) -> dict:
    """fp32 state-dict (numpy) of the text tower + text_projection (HF names: ``text_model.embeddings.*``,
    ``text_model.encoder.layers.{i}.*``, ``text_model.final_layer_norm``, ``text_projection``), same
    scale choices as vision_weights.  ``cfg`` is a config.TextConfig."""
    D, F, E, T, V = cfg.width, cfg.mlp, cfg.embed, cfg.context_length, cfg.vocab_size
    w = {}
    pre = "text_model."
    w[pre + "embeddings.token_embedding.weight"] = _normal(seed, "t.tok", (V, D), 0.5)
    w[pre + "embeddings.position_embedding.weight"] = _normal(seed, "t.pos", (T, D), 0.2)
    w[pre + "final_layer_norm.weight"] = _normal(seed, "t.final.w", (D,), 0.1, 1.0)
    w[pre + "final_layer_norm.bias"] = _normal(seed, "t.final.b", (D,), 0.05)
    depth = float(cfg.layers)
    for i in range(cfg.layers):
        lp = f"{pre}encoder.layers.{i}."
        for nm in ("layer_norm1", "layer_norm2"):
            w[lp + nm + ".weight"] = _normal(seed, f"t.{i}.{nm}.w", (D,), 0.1, 1.0)
            w[lp + nm + ".bias"] = _normal(seed, f"t.{i}.{nm}.b", (D,), 0.05)
        qk_std = 1.2 / np.sqrt(D)
        for nm, std in (("q_proj", qk_std), ("k_proj", qk_std), ("v_proj", 1.0 / np.sqrt(D)),
                        ("out_proj", 1.0 / np.sqrt(D) / np.sqrt(depth))):
            w[lp + f"self_attn.{nm}.weight"] = _normal(seed, f"t.{i}.{nm}.w", (D, D), std)
            w[lp + f"self_attn.{nm}.bias"] = _normal(seed, f"t.{i}.{nm}.b", (D,), 0.02)
        w[lp + "mlp.fc1.weight"] = _normal(seed, f"t.{i}.fc1.w", (F, D), 1.0 / np.sqrt(D))
        w[lp + "mlp.fc1.bias"] = _normal(seed, f"t.{i}.fc1.b", (F,), 0.02)
        w[lp + "mlp.fc2.weight"] = _normal(seed, f"t.{i}.fc2.w", (D, F), 1.0 / np.sqrt(F) / np.sqrt(depth))
        w[lp + "mlp.fc2.bias"] = _normal(seed, f"t.{i}.fc2.b", (D,), 0.02)
    w["text_projection.weight"] = _normal(seed, "t.proj", (E, D), 1.0 / np.sqrt(D))
    return w


def token_ids(n_prompts: int, cfg, seed: int = 0) -> np.ndarray:
    """[K, context] int32 CLIP-style token rows: <sot>=V-2, 3..12 random word tokens, <eot>=V-1 (the
    row maximum, which is how the pooled position is found: modeling_clip.py argmax(input_ids)), 0 pad."""
    rng = _rng(seed, f"ids{n_prompts}")
    T, V = cfg.context_length, cfg.vocab_size
    ids = np.zeros((n_prompts, T), dtype=np.int32)
    for k in range(n_prompts):
        n = int(rng.integers(3, 13))
        ids[k, 0] = V - 2
        ids[k, 1:1 + n] = rng.integers(1, V - 2, n)
        ids[k, 1 + n] = V - 1
    return ids


def lora_init(cfg, seed: int = 0, targets=None, tower="vision_model") -> dict:
    """LoRA A (xavier_normal: std = sqrt(2/(D+r)), clip/custom_clip.py:152-153,184-187) and
    B = 0 (peft default) for every layer, keyed like the reference's parameter names
    (``tower`` = "vision_model" or "text_model")."""
    D, r = cfg.width, cfg.rank
    std = np.sqrt(2.0 / (D + r))
    out = {}
    tag = "" if tower == "vision_model" else "t."
    if targets is None:
        from .config import ordered_targets
        targets = ordered_targets(cfg)
    for i in range(cfg.layers):
        for t in targets:
            base = f"{tower}.encoder.layers.{i}.self_attn.{t}."
            out[base + "lora_A.default.weight"] = _normal(seed, f"{tag}{i}.{t}.A", (r, D), std)
            out[base + "lora_B.default.weight"] = np.zeros((D, r), dtype=np.float32)
    return out


def views(cfg: VitConfig, n_views: int, seed: int = 0) -> np.ndarray:
    """[N,3,S,S] fp32 normalised 'uint8 noise' views (synthetic inputs).

    View 0 plays the un-augmented image (ttl.py:327); the others add per-view jitter to a
    shared low-frequency base so that view entropies differ but are correlated, like
    crops of one photo."""
    S = cfg.image_size
    rng = _rng(seed, f"views{n_views}x{S}")
    base = rng.integers(0, 256, size=(1, 3, S // 8, S // 8)).astype(np.float32)
    base = np.repeat(np.repeat(base, 8, axis=2), 8, axis=3)
    jit = rng.integers(-64, 65, size=(n_views, 3, S, S)).astype(np.float32)
    scale = rng.uniform(0.6, 1.0, size=(n_views, 1, 1, 1)).astype(np.float32)
    px = np.clip(base * scale + jit, 0, 255) / np.float32(255.0)
    px = (px - CLIP_MEAN[None, :, None, None]) / CLIP_STD[None, :, None, None]
    return np.ascontiguousarray(px.astype(np.float32))


def text_features(n_classes: int, embed: int, seed: int = 0) -> np.ndarray:
    """Unit-norm [K,E] class embeddings standing in for the cached text tower output
    (clip/custom_clip.py:651-663) when no text encoder is run."""
    t = _normal(seed, f"text{n_classes}x{embed}", (n_classes, embed), 1.0)
    t /= np.linalg.norm(t, axis=-1, keepdims=True)
    return t.astype(np.float32)


def checksum(arrs) -> str:
    """sha256 over a dict (sorted by key) or list of arrays; stored in fixtures."""
    h = hashlib.sha256()
    items = sorted(arrs.items()) if isinstance(arrs, dict) else enumerate(arrs)
    for k, a in items:
        h.update(str(k).encode())
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()
