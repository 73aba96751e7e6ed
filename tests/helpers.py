"""Shared helpers for the parity tests (CPU oracle side and GPU side)."""
import os

import numpy as np

from ttl_amd import synth
from ttl_amd.config import get_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_WEIGHTS = {}


def load_case(name):
    """-> (golden npz, cfg, W, x, lora0, tfeat) rebuilt from seeds and checked by sha256."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = get_config(str(g["arch"])).replace(rank=int(g["rank"]))
    if "lora_targets" in g.files:          # fixtures with adapters beyond the reference's q_proj / v_proj
        cfg = cfg.replace(lora_targets=tuple(str(t) for t in g["lora_targets"]))
    variant = str(g["weights_variant"]) if "weights_variant" in g.files else None      # "outliers": CLIP-like activation statistics
    wkey = (cfg.name, int(g["weight_seed"]), variant)
    if wkey not in _WEIGHTS:       # (fixtures of one model share the frozen weights: generate and checksum them once per process)
        W = synth.vision_weights(cfg, int(g["weight_seed"]), variant=variant)
        assert synth.checksum(W) == str(g["weights_sha256"]), "synthetic weights drifted from the fixture"
        if len(_WEIGHTS) >= 2:     # keep at most TWO models (ViT-L/14 is 1.2 GB of fp32)
            _WEIGHTS.pop(next(iter(_WEIGHTS)))
        _WEIGHTS[wkey] = (W, str(g["weights_sha256"]))
    W, sha = _WEIGHTS[wkey]
    assert sha == str(g["weights_sha256"]), "fixtures of one (arch, seed, variant) disagree on the weights"
    x = synth.views(cfg, int(g["n_views"]), int(g["view_seed"]))
    assert synth.checksum([x]) == str(g["x_sha256"]), "synthetic views drifted from the fixture"
    lora0 = synth.lora_init(cfg, 0)
    for k in g.files:
        if k.startswith("lora0/"):
            lora0[k[6:]] = g[k]
    return g, cfg, W, x, lora0, g["text_features"]


def episode_kwargs(g):
    return dict(objective=str(g["objective"]), mode=str(g["mode"]), rho=float(g["rho"]),
                margin=float(g["margin"]), n_updates=int(g["n_updates"]), lr=float(g["lr"]))


def max_rel(a, b):
    """max |a-b| / max |b|  (tensor-level relative error)."""
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()
                 / (np.abs(b).max() + 1e-30))


def adamw_first_step(p, g, lr, wd=1e-2, eps=1e-8):
    """Closed form of AdamW step 1 from zero state: p(1-lr*wd) - lr*g/(|g|+eps)."""
    p, g = np.asarray(p, np.float64), np.asarray(g, np.float64)
    return p * (1.0 - lr * wd) - lr * g / (np.abs(g) + eps)


def check_lora_step(new, ref, grad, lr, tol, name="", dg=None, eps=1e-8):
    """Compare post-step LoRA weights with the reference's.

    The first AdamW step from zero state is sign-like, f(g) = -lr*g/(|g|+eps): an
    element whose gradient magnitude is below the gradient error can land on the other side
    (+-lr).  For a gradient known to within +-dg the exact worst case is
    max(|f(g+dg)-f(g)|, |f(g-dg)-f(g)|) (f is monotone), which is ~lr*eps*dg/g^2 (negligible)
    where |g| >> dg and up to 2*lr where |g| <= dg.  ``dg`` is an absolute bound (callers pass
    the measured max gradient error); ``tol`` is relative to the tensor's max.
    With grad=None only ``tol`` applies."""
    new, ref = np.asarray(new, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    allowed = np.full(ref.shape, tol * scale)
    if grad is not None:
        g = np.asarray(grad, np.float64)
        d = tol * np.abs(g).max() if dg is None else dg
        f = lambda t: -lr * t / (np.abs(t) + eps)
        allowed = allowed + np.maximum(np.abs(f(g + d) - f(g)), np.abs(f(g - d) - f(g)))
    err = np.abs(new - ref)
    bad = err > allowed
    assert not bad.any(), (name, int(bad.sum()), float((err - allowed).max()))


def load_text_case(name):
    """--lora_encoder text fixtures -> (golden npz, vcfg, tcfg, Wv, Wt, x, ids, lora0) rebuilt from seeds."""
    from ttl_amd.config import get_text_config
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    arch = str(g["arch"])
    vcfg, tcfg = get_config(arch), get_text_config(arch)
    Wv = synth.vision_weights(vcfg, int(g["weight_seed"]))
    assert synth.checksum(Wv) == str(g["weights_sha256"]), "synthetic weights drifted from the fixture"
    Wt = synth.text_weights(tcfg, int(g["weight_seed"]))
    x = synth.views(vcfg, int(g["n_views"]), int(g["view_seed"]))
    ids = g["ids"]
    assert np.array_equal(ids, synth.token_ids(int(g["n_classes"]), tcfg, int(g["ids_seed"])))
    lora0 = synth.lora_init(tcfg, 0, tower="text_model")
    for k in g.files:
        if k.startswith("lora0/"):
            lora0[k[6:]] = g[k]
    return g, vcfg, tcfg, Wv, Wt, x, ids, lora0


def write_tiny_hf_checkpoint(path, arch="tiny", seed=0):
    """A locally WRITTEN HF-format CLIP checkpoint directory of a reduced geometry: config.json + model.safetensors from
    ``save_pretrained`` and vocab.json / merges.txt of a byte-level CLIP BPE (256 byte symbols, their word-final forms, the two
    specials) — what ``CLIPModel.from_pretrained`` / ``CLIPTokenizer.from_pretrained`` read at clip/custom_clip.py:581 of the
    reference, without any download.  -> (the CLIPModel that was saved, vocab dict)."""
    import json
    import torch
    from tokenizers import pre_tokenizers
    from transformers import CLIPConfig, CLIPModel
    from ttl_amd.config import get_text_config
    cfg, tcfg = get_config(arch), get_text_config(arch)
    chars = sorted(pre_tokenizers.ByteLevel.alphabet())
    vocab = {t: i for i, t in enumerate(chars + [c + "</w>" for c in chars] + ["<|startoftext|>", "<|endoftext|>"])}
    conf = CLIPConfig(
        vision_config=dict(hidden_size=cfg.width, intermediate_size=cfg.mlp, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                           image_size=cfg.image_size, patch_size=cfg.patch_size, projection_dim=cfg.embed, hidden_act="quick_gelu"),
        text_config=dict(hidden_size=tcfg.width, intermediate_size=tcfg.mlp, num_hidden_layers=tcfg.layers, num_attention_heads=tcfg.heads,
                         vocab_size=len(vocab), max_position_embeddings=77, projection_dim=cfg.embed, hidden_act="quick_gelu",
                         bos_token_id=len(vocab) - 2, eos_token_id=len(vocab) - 1, pad_token_id=len(vocab) - 1),
        projection_dim=cfg.embed)
    gen = torch.random.get_rng_state()
    torch.manual_seed(seed)
    ref = CLIPModel(conf).float().eval()
    with torch.no_grad():       # HF's default init gives near-identical features: spread the weights like synth.py does
        for n, p in ref.named_parameters():
            if p.dim() >= 2 and "embedding" not in n:
                p.normal_(0, 1.2 / p.shape[-1] ** 0.5)
    torch.random.set_rng_state(gen)
    os.makedirs(path, exist_ok=True)
    ref.save_pretrained(str(path))
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(path, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    with open(os.path.join(path, "tokenizer_config.json"), "w") as f:
        json.dump({"tokenizer_class": "CLIPTokenizer", "model_max_length": 77}, f)
    return ref, vocab
