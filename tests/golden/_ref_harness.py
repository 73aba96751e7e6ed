"""Import plumbing that lets the *unmodified* reference (/root/reference) run in the build
container, which lacks torchvision / peft / ftfy / pretrained weights.

Used only by tests/golden/make_golden.py to produce the committed fixtures.  Nothing here
is shipped, and nothing here is read on the GPU box.

What is stubbed and why it does not touch the arithmetic of the path:
  * torchvision, ftfy, cv2: import-time names only (class names are ASCII; no transform or
    dataset is ever executed on the path under test).
  * peft: minimal stand-in implementing the documented LoRA-Linear semantics the reference
    relies on (lora_A/lora_B ModuleDict{"default"}, scaling = alpha/r, B zero-init,
    dropout inactive in eval, get_peft_model(m,cfg).base_model.model is m,
    prepare_model_for_int8_training = freeze all + fp32 cast).
  * CLIPModel.from_pretrained: returns a CLIPModel of the requested geometry whose vision
    tower / projection / logit_scale are loaded from ttl_amd.synth.vision_weights(seed)
    (no network); get_image_features/get_text_features unwrap .pooler_output (Q17).
  * clip.load: tiny stand-in exposing only the attributes PromptLearner touches.
"""
import math
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None


def install_import_stubs():
    import transformers  # noqa: F401  (must be imported before torchvision is faked)

    class _Interp:
        BICUBIC = 3
        BILINEAR = 2
    names = ["Compose", "Resize", "CenterCrop", "ToTensor", "Normalize", "RandomResizedCrop",
             "RandomHorizontalFlip", "Lambda", "ToPILImage"]
    class _Resize:   # torchvision.transforms.Resize on a tensor (>= 0.17): bilinear + antialias
        def __init__(self, size, *a, **k):
            self.size = tuple(size) if isinstance(size, (tuple, list)) else (size, size)

        def __call__(self, x):
            import torch.nn.functional as Fn
            return Fn.interpolate(x, size=self.size, mode="bilinear", align_corners=False, antialias=True)
    tfm = _mod("torchvision.transforms", InterpolationMode=_Interp,
               **{n: type(n, (_Anything,), {}) for n in names})
    tfm.Resize = _Resize
    models = _mod("torchvision.models")
    dsets = _mod("torchvision.datasets", ImageFolder=type("ImageFolder", (_Anything,), {}),
                 VisionDataset=type("VisionDataset", (_Anything,), {}))
    utils = _mod("torchvision.utils")
    tv = _mod("torchvision", transforms=tfm, models=models, datasets=dsets, utils=utils)
    tv.__path__ = []
    _mod("ftfy", fix_text=lambda s: s)
    _mod("cv2")

    # ---- peft stand-in -------------------------------------------------------------
    class LoraConfig:
        def __init__(self, r=8, lora_alpha=8, target_modules=None, lora_dropout=0.0,
                     bias="none", task_type=None, **kw):
            self.r, self.lora_alpha = r, lora_alpha
            # TARGET_MODULES_OVERRIDE: harness-side override of the list the reference hard-codes (clip/custom_clip.py:586),
            # used ONLY by the k_proj / out_proj fixtures of make_golden.py (the reference itself is not edited)
            self.target_modules, self.lora_dropout = list(TARGET_MODULES_OVERRIDE or target_modules), lora_dropout

    class LoraLinear(nn.Module):
        def __init__(self, base: nn.Linear, r, alpha, dropout):
            super().__init__()
            self.base_layer = base
            self.weight, self.bias = base.weight, base.bias
            self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False)})
            self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False)})
            self.lora_dropout = nn.ModuleDict({"default": nn.Dropout(dropout)})
            self.scaling = {"default": alpha / r}
            nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B["default"].weight)

        def forward(self, x):
            y = self.base_layer(x)
            d = self.lora_dropout["default"](x)
            return y + self.lora_B["default"](self.lora_A["default"](d)) * self.scaling["default"]

    def get_peft_model(model, cfg):
        for parent in list(model.modules()):
            for cname, child in list(parent.named_children()):
                if isinstance(child, nn.Linear) and any(cname == t for t in cfg.target_modules):
                    setattr(parent, cname, LoraLinear(child, cfg.r, cfg.lora_alpha, cfg.lora_dropout))
        class _Tuner(nn.Module):
            def __init__(self, m):
                super().__init__()
                self.model = m

        class _Peft(nn.Module):
            def __init__(self, m):
                super().__init__()
                self.base_model = _Tuner(m)
        return _Peft(model)

    def prepare_model_for_int8_training(model, use_gradient_checkpointing=True):
        for p in model.parameters():
            p.requires_grad = False
            if p.dtype in (torch.float16, torch.bfloat16):
                p.data = p.data.to(torch.float32)
        return model

    _mod("peft", LoraConfig=LoraConfig, get_peft_model=get_peft_model,
         prepare_model_for_int8_training=prepare_model_for_int8_training,
         TaskType=types.SimpleNamespace())


TARGET_MODULES_OVERRIDE = None
# synth.vision_weights variant the patched CLIPModel.from_pretrained loads (None, or "outliers": CLIP-like activation statistics)
WEIGHTS_VARIANT = None


def make_clip_model(cfg, seed, text_seed=1234, text_cfg=None):
    """HF CLIPModel of geometry ``cfg`` with the synthetic vision weights loaded.  With ``text_cfg``
    (config.TextConfig; the --lora_encoder text fixtures) the text tower has that geometry and carries
    ttl_amd.synth.text_weights(text_cfg, seed); without it, a small seeded random text tower (the image-
    mode fixtures only store its output features)."""
    from transformers import CLIPConfig, CLIPModel
    from ttl_amd import synth
    torch.manual_seed(text_seed)
    if text_cfg is not None:
        conf = CLIPConfig(
            vision_config=dict(hidden_size=cfg.width, intermediate_size=cfg.mlp,
                               num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                               image_size=cfg.image_size, patch_size=cfg.patch_size,
                               hidden_act="quick_gelu", layer_norm_eps=cfg.ln_eps,
                               projection_dim=cfg.embed),
            text_config=dict(hidden_size=text_cfg.width, intermediate_size=text_cfg.mlp,
                             num_hidden_layers=text_cfg.layers, num_attention_heads=text_cfg.heads,
                             eos_token_id=2, vocab_size=text_cfg.vocab_size, hidden_act="quick_gelu",
                             max_position_embeddings=text_cfg.context_length, layer_norm_eps=text_cfg.ln_eps,
                             projection_dim=text_cfg.embed),
            projection_dim=cfg.embed)
        conf._attn_implementation = "eager"
        model = CLIPModel(conf).float().eval()
        W = dict(synth.vision_weights(cfg, seed, variant=WEIGHTS_VARIANT))
        W.update(synth.text_weights(text_cfg, seed))
        sd = model.state_dict()
        for k, a in W.items():
            assert k in sd and tuple(sd[k].shape) == a.shape, (k, a.shape)
            sd[k].copy_(torch.from_numpy(a))
        model.load_state_dict(sd)
        gi, gt = model.get_image_features, model.get_text_features
        model.get_image_features = lambda *a, **k: _unwrap(gi(*a, **k))
        model.get_text_features = lambda *a, **k: _unwrap(gt(*a, **k))
        return model
    text_w = 64 if cfg.width <= 128 else 512
    conf = CLIPConfig(
        vision_config=dict(hidden_size=cfg.width, intermediate_size=cfg.mlp,
                           num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                           image_size=cfg.image_size, patch_size=cfg.patch_size,
                           hidden_act="quick_gelu", layer_norm_eps=cfg.ln_eps,
                           projection_dim=cfg.embed),
        text_config=dict(hidden_size=text_w, intermediate_size=4 * text_w,
                         num_hidden_layers=2 if cfg.width <= 128 else 12,
                         num_attention_heads=max(1, text_w // 64), eos_token_id=2,
                         vocab_size=49408, max_position_embeddings=77,
                         projection_dim=cfg.embed),
        projection_dim=cfg.embed)
    conf._attn_implementation = "eager"
    model = CLIPModel(conf).float().eval()
    W = synth.vision_weights(cfg, seed, variant=WEIGHTS_VARIANT)
    sd = model.state_dict()
    for k, a in W.items():
        assert k in sd and tuple(sd[k].shape) == a.shape, (k, a.shape)
        sd[k].copy_(torch.from_numpy(a))
    model.load_state_dict(sd)
    # text tower: make it non-degenerate (HF default init gives near-identical features)
    with torch.no_grad():
        for n, p in model.text_model.named_parameters():
            if p.dim() >= 2 and "embedding" not in n:
                p.normal_(0, 1.5 / math.sqrt(p.shape[-1]))
        model.text_projection.weight.normal_(0, 1.0 / math.sqrt(text_w))
    gi, gt = model.get_image_features, model.get_text_features
    model.get_image_features = lambda *a, **k: _unwrap(gi(*a, **k))
    model.get_text_features = lambda *a, **k: _unwrap(gt(*a, **k))
    return model


def _unwrap(o):
    return o.pooler_output if hasattr(o, "pooler_output") else o


class _FakeOpenAIClip(nn.Module):
    """Only what PromptLearner touches (clip/custom_clip.py:234-238,264,317,362-365)."""

    def __init__(self):
        super().__init__()
        self.visual = nn.Module()
        self.visual.conv1 = nn.Conv2d(3, 4, 1, bias=False)
        self.ln_final = nn.LayerNorm(8)
        self.token_embedding = nn.Embedding(49408, 8)
        self.dtype = torch.float32


def import_reference(cfg, seed, text_cfg=None):
    """Returns (ttl, deyo, custom_clip) reference modules, patched to build ``cfg`` (+ ``text_cfg``)."""
    install_import_stubs()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    argv = sys.argv
    sys.argv = ["ttl.py"]
    try:
        import clip as ref_clip
        import clip.clip as ref_clip_clip
        import clip.custom_clip as ref_cc
        import deyo as ref_deyo
        import ttl as ref_ttl
    finally:
        sys.argv = argv
    fake_load = lambda *a, **k: (_FakeOpenAIClip(), 8, None)
    for m in (ref_clip, ref_clip_clip, ref_cc):
        m.load = fake_load
    from transformers import CLIPModel
    ref_cc.CLIPModel = type("PatchedCLIPModel", (), {
        "from_pretrained": staticmethod(lambda *a, **k: make_clip_model(cfg, seed, text_cfg=text_cfg))})
    return ref_ttl, ref_deyo, ref_cc
