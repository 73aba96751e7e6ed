"""Geometry of the CLIP image tower the hot path runs on.

The reference hard-codes ``openai/clip-vit-base-patch16`` (clip/custom_clip.py:581) and
rank 16 / alpha 32 / targets q_proj,v_proj (clip/custom_clip.py:583-590).  The same
numbers are the defaults here; ViT-L/14 and reduced test geometries use the same code.
"""
from dataclasses import dataclass, asdict


@dataclass(frozen=True)
class VitConfig:
    name: str = "ViT-B/16"
    image_size: int = 224
    patch_size: int = 16
    width: int = 768          # D
    heads: int = 12           # H, head dim is always 64 on this path
    mlp: int = 3072           # F
    layers: int = 12          # L
    embed: int = 512          # E (projection dim)
    ln_eps: float = 1e-5
    # LoRA (peft LoraConfig at clip/custom_clip.py:583-590)
    rank: int = 16
    lora_alpha: float = 32.0
    # first / last encoder layer whose q/v adapters train (ttl.py:159-161, --layer_range)
    layer_lo: int = 9
    layer_hi: int = 11
    # attention projections that carry an adapter.  The reference ships ["q_proj", "v_proj"] (clip/custom_clip.py:586, the k
    # slots of LoRA_AB are commented out at :181-182,216-217); k_proj / out_proj are what BASELINE.json's north_star adds.
    lora_targets: tuple = ("q_proj", "v_proj")

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def tokens(self) -> int:  # T
        return self.grid * self.grid + 1

    @property
    def head_dim(self) -> int:
        return self.width // self.heads

    @property
    def patch_k(self) -> int:  # 3*P*P
        return 3 * self.patch_size * self.patch_size

    @property
    def scaling(self) -> float:  # peft: lora_alpha / r
        return self.lora_alpha / self.rank

    def replace(self, **kw) -> "VitConfig":
        d = asdict(self)
        d.update(kw)
        return VitConfig(**d)


@dataclass(frozen=True)
class TextConfig:
    """Geometry of the CLIP text tower (``--lora_encoder text``, clip/custom_clip.py:602-607): same
    pre-LN encoder layers as the image tower, causal attention over ``context_length`` tokens, pooled at
    the end-of-text token.  Field names shared with VitConfig mean the same thing."""
    name: str = "ViT-B/16-text"
    context_length: int = 77
    vocab_size: int = 49408
    width: int = 512
    heads: int = 8
    mlp: int = 2048
    layers: int = 12
    embed: int = 512
    ln_eps: float = 1e-5
    rank: int = 16
    lora_alpha: float = 32.0
    layer_lo: int = 9
    layer_hi: int = 11
    lora_targets: tuple = ("q_proj", "v_proj")

    @property
    def tokens(self) -> int:
        return self.context_length

    @property
    def head_dim(self) -> int:
        return self.width // self.heads

    @property
    def scaling(self) -> float:
        return self.lora_alpha / self.rank

    def replace(self, **kw) -> "TextConfig":
        d = asdict(self)
        d.update(kw)
        return TextConfig(**d)


VIT_B16 = VitConfig()
VIT_B32 = VitConfig(name="ViT-B/32", patch_size=32)            # the run script's other ARCH option (scripts/test_ttl.sh:7)
VIT_L14 = VitConfig(name="ViT-L/14", patch_size=14, width=1024, heads=16, mlp=4096,
                    layers=24, embed=768, layer_lo=21, layer_hi=23)
# reduced geometries used by the parity fixtures (tests/golden/make_golden.py)
VIT_TINY = VitConfig(name="tiny", image_size=64, patch_size=16, width=128, heads=2, mlp=512,
                     layers=4, embed=64, layer_lo=1, layer_hi=3)
VIT_TINY197 = VIT_TINY.replace(name="tiny197", image_size=224)
VIT_TINY_MID = VIT_TINY.replace(name="tiny_mid", layer_lo=1, layer_hi=2)    # --layer_range that stops below the top layer
VIT_TINY_ALL = VIT_TINY.replace(name="tiny_all", layer_lo=0, layer_hi=3)    # every layer trained (get_coop's own default)

TEXT_B16 = TextConfig()
TEXT_L14 = TextConfig(name="ViT-L/14-text", width=768, heads=12, mlp=3072, embed=768)
TEXT_TINY = TextConfig(name="tiny-text", width=128, heads=2, mlp=512, layers=4, embed=64, layer_lo=1, layer_hi=3)
TEXT_ARCHS = {"ViT-B/16": TEXT_B16, "ViT-B/32": TEXT_B16, "ViT-L/14": TEXT_L14, "tiny": TEXT_TINY, "tiny197": TEXT_TINY, "tiny_mid": TEXT_TINY,
              "tiny_all": TEXT_TINY}

ARCHS = {"ViT-B/16": VIT_B16, "ViT-B/32": VIT_B32, "ViT-L/14": VIT_L14, "tiny": VIT_TINY, "tiny197": VIT_TINY197, "tiny_mid": VIT_TINY_MID,
         "tiny_all": VIT_TINY_ALL}


LORA_TARGET_ORDER = ("q_proj", "k_proj", "v_proj", "out_proj")      # order of the parameter groups inside a layer
LORA_TARGET_BITS = {"q_proj": 1, "k_proj": 2, "v_proj": 4, "out_proj": 8}


def ordered_targets(cfg):
    """cfg.lora_targets in canonical order (q, k, v, out)."""
    bad = set(cfg.lora_targets) - set(LORA_TARGET_ORDER)
    if bad or not cfg.lora_targets:
        raise ValueError(f"lora_targets must be a non-empty subset of {LORA_TARGET_ORDER}, got {cfg.lora_targets}")
    return tuple(t for t in LORA_TARGET_ORDER if t in cfg.lora_targets)


def trainable_names(cfg, tower="vision_model"):
    """State-dict names of the bound LoRA buffer's tensors, in its order (ttl.py:195-213: per layer q.A, q.B, v.A, v.B; with
    k_proj / out_proj adapters configured: q, k, v, out)."""
    return [f"{tower}.encoder.layers.{i}.self_attn.{pj}.lora_{ab}.default.weight"
            for i in range(cfg.layer_lo, cfg.layer_hi + 1) for pj in ordered_targets(cfg) for ab in ("A", "B")]


def targets_mask(cfg):
    return sum(LORA_TARGET_BITS[t] for t in ordered_targets(cfg))


def get_config(arch: str) -> VitConfig:
    if arch not in ARCHS:
        raise ValueError(f"unsupported arch {arch!r}; known: {sorted(ARCHS)}")
    return ARCHS[arch]


def get_text_config(arch: str) -> TextConfig:
    """Text tower that pairs with image tower ``arch`` (openai/clip-vit-* checkpoints)."""
    if arch not in TEXT_ARCHS:
        raise ValueError(f"unsupported arch {arch!r}; known: {sorted(TEXT_ARCHS)}")
    return TEXT_ARCHS[arch]
