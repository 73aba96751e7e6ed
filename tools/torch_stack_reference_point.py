"""Reference point: the reference's software stack on THIS GPU — HF transformers CLIP (sdpa attention) + a peft-style LoRA
Linear + torch.autocast(fp16) + GradScaler + torch.optim.AdamW, i.e. what ttl.py:338-352 / deyo.py:92-196 execute per image
when the reference itself runs on an MI355X (rocBLAS / hipBLASLt GEMMs, PyTorch's own attention and elementwise kernels).
Random-init weights of the ViT-B/16 geometry and synthetic views, as in bench.py (no checkpoint / dataset offline).
Not the reference's code: an independent restatement of the same per-image work with the same libraries, for the
"how fast is the PyTorch path here" figure quoted in BASELINE.md.

    python tools/torch_stack_reference_point.py [--views 64] [--classes 200] [--images 12] [--text-per-forward]
"""
import argparse
import json
import math
import time

import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers import CLIPTextConfig, CLIPTextModelWithProjection, CLIPVisionConfig, CLIPVisionModelWithProjection


class LoRALinear(nn.Module):
    """peft.tuners.lora.Linear forward with dropout 0: base(x) + (alpha / r) * B(A(x))."""

    def __init__(self, base, r, alpha):
        super().__init__()
        self.base = base
        self.A = nn.Linear(base.in_features, r, bias=False)
        self.B = nn.Linear(r, base.out_features, bias=False)
        nn.init.kaiming_uniform_(self.A.weight, a=math.sqrt(5))
        nn.init.zeros_(self.B.weight)
        self.scaling = alpha / r

    def forward(self, x):
        return self.base(x) + self.scaling * self.B(self.A(x))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--classes", type=int, default=200)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--images", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--text-per-forward", action="store_true",
                    help="recompute the K prompts' text features in every forward like the reference (clip/custom_clip.py:651-663)")
    a = ap.parse_args()
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    torch.manual_seed(0)
    vcfg = CLIPVisionConfig(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                            image_size=224, patch_size=16, projection_dim=512, hidden_act="quick_gelu")
    vis = CLIPVisionModelWithProjection(vcfg).to(dev).eval()
    for p in vis.parameters():
        p.requires_grad_(False)
    lora = []
    for i in (9, 10, 11):                       # --layer_range 9 11, q_proj / v_proj (clip/custom_clip.py:586)
        att = vis.vision_model.encoder.layers[i].self_attn
        for name in ("q_proj", "v_proj"):
            m = LoRALinear(getattr(att, name), a.rank, 2 * a.rank).to(dev)
            setattr(att, name, m)
            lora += [m.A.weight, m.B.weight]
    init = [p.detach().clone() for p in lora]
    opt = torch.optim.AdamW(lora, lr=5e-3)       # ttl.py:218
    opt_state = {k: (v.copy() if isinstance(v, dict) else v) for k, v in opt.state_dict().items()}
    scaler = torch.amp.GradScaler("cuda", init_scale=1000)   # ttl.py:222
    text = None
    if a.text_per_forward:
        tcfg = CLIPTextConfig(hidden_size=512, intermediate_size=2048, num_hidden_layers=12, num_attention_heads=8,
                              max_position_embeddings=77, vocab_size=49408, projection_dim=512, hidden_act="quick_gelu",
                              eos_token_id=49407, bos_token_id=49406, pad_token_id=0)
        text = CLIPTextModelWithProjection(tcfg).to(dev).eval()
        ids = torch.randint(1, 49000, (a.classes, 77), device=dev)
        ids[:, 0] = 49406
        ids[:, 10] = 49407
    tfeat_cached = F.normalize(torch.randn(a.classes, 512, device=dev), dim=-1)
    logit_scale = math.log(1 / 0.07)
    pool = [torch.randn(a.views, 3, 224, 224, device=dev) for _ in range(4)]

    def forward(x):
        with torch.autocast("cuda", dtype=torch.float16):
            if text is not None:
                with torch.no_grad():
                    tf = F.normalize(text(input_ids=ids).text_embeds, dim=-1)
            else:
                tf = tfeat_cached
            f = F.normalize(vis(pixel_values=x).image_embeds, dim=-1)
            return math.exp(logit_scale) * f @ tf.t()

    def episode(x):
        with torch.no_grad():                                   # LoRA_reset + optimizer.load_state_dict (ttl.py:338-344)
            for p, p0 in zip(lora, init):
                p.copy_(p0)
        opt.load_state_dict(opt_state)
        logits = forward(x)                                     # deyo.py:97
        ent = -(logits.softmax(1) * logits.log_softmax(1)).sum(1)
        idx = torch.where(ent <= math.log(1000))[0]            # deyo.py:107 (filter_ent = 0)
        if idx.numel():
            e = ent[idx]
            coeff = 1 / torch.exp(e.detach() - 0.4)
            loss = (e * coeff).mean(0)
            opt.zero_grad()
            scaler.scale(loss).backward()                      # deyo.py:186-188
            scaler.step(opt)
            scaler.update()
        with torch.no_grad():
            return forward(x[:1])                               # ttl.py:352

    for i in range(a.warmup):
        episode(pool[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.images):
        out = episode(pool[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.images
    print(json.dumps({"stack": f"torch {torch.__version__} + transformers CLIP (attn={vcfg._attn_implementation}) + LoRA Linear, autocast fp16",
                      "views": a.views, "classes": a.classes, "text_per_forward": bool(a.text_per_forward),
                      "ms_per_image": round(dt * 1e3, 2), "images_per_sec": round(1 / dt, 2), "images": a.images,
                      "finite": bool(torch.isfinite(out).all().item())}))


if __name__ == "__main__":
    main()
