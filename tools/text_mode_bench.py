"""Throughput of the --lora_encoder text episode: image features of 64 views (forward only) +
text tower over K prompts x 77 tokens with LoRA backward + AdamW + adapted text features."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"))
import numpy as np
import torch
from ttl_amd import synth
from ttl_amd.config import get_config, get_text_config
from ttl_amd.custom_clip import build_text_mode_engine
from ttl_amd.driver import EpisodePipeline

arch = "ViT-B/16"
PREC = sys.argv[1] if len(sys.argv) > 1 else "fp16"       # operand build (fp16 = the headline build)
print("operand build:", PREC)
vcfg, tcfg = get_config(arch), get_text_config(arch)
Wv, Wt = synth.vision_weights(vcfg, 0), synth.text_weights(tcfg, 0)
lora = synth.lora_init(tcfg, 0, tower="text_model")
names = [f"text_model.encoder.layers.{i}.self_attn.{pj}.lora_{ab}.default.weight"
         for i in range(tcfg.layer_lo, tcfg.layer_hi + 1) for pj in ("q_proj", "v_proj") for ab in ("A", "B")]
dev = torch.device("cuda:0")
views = [torch.from_numpy(synth.views(vcfg, 64, 1000 + j)).to(dev) for j in range(4)]
for K in (200, 1000):
    ids = synth.token_ids(K, tcfg, 3)
    for streams in (1, 2, 3):
        fac = lambda: build_text_mode_engine(vcfg, tcfg, Wv, Wt, ids, 100.0, dev, 64, K, PREC)
        pipe = EpisodePipeline(vcfg, None, names, lora, None, 100.0, dev, n_streams=streams, max_views=64, precision=PREC, engine_factory=fac, n_classes=K)
        for i in range(6):
            pipe.submit(views[i % 4], n_updates=1)
        pipe.synchronize()
        n = 60
        t0 = time.perf_counter()
        for i in range(n):
            pipe.submit(views[i % 4], n_updates=1)
        pipe.synchronize()
        dt = time.perf_counter() - t0
        # algorithmic FLOPs: image fwd 2.263T + text fwd (12 layers) + text bwd (3 layers, ~2x fwd of those minus) + resume fwd (3 layers)
        M = K * 77
        D, F = tcfg.width, tcfg.mlp
        lay = 2 * M * (4 * D * D + 2 * D * F) + 4 * K * tcfg.heads * 77 * 77 * 64
        fl = 2.263e12 + 12 * lay + 2 * 2 * lay + 3 * lay
        print(f"K={K} streams={streams}: {n/dt:.1f} images/s ({1e3*dt/n:.2f} ms/image), ~{fl/1e12:.2f} TFLOP/image -> {fl*n/dt/1e12:.0f} TFLOP/s")
        if streams == 1:
            eng = pipe.slots[0]["eng"]
            for e in (eng.img, eng.txt):
                e.profile_enable(True)
            for i in range(4):
                pipe.submit(views[i % 4], n_updates=1)
            pipe.synchronize()
            for nm, e in (("image ctx", eng.img), ("text ctx", eng.txt)):
                ms, cnt, gf = e.profile_read()
                e.profile_enable(False)
                print("   ", nm, {k: round(v / 4, 3) for k, v in ms.items()}, "gemm TF/s", round(gf / max(ms["gemm"], 1e-9) / 1e9, 0))
        pipe.close()
