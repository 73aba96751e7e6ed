"""Host-side phase timing of the decoded-image -> GPU views -> episode loop (where does the CPU time go?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from ttl_amd import synth, views as V
from ttl_amd.config import get_config
from ttl_amd.driver import EpisodePipeline

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = get_config("ViT-B/16")
Wt = synth.vision_weights(cfg, 0)
lora0 = synth.lora_init(cfg, 1)
tf = torch.from_numpy(synth.text_features(200, cfg.embed, 2))
names = [f"vision_model.encoder.layers.{i}.self_attn.{pj}.lora_{ab}.default.weight"
         for i in range(cfg.layer_lo, cfg.layer_hi + 1) for pj in ("q_proj", "v_proj") for ab in ("A", "B")]
pipe = EpisodePipeline(cfg, Wt, names, lora0, tf, float(np.exp(Wt["logit_scale"])), "cuda:0", n_streams=streams)
img = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (375, 500, 3), dtype=np.uint8)).pin_memory()
gen = torch.Generator().manual_seed(0)
T = dict(boxes=0.0, h2d=0.0, views=0.0, submit=0.0)
n = 300
for it in range(n + 20):
    if it == 20:
        torch.cuda.synchronize(); T = {k: 0.0 for k in T}; t_all = time.perf_counter()
    t0 = time.perf_counter(); b = V.draw_boxes(375, 500, 64, gen)
    t1 = time.perf_counter(); d = img.to("cuda:0", non_blocking=True); tgt = torch.tensor([3]).to("cuda:0")
    t2 = time.perf_counter(); v = V.make_views(d, b, 224)
    t3 = time.perf_counter(); pipe.submit(v, target=tgt, n_updates=1)
    t4 = time.perf_counter()
    T["boxes"] += t1 - t0; T["h2d"] += t2 - t1; T["views"] += t3 - t2; T["submit"] += t4 - t3
pipe.synchronize()
dt = time.perf_counter() - t_all
print(f"streams={streams}: {n/dt:.1f} img/s; host ms/image:", {k: round(v / n * 1e3, 3) for k, v in T.items()})
