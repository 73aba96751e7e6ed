"""Throughput with S independent episodes in flight on S HIP streams (S contexts on one GPU)."""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd")]
from ttl_amd import synth
from ttl_amd.config import get_config
from ttl_amd.engine import TTLEngine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = get_config("ViT-B/16")
W = synth.vision_weights(cfg, 0)
lora = synth.lora_init(cfg, 0)
names = [f"vision_model.encoder.layers.{i}.self_attn.{pj}.lora_{ab}.default.weight"
         for i in range(cfg.layer_lo, cfg.layer_hi + 1) for pj in ("q_proj", "v_proj") for ab in ("A", "B")]
tf = torch.from_numpy(synth.text_features(200, cfg.embed))
ctxs = []
for s in range(S):
    eng = TTLEngine(cfg, 64, 200, "cuda:0")
    eng.load_weights(W); eng.set_text_features(tf, 100.0)
    flat = torch.cat([torch.from_numpy(lora[k]).reshape(-1) for k in names]).cuda()
    eng.bind_lora(flat)
    ctxs.append(dict(eng=eng, flat=flat, snap=flat.clone(), m=torch.zeros_like(flat), v=torch.zeros_like(flat),
                     x=torch.from_numpy(synth.views(cfg, 64, 3 + s)).cuda(), stream=torch.cuda.Stream()))
torch.cuda.synchronize()
def run(n):
    for i in range(n):
        c = ctxs[i % S]
        with torch.cuda.stream(c["stream"]):
            c["eng"].episode(c["x"], c["snap"], c["m"], c["v"])
    torch.cuda.synchronize()
run(6)
t0 = time.time(); N = 60; run(N); dt = (time.time() - t0) / N
print(f"streams={S}: {dt*1e3:.3f} ms/image -> {1/dt:.1f} img/s")
