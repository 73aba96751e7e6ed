"""Quick timing of the B/16 episode + per-class profile (run on the GPU box).
    python tools/quick_bench.py [arch views classes]      TTL_PRECISION=fp16|bf16 picks the operand build (default fp16, the headline)"""
import sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd")]
from ttl_amd import synth
from ttl_amd.config import get_config
from ttl_amd.engine import TTLEngine

arch = sys.argv[1] if len(sys.argv) > 1 else "ViT-B/16"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
cfg = get_config(arch)
t0 = time.time()
W = synth.vision_weights(cfg, 0)
print("weights gen", time.time() - t0)
PREC = os.environ.get("TTL_PRECISION", "fp16")
print("operand build:", PREC)
eng = TTLEngine(cfg, N, K, "cuda:0", precision=PREC)
eng.load_weights(W)
eng.set_text_features(torch.from_numpy(synth.text_features(K, cfg.embed)), 100.0)
lora = synth.lora_init(cfg, 0)
names = []
for i in range(cfg.layer_lo, cfg.layer_hi + 1):
    for pj in ("q_proj", "v_proj"):
        for ab in ("A", "B"):
            names.append(f"vision_model.encoder.layers.{i}.self_attn.{pj}.lora_{ab}.default.weight")
flat = torch.cat([torch.from_numpy(lora[k]).reshape(-1) for k in names]).cuda()
eng.bind_lora(flat)
snap = flat.clone(); m = torch.zeros_like(flat); v = torch.zeros_like(flat)
x = torch.from_numpy(synth.views(cfg, N, 3)).cuda()
for _ in range(3):
    eng.episode(x, snap, m, v)
torch.cuda.synchronize()
steps = 20
t0 = time.time()
for _ in range(steps):
    eng.episode(x, snap, m, v)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print(f"episode {dt*1e3:.3f} ms  -> {1/dt:.1f} img/s")
eng.profile_enable(True)
for _ in range(3):
    eng.episode(x, snap, m, v)
ms, cnt, fl = eng.profile_read()
eng.profile_enable(False)
tot = sum(ms.values())
for k in ms:
    print(f"  {k:24s} {ms[k]/3:8.3f} ms/ep  {cnt[k]//3:4d} launches")
print(f"  sum {tot/3:.3f} ms/ep ; gemm {fl/3/1e12:.3f} TFLOP/ep -> {fl/ (ms['gemm']*1e-3)/1e12:.1f} TFLOP/s")
