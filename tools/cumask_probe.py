"""Probe: do CU-masked streams (hipExtStreamCreateWithCUMask) help episodes in flight overlap?
Each slot of the pipeline gets a stream restricted to a subset of CUs; compares images/s with the default streams."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"))
import numpy as np, torch
from ttl_amd import synth
from ttl_amd.config import get_config
from ttl_amd.driver import EpisodePipeline

hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (C.c_uint32 * 8)(*[sum(((b >> (32 * w + i)) & 1) << i for i in range(32)) for w in range(8) for b in [bits]])
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

cfg = get_config("ViT-B/16")
lora = synth.lora_init(cfg, 0)
names = [f"vision_model.encoder.layers.{i}.self_attn.{pj}.lora_{ab}.default.weight"
         for i in range(cfg.layer_lo, cfg.layer_hi + 1) for pj in ("q_proj", "v_proj") for ab in ("A", "B")]
W = synth.vision_weights(cfg, 0)
tf = torch.from_numpy(synth.text_features(200, cfg.embed))
dev = torch.device("cuda:0")
pool = [torch.from_numpy(synth.views(cfg, 64, 1000 + j)).to(dev) for j in range(4)]

def run(nstreams, masks, label):
    pipe = EpisodePipeline(cfg, W, names, lora, tf, 100.0, dev, n_streams=nstreams, max_views=64)
    if masks:
        for sl, m in zip(pipe.slots, masks):
            sl["stream"] = masked_stream(m)
    for i in range(12): pipe.submit(pool[i % 4], n_updates=1)
    pipe.synchronize(); torch.cuda.synchronize()
    n = 150; t0 = time.perf_counter()
    for i in range(n): pipe.submit(pool[i % 4], n_updates=1)
    pipe.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {n/dt:.1f} images/s", flush=True)
    pipe.close()

ALL = (1 << 256) - 1
def bits(idx): return sum(1 << i for i in idx)
run(3, None, "3 streams, no masks")
run(1, [bits(range(0, 128))], "1 stream, CUs 0-127 (contiguous half)")
run(1, [bits(range(0, 256, 2))], "1 stream, even CUs")
run(2, [bits(range(0, 128)), bits(range(128, 256))], "2 streams, contiguous halves")
run(2, [bits(range(0, 256, 2)), bits(range(1, 256, 2))], "2 streams, even/odd CUs")
run(4, [bits(range(k * 64, (k + 1) * 64)) for k in range(4)], "4 streams, contiguous quarters")
run(3, [bits(range(0, 86)), bits(range(86, 171)), bits(range(171, 256))], "3 streams, contiguous thirds")
run(3, [ALL, ALL, bits(range(0, 64))], "3 streams: two full + one on CUs 0-63")
