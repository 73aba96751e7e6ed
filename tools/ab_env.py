"""In-situ A/B of settings a context reads from the environment WHEN IT IS CREATED (TTL_QKV_HEAD_MAJOR, TTL_SHARE_WEIGHTS ...):
every variant gets its own EpisodePipeline in ONE process, timed blocks alternate between them (guide §5.4 rule 24), and the
single-stream per-class device times of one engine per variant are printed beside the rates.

    python tools/ab_env.py base: hm0:TTL_QKV_HEAD_MAJOR=0 [--streams 3 --block 60 --rounds 5 --precision bf16]
"""
import argparse, json, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd")]
import torch
from ttl_amd import synth
from ttl_amd.config import get_config
from ttl_amd.driver import EpisodePipeline

ap = argparse.ArgumentParser()
ap.add_argument("variants", nargs="+", help="NAME:ENV=VAL,ENV=VAL (empty list = defaults)")
ap.add_argument("--streams", type=int, default=3)
ap.add_argument("--block", type=int, default=60)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--views", type=int, default=64)
ap.add_argument("--classes", type=int, default=200)
ap.add_argument("--precision", default="bf16")
ap.add_argument("--targets", default="q_proj,v_proj")
a = ap.parse_args()

cfg = get_config("ViT-B/16").replace(lora_targets=tuple(a.targets.split(",")))
W = synth.vision_weights(cfg, 0)
lora = synth.lora_init(cfg, 0)
from ttl_amd.config import trainable_names
names = trainable_names(cfg)
tf = torch.from_numpy(synth.text_features(a.classes, cfg.embed))
dev = torch.device("cuda:0")
pool = [torch.from_numpy(synth.views(cfg, a.views, 1000 + j)).to(dev) for j in range(4)]

pipes = {}
for v in a.variants:
    name, _, envs = v.partition(":")
    kv = dict(e.split("=", 1) for e in envs.split(",") if e)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    pipes[name] = EpisodePipeline(cfg, W, names, lora, tf, 100.0, dev, n_streams=a.streams, max_views=a.views, precision=a.precision)
    for k, o in old.items():
        if o is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = o


def block(p, n):
    for i in range(n):
        p.submit(pool[i % 4])
    p.synchronize()


for p in pipes.values():
    block(p, 12)
rates = {k: [] for k in pipes}
for r in range(a.rounds):
    for k, p in pipes.items():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        block(p, a.block)
        rates[k].append(a.block / (time.perf_counter() - t0))
out = {}
for k, p in pipes.items():
    eng = p.slots[0]["eng"]
    sl = p.slots[0]
    eng.profile_enable(True)
    with torch.cuda.stream(sl["stream"]):
        for i in range(4):
            eng.episode(pool[i % 4], sl["snap"], sl["m"], sl["v"])
    sl["stream"].synchronize()
    ms, cnt, fl = eng.profile_read()
    eng.profile_enable(False)
    out[k] = {"images_per_s_median": round(statistics.median(rates[k]), 2), "min": round(min(rates[k]), 2), "max": round(max(rates[k]), 2),
              "class_ms_per_image_1stream": {c: round(t / 4, 4) for c, t in ms.items()}}
    print(k, json.dumps(out[k]), flush=True)
print("AB_JSON " + json.dumps(out))
