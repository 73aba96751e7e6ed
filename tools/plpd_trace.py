"""--filter_plpd 1 through the fused pipeline, for a rocprofv3 kernel trace (round-4 review item 4: no at::native kernel between the
first and the last launch of an episode).

    rocprofv3 --kernel-trace --stats -d gpurun_out/plpd_prof -o p --output-format csv -- python3 tools/plpd_trace.py [patch|pixel|occ]
    python3 tools/plpd_trace.py --summarize gpurun_out/plpd_prof    -> kernels by owner, and every foreign kernel dispatched after the
                                                                       first episode_reset_kernel (i.e. inside the episode stream)
"""
import csv, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd")]

OWN = re.compile(r"gemm|attn|ln_|layernorm|skinny|wgrad|head_mm|adamw|episode_reset|deyo_|row_stats|select_kernel|topk_hits|plpd_|refresh_kernel|"
                 r"im2col|cls_rows|splitk|cast_|transpose|unit_rows|gather_rows|text_embed|views_kernel|coef_kernel|scaler_|tpt_|reset_kernel")


def summarize(d):
    trace = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("kernel_trace.csv")]
    rows = sorted(csv.DictReader(open(trace[0])), key=lambda r: int(r["Start_Timestamp"]))
    first = next(i for i, r in enumerate(rows) if "episode_reset_kernel" in r["Kernel_Name"])
    inside = rows[first:]
    own = [r for r in inside if OWN.search(r["Kernel_Name"])]
    rt = [r for r in inside if r["Kernel_Name"].startswith("__amd_rocclr")]
    foreign = [r for r in inside if r not in own and r not in rt]
    n_ep = sum("episode_reset_kernel" in r["Kernel_Name"] for r in rows)
    print(f"episodes traced: {n_ep}; kernels from the first episode on: {len(inside)} = {len(own)} of this library + {len(rt)} HIP-runtime "
          f"copy / fill kernels (hipMemcpyAsync / hipMemsetAsync nodes) + {len(foreign)} foreign")
    by = {}
    for r in own:
        if "plpd_" in r["Kernel_Name"]:
            k = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
            by.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(by.items()):
        print(f"  {k:28s} {len(v):5d} launches, avg {sum(v) / len(v):7.2f} us")
    for r in foreign:
        print("  FOREIGN:", r["Kernel_Name"][:120])
    return len(foreign)


if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    sys.exit(1 if summarize(sys.argv[2]) else 0)

import time
import torch
from ttl_amd import synth
from ttl_amd.config import get_config, trainable_names
from ttl_amd.driver import EpisodePipeline
aug = sys.argv[1] if len(sys.argv) > 1 else "patch"
graph = "--graph" in sys.argv           # replay every episode as one HIP graph (the throughput figure; the trace uses plain launches)
cfg = get_config("ViT-B/16")
names = trainable_names(cfg)
dev = torch.device("cuda:0")
pipe = EpisodePipeline(cfg, synth.vision_weights(cfg, 0), names, synth.lora_init(cfg, 0), torch.from_numpy(synth.text_features(200, cfg.embed)),
                       100.0, dev, n_streams=3, max_views=64, precision="fp16", use_graph=graph)
views = [torch.from_numpy(synth.views(cfg, 64, 1000 + j)).to(dev) for j in range(3)]
tgt = torch.zeros(1, dtype=torch.int64, device=dev)
spec = dict(aug_type=aug, threshold=0.2, patch_len=4, occlusion_size=112, row_start=56, column_start=56)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
n = 30
host = {"draw": 0.0, "struct": 0.0, "n": 0}
if "--host-timing" in sys.argv:      # where the host's time per image goes: the permutation draw, and all of _plpd_struct (draw + staging copy)
    import ttl_amd.deyo as _D
    _draw, _struct = _D.draw_plpd_perms, pipe._plpd_struct
    def draw(*a_, **k_):
        t = time.perf_counter(); r = _draw(*a_, **k_); host["draw"] += time.perf_counter() - t; return r
    def struct(*a_, **k_):
        t = time.perf_counter(); r = _struct(*a_, **k_); host["struct"] += time.perf_counter() - t; host["n"] += 1; return r
    _D.draw_plpd_perms, pipe._plpd_struct = draw, struct
    print("torch threads:", torch.get_num_threads(), "cpus:", os.cpu_count(), flush=True)
for i in range(6):       # warm-up: auxiliary contexts, graph capture
    pipe.submit(views[i % 3], target=tgt, want_output=False, plpd=dict(spec=spec, n_candidates=64), n_updates=1)
pipe.synchronize()
t0 = time.perf_counter()
for i in range(n):
    pipe.submit(views[i % 3], target=tgt, want_output=False, plpd=dict(spec=spec, n_candidates=64), n_updates=1)
pipe.synchronize()
print(f"{aug}: {n / (time.perf_counter() - t0):.1f} images/s with the PLPD stage (64 views, K=200, 3 episodes in flight, {'graph replay' if graph else 'plain launches'})", flush=True)
pipe.close()
if host["n"]:
    print(f"   host per image: permutation draw {1e3 * host['draw'] / host['n']:.2f} ms, _plpd_struct in all {1e3 * host['struct'] / host['n']:.2f} ms", flush=True)
