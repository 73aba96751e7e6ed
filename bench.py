#!/usr/bin/env python3
"""bench.py — test images/sec of TTL's per-sample hot path on MI355X.

One "step" = one test image = reset -> 64-view CLIP ViT-B/16 forward -> entropy-weighted loss ->
backward into the rank-16 LoRA adapters of layers 9-11 -> AdamW -> adapted 1-view inference
(the loop body of the reference's ttl.py:338-352), on synthetic views already resident in HBM.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  Multi-GPU: images shard across ranks (weak scaling, no data-path
collective); the only collectives are the timing barrier/max and the 3-int accuracy all-reduce
(ttl_amd.driver.ImageShard, the same object the evaluation loop and the gloo tests use).
`--gpus N` started as a plain `python bench.py` launches the N rank processes itself (before any
GPU call) through torch.distributed.run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd")]

PEAK_BF16_TFLOPS = 2500.0   # dense bf16/fp16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--arch", default="ViT-B/16")
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--classes", type=int, default=200)       # ImageNet-A label-set size (configs[1])
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--updates", type=int, default=1)
    ap.add_argument("--pool", type=int, default=4, help="distinct pre-staged view batches per rank")
    ap.add_argument("--streams", type=int, default=3, help="independent episodes in flight per GPU (HIP streams)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp16"], help="MFMA operand dtype")
    ap.add_argument("--graph", type=int, default=0, help="1: replay every episode as one HIP graph launch (host-bound small-view runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the untimed parity check against the reference-generated fixture")
    ap.add_argument("--no-fp16-leg", action="store_true", help="skip the second timed leg on the fp16-operand build")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only for smoke-testing the N>1 logic on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true", help="smoke test: put every rank on cuda:0 (with --backend gloo)")
    return ap.parse_args()


def self_spawn(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, BEFORE anything touches the GPU (a process
    that has initialised HIP must never exec/replace itself; a child process is fine)."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def episode_flops(cfg, n_views, n_classes):
    """Algorithmic FLOPs of one test image (SURVEY.md §8d / BASELINE.md §2)."""
    D, F, T, H, L, E, r, P = cfg.width, cfg.mlp, cfg.tokens, cfg.heads, cfg.layers, cfg.embed, cfg.rank, cfg.patch_size

    def fwd(n):
        M = n * T
        layer = 2 * M * (4 * D * D + 2 * D * F) + 4 * n * H * T * T * 64
        lora = 8 * M * D * r
        nt = cfg.layer_hi - cfg.layer_lo + 1
        return 2 * n * (T - 1) * 3 * P * P * D + L * layer + nt * lora + 2 * n * (D * E + E * n_classes)

    M = n_views * T
    nt = cfg.layer_hi - cfg.layer_lo + 1
    full = 2 * M * (4 * D * D + 2 * D * F) + 8 * n_views * H * T * T * 64 + 16 * M * D * r
    first = 2 * M * (D * D + 2 * D * F) + 6 * n_views * H * T * T * 64 + 12 * M * D * r
    bwd = (nt - 1) * full + first + 2 * n_views * (D * E + E * n_classes)
    return fwd(n_views) + bwd + fwd(1)


def non_gemm_flops_executed(cfg, n_views, n_classes):
    """FLOPs the build executes outside its GEMM kernels for one image (attention, LoRA skinny / wgrad products, head),
    with the shortcuts of DESIGN §3.5 applied: last layer forward for the CLS query only, top-layer attention backward
    rank-1.  Added to the measured GEMM FLOPs this gives the EXECUTED work (the algorithmic figure above also counts
    what the CLS-only last layer and top-layer backward skip)."""
    D, T, H, L, E, r = cfg.width, cfg.tokens, cfg.heads, cfg.layers, cfg.embed, cfg.rank
    nt = cfg.layer_hi - cfg.layer_lo + 1
    n, M = n_views, n_views * T
    attn_f = (L - 1) * 4 * n * H * T * T * 64 + 4 * n * H * T * 64            # dense layers + the CLS query of the last one
    attn_b = max(nt - 2, 0) * 8 * n * H * T * T * 64 + (6 * n * H * T * T * 64 if nt > 1 else 0) + 8 * n * H * T * 64
    lora = nt * (4 * M * D * r) + nt * (4 * M * D * r + 8 * M * D * r)         # fwd down-proj; bwd dU + 4 weight gradients
    head = 3 * 2 * n * (D * E + E * n_classes)
    attn_1 = (L - cfg.layer_lo - 1) * 4 * H * T * T * 64                       # resumed 1-view inference
    return attn_f + attn_b + lora + head + attn_1


def cpu_baseline(cfg, n_classes, full_views=64, budget_s=25.0):
    """Oracle (numpy fp32 restatement, validated against the reference goldens) timed on this host inside a time budget:
    a 4-view probe sizes the sample, then one warm-up and as many timed episodes as fit (>= 2) run at the largest view
    count that fits; cost is linear in views, so the result is scaled to the full view count.  A reported baseline."""
    from oracle import ttl_oracle as O
    from ttl_amd import synth
    W = synth.vision_weights(cfg, 0)
    lora = synth.lora_init(cfg, 0)
    tf = synth.text_features(n_classes, cfg.embed)
    t0 = time.time()
    O.episode(cfg, W, lora, synth.views(cfg, 4, 11), tf, prec="fp32")
    probe = time.time() - t0                                     # (also the warm-up: BLAS threads, page-in)
    sample_views = full_views
    while sample_views > 4 and probe * sample_views / 4 * 3 > budget_s:   # room for 3 episodes
        sample_views //= 2
    x = synth.views(cfg, sample_views, 11)
    times = []
    t_start = time.time()
    while len(times) < 2 or (time.time() - t_start + (times[-1] if times else 0) < budget_s and len(times) < 10):
        t0 = time.time()
        O.episode(cfg, W, lora, x, tf, prec="fp32")
        times.append(time.time() - t0)
    dt = sorted(times)[len(times) // 2]
    t_img = dt * full_views / sample_views
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next(l.split(":", 1)[1].strip() for l in f if l.startswith("model name"))
    except Exception:
        pass
    out = {"value": round(1.0 / t_img, 5), "unit": "images/sec",
           "cores": os.cpu_count(), "kind": "port", "cpu_model": cpu_model,
           "sample": f"oracle/ttl_oracle.py (numpy fp32, BLAS threads = all cores): 1 warm-up (4 views) + {len(times)} timed episodes on "
                     f"{sample_views} of {full_views} views, K={n_classes}, median {dt:.2f} s (min {min(times):.2f}, max {max(times):.2f}), "
                     f"scaled by {full_views}/{sample_views} (cost is linear in views; text features cached like the GPU path)"}
    # The reference recomputes the K class-text features in EVERY forward (clip/custom_clip.py:669-671, Q12): twice per
    # image.  Time the text tower's restatement on a few prompts and scale linearly in K for that figure.
    try:
        from ttl_amd.config import get_text_config
        tcfg = get_text_config(cfg.name)
        Wt = synth.text_weights(tcfg, 0)
        kp = min(n_classes, 32)
        ids = synth.token_ids(kp, tcfg, 3)
        net = O.TextOracle(tcfg, Wt, synth.lora_init(tcfg, 0, tower="text_model"), "fp32")
        net.trained = lambda i: False
        t0 = time.time()
        net.forward(ids)
        t_text = (time.time() - t0) * n_classes / kp
        out["reference_faithful"] = {"value": round(1.0 / (t_img + 2.0 * t_text), 5), "unit": "images/sec",
                                     "note": f"+ 2 text-tower forwards of K={n_classes} prompts per image as the reference does "
                                             f"(timed on {kp} prompts: {t_text:.1f} s per K-prompt forward after scaling)"}
    except Exception as e:      # never let the secondary figure break the bench line
        out["reference_faithful"] = {"value": None, "note": f"not measured: {e}"}
    return out


def parity_check(precision):
    """Untimed: the reference-generated fixture b16_n64_k200_ent0 (ViT-B/16, 64 views, K=200: the benched workload's
    shape) through the benched build.  metric = max|a-b| / max|b| per tensor (tests/helpers.max_rel)."""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_case, episode_kwargs, max_rel
    from ttl_amd.engine import TTLEngine
    g, cfg, W, x, lora0, tf = load_case("b16_n64_k200_ent0")
    kw = episode_kwargs(g)
    names = [f"vision_model.encoder.layers.{i}.self_attn.{pj}.lora_{ab}.default.weight"
             for i in range(cfg.layer_lo, cfg.layer_hi + 1) for pj in ("q_proj", "v_proj") for ab in ("A", "B")]
    eng = TTLEngine(cfg, x.shape[0], tf.shape[0], "cuda", precision)
    eng.load_weights(W)
    eng.set_text_features(torch.from_numpy(tf), float(np.exp(W["logit_scale"])))
    flat = torch.cat([torch.from_numpy(lora0[k]).reshape(-1) for k in names]).cuda().contiguous()
    eng.bind_lora(flat)
    snap, m, v = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat)
    l1, l0 = eng.episode(torch.from_numpy(x).cuda(), snap, m, v, n_updates=kw["n_updates"], objective=kw["objective"],
                         mode=1 if kw["mode"] == "topk" else 0, rho=kw["rho"], margin=kw["margin"], lr=kw["lr"], want_logits0=True)
    torch.cuda.synchronize()
    idx, _ = eng.last_selection(x.shape[0])
    grads, off, gerr, werr = eng.grads.cpu().numpy(), 0, 0.0, 0.0
    new = flat.cpu().numpy()
    for k in names:
        n = lora0[k].size
        gref = g["grad/" + k]
        if np.abs(gref).max() > 0:
            gerr = max(gerr, max_rel(grads[off:off + n].reshape(gref.shape), gref))
        werr = max(werr, float(np.abs(new[off:off + n].reshape(gref.shape) - g["lora1/" + k]).max()))
        off += n
    out = {"fixture": "tests/golden/b16_n64_k200_ent0.npz (written by the reference's own test_time_tuning, fp32 CPU)",
           "dtype": precision, "metric": "max|a-b|/max|b| per tensor",
           "logits_max_rel": round(max_rel(l0.cpu().numpy(), g["logits0"]), 6),
           "adapted_logits_max_rel": round(max_rel(l1.cpu().numpy(), g["logits1"]), 6),
           "grad_max_rel": round(gerr, 6), "lora_weights_max_abs_diff": round(werr, 8),
           "mask_exact": bool(np.array_equal(np.sort(idx), np.sort(np.asarray(g["idx"]).reshape(-1)))),
           "top1_equal": bool(int(l1.argmax()) == int(g["top5"][0, 0])),
           "north_star_tolerance": 1e-3}
    eng.close()
    return out


def main():
    a = parse_args()
    self_spawn(a)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}")
    if not a.same_device and torch.cuda.device_count() < world:     # (device_count does not initialise the GPU)
        raise SystemExit(f"--gpus {a.gpus} but only {torch.cuda.device_count()} devices are visible")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    if a.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus says {a.gpus}")

    from ttl_amd import synth
    from ttl_amd.config import get_config
    from ttl_amd.driver import EpisodePipeline, ImageShard
    shard = ImageShard(rank, world)
    ranks_seen = shard.ranks_seen(dev)
    if ranks_seen != a.gpus:
        raise SystemExit(f"{ranks_seen} ranks answered the all-reduce, --gpus says {a.gpus}")

    cfg = get_config(a.arch).replace(rank=a.rank)
    lora = synth.lora_init(cfg, 0)
    weights = synth.vision_weights(cfg, 0)
    names = [f"vision_model.encoder.layers.{i}.self_attn.{pj}.lora_{ab}.default.weight"
             for i in range(cfg.layer_lo, cfg.layer_hi + 1) for pj in ("q_proj", "v_proj") for ab in ("A", "B")]
    tfeat = torch.from_numpy(synth.text_features(a.classes, cfg.embed))
    # synthetic inputs of the workload's shape, already resident in HBM (data: synthetic).  Item i of the global stream
    # belongs to rank i % world (ImageShard) and is view batch i % pool with label (7 * (i % pool)) % classes: what an item
    # is does not depend on the number of ranks, so the accuracy accumulator of N ranks x K steps equals 1 rank x N*K steps.
    pool = [torch.from_numpy(synth.views(cfg, a.views, 1000 + j)).to(dev) for j in range(a.pool)]
    labels = [torch.tensor([(7 * j) % a.classes], device=dev) for j in range(a.pool)]

    def timed_run(precision, steps, warmup):
        pipe = EpisodePipeline(cfg, weights, names, lora, tfeat, 100.0, dev, n_streams=a.streams, max_views=a.views,
                               precision=precision, use_graph=bool(a.graph))

        def step(i):
            pipe.submit(pool[i % a.pool], target=labels[i % a.pool], n_updates=a.updates)

        def fence():
            pipe.synchronize()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize()

        for i in shard.indices(world * warmup):     # this rank's items of the warm-up stream
            step(i)
        fence()
        pipe.reset_totals()
        t0 = time.perf_counter()
        for i in shard.indices(world * steps):      # ... and of the timed stream: exactly `steps` per rank
            step(i)
        fence()
        dt = time.perf_counter() - t0
        T = float(shard.max(torch.tensor([dt], dtype=torch.float64, device=dev)).item())
        return pipe, step, T

    pipe, step, T = timed_run(a.precision, a.steps, a.warmup)
    acc = shard.accuracy(pipe.totals())                       # C1: accuracy accumulator of the timed stream (the path's only collective)
    eng = pipe.slots[0]["eng"]

    # ---- roofline of the dominant kernel (the big-M MFMA GEMM): HIP events on the launch streams.
    # pass A: the same S-streams-in-flight regime as the timed region; pass B: one stream alone (kernel durations
    # without a second episode sharing the CUs — what rocprofv3 --kernel-trace reports too, it serialises kernels).
    roof = None
    executed = None
    if rank == 0:
        def profiled(run, engines):
            for e in engines:
                e.profile_enable(True)
            run()
            tot_ms, tot_cnt, tot_fl = {}, {}, 0.0
            profiled.bytes, profiled.flops_all = 0.0, 0.0
            for e in engines:
                ms, cnt, fl = e.profile_read()
                profiled.bytes += e.last_gemm_bytes
                profiled.flops_all += e.last_gemm_flops_all
                e.profile_enable(False)
                for k in ms:
                    tot_ms[k] = tot_ms.get(k, 0.0) + ms[k]
                    tot_cnt[k] = tot_cnt.get(k, 0) + cnt[k]
                tot_fl += fl
            return tot_ms, tot_cnt, tot_fl
        nprof = 6

        def run_all():
            for i in range(nprof):
                step(i)
            pipe.synchronize()

        def run_one():
            sl = pipe.slots[0]
            for i in range(nprof):
                with torch.cuda.stream(sl["stream"]):
                    eng.episode(pool[i % a.pool], sl["snap"], sl["m"], sl["v"], n_updates=a.updates)
            sl["stream"].synchronize()
        graph_mode, pipe.use_graph = pipe.use_graph, False     # per-launch events need real launches, not a graph replay
        ms, cnt, gflops = profiled(run_all, [sl["eng"] for sl in pipe.slots])
        pipe.use_graph = graph_mode
        alg_bytes = profiled.bytes / max(cnt["gemm"], 1)
        ms1, cnt1, gflops1 = profiled(run_one, [eng])
        executed = profiled.flops_all / nprof + non_gemm_flops_executed(cfg, a.views, a.classes) * a.updates
        # HBM-side traffic of the same launches: PMC passes cannot run inside this process, so the figure is STATIC: read
        # from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE measurement of this command (tools/pmc_traffic.py)
        traffic, traffic_src, traffic_regime = None, None, None
        import glob
        cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_gemm_traffic.json")))
        if cand and a.arch == "ViT-B/16" and a.views == 64 and a.classes == 200 and a.rank == 16 and a.updates == 1:
            try:
                tj = json.load(open(cand[-1]))
                traffic = tj["traffic_bytes_per_launch"]
                traffic_src = "profiles/" + os.path.basename(cand[-1])
                traffic_regime = "static: " + tj.get("regime", "rocprofv3 --pmc, streams=1, separate FETCH_SIZE / WRITE_SIZE passes")
            except Exception:
                traffic = None
        ach = gflops / (ms["gemm"] * 1e-3) / 1e12
        ach1 = gflops1 / (ms1["gemm"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": round(ach1, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach1 / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM-side bytes per GEMM launch",
                "traffic_source": traffic_src, "traffic_regime": traffic_regime, "algorithmic_bytes_per_launch": round(alg_bytes),
                "kernel": "gemm_big_kernel<5,3,EPI> (160x256x64 tiles, 8 waves, one persistent block per CU) + gemm_kernel<160,2,2,2,EPI> "
                          "for the MLP-dgrad / patch-embed epilogues: every big-M (M >= 1024) GEMM launch of an episode; the small-M "
                          "launches (1-view inference, CLS-row GEMMs of the last layer) are class gemm_small_m",
                "regime": "one episode at a time (kernel alone on the chip; HIP events on the launch stream)",
                "flops_per_launch": round(gflops1 / max(cnt1["gemm"], 1)),
                "avg_launch_us": round(1e3 * ms1["gemm"] / max(cnt1["gemm"], 1), 2),
                "launches_per_image": cnt1["gemm"] // nprof,
                "class_ms_per_image": {k: round(val / nprof, 3) for k, val in ms1.items()},
                "episodes_in_flight": {"streams": a.streams, "achieved_per_launch": round(ach, 1),
                                       "avg_launch_us": round(1e3 * ms["gemm"] / max(cnt["gemm"], 1), 2),
                                       "note": "per-launch rate while another episode's kernels share the CUs (the timed region's regime)",
                                       "class_ms_per_image": {k: round(val / nprof, 3) for k, val in ms.items()}}}
    pipe.close()

    # ---- second timed leg on the fp16-operand build (the one that meets the 1e-3 parity tolerance), same process
    fp16_leg = None
    if a.precision == "bf16" and not a.no_fp16_leg and world == 1:
        pipe16, _, T16 = timed_run("fp16", max(a.steps // 2, 10), max(a.warmup // 2, 5))
        n16 = max(a.steps // 2, 10)
        fp16_leg = {"value": round(world * n16 / T16, 2), "unit": "images/sec", "ms_per_step": round(1e3 * T16 / n16, 4),
                    "steps": n16, "note": "libttl_hip_fp16.so: IEEE-half MFMA operands (the reference's autocast dtype, ttl.py:79), "
                                          "same kernels, same MFMA rate; meets the 1e-3 tolerance (tests/test_gpu_path.py)"}
        pipe16.close()

    if rank == 0:
        value = world * a.steps / T
        flops = episode_flops(cfg, a.views, a.classes) * a.updates  # (1-view inference counted once per update: <2%)
        out = {
            "metric": "test images/sec (64-view TTA, 1 step), CLIP ViT-B/16 r=16",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "ranks_seen": ranks_seen, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * T / a.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{cfg.name} r={cfg.rank}, {a.views} views, {a.updates} TTA step, K={a.classes} "
                                   f"(ImageNet-A shape), layers {cfg.layer_lo}-{cfg.layer_hi}, episodic reset + adapted "
                                   f"1-view inference; views pre-staged in HBM; {a.steps} images/rank",
                       "arch": cfg.name, "views": a.views, "classes": a.classes, "rank": cfg.rank, "updates": a.updates,
                       "streams_per_gpu": a.streams, "hip_graph": bool(a.graph),
                       "parallelism": f"image-sharded x{world} (item i -> rank i % {world}), {a.streams} episodes in flight per GPU"},
            "tflop_per_image": round(flops / 1e12, 3),
            "tflop_per_image_executed": None if executed is None else round(executed / 1e12, 3),
            "whole_path_tflops_per_gpu": round(flops * value / world / 1e12, 1),
            "whole_path_frac_of_bf16_peak": round(flops * value / world / 1e12 / PEAK_BF16_TFLOPS, 4),
            "whole_path_frac_executed": None if executed is None else round(executed * value / world / 1e12 / PEAK_BF16_TFLOPS, 4),
            "accuracy_accumulator": {"top1_hits": acc["hits1"], "top5_hits": acc["hits5"], "images": acc["count"],
                                     "note": "synthetic labels: exercises the sharded accumulator + all-reduce, not a quality number; "
                                             "no pretrained checkpoint / dataset exists offline, so README top-1 is not reproducible here"},
            "roofline": roof,
        }
        if fp16_leg:
            out["fp16"] = fp16_leg
        if not a.no_parity and cfg.name == "ViT-B/16":
            try:
                out["parity"] = parity_check(a.precision)
                if a.precision == "bf16" and not a.no_fp16_leg:
                    out["parity_fp16"] = parity_check("fp16")
            except Exception as e:      # a missing fixture must not hide the timing; it is reported instead
                out["parity"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, a.classes)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
