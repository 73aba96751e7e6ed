"""Evaluation loop over a dataset — this build's counterpart of ``test_time_adapt_eval``
(ttl.py:300-363), with the per-image body fused into one enqueue and images sharded over ranks.

    export PYTHONPATH=ttl-test-time-low-rank-adaptation_amd
    python -m ttl_amd.eval --arch ViT-B/16 --images 256 --views 64 --classes 200
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m ttl_amd.eval ...

Semantics kept from the reference loop: model.eval(); per image LoRA reset + empty Adam state
(ttl.py:338-344), adaptation on all views (ttl.py:347), prediction on view 0 with the adapted
weights (ttl.py:350-352), top-1 / top-5 as hit percentages (utils/tools.py:88-102).  What differs:
text features are cached per dataset (Q12), ``n_streams`` images are in flight per GPU, and
rank r takes the items i with i % world == r; ONE all-reduce(SUM) of [hits1, hits5, count]
ends the dataset (RCCL over xGMI on GPUs).
"""
import argparse
import itertools
import json
import time

import torch
import torch.distributed as dist

from . import synth
from .driver import EpisodePipeline, ImageShard, dist_env


def episode_kwargs_from_args(args):
    """Reference CLI semantics -> fused-episode arguments (incl. tta_steps**2 on the DeYO branch, Q6).  ``--filter_plpd 1``
    travels separately (EpisodePipeline.submit(plpd=...): per-image permutations)."""
    if getattr(args, "reweight_plpd", 0):
        raise NotImplementedError("reweight_plpd: the term is commented out in the reference (deyo.py:176)")
    deyo = bool(args.deyo_selection) and args.lora_encoder != 'prompt'
    return dict(n_updates=(args.tta_steps ** 2 if deyo else args.tta_steps), objective="deyo" if deyo else "tpt",
                mode=1 if getattr(args, "filter_ent", 0) else 0, rho=args.selection_p, margin=args.deyo_margin_e0,
                reweight=float(getattr(args, "reweight_ent", 1)), lr=args.lr)


def test_time_adapt_eval(val_loader, model, model_state, optimizer, optim_state, scaler, args, n_streams=3,
                         rank=0, world=1, gpu_augmenter=None, progress=None):
    """Same call shape as ttl.py:300.  ``val_loader`` yields (images, target) with images either a
    list of [1,3,S,S] tensors (view 0 first, like AugMixAugmenter) or one [N,3,S,S] tensor — or, with
    ``gpu_augmenter`` (views.GpuAugMixAugmenter), one decoded uint8 [H,W,3] image whose views are then
    generated on the GPU (bit-exact with the host Pillow pipeline for the same crop boxes).
    ``optimizer`` supplies the AdamW hyper-parameters; ``model_state``/``optim_state``/``scaler`` are
    accepted for signature compatibility (the fused episode resets LoRA and Adam state itself).
    ``progress`` (driver.ShardProgress): per-rank resume file — items this rank has accounted for are skipped and its saved
    [hits1, hits5, count] continues (a sharded 50k-image run that dies does not start over).
    Returns [top1, top5] in percent, identical on every rank."""
    from .deyo import _adam_hparams
    model.eval()
    eng = model._ensure_engine()
    plpd = None
    if getattr(args, "filter_plpd", 0) or getattr(args, "reweight_plpd", 0):
        # --filter_plpd 1 (deyo.py:115-151).  Round 5: the destroyed views, the second forward and the keep mask are a stage of the
        # fused episode (ttl_episode_args.plpd) whenever the host knows how many views the first selection stage yields (always in
        # top-rho mode; in threshold mode for K <= 1000 classes) and the objective is DeYO; otherwise — and with
        # TTL_PLPD_STEPWISE=1 — the reference-shaped per-image loop (ttl.py:338-352) on the step-wise entry points
        import os
        from .deyo import plpd_candidates, plpd_spec
        deyo = bool(args.deyo_selection) and args.lora_encoder != 'prompt'
        n_cls0 = int(model.prompt_learner.tokenized_prompts.shape[0]) if model.lora_encoder == 'text' else int(model.text_features.shape[0])
        # (fused stage: the first-stage candidate count must be known without looking at the logits — top-rho mode, or threshold
        #  mode with K <= 1000 classes; the count itself is taken per image from the loader's actual view count below)
        ok = deyo and not getattr(args, "reweight_plpd", 0) and plpd_candidates(args, 1, n_cls0) is not None and os.environ.get("TTL_PLPD_STEPWISE", "0") != "1"
        if not ok:
            return _host_loop_eval(val_loader, model, optimizer, optim_state, scaler, args, rank, world, gpu_augmenter)
        plpd = plpd_spec(args)
    _, lr, betas, eps, wd = _adam_hparams(optimizer, model)
    kw = episode_kwargs_from_args(args)
    kw.update(lr=lr, betas=betas, eps=eps, weight_decay=wd)
    names = [f"p{i}" for i in range(len(model.trainable_lora_parameters()))]
    init = {n: t for n, t in zip(names, _split(model.snapshot_flat(), model.trainable_lora_parameters()))}
    # the slots (contexts, arenas, weight images) outlive one dataset: the reference loops over set_ids
    # on one model (ttl.py:262-298); only class-text features and the LoRA snapshot change
    key = (int(n_streams), model.precision, eng.max_views)
    cache = model.__dict__.setdefault("_episode_pipelines", {})
    pipe = cache.get(key)
    scale = float(model.logit_scale.exp())
    text_mode = model.lora_encoder == 'text'
    prompts = model.prompt_learner.tokenized_prompts
    n_cls = int(prompts.shape[0]) if text_mode else int(model.text_features.shape[0])
    if pipe is not None and pipe.max_classes >= n_cls:
        if text_mode:
            pipe.rebind(init, prompts=prompts)
        else:
            pipe.rebind(init, model.text_features, scale)
    else:
        if pipe is not None:
            pipe.close()
        factory = None
        if text_mode:    # clip/custom_clip.py:602-607: adapters on the text tower, image tower forward-only
            from .custom_clip import build_text_mode_engine
            factory = lambda: build_text_mode_engine(model.cfg, model.tcfg, model._vision_state, model._text_state, prompts,
                                                     scale, eng.device, eng.max_views, n_cls, model.precision)
        pipe = cache[key] = EpisodePipeline(model.cfg, model._vision_state, names, init, model.text_features, scale,
                                            eng.device, n_streams=n_streams, max_views=eng.max_views,
                                            precision=model.precision, engine_factory=factory, n_classes=n_cls)
    dev = eng.device
    shard = ImageShard(rank, world)
    start, acc0 = (0, [0, 0, 0]) if progress is None else progress.resume()
    acc0 = torch.tensor(acc0, dtype=torch.int64, device=dev)
    n_seen = 0
    for i, (images, target) in enumerate(val_loader):
        n_seen = i + 1
        if not shard.owns(i) or i < start:
            continue
        if gpu_augmenter is not None and torch.is_tensor(images) and images.dtype == torch.uint8:
            images = gpu_augmenter(images.to(dev, non_blocking=True), i)                    # datautils.py:141-157 on the GPU
        elif isinstance(images, (list, tuple)):
            images = torch.cat([im.to(dev, non_blocking=True) for im in images], dim=0)     # ttl.py:324-336
        else:
            images = images.to(dev, non_blocking=True)
            if images.dim() == 5:
                images = images.squeeze(0)
        tgt = torch.as_tensor(target).reshape(-1)[:1].to(dev, dtype=torch.int64)     # the device-side hit count reads an int64 label
        if plpd is not None:
            from .deyo import plpd_candidates
            nc = plpd_candidates(args, images.shape[0], n_cls)
            if nc:
                pipe.submit(images, target=tgt, plpd=dict(spec=plpd, n_candidates=nc), **kw)
            else:       # int(N * selection_p) == 0 (fewer than 10 views at rho 0.1): the reference returns before the PLPD stage and
                pipe.submit(images, target=tgt, **kw)       # the update (deyo.py:110-113); the episode's empty top-rho selection does the same
        else:
            pipe.submit(images, target=tgt, **kw)
        if progress is not None:
            progress.note(i, lambda: (pipe.totals() + acc0).tolist())
    totals = pipe.totals() + acc0
    if progress is not None:
        progress.write(n_seen, totals.tolist())
    r = shard.accuracy(totals)
    return [r["top1"], r["top5"]]


test_time_adapt_eval.__test__ = False  # not a pytest test


def _host_loop_eval(val_loader, model, optimizer, optim_state, scaler, args, rank, world, gpu_augmenter):
    """ttl.py:321-356 image by image on the step-wise HIP entry points (forward / loss / backward / AdamW)."""
    import copy
    from .driver import topk_hits
    from .ttl import test_time_tuning
    dev = model._ensure_engine().device
    shard = ImageShard(rank, world)
    if optim_state is None:
        optim_state = copy.deepcopy(optimizer.state_dict())
    acc = torch.zeros(3, dtype=torch.int64, device=dev)
    for i, (images, target) in enumerate(val_loader):
        if not shard.owns(i):
            continue
        if gpu_augmenter is not None and torch.is_tensor(images) and images.dtype == torch.uint8:
            images = gpu_augmenter(images.to(dev, non_blocking=True), i)
        elif isinstance(images, (list, tuple)):
            images = torch.cat([im.to(dev, non_blocking=True) for im in images], dim=0)
        else:
            images = images.to(dev, non_blocking=True)
            if images.dim() == 5:
                images = images.squeeze(0)
        with torch.no_grad():
            model.LoRA_reset()                                                    # ttl.py:341-343
        optimizer.load_state_dict(optim_state)                                    # ttl.py:344
        test_time_tuning(model, images, optimizer, scaler, args)                  # ttl.py:347
        with torch.no_grad():
            out = model(images[:1])                                               # ttl.py:350-352
        h1, h5 = topk_hits(out, torch.as_tensor(target).reshape(-1)[:1].to(dev))
        acc[0] += h1
        acc[1] += h5
        acc[2] += 1
    r = shard.accuracy(acc)
    return [r["top1"], r["top5"]]


def _split(flat, like):
    out, off = [], 0
    for p in like:
        out.append(flat[off:off + p.numel()].view(p.shape))
        off += p.numel()
    return out


class SyntheticViews:
    """Deterministic stand-in for the AugMix view generator: item i -> ([N,3,S,S] views, label)."""

    def __init__(self, cfg, n_items, n_views, n_classes, seed=0):
        self.cfg, self.n, self.v, self.k, self.seed = cfg, n_items, n_views, n_classes, seed

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield torch.from_numpy(synth.views(self.cfg, self.v, self.seed * 100003 + i)), i % self.k


class SyntheticImages:
    """Deterministic decoded uint8 [H,W,3] images (ImageNet-like 375x500) for the GPU view generator."""

    def __init__(self, n_items, n_classes, height=375, width=500, seed=0, pool=16):
        import numpy as np
        self.n, self.k = n_items, n_classes
        rng = np.random.default_rng(seed)
        # a small pool of distinct images in pinned memory, cycled: stands in for a decoded-image queue
        self.pool = [torch.from_numpy(rng.integers(0, 256, (height, width, 3), dtype=np.uint8)) for _ in range(min(pool, n_items))]
        if torch.cuda.is_available():
            self.pool = [t.pin_memory() for t in self.pool]

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            yield self.pool[i % len(self.pool)], i % self.k


RESUME_TAG_FIELDS = ("arch", "images", "views", "classes", "rank", "lr", "tta_steps", "selection_p", "filter_ent", "deyo_selection",
                     "deyo_margin_e0", "reweight_ent", "streams", "precision", "gpu_views", "lora_encoder", "seed",
                     "filter_plpd", "plpd_threshold", "aug_type", "patch_len", "occlusion_size", "row_start", "column_start")


def resume_tag(a):
    """Identity of a run for driver.ShardProgress: every argument that affects a result (precision, objective, step count,
    learning rate, selection / margin / reweighting, view generation and its seed) or the work split (streams only changes
    the order of execution, but it is cheap to be strict).  Two runs with different tags never share a progress file."""
    return "|".join(f"{k}={getattr(a, k, None)!r}" for k in RESUME_TAG_FIELDS)


def main():
    ap = argparse.ArgumentParser(description="TTL evaluation on synthetic views (no datasets on the GPU box)")
    ap.add_argument("--arch", default="ViT-B/16")
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--views", type=int, default=64)
    ap.add_argument("--classes", type=int, default=200)
    ap.add_argument("--rank", type=int, default=16)
    ap.add_argument("--lr", type=float, default=5e-3)
    ap.add_argument("--tta_steps", type=int, default=1)
    ap.add_argument("--selection_p", type=float, default=0.1)
    ap.add_argument("--filter_ent", type=int, default=0)
    ap.add_argument("--deyo_selection", default=True)
    ap.add_argument("--deyo_margin_e0", type=float, default=0.4)
    ap.add_argument("--reweight_ent", type=int, default=1)
    # the PLPD filter of DeYO, with the reference's CLI names and defaults (ttl.py:408-422)
    ap.add_argument("--filter_plpd", type=int, default=0)
    ap.add_argument("--reweight_plpd", type=int, default=0)
    ap.add_argument("--plpd_threshold", type=float, default=0.2)
    ap.add_argument("--aug_type", default="patch", choices=["patch", "pixel", "occ"])
    ap.add_argument("--patch_len", type=int, default=6)
    ap.add_argument("--occlusion_size", type=int, default=112)
    ap.add_argument("--row_start", type=int, default=56)
    ap.add_argument("--column_start", type=int, default=56)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--precision", default=None, choices=["bf16", "fp16", "strict"],
                    help="MFMA operand dtype; default: ttl_amd._lib.DEFAULT_PRECISION = fp16 (the reference's autocast dtype), or TTL_PRECISION")
    ap.add_argument("--gpu_views", type=int, default=0, help="1: decoded uint8 images in, views generated on the GPU")
    ap.add_argument("--lora_encoder", default="image", choices=["image", "text"])
    ap.add_argument("--seed", type=int, default=0, help="seed of the synthetic data and of the GPU view generator's crop boxes")
    ap.add_argument("--resume_file", default=None, help="prefix of the per-rank progress files (driver.ShardProgress): a run that "
                    "is started again with the same arguments continues after the last recorded image of every rank")
    a = ap.parse_args()
    from . import _lib
    a.precision = _lib.resolve_precision(a.precision)      # the resume tag and the result line name the build that ran
    rank, local, world = dist_env()
    from .driver import ShardProgress, pin_to_gpu_numa_node
    pin_to_gpu_numa_node(local, world)          # before the first GPU call: host threads next to the device's PCIe root
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from .config import get_config
    from .custom_clip import ClipTestTimeTuning
    cfg = get_config(a.arch)
    model = ClipTestTimeTuning(local, [f"class {i}" for i in range(a.classes)], None, arch=a.arch,
                               layer_range=[cfg.layer_lo, cfg.layer_hi], init_method="xavier", lora_encoder=a.lora_encoder,
                               rank=a.rank, max_views=a.views, max_classes=a.classes, precision=a.precision)
    opt = torch.optim.AdamW([{"params": [p]} for p in model.trainable_lora_parameters()], lr=a.lr)
    aug = None
    if a.gpu_views:
        from .views import GpuAugMixAugmenter
        data = SyntheticImages(a.images, a.classes, seed=a.seed)
        aug = GpuAugMixAugmenter(a.views - 1, model.cfg.image_size, precision=a.precision, seed=a.seed)
    else:
        data = SyntheticViews(model.cfg, a.images, a.views, a.classes, seed=a.seed)
    # untimed first pass: builds the per-stream contexts (weight images, arenas), like loading the model
    test_time_adapt_eval(itertools.islice(iter(data), 2 * world), model, None, opt, None, None, a, n_streams=a.streams, rank=rank,
                         world=world, gpu_augmenter=aug)
    torch.cuda.synchronize()
    t0 = time.time()
    progress, resumed_at = None, 0
    if a.resume_file:
        # the tag names EVERY argument that changes what an item's result is (or which items this rank owns): a progress
        # file written under any other setting is ignored instead of being mixed into this run's accumulator
        progress = ShardProgress(a.resume_file, rank, world, tag=resume_tag(a))
        resumed_at = progress.resume()[0]
        if resumed_at:
            print(f"[rank {rank}] resuming from index {resumed_at} of {a.images} ({a.resume_file})", flush=True)
    top1, top5 = test_time_adapt_eval(data, model, None, opt, None, None, a, n_streams=a.streams, rank=rank, world=world, progress=progress,
                                      gpu_augmenter=aug)
    dt = time.time() - t0
    # throughput over the images this run really processed (a resumed run skips what the file already accounts for)
    done = torch.tensor([sum(1 for i in range(rank, a.images, world) if i >= resumed_at)], dtype=torch.int64, device=f"cuda:{local}")
    done = int(ImageShard(rank, world).sum(done).item())
    if rank == 0:
        print(json.dumps({"top1": top1, "top5": top5, "images": a.images, "images_processed_this_run": done,
                          "resumed_from_index": resumed_at, "world": world, "gpu_views": bool(a.gpu_views),
                          "images_per_sec_incl_input_generation": round(done / dt, 2) if done else None}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
