"""Sharded evaluation loop — this build's counterpart of ``test_time_adapt_eval`` (ttl.py:300-363).

Test images are independent episodes (LoRA + Adam state are reset before every image,
ttl.py:338-344), so they shard embarrassingly: rank r of W takes the indices i with i % W == r, runs
the single-GPU loop, and ONE all-reduce(SUM) of [top1 hits, top5 hits, count] (3 x int64) per
dataset produces the accuracy (RCCL over xGMI on GPUs; gloo in the CPU tests).  There is no other
collective on the path: nothing about an episode crosses GPUs.
"""
import os

import torch
import torch.distributed as dist

from . import _lib


def dist_env():
    """(rank, local_rank, world) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def _visible_gpus(n_physical, env):
    """Indices (into the runtime's physical enumeration) of the GPUs this process sees, in the order HIP numbers them, or None when the
    environment cannot be resolved safely.  ROCR_VISIBLE_DEVICES filters / reorders the physical list first; HIP_VISIBLE_DEVICES (or
    CUDA_VISIBLE_DEVICES, which HIP honours too) then indexes INTO that subset.  Entries that are not plain indices (GPU-xxxx UUIDs),
    out-of-range indices, or HIP_ and CUDA_ lists that disagree -> None: no pin is better than a pin to the wrong socket."""
    def parse(name):
        v = env.get(name)
        if v is None or v.strip() == "":
            return None
        out = []
        for t in v.split(","):
            t = t.strip()
            if not t.isdigit():
                return False
            out.append(int(t))
        return out
    rocr, hipv, cudav = parse("ROCR_VISIBLE_DEVICES"), parse("HIP_VISIBLE_DEVICES"), parse("CUDA_VISIBLE_DEVICES")
    if rocr is False or hipv is False or cudav is False:
        return None
    if hipv is not None and cudav is not None and hipv != cudav:
        return None
    ids = list(range(n_physical))
    for lst in (rocr, hipv if hipv is not None else cudav):
        if lst is None:
            continue
        if any(i >= len(ids) for i in lst):
            return None
        ids = [ids[i] for i in lst]
    return ids


def _gpus_from_kfd(sysfs):
    """[(numa node or None, pci address or None)] of the GPUs in the runtime's enumeration order: KFD topology nodes with
    simd_count > 0; a node's drm_render_minor leads to /sys/class/drm/renderD<minor>/device/numa_node.  A node whose properties
    file may not be read (EPERM: a container's device cgroup admits only the leased GPUs — seen on the 1-GPU bench boxes, where 7 of
    the 8 GPU nodes answer "Operation not permitted") is a GPU the runtime of this process does not enumerate either: skipped."""
    base = os.path.join(sysfs, "class/kfd/kfd/topology/nodes")
    gpus = []
    for n in sorted(os.listdir(base), key=int):
        try:
            props = dict(l.split()[:2] for l in open(os.path.join(base, n, "properties")) if len(l.split()) >= 2)
        except PermissionError:
            continue
        if int(props.get("simd_count", 0)) > 0:
            dev = os.path.join(sysfs, f"class/drm/renderD{int(props['drm_render_minor'])}/device")
            gpus.append((int(open(os.path.join(dev, "numa_node")).read()), os.path.basename(os.path.realpath(dev))))
    return gpus


def _gpus_from_drm(sysfs, devfs=None):
    """The same list without KFD (containers that hide /sys/class/kfd): AMD (vendor 0x1002) display / accelerator functions behind
    /sys/class/drm/renderD*, ordered by PCI address — the order the ROCm runtime enumerates a node's GPUs in on every box seen.
    sysfs shows every GPU of the host even when the container may open only some: with ``devfs`` (default /dev/dri for the real
    /sys) only the render nodes this process can open count."""
    base = os.path.join(sysfs, "class/drm")
    if devfs is None and sysfs == "/sys":
        devfs = "/dev/dri"
    found = {}
    for c in os.listdir(base):
        if not c.startswith("renderD"):
            continue
        dev = os.path.join(base, c, "device")
        try:
            if int(open(os.path.join(dev, "vendor")).read(), 16) != 0x1002:
                continue
            cls = int(open(os.path.join(dev, "class")).read(), 16) >> 16 if os.path.exists(os.path.join(dev, "class")) else 0x03
            if cls not in (0x03, 0x12):                   # display controller / processing accelerator
                continue
            if devfs and os.path.isdir(devfs) and not os.access(os.path.join(devfs, c), os.R_OK | os.W_OK):
                continue
            found[os.path.basename(os.path.realpath(dev))] = int(open(os.path.join(dev, "numa_node")).read())
        except (OSError, ValueError):
            continue
    def key(addr):
        try:
            dom, bus, rest = addr.split(":")
            d, f = rest.split(".")
            return (int(dom, 16), int(bus, 16), int(d, 16), int(f, 16))
        except ValueError:
            return (1 << 30, 0, 0, 0)
    return [(found[a], a) for a in sorted(found, key=key)]


def _gpus_from_rocm_smi(timeout=20):
    """Last resort: ask `rocm-smi --showtoponuma --json` from a CHILD process (this process must not touch the GPU before it has
    pinned itself).  -> [(numa node, None)] in card order."""
    import json
    import subprocess
    r = subprocess.run(["rocm-smi", "--showtoponuma", "--json"], capture_output=True, text=True, timeout=timeout)
    d = json.loads(r.stdout[r.stdout.index("{"):])
    cards = sorted((k for k in d if k.startswith("card")), key=lambda k: int(k[4:]))
    out = []
    for k in cards:
        node = next((v for kk, v in d[k].items() if "numa node" in kk.lower()), None)
        out.append((int(node), None))
    return out


def gpu_numa_cpus(local_rank, sysfs="/sys", env=None, with_source=False, allow_smi=None):
    """(numa node, set of CPUs) of the NUMA node the ``local_rank``-th VISIBLE GPU hangs off, found without touching the GPU.
    Sources, in order, each tried only when the one before cannot answer: (1) KFD topology -> drm_render_minor -> numa_node;
    (2) /sys/class/drm/card*/device/{vendor,numa_node} ordered by PCI address; (3) `rocm-smi --showtoponuma --json` in a child
    process (only with the real /sys, or ``allow_smi``).  (None, None) when none of them yields a node >= 0 whose cpulist can be read,
    or when the *_VISIBLE_DEVICES environment cannot be resolved (see _visible_gpus).  with_source: a third element naming what
    answered or why nothing did."""
    env = os.environ if env is None else env
    if allow_smi is None:
        allow_smi = (sysfs == "/sys")
    why = []
    sources = [("kfd", lambda: _gpus_from_kfd(sysfs)), ("drm-pci-order", lambda: _gpus_from_drm(sysfs))]
    if allow_smi:
        sources.append(("rocm-smi", _gpus_from_rocm_smi))
    result = (None, None, None)
    for name, fn in sources:
        try:
            gpus = fn()
        except Exception as e:      # unreadable / absent: next source
            why.append(f"{name}: {type(e).__name__}")
            continue
        if not gpus:
            why.append(f"{name}: no GPUs listed")
            continue
        vis = _visible_gpus(len(gpus), env)
        if vis is None:
            result = (None, None, "unresolvable *_VISIBLE_DEVICES (non-index entries, out-of-range, or HIP/CUDA lists that differ)")
            break
        if local_rank >= len(vis):
            why.append(f"{name}: local rank {local_rank} but {len(vis)} visible GPUs")
            continue
        node = gpus[vis[local_rank]][0]
        if node is None or node < 0:
            why.append(f"{name}: numa_node {node}")
            continue
        try:
            cpus = _parse_cpulist(open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")).read())
        except (OSError, ValueError):
            why.append(f"{name}: node{node}/cpulist unreadable")
            continue
        result = (node, cpus, name)
        break
    if result[2] is None:
        result = (None, None, "; ".join(why) or "no source")
    return result if with_source else result[:2]


def pin_to_gpu_numa_node(local_rank, n_local_ranks=1, sysfs="/sys", env=None, apply=True, allow_smi=None):
    """Restrict this process (call it BEFORE torch / HIP start their threads and before any GPU call) to the CPUs of its GPU's
    NUMA node, so the episode's ~130 kernel enqueues per image and the allocator's host threads stay next to the device's PCIe
    root.  Several ranks on one node share its CPUs (the scheduler spreads them).  Returns a small record for the bench line
    (which source answered, or why none did); never raises: an unreadable topology leaves the affinity alone."""
    node, cpus, src = gpu_numa_cpus(local_rank, sysfs, env, with_source=True, allow_smi=allow_smi)
    rec = {"local_rank": int(local_rank), "numa_node": node, "applied": False}
    if node is None:
        rec["reason"] = "GPU -> NUMA node not known: " + str(src)
        return rec
    rec["source"] = src
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:          # pragma: no cover
        rec["reason"] = "no sched_getaffinity on this platform"
        return rec
    target = cpus & allowed
    if len(target) < 2:
        rec["reason"] = f"only {len(target)} of the node's CPUs are allowed for this process"
        return rec
    rec["cpus"] = len(target)
    if apply:
        os.sched_setaffinity(0, target)
        rec["applied"] = True
    return rec


class ShardProgress:
    """Per-rank progress file of a sharded evaluation: {tag, rank, world, next_index, acc}.  The reference's loop
    (ttl.py:321-356) keeps its AverageMeters in memory only — a 50k-image run that dies starts over.  Test images are
    independent episodes and the view RNG is keyed by (seed, global index), so a rank can pick up at the first index it has
    not accounted for: ``resume()`` -> (first index to run, accumulator so far); ``note(i, totals_fn)`` after item i has been
    submitted — every ``every`` owned items the accumulator is fetched (that drains the streams) and the file replaced
    atomically.  A file written for another tag / rank / world size is ignored."""

    def __init__(self, path, rank, world, tag="", every=256):
        self.path = f"{path}.rank{int(rank)}of{int(world)}.json"
        self.key = dict(tag=str(tag), rank=int(rank), world=int(world))
        self.every, self._since = max(int(every), 1), 0

    def resume(self):
        import json
        try:
            with open(self.path) as f:
                d = json.load(f)
            if all(d.get(k) == v for k, v in self.key.items()):
                return int(d["next_index"]), [int(v) for v in d["acc"]]
        except (OSError, ValueError, KeyError, TypeError):
            pass
        return 0, [0, 0, 0]

    def write(self, next_index, acc):
        import json
        tmp = self.path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(dict(self.key, next_index=int(next_index), acc=[int(v) for v in acc]), f)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, self.path)

    def note(self, i, totals_fn):
        self._since += 1
        if self._since >= self.every:
            self._since = 0
            self.write(i + 1, totals_fn())


def shard_indices(n_items, rank, world):
    """Strided partition: invariant per-index work assignment, balanced to within one item."""
    return range(rank, n_items, world)


class ImageShard:
    """Which test images a rank runs, and the path's collectives.  ONE object used by bench.py,
    ``eval.test_time_adapt_eval`` and ``evaluate_sharded``, so the world-size-2 gloo tests exercise the code the
    GPU runs use.  Item ``i`` of a dataset belongs to rank ``i % world`` (ttl.py:321's loop, strided)."""

    def __init__(self, rank=0, world=1, group=None):
        if not 0 <= rank < max(world, 1):
            raise ValueError(f"rank {rank} outside world {world}")
        self.rank, self.world, self.group = int(rank), max(int(world), 1), group

    @classmethod
    def from_env(cls, group=None):
        rank, _, world = dist_env()
        return cls(rank, world, group)

    def owns(self, i):
        return i % self.world == self.rank

    def indices(self, n_items):
        return shard_indices(n_items, self.rank, self.world)

    def _reduce(self, t, op):
        if self.world == 1:
            return t
        # gloo (CPU tests, 1-GPU smoke runs) reduces host tensors; nccl (= RCCL over xGMI) device tensors
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            c = t.cpu()
            dist.all_reduce(c, op=op, group=self.group)
            return c.to(t.device)
        dist.all_reduce(t, op=op, group=self.group)
        return t

    def sum(self, t):
        """C1: all-reduce(SUM) of the [hits1, hits5, count] accumulator — the only collective of the path."""
        return self._reduce(t, dist.ReduceOp.SUM)

    def max(self, t):
        return self._reduce(t, dist.ReduceOp.MAX)

    def gather(self, t):
        """[world, t.numel()]: row r = rank r's ``t`` (measurement only — bench.py's per-rank rates; an all-reduce(SUM) of a
        matrix in which every rank fills its own row, so it rides the same collective the accumulator uses)."""
        if self.world == 1:
            return t.reshape(1, -1).clone()
        buf = torch.zeros((self.world, t.numel()), dtype=t.dtype, device=t.device)
        buf[self.rank] = t.reshape(-1)
        return self._reduce(buf, dist.ReduceOp.SUM)

    def ranks_seen(self, device="cpu"):
        """Number of ranks that really take part (all-reduce of ones): bench.py prints it next to n_gpus."""
        return int(self.sum(torch.ones(1, dtype=torch.int64, device=device)).item())

    def accuracy(self, acc):
        """acc: int64 [hits1, hits5, count] of THIS rank -> dict identical on every rank."""
        hits1, hits5, count = (int(v) for v in self.sum(acc).tolist())
        return dict(hits1=hits1, hits5=hits5, count=count,
                    top1=100.0 * hits1 / max(count, 1), top5=100.0 * hits5 / max(count, 1))


def topk_hits(logits, target, ks=(1, 5)):
    """utils/tools.py:88-102 ``accuracy`` as integer hit counts (device tensors, no host sync)."""
    k = min(max(ks), logits.shape[1])
    pred = logits.topk(k, dim=1).indices                        # [B,k]
    eq = pred.eq(target.view(-1, 1))
    return [eq[:, :min(kk, k)].any(dim=1).sum() for kk in ks]


def evaluate_sharded(predict_fn, n_items, label_fn, rank=0, world=1, device="cpu", group=None, progress=None):
    """predict_fn(i) -> logits [1,K] after adaptation on test item i; label_fn(i) -> int.
    ``progress``: a ShardProgress — items this rank has already accounted for are skipped and its saved accumulator continues.
    Returns dict(top1, top5, count, hits1, hits5) — identical on every rank."""
    shard = ImageShard(rank, world, group)
    acc = torch.zeros(3, dtype=torch.int64, device=device)      # [hits1, hits5, count]
    start = 0
    if progress is not None:
        start, acc0 = progress.resume()
        acc += torch.tensor(acc0, dtype=torch.int64, device=device)
    for i in shard.indices(n_items):
        if i < start:
            continue
        logits = predict_fn(i)
        tgt = torch.tensor([label_fn(i)], device=logits.device)
        h1, h5 = topk_hits(logits, tgt)
        acc[0] += h1.to(acc.device)
        acc[1] += h5.to(acc.device)
        acc[2] += 1
        if progress is not None:
            progress.note(i, lambda: acc.tolist())
    if progress is not None:
        progress.write(n_items, acc.tolist())
    return shard.accuracy(acc)


class EpisodeRunner:
    """Fused per-image episode on a ``ClipTestTimeTuning`` model: reset -> n_updates x (forward,
    loss, LoRA backward, AdamW) -> adapted 1-view inference, as one enqueue (ttl_episode)."""

    def __init__(self, model, args):
        if getattr(args, "reweight_plpd", 0):
            raise NotImplementedError("reweight_plpd is commented out in the reference (deyo.py:176)")
        self.plpd = None
        if getattr(args, "filter_plpd", 0):      # deyo.py:115-151 inside the fused episode (round 5): ttl_episode_args.plpd
            from .deyo import plpd_spec
            if not (bool(args.deyo_selection) and args.lora_encoder != 'prompt'):
                raise NotImplementedError("the PLPD filter belongs to the DeYO objective (deyo.py:115)")
            self.plpd = plpd_spec(args)
        self.model = model
        self.args = args
        self.eng = model._ensure_engine()
        self.snap = model.snapshot_flat()
        deyo = bool(args.deyo_selection) and args.lora_encoder != 'prompt'
        self.kw = dict(
            n_updates=(args.tta_steps ** 2 if deyo else args.tta_steps),
            objective="deyo" if deyo else "tpt",
            mode=1 if getattr(args, "filter_ent", 0) else 0, rho=args.selection_p, margin=args.deyo_margin_e0,
            reweight=float(getattr(args, "reweight_ent", 1)), lr=args.lr)

    def __call__(self, views):
        """views [N,3,S,S] (view 0 = the un-augmented image) -> logits [1,K] with adapted weights."""
        m = self.model
        if self.plpd is None:
            return self.eng.episode(views, self.snap, m._opt_m, m._opt_v, **self.kw)
        from .deyo import draw_plpd_perms, plpd_candidates
        nc = plpd_candidates(self.args, views.shape[0], self.eng.n_classes)
        if nc is None:
            raise NotImplementedError("the first-stage selection count is data-dependent here (threshold mode with more than 1000 "
                                      "classes): use ttl.test_time_tuning (step-wise path)")
        if nc == 0:      # int(N * selection_p) == 0: the reference returns before the PLPD stage and the update (deyo.py:110-113)
            return self.eng.episode(views, self.snap, m._opt_m, m._opt_v, **self.kw)
        text = m.lora_encoder == 'text'
        perm = draw_plpd_perms(self.plpd, self.kw["n_updates"], nc, views.shape[-1], self.eng.device)
        st = self.eng.plpd_struct(self.plpd, perm, nc, None if text else m._aux_engine()) if not text else \
            self.eng.txt.plpd_struct(self.plpd, perm, nc, None)
        return self.eng.episode(views, self.snap, m._opt_m, m._opt_v, plpd=st, **self.kw)


class EpisodePipeline:
    """S independent episodes in flight on S HIP streams of ONE GPU.

    Episodes of different test images share nothing but the frozen weights (ttl.py:338-344 resets
    LoRA + Adam state per image), so the launch tails, small kernels and HBM-bound phases of one
    image overlap with the MFMA-bound phases of another: +19 % images/s at S = 2 and +23 % at S = 3 on
    MI355X (S = 4 is slower again; bench.py --streams).  Every slot owns a context (its activation arena), its LoRA /
    gradient / Adam buffers and a stream; results are identical to running the slots one by one.
    """

    def __init__(self, cfg, weights, lora_names, lora_init, text_features, logit_scale_exp, device, n_streams=2,
                 max_views=64, precision=None, engine_factory=None, n_classes=None, use_graph=False):
        """``engine_factory`` (optional): callable returning a ready engine (weights loaded, peer features /
        prompts set, LoRA unbound) with ``bind_lora`` / ``episode`` / ``close`` — used for
        ``--lora_encoder text`` (custom_clip.build_text_mode_engine); default: an image-tower TTLEngine."""
        from .engine import TTLEngine
        self.slots = []
        dev = torch.device(device)
        # one copy of the frozen weights per GPU (the reference's single `model`, ttl.py:178-179): slots after the first read
        # slot 0's images (ttl_ctx_create_shared).  TTL_SHARE_WEIGHTS=0: a private copy per slot (A/B, tools/ab_env.py).
        share = os.environ.get("TTL_SHARE_WEIGHTS", "1") != "0"
        # TTL_STREAM_PRIOS="-1,0,0" (experiments): a HIP stream priority per slot (lower = more urgent); default: all equal
        prios = [int(v) for v in os.environ.get("TTL_STREAM_PRIOS", "").split(",") if v.strip()]
        for _ in range(max(1, int(n_streams))):
            if engine_factory is not None:
                eng = engine_factory()
            else:
                owner = self.slots[0]["eng"] if (share and self.slots) else None
                eng = TTLEngine(cfg, max_views, text_features.shape[0], dev, precision, share_from=owner)
                if owner is None:
                    eng.load_weights(weights)
                eng.set_text_features(text_features, logit_scale_exp)
            flat = torch.cat([torch.as_tensor(lora_init[k]).reshape(-1).float() for k in lora_names]).to(dev).contiguous()
            eng.bind_lora(flat)
            self.slots.append(dict(eng=eng, flat=flat, snap=flat.clone(), m=torch.zeros_like(flat), v=torch.zeros_like(flat),
                                   stream=torch.cuda.Stream(device=dev, priority=prios[len(self.slots) % len(prios)]) if prios
                                   else torch.cuda.Stream(device=dev),
                                   acc=torch.zeros(3, dtype=torch.int64, device=dev)))   # [hits1, hits5, count]
        # every context learns how many episodes share the GPU: with others in flight a GEMM launch is chosen by its CU-time, not its
        # makespan (include/ttl_hip.h ttl_ctx_set_concurrency; text mode: both towers)
        # (TTL_CONCURRENCY: profiling runs replay ONE stream with the kernels the three-stream run picks)
        self.concurrency = max(len(self.slots), int(os.environ.get("TTL_CONCURRENCY", "0") or 0))
        for sl in self.slots:
            for e in (sl["eng"], getattr(sl["eng"], "img", None), getattr(sl["eng"], "txt", None)):
                if e is not None and hasattr(e, "set_concurrency"):
                    e.set_concurrency(self.concurrency)
        self._next = 0
        self._text_features, self._logit_scale, self._text_version = text_features, logit_scale_exp, 0     # (for the PLPD auxiliary contexts)
        # use_graph: every slot replays its episode as ONE hipGraphLaunch (captured on first use per argument set)
        # over a slot-owned copy of the views: the host then spends ~0.1 ms per image instead of ~2.7 ms of kernel
        # enqueues, which is what bounds runs with few views (8 views = < 1 ms of GPU time).  Image-tower slots only.
        self.use_graph = bool(use_graph) and engine_factory is None
        self.max_classes = int(n_classes if n_classes is not None else text_features.shape[0])
        self.lora_names = list(lora_names)
        torch.cuda.synchronize(dev)

    def rebind(self, lora_init, text_features=None, logit_scale_exp=None, prompts=None):
        """New dataset on the same frozen weights (the reference loops over set_ids, ttl.py:262-298):
        fresh class-text features (image mode) or tokenized prompts (text mode), LoRA snapshot and
        accuracy accumulators; contexts and arenas stay."""
        n_new = int(prompts.shape[0] if prompts is not None else text_features.shape[0])
        if n_new > self.max_classes:
            raise ValueError("more classes than the pipeline was built for")
        self.synchronize()
        if text_features is not None:
            self._text_features, self._logit_scale, self._text_version = text_features, logit_scale_exp, self._text_version + 1
        for sl in self.slots:
            if prompts is not None:
                sl["eng"].txt.set_prompts(prompts)
            else:
                sl["eng"].set_text_features(text_features, logit_scale_exp)
            flat = torch.cat([torch.as_tensor(lora_init[k]).reshape(-1).float() for k in self.lora_names]).to(sl["flat"].device)
            sl["flat"].copy_(flat)
            sl["snap"].copy_(flat)
            sl["m"].zero_()
            sl["v"].zero_()
            sl["acc"].zero_()
            sl["gkey"] = None          # class count is baked into a captured graph: recapture on next use
        self._next = 0
        torch.cuda.synchronize(self.slots[0]["flat"].device)

    def _plpd_struct(self, sl, plpd, n_updates, views):
        """ttl_plpd_args of one image on slot ``sl``: the slot's auxiliary context (image mode; created on first use on the
        owner's weight images, bound to the slot's adapter buffer) and this image's permutations, drawn on the host like the
        reference draws them and copied into a slot-owned device buffer (whose address a captured graph keeps)."""
        from .deyo import draw_plpd_perms
        from .engine import TTLEngine
        spec, nc = plpd["spec"], int(plpd["n_candidates"])
        eng = sl["eng"]
        text = hasattr(eng, "txt")
        aux = None
        if not text:
            aux = sl.get("aux")
            if aux is None:
                aux = sl["aux"] = TTLEngine(eng.cfg, eng.max_views, eng.max_classes, eng.device, eng.precision,
                                            share_from=self.slots[0]["eng"])          # (slot 0 owns its weight images)
                aux.bind_lora(sl["flat"])
                aux.set_concurrency(self.concurrency)
                sl["aux_classes"] = None
            if sl.get("aux_classes") != self._text_version:
                aux.set_text_features(self._text_features, self._logit_scale)
                sl["aux_classes"] = self._text_version
        from .deyo import plpd_perm_shape
        shape = plpd_perm_shape(spec, n_updates, nc, views.shape[-1])
        buf = None
        if shape is not None:
            buf = sl.get("permbuf")
            if buf is None or tuple(buf.shape) != shape:
                buf = sl["permbuf"] = torch.empty(shape, dtype=torch.int32, device=views.device)
                sl["permpin"] = [torch.empty(shape, dtype=torch.int32).pin_memory() for _ in range(2)]     # (pinning per image costs ms)
                sl["permev"] = [None, None]
                sl["permi"] = 0
                sl["gkey"] = None                       # a captured graph holds the old buffer's address
            k = sl["permi"] = sl["permi"] ^ 1            # two staging buffers: the copy of image i-2 of this slot has long completed
            if sl["permev"][k] is not None:
                sl["permev"][k].synchronize()
            draw_plpd_perms(spec, n_updates, nc, views.shape[-1], "cpu", out=sl["permpin"][k])      # straight into the pinned staging buffer
            buf.copy_(sl["permpin"][k], non_blocking=True)       # on the slot's stream, ahead of the episode
            sl["permev"][k] = torch.cuda.Event()
            sl["permev"][k].record()
        owner = eng.txt if text else eng
        return owner.plpd_struct(spec, buf, nc, aux)

    def submit(self, views, target=None, persistent_input=False, want_output=True, plpd=None, **episode_kw):
        """Enqueue one episode on the next slot's stream; returns the (future) logits1 tensor [1,K] (None with
        ``want_output=False``: the caller only wants the accuracy accumulator — bench.py — and no copy is made).
        With ``target`` (device int64 [1]) the slot's [hits1, hits5, count] accumulator is updated INSIDE the episode's enqueue
        (ttl_episode_args.target / hits_out: one hit-count launch of the library, utils/tools.py:88-102 semantics; no torch
        kernel runs on the hot path and nothing synchronises).  The caller must not overwrite ``views`` / ``target`` until the
        slot's stream has caught up.  ``persistent_input`` (graph replay only): ``views`` and ``target`` are buffers the caller
        keeps alive and re-submits (bench.py's pre-staged batches) — the slot captures one graph per such (views, target) pair and
        replays it in place, instead of copying every batch into the slot's own input buffer first (38.5 MB per image at 64 views)."""
        sl = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        sl["stream"].wait_stream(torch.cuda.current_stream())
        views.record_stream(sl["stream"])          # the caching allocator must not recycle it under the slot's stream
        hits = sl["acc"] if target is not None else None
        if target is not None:
            if not (torch.is_tensor(target) and target.is_cuda and target.dtype == torch.int64 and target.numel() >= 1 and target.is_contiguous()):
                raise _lib.TtlError("target must be a contiguous device int64 tensor (topk_hits_kernel reads 8 bytes per label); "
                                    "coerce with .to(device, dtype=torch.int64)")
            target.record_stream(sl["stream"])
        with torch.cuda.stream(sl["stream"]):
            if plpd is not None:       # --filter_plpd 1 (deyo.py:115-151) inside the fused episode
                episode_kw = dict(episode_kw, plpd=self._plpd_struct(sl, plpd, int(episode_kw.get("n_updates", 1)), views))
            if self.use_graph:
                key = (tuple(views.shape), target is not None, tuple(sorted((k, v) for k, v in episode_kw.items() if k != "plpd")),
                       None if plpd is None else (tuple(sorted(plpd["spec"].items())), plpd["n_candidates"]))
                if sl.get("gkey") != key:          # another shape / argument set: every captured graph of the slot is stale
                    sl["gkey"], sl["graph"], sl["xbuf"], sl["graphs_in_place"] = key, None, None, {}
                    sl["obuf"] = torch.empty((1, sl["eng"].n_classes), dtype=torch.float32, device=views.device)
                    sl["tbuf"] = torch.zeros(1, dtype=torch.int64, device=views.device) if target is not None else None
                inplace = sl["graphs_in_place"]
                pkey = (views.data_ptr(), target.data_ptr() if target is not None else 0)
                if persistent_input and (pkey in inplace or len(inplace) < 16):
                    g = inplace.get(pkey)
                    if g is None:                   # (the capture runs this episode already: its hit is counted)
                        g = inplace[pkey] = (sl["eng"].episode_graph(views, sl["snap"], sl["m"], sl["v"], sl["obuf"], target=target,
                                                                     hits=hits, **episode_kw), views, target)
                    else:
                        g[0]()
                elif sl["graph"] is None:
                    sl["xbuf"] = torch.empty_like(views)
                    sl["xbuf"].copy_(views)
                    if target is not None:
                        sl["tbuf"].copy_(target.reshape(-1)[:1])
                    sl["graph"] = sl["eng"].episode_graph(sl["xbuf"], sl["snap"], sl["m"], sl["v"], sl["obuf"], target=sl["tbuf"],
                                                          hits=hits, **episode_kw)      # the capture ran this episode already
                else:
                    sl["xbuf"].copy_(views, non_blocking=True)
                    if target is not None:
                        sl["tbuf"].copy_(target.reshape(-1)[:1], non_blocking=True)
                    sl["graph"]()
                out = sl["obuf"].clone() if want_output else None       # obuf is overwritten by the slot's next episode
            else:
                out = sl["eng"].episode(views, sl["snap"], sl["m"], sl["v"], target=target, hits=hits, **episode_kw)
                if not want_output:
                    out = None
        return out

    def synchronize(self):
        for sl in self.slots:
            sl["stream"].synchronize()

    def reset_totals(self):
        self.synchronize()
        for sl in self.slots:
            sl["acc"].zero_()

    def totals(self):
        """Sum of the per-slot accuracy accumulators (device int64 [3]) after draining the streams."""
        self.synchronize()
        return torch.stack([sl["acc"] for sl in self.slots]).sum(0)

    def close(self):
        for sl in reversed(self.slots):          # the owner of the shared weight images (slot 0) goes last
            if sl.get("aux") is not None:
                sl["aux"].close()
            sl["eng"].close()
