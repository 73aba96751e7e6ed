#!/usr/bin/env python3
"""Generate the committed golden fixtures by running the REFERENCE itself
(/root/reference, imported unmodified through tests/golden/_ref_harness.py) on CPU fp32.

    python tests/golden/make_golden.py [case ...]

Runs only in the build container (the reference never travels to the GPU box); the .npz
files it writes next to itself are data: inputs, seeds, checksums and the reference's
outputs.  tests/test_oracle_golden.py pins oracle/ttl_oracle.py against them and the
``-m gpu`` tests compare the HIP path against the same files.
"""
import argparse
import copy
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"))
sys.path.insert(0, HERE)

from ttl_amd import synth  # noqa: E402
from ttl_amd.config import get_config  # noqa: E402
import _ref_harness as H  # noqa: E402

CIFAR10 = ["airplane", "automobile", "bird", "cat", "deer", "dog", "frog", "horse", "ship", "truck"]

# chosen after a first run with the CLI default 0.2 so that the second stage keeps about half of the 64 candidates of this synthetic
# model (its 64 PLPD values run from -0.26 to +0.04; -0.056 sits in the widest gap near the median, 0.0043 from either neighbour: 25 views
# survive); TTL_PLPD_B16_THRESHOLD overrides it for that first look
PLPD_B16_THRESHOLD = float(os.environ.get("TTL_PLPD_B16_THRESHOLD", "-0.056"))

CASES = {
    # name: (arch, n_views, n_classes, overrides)
    "tiny_deyo": ("tiny", 8, 10, {}),
    "tiny_topk": ("tiny", 64, 10, {"filter_ent": 1}),
    "tiny_steps2": ("tiny", 8, 10, {"tta_steps": 2}),
    "tiny_r32": ("tiny", 8, 10, {"rank": 32}),
    "tiny_tpt": ("tiny", 64, 10, {"deyo_selection": False, "tta_steps": 2}),
    "tiny197_deyo": ("tiny197", 8, 10, {}),
    "tiny_plpd": ("tiny", 64, 10, {"filter_plpd": 1, "plpd_threshold": 0.17}),
    "tiny_mid_deyo": ("tiny_mid", 8, 10, {}),
    "tiny_all_deyo": ("tiny_all", 8, 10, {}),
    "tiny_plpd_occ": ("tiny", 64, 10, {"filter_plpd": 1, "aug_type": "occ", "occlusion_size": 24, "row_start": 16,
                                       "column_start": 20, "plpd_threshold": -0.026}),
    "tiny_plpd_pixel": ("tiny", 64, 10, {"filter_plpd": 1, "aug_type": "pixel", "plpd_threshold": -0.129}),
    "b16_n8_k10": ("ViT-B/16", 8, 10, {}),
    "b16_n64_k200_ent0": ("ViT-B/16", 64, 200, {}),
    "b16_n64_k200_ent1": ("ViT-B/16", 64, 200, {"filter_ent": 1}),
    # BASELINE.json configs[2]'s single-GPU workload: the 1000 ImageNet labels
    "b16_n64_k1000_ent0": ("ViT-B/16", 64, 1000, {}),
    "b16_n64_k1000_ent1": ("ViT-B/16", 64, 1000, {"filter_ent": 1}),
    # BASELINE.json configs[3]'s geometry (patch 14, T = 257, D = 1024, 16 heads, 24 layers, E = 768; adapters on layers 21-23) at a
    # view count the CPU reference finishes in seconds
    "l14_n4_k10": ("ViT-L/14", 4, 10, {}),
    # the run script's other --arch option (patch 32, T = 50)
    "b32_n8_k10": ("ViT-B/32", 8, 10, {}),
    # BASELINE.json configs[4]'s features at a CPU-sized view count: rank 32, --tta_steps 2 (= 4 optimizer updates, Q6), top-rho
    # selection (int(16 * 0.1) = 1 view) on the full ViT-B/16 geometry
    "b16_r32_n16_steps2": ("ViT-B/16", 16, 10, {"rank": 32, "tta_steps": 2, "filter_ent": 1}),
    # BASELINE.json configs[4] at its full size on one GPU: rank 32, 128 views, --tta_steps 2 (= 4 updates), top-rho selection
    # (int(128 * 0.1) = 12 views), the 1000 ImageNet-Sketch labels: ~10 minutes and ~25 GB of the reference on CPU
    "b16_r32_n128_k1000_steps2": ("ViT-B/16", 128, 1000, {"rank": 32, "tta_steps": 2, "filter_ent": 1}),
    # adapters on all four attention projections (BASELINE.json north_star).  The reference hard-codes q_proj + v_proj in its
    # LoraConfig (clip/custom_clip.py:586): these two cases run the UNMODIFIED reference with the harness's peft stand-in told to
    # wrap k_proj / out_proj too (_ref_harness.TARGET_MODULES_OVERRIDE); the reference's LoRA_AB still (re-)initialises q and v
    # only, so k / out keep the stand-in's kaiming A and zero B.
    "tiny_qkvo_deyo": ("tiny", 8, 10, {"target_modules": ["q_proj", "k_proj", "v_proj", "out_proj"]}),
    "tiny_qkvo_steps2": ("tiny", 8, 10, {"target_modules": ["q_proj", "k_proj", "v_proj", "out_proj"], "tta_steps": 2}),
    # the same on the full ViT-B/16 geometry with every product live in ONE update: the adapters' B matrices (zero after the
    # reference's init, so that dA == 0 and the K-extension columns are idle) are set to seeded N(0, 0.02^2) values by the
    # harness AFTER the reference's LoRA_reset and BEFORE its test_time_tuning — input state, the reference's code is untouched
    "b16_n8_k10_qkvo": ("ViT-B/16", 8, 10, {"target_modules": ["q_proj", "k_proj", "v_proj", "out_proj"], "lora_B_std": 0.02}),
    # the same adapter set on the benched workload (64 views, K = 200): what `bench.py --lora-targets qkvo` runs
    "b16_n64_k200_qkvo": ("ViT-B/16", 64, 200, {"target_modules": ["q_proj", "k_proj", "v_proj", "out_proj"], "lora_B_std": 0.02}),
    # BASELINE.json configs[3] at its full view count (ViT-L/14, 64 views, K = 200): ~20 minutes of the reference on CPU
    "l14_n64_k200": ("ViT-L/14", 64, 200, {}),
    # a multi-update episode with adapters on all four projections and NON-ZERO B from the start: every gradient (k_proj's included,
    # which is fp32 noise while B == 0) is a real signal in every update, so the oracle is pinned tightly through the resumed forward
    # and the multi-step backward of the q/k/v/out path (round-3 advisor; tiny_qkvo_steps2 can only be compared loosely)
    "tiny_qkvo_steps2_b": ("tiny", 8, 10, {"target_modules": ["q_proj", "k_proj", "v_proj", "out_proj"], "tta_steps": 2, "lora_B_std": 0.05}),
    # CLIP-like activation statistics (synth.add_activation_outliers: a few residual channels 30-100x the rest on the CLS / one patch
    # token / every token, LayerNorm gains far from 1 on them) — what the checkpoint the reference loads (clip/custom_clip.py:581) is
    # known for and Gaussian weights do not show; every other fixture is pinned on Gaussian weights
    "b16_n8_k10_outliers": ("ViT-B/16", 8, 10, {"weights_variant": "outliers"}),
    "b16_n64_k200_outliers": ("ViT-B/16", 64, 200, {"weights_variant": "outliers"}),
    "b16_n64_k200_outliers_ent1": ("ViT-B/16", 64, 200, {"weights_variant": "outliers", "filter_ent": 1}),
    "tiny_outliers": ("tiny", 8, 10, {"weights_variant": "outliers"}),
    # round 5: the TPT objective (ttl.py:87-108: --deyo_selection False, top-rho selection + avg_entropy) and the PLPD filter with the
    # reference's DEFAULT --patch_len 6 (224 -> 222 -> 224: both antialiased resizes are live) at the benched size, 64 views / K = 200
    "b16_n64_k200_tpt": ("ViT-B/16", 64, 200, {"deyo_selection": False}),
    "b16_n64_k200_plpd": ("ViT-B/16", 64, 200, {"filter_plpd": 1, "plpd_threshold": PLPD_B16_THRESHOLD}),
}


def default_args(**over):
    """Defaults of the reference CLI (ttl.py:365-424)."""
    a = argparse.Namespace(
        arch="ViT-B/16", batch_size=64, lr=5e-3, gpu=0, tpt=True, selection_p=0.1, tta_steps=1,
        n_ctx=4, ctx_init="a_photo_of_a", cocoop=False, seed=0, layer_range=[9, 11],
        init_method="xavier", lora_encoder="image", rank=16, deyo_selection=True,
        aug_type="patch", occlusion_size=112, patch_len=6, row_start=56, column_start=56,
        deyo_margin=0.5, deyo_margin_e0=0.4, plpd_threshold=0.2, fishers=0, filter_ent=0,
        filter_plpd=0, reweight_ent=1, reweight_plpd=0)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def build_reference(case):
    arch, n_views, n_cls, over = CASES[case]
    cfg = get_config(arch)
    rank = over.get("rank", 16)
    cfg = cfg.replace(rank=rank)
    over = dict(over)
    targets = over.pop("target_modules", None)
    H.TARGET_MODULES_OVERRIDE = targets
    H.WEIGHTS_VARIANT = over.pop("weights_variant", None)
    if targets:
        cfg = cfg.replace(lora_targets=tuple(targets))
    args = default_args(**over)
    args.layer_range = [cfg.layer_lo, cfg.layer_hi]
    args.batch_size = n_views
    ttl, deyo, cc = H.import_reference(cfg, seed=0)
    if n_cls == 10:
        classnames = CIFAR10
    else:
        from data.imagnet_prompts import imagenet_classes
        classnames = list(imagenet_classes[:n_cls])
    torch.manual_seed(args.seed)
    # direct ctor so that ``rank`` is honoured (get_coop drops it, Q7); same call otherwise
    model = cc.ClipTestTimeTuning("cpu", classnames, None, arch="ViT-B/16", n_ctx=args.n_ctx,
                                  ctx_init=args.ctx_init, layer_range=args.layer_range,
                                  init_method=args.init_method, lora_encoder="image", rank=rank)
    # ---- ttl.py:151-163 requires_grad filter (restated: main_worker cannot be called) ----
    for name, p in model.named_parameters():
        on = ("image_encoder" in name and ("lora_A" in name or "lora_B" in name)
              and any(f"layers.{i}." in name for i in range(args.layer_range[0], args.layer_range[1] + 1)))
        p.requires_grad_(on)
    # ---- ttl.py:189-220 optimizer groups ----
    groups = []
    for i, layer in enumerate(model.image_encoder.vision_model.encoder.layers):
        if args.layer_range[0] <= i <= args.layer_range[1]:
            for pj in ("q_proj", "k_proj", "v_proj", "out_proj"):      # ttl.py:195-213 lists q and v; k / out when wrapped
                m = getattr(layer.self_attn, pj)
                if hasattr(m, "lora_A"):
                    groups += [{"params": m.lora_A.parameters()}, {"params": m.lora_B.parameters()}]
    opt = torch.optim.AdamW(groups, lr=args.lr)
    opt_state = copy.deepcopy(opt.state_dict())
    scaler = torch.cuda.amp.GradScaler(init_scale=1000)   # disabled on CPU (Q14)
    return cfg, args, ttl, deyo, cc, model, opt, opt_state, scaler, n_views, n_cls


def lora_named(model):
    out = {}
    for n, p in model.named_parameters():
        if "image_encoder" in n and ("lora_A" in n or "lora_B" in n):
            out[n.replace("image_encoder.", "")] = p
    return out


def run_case(case):
    cfg, args, ttl, deyo, cc, model, opt, opt_state, scaler, N, K = build_reference(case)
    x = torch.from_numpy(synth.views(cfg, N, seed=7))
    model.eval()
    rec = {"logits": []}
    hook = model.register_forward_hook(lambda m, i, o: rec["logits"].append(o.detach().clone()))
    taps = {}
    hooks = []
    if cfg.width <= 128 and "plpd_" not in case:      # (the extra PLPD variants only need the step-level records)
        vm = model.image_encoder.vision_model
        first = {"done": False}

        def tap(name):
            def f(m, i, o):
                if name not in taps:
                    taps[name] = (o[0] if isinstance(o, tuple) else o).detach().clone()
            return f
        hooks.append(vm.embeddings.register_forward_hook(tap("embed")))
        hooks.append(vm.pre_layrnorm.register_forward_hook(tap("pre_ln")))
        for i, l in enumerate(vm.encoder.layers):
            hooks.append(l.register_forward_hook(tap(f"layer{i}")))
        hooks.append(vm.post_layernorm.register_forward_hook(tap("pooled")))
    lora0 = {k: v.detach().clone().numpy() for k, v in lora_named(model).items()}
    # ---- the per-image sequence of ttl.py:338-352 ----
    with torch.no_grad():
        model.LoRA_reset()
        if getattr(args, "lora_B_std", 0):
            gen = torch.Generator().manual_seed(1234)
            for k, p in lora_named(model).items():
                if "lora_B" in k and any(f"layers.{i}." in k for i in range(cfg.layer_lo, cfg.layer_hi + 1)):
                    p.copy_(torch.randn(p.shape, generator=gen) * args.lora_B_std)
                    lora0[k] = p.detach().clone().numpy()
    opt.load_state_dict(opt_state)
    torch.manual_seed(4321)        # the PLPD patch permutation draws torch.rand from the CPU generator
    ttl.test_time_tuning(model, x, opt, scaler, args)
    grads = {k: (p.grad.detach().clone().numpy() if p.grad is not None else None)
             for k, p in lora_named(model).items()}
    lora1 = {k: v.detach().clone().numpy() for k, v in lora_named(model).items()}
    with torch.no_grad():
        out1 = model(x[:1])
    hook.remove()
    for h in hooks:
        h.remove()
    tfeat = model.text_features.detach().numpy()
    z0 = rec["logits"][0]
    # entropy / selection / loss of the FIRST update, through the reference's own functions
    Hs = deyo.softmax_entropy(z0)
    if args.deyo_selection:
        if args.filter_ent:
            idx = torch.argsort(Hs, descending=False)[:int(Hs.size()[0] * args.selection_p)]
        else:
            idx = torch.where(Hs <= math.log(1000))[0]
        e = Hs[idx]
        coeff = args.reweight_ent * (1 / torch.exp(e.clone().detach() - args.deyo_margin_e0))
        loss = e.mul(coeff).mean(0)
    else:
        sel, idx = ttl.select_confident_samples(z0, args.selection_p)
        coeff = torch.zeros(0)
        loss = ttl.avg_entropy(sel.float())
    n_updates = args.tta_steps ** 2 if args.deyo_selection else args.tta_steps
    n_fwd = n_updates * (2 if args.filter_plpd else 1) + 1
    assert len(rec["logits"]) == n_fwd, (len(rec["logits"]), n_fwd)
    if args.filter_plpd:    # second stage of the first update, through the reference's formulae (deyo.py:137-151)
        zp = rec["logits"][1]
        prob, prob_p = z0[idx].softmax(1), zp.softmax(1)
        cls1 = prob.argmax(dim=1)
        plpd = (torch.gather(prob, 1, cls1.reshape(-1, 1)) - torch.gather(prob_p, 1, cls1.reshape(-1, 1))).reshape(-1)
        ids2 = torch.where(plpd > args.plpd_threshold)[0]
        e2 = Hs[idx][ids2]
        coeff = args.reweight_ent * (1 / torch.exp(e2.clone().detach() - args.deyo_margin_e0))
        loss = e2.mul(coeff).mean(0)
        extra = dict(plpd=plpd.numpy(), idx2=idx[ids2].numpy().astype(np.int64), logits_prime=zp.numpy(),
                     plpd_threshold=args.plpd_threshold, patch_len=args.patch_len, rng_seed=4321, aug_type=args.aug_type,
                     occlusion_size=args.occlusion_size, row_start=args.row_start, column_start=args.column_start)
    else:
        extra = {}
    trained = [k for k in lora0 if any(f"layers.{i}." in k for i in range(cfg.layer_lo, cfg.layer_hi + 1))]
    out = dict(
        arch=cfg.name, rank=cfg.rank, lora_targets=np.array(list(cfg.lora_targets)), n_views=N, n_classes=K, weight_seed=0, view_seed=7,
        weights_sha256=synth.checksum(synth.vision_weights(cfg, 0, variant=H.WEIGHTS_VARIANT)),
        x_sha256=synth.checksum([x.numpy()]),
        objective="deyo" if args.deyo_selection else "tpt",
        mode="topk" if (args.filter_ent or not args.deyo_selection) else "le_thresh",
        rho=args.selection_p, margin=args.deyo_margin_e0, n_updates=n_updates, lr=args.lr,
        text_features=tfeat, logits0=z0.numpy(), H=Hs.numpy(), idx=idx.numpy().astype(np.int64),
        coeff=coeff.numpy(), loss=np.float32(loss.item()),
        logits_last=rec["logits"][n_fwd - 2 - (1 if args.filter_plpd else 0)].numpy(),
        logits1=out1.numpy(), top5=torch.topk(out1, min(5, K), dim=1).indices.numpy())
    out.update(extra)
    if H.WEIGHTS_VARIANT:
        out["weights_variant"] = H.WEIGHTS_VARIANT
    for k in (lora0 if cfg.width <= 128 else trained):
        out["lora0/" + k] = lora0[k]
    for k in trained:
        out["grad/" + k] = grads[k]
        out["lora1/" + k] = lora1[k]
    for k, v in taps.items():
        out["tap/" + k] = v.numpy()
    if cfg.width <= 128 and N <= 8 and cfg.image_size <= 64:
        out["x"] = x.numpy()
    path = os.path.join(HERE, case + ".npz")
    np.savez_compressed(path, **out)
    print(f"{case}: wrote {path} ({os.path.getsize(path)/1e6:.2f} MB) loss={loss.item():.6f} "
          f"n_sel={idx.numel()} H=[{Hs.min():.3f},{Hs.max():.3f}]")


def run_unit():
    """Known-answer vectors for the loss-side functions and AdamW."""
    cfg = get_config("tiny")
    ttl, deyo, cc = H.import_reference(cfg, 0)
    g = torch.Generator().manual_seed(5)
    out = {}
    for name, (N, K, scale) in {"a": (64, 1000, 3.0), "b": (64, 200, 6.0), "c": (8, 10, 2.0),
                                "d": (128, 1000, 4.0)}.items():
        z = torch.randn(N, K, generator=g) * scale
        if name == "a":  # near-tie across the rho boundary: views 3 and 11 almost equal
            z[11] = z[3] + 1e-4 * torch.randn(K, generator=g)
        out[f"{name}/z"] = z.numpy()
        Hs = deyo.softmax_entropy(z)
        out[f"{name}/H"] = Hs.numpy()
        sel, idx = ttl.select_confident_samples(z, 0.1)
        out[f"{name}/topk_idx"] = idx.numpy().astype(np.int64)
        if idx.numel():
            out[f"{name}/avg_entropy"] = np.float32(ttl.avg_entropy(sel.float()).item())
            zz = z.clone().requires_grad_(True)
            ttl.avg_entropy(zz[idx].float()).backward()
            out[f"{name}/tpt_dz"] = zz.grad.numpy()
        for mode in ("le_thresh", "topk"):
            zz = z.clone().requires_grad_(True)
            e = deyo.softmax_entropy(zz)
            ids = (torch.where(e <= math.log(1000))[0] if mode == "le_thresh"
                   else torch.argsort(e, descending=False)[:int(N * 0.1)])
            if ids.numel() == 0:
                continue
            e = e[ids]
            coeff = 1 * (1 / torch.exp(e.clone().detach() - 0.4))
            loss = e.mul(coeff).mean(0)
            loss.backward()
            out[f"{name}/{mode}/idx"] = ids.numpy().astype(np.int64)
            out[f"{name}/{mode}/loss"] = np.float32(loss.item())
            out[f"{name}/{mode}/dz"] = zz.grad.numpy()
    # AdamW: 3 steps on random grads, reference hyper-parameters (ttl.py:218)
    p = torch.nn.Parameter(torch.randn(16, 96, generator=g) * 0.05)
    opt = torch.optim.AdamW([p], lr=5e-3)
    out["adamw/p0"] = p.detach().clone().numpy()
    for t in range(3):
        gr = torch.randn(16, 96, generator=g) * (10.0 ** (-t * 2))
        p.grad = gr.clone()
        opt.step()
        out[f"adamw/g{t}"] = gr.numpy()
        out[f"adamw/p{t + 1}"] = p.detach().clone().numpy()
    path = os.path.join(HERE, "unit_loss_adamw.npz")
    np.savez_compressed(path, **out)
    print("unit: wrote", path, f"({os.path.getsize(path)/1e6:.2f} MB)")


def run_ties():
    """Exact entropy ties across the rho*N boundary (bit-identical logit rows): what the reference's torch.argsort
    (deyo.py:105, ttl.py:46; stable=False) returns for them, so the oracle's and the kernel's tie rule is pinned."""
    cfg = get_config("tiny")
    ttl, deyo, cc = H.import_reference(cfg, 0)
    g = torch.Generator().manual_seed(11)
    out = {}
    for name, (N, K, rho, dup) in {"t1": (64, 200, 0.1, 3), "t2": (64, 1000, 0.1, 5), "t3": (16, 10, 0.25, 4)}.items():
        z = torch.randn(N, K, generator=g) * 3.0
        order = torch.argsort(deyo.softmax_entropy(z))
        k = int(N * rho)
        src = int(order[k - 2])                       # rank k-2: the tied group straddles ranks k-2 .. k-3+dup
        hi = [int(i) for i in order[-(dup - 1):]]     # overwrite the highest-entropy rows (indices on both sides of src)
        for i in hi:
            z[i] = z[src]
        e = deyo.softmax_entropy(z)
        assert int((e == e[src]).sum()) == dup
        out[f"{name}/z"] = z.numpy()
        out[f"{name}/rho"] = np.float64(rho)
        out[f"{name}/H"] = e.numpy()
        out[f"{name}/topk/idx"] = torch.argsort(e, descending=False)[:k].numpy().astype(np.int64)     # deyo.py:105
        out[f"{name}/tpt_idx"] = ttl.select_confident_samples(z, rho)[1].numpy().astype(np.int64)          # ttl.py:43-47
        out[f"{name}/tied_rows"] = np.array(sorted([src] + hi), np.int64)
    path = os.path.join(HERE, "unit_ties.npz")
    np.savez_compressed(path, **out)
    print("ties: wrote", path, {k: out[k].tolist() for k in out if k.endswith("idx") or k.endswith("rows")})


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or (["unit", "ties"] + list(CASES))
    for c in which:
        if c == "unit":
            run_unit()
        elif c == "ties":
            run_ties()
        else:
            # one process per case: the reference is patched per geometry at import
            if len(which) > 1:
                rc = os.system(f"{sys.executable} {os.path.abspath(__file__)} {c}")
                if rc:
                    sys.exit(1)
            else:
                run_case(c)
