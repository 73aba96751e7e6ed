#!/usr/bin/env python3
"""Golden fixtures for ``--lora_encoder text`` (clip/custom_clip.py:602-607,672-678; ttl.py:143-147,
190-192): the REFERENCE itself, imported unmodified through _ref_harness.py, run on CPU fp32 with the
synthetic image- and text-tower weights of ttl_amd.synth.

    python tests/golden/make_golden_text.py [case ...]

The prompts are token rows from synth.token_ids (the reference's tokenizer needs its BPE vocabulary and
real class names; what the path under test consumes is ``prompt_learner.tokenized_prompts``, an int
tensor, which is set directly).  Build container only; the .npz files are data.
"""
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
from ttl_amd.config import get_config, get_text_config  # noqa: E402
import _ref_harness as H  # noqa: E402
from make_golden import default_args, CIFAR10  # noqa: E402

CASES = {
    # name: (arch, n_views, n_prompts, overrides)
    "tiny_text_deyo": ("tiny", 8, 10, {}),
    "tiny_text_topk": ("tiny", 64, 10, {"filter_ent": 1}),
    "tiny_text_steps2": ("tiny", 8, 10, {"tta_steps": 2}),
    "tiny_text_tpt": ("tiny", 64, 10, {"deyo_selection": False, "tta_steps": 2}),
    "tiny_text_plpd": ("tiny", 64, 10, {"filter_plpd": 1, "plpd_threshold": 0.05}),
    "b16_text_n8_k10": ("ViT-B/16", 8, 10, {}),
    "b16_text_n64_k200": ("ViT-B/16", 64, 200, {}),
}


def lora_named(model):
    return {n.replace("text_encoder.", ""): p for n, p in model.named_parameters()
            if "text_encoder" in n and ("lora_A" in n or "lora_B" in n)}


def run_case(case):
    arch, N, K, over = CASES[case]
    cfg, tcfg = get_config(arch), get_text_config(arch)
    args = default_args(**over)
    args.lora_encoder = "text"
    args.layer_range = [tcfg.layer_lo, tcfg.layer_hi]
    args.batch_size = N
    ttl, deyo, cc = H.import_reference(cfg, seed=0, text_cfg=tcfg)
    torch.manual_seed(args.seed)
    classnames = CIFAR10 if K == 10 else [f"c{i}" for i in range(K)]
    model = cc.ClipTestTimeTuning("cpu", classnames, None, arch="ViT-B/16", n_ctx=args.n_ctx, ctx_init=args.ctx_init,
                                  layer_range=args.layer_range, init_method=args.init_method, lora_encoder="text", rank=16)
    ids = torch.from_numpy(synth.token_ids(K, tcfg, seed=3).astype(np.int64))
    model.prompt_learner.tokenized_prompts = ids             # what get_text_features reads (custom_clip.py:655)
    # ---- ttl.py:143-163 requires_grad filter with lora_enc = 'text_encoder' ----
    for name, p in model.named_parameters():
        on = ("text_encoder" in name and ("lora_A" in name or "lora_B" in name)
              and any(f"layers.{i}." in name for i in range(args.layer_range[0], args.layer_range[1] + 1)))
        p.requires_grad_(on)
    # ---- ttl.py:189-220 optimizer groups over model.text_encoder.text_model.encoder.layers ----
    groups = []
    for i, layer in enumerate(model.text_encoder.text_model.encoder.layers):
        if args.layer_range[0] <= i <= args.layer_range[1]:
            groups += [{"params": layer.self_attn.q_proj.lora_A.parameters()},
                       {"params": layer.self_attn.q_proj.lora_B.parameters()},
                       {"params": layer.self_attn.v_proj.lora_A.parameters()},
                       {"params": layer.self_attn.v_proj.lora_B.parameters()}]
    opt = torch.optim.AdamW(groups, lr=args.lr)
    opt_state = copy.deepcopy(opt.state_dict())
    scaler = torch.cuda.amp.GradScaler(init_scale=1000)
    x = torch.from_numpy(synth.views(cfg, N, seed=7))
    model.eval()
    rec = {"logits": []}
    hook = model.register_forward_hook(lambda m, i, o: rec["logits"].append(o.detach().clone()))
    lora0 = {k: v.detach().clone().numpy() for k, v in lora_named(model).items()}
    with torch.no_grad():
        model.LoRA_reset()
    opt.load_state_dict(opt_state)
    torch.manual_seed(4321)        # the PLPD patch permutation draws torch.rand from the CPU generator
    ttl.test_time_tuning(model, x, opt, scaler, args)
    grads = {k: (p.grad.detach().clone().numpy() if p.grad is not None else None) for k, p in lora_named(model).items()}
    lora1 = {k: v.detach().clone().numpy() for k, v in lora_named(model).items()}
    with torch.no_grad():
        out1 = model(x[:1])
    hook.remove()
    z0 = rec["logits"][0]
    Hs = deyo.softmax_entropy(z0)
    if not args.deyo_selection:                      # TPT objective on the text LoRA (ttl.py:87-108)
        sel, idx = ttl.select_confident_samples(z0, args.selection_p)
        coeff = torch.zeros(0)
        loss = ttl.avg_entropy(sel.float())
    else:
        if args.filter_ent:
            idx = torch.argsort(Hs, descending=False)[:int(Hs.size()[0] * args.selection_p)]
        else:
            idx = torch.where(Hs <= math.log(1000))[0]
        e = Hs[idx]
        coeff = args.reweight_ent * (1 / torch.exp(e.clone().detach() - args.deyo_margin_e0))
        loss = e.mul(coeff).mean(0)
    n_updates = args.tta_steps ** 2 if args.deyo_selection else args.tta_steps
    extra = {}
    if args.filter_plpd:           # second stage of the update, through the reference's formulae (deyo.py:137-151)
        assert len(rec["logits"]) == 2 * n_updates + 1
        zp = rec["logits"][1]
        prob, prob_p = z0[idx].softmax(1), zp.softmax(1)
        cls1 = prob.argmax(dim=1)
        plpd = (torch.gather(prob, 1, cls1.reshape(-1, 1)) - torch.gather(prob_p, 1, cls1.reshape(-1, 1))).reshape(-1)
        ids2 = torch.where(plpd > args.plpd_threshold)[0]
        e2 = Hs[idx][ids2]
        coeff = args.reweight_ent * (1 / torch.exp(e2.clone().detach() - args.deyo_margin_e0))
        loss = e2.mul(coeff).mean(0)
        extra = dict(plpd=plpd.numpy(), idx2=idx[ids2].numpy().astype(np.int64), logits_prime=zp.numpy(),
                     plpd_threshold=args.plpd_threshold, patch_len=args.patch_len, rng_seed=4321)
    else:
        assert len(rec["logits"]) == n_updates + 1
    with torch.no_grad():
        fimg = model.image_encoder(x)
        fimg = fimg / fimg.norm(dim=-1, keepdim=True)
    trained = [k for k in lora0 if any(f"layers.{i}." in k for i in range(tcfg.layer_lo, tcfg.layer_hi + 1))]
    out = dict(arch=cfg.name, mode_encoder="text", rank=16, n_views=N, n_classes=K, weight_seed=0, view_seed=7, ids_seed=3,
               ids=ids.numpy().astype(np.int32),
               weights_sha256=synth.checksum(synth.vision_weights(cfg, 0)),
               objective="deyo" if args.deyo_selection else "tpt",
               mode="topk" if (args.filter_ent or not args.deyo_selection) else "le_thresh", rho=args.selection_p,
               margin=args.deyo_margin_e0, n_updates=n_updates, lr=args.lr,
               image_features=fimg.numpy(), text_features_after=model.text_features.detach().numpy(),
               logits0=z0.numpy(), H=Hs.numpy(), idx=idx.numpy().astype(np.int64), coeff=coeff.numpy(),
               loss=np.float32(loss.item()), logits_last=rec["logits"][(2 if args.filter_plpd else 1) * n_updates - (2 if args.filter_plpd else 1)].numpy(),
               logits1=out1.numpy(), top5=torch.topk(out1, min(5, K), dim=1).indices.numpy())
    out.update(extra)
    for k in (lora0 if cfg.width <= 128 else trained):
        out["lora0/" + k] = lora0[k]
    for k in trained:
        out["grad/" + k] = grads[k]
        out["lora1/" + k] = lora1[k]
    path = os.path.join(HERE, case + ".npz")
    np.savez_compressed(path, **out)
    print(f"{case}: wrote {path} ({os.path.getsize(path)/1e6:.2f} MB) loss={loss.item():.6f} n_sel={idx.numel()} "
          f"H=[{Hs.min():.3f},{Hs.max():.3f}]")


if __name__ == "__main__":
    torch.set_num_threads(8)
    which = sys.argv[1:] or list(CASES)
    for c in which:
        if len(which) > 1:
            if os.system(f"{sys.executable} {os.path.abspath(__file__)} {c}"):
                sys.exit(1)
        else:
            run_case(c)
