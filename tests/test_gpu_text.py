"""-m gpu: --lora_encoder text through the C ABI — text tower forward, LoRA backward and the
fused text-mode episode vs the bf16-emulating oracle (tight) and the reference-generated fixtures."""
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import ttl_oracle as O
from helpers import load_text_case, episode_kwargs, max_rel, check_lora_step

pytestmark = pytest.mark.gpu

TINY = ["tiny_text_deyo", "tiny_text_topk", "tiny_text_steps2"]
TINY_ALL = TINY + ["tiny_text_tpt"]


def make(vcfg, tcfg, Wv, Wt, lora0, n_views, n_prompts, precision="bf16"):
    from ttl_amd.engine import TTLEngine, TextTowerEngine
    img = TTLEngine(vcfg, max_views=n_views, max_classes=1, device="cuda:0", precision=precision)
    img.load_weights(Wv)                                   # no bind_lora: the image tower has no adapters in this mode
    txt = TextTowerEngine(tcfg, max_prompts=n_prompts, max_views=n_views, device="cuda:0", precision=precision)
    txt.load_weights(Wt)
    txt.set_logit_scale(float(np.exp(Wv["logit_scale"])))
    names = O.trainable_names(tcfg, "text_model")
    flat = torch.cat([torch.from_numpy(lora0[k]).reshape(-1) for k in names]).cuda().contiguous()
    txt.bind_lora(flat)
    return img, txt, flat, names


def split(flat, like, names):
    out, off = {}, 0
    a = flat.detach().cpu().numpy()
    for k in names:
        n = like[k].size
        out[k] = a[off:off + n].reshape(like[k].shape)
        off += n
    return out


@pytest.mark.parametrize("name", TINY + ["b16_text_n8_k10"])
def test_text_forward_and_backward(name):
    g, vcfg, tcfg, Wv, Wt, x, ids, lora0 = load_text_case(name)
    N, K = x.shape[0], ids.shape[0]
    img, txt, flat, names = make(vcfg, tcfg, Wv, Wt, lora0, N, K)
    f = img.features(torch.from_numpy(x).cuda())
    fh = torch.nn.functional.normalize(f, dim=-1).cpu().numpy()
    assert max_rel(fh, g["image_features"]) < 2e-2
    txt.set_prompts(ids)
    # feed the reference's own image features so the text side is compared in isolation
    txt.set_image_features(torch.from_numpy(g["image_features"]).cuda(), normalize=False)
    z, t = txt.forward(save=True, want_features=True)
    net = O.TextOracle(tcfg, Wt, lora0, "bf16")
    save = {}
    tb = net.forward(ids, save)
    thb = tb / np.linalg.norm(tb, axis=-1, keepdims=True)
    S = np.float32(np.exp(Wv["logit_scale"]))
    zb = S * g["image_features"] @ thb.T
    assert max_rel(t.cpu().numpy(), tb) < 1.2e-2, ("text features vs bf16 oracle", max_rel(t.cpu().numpy(), tb))
    assert max_rel(z.cpu().numpy(), zb) < 1.2e-2
    assert max_rel(z.cpu().numpy(), g["logits0"]) < 3e-2, ("vs reference fp32", max_rel(z.cpu().numpy(), g["logits0"]))
    # backward from the reference's loss gradient at the reference's logits
    L = O.deyo_loss_and_grad(g["logits0"], str(g["mode"]), float(g["rho"]), float(g["margin"]), 1.0)
    txt.backward(torch.from_numpy(L["dz"]).cuda())
    torch.cuda.synchronize()
    grads = split(txt.grads, lora0, names)
    gb = net.backward(L["dz"], g["image_features"], save, S)
    for k in names:
        if np.abs(gb[k]).max() == 0:
            assert not grads[k].any(), k
        else:
            assert max_rel(grads[k], gb[k]) < 2.5e-2, (k, max_rel(grads[k], gb[k]))
            if int(g["n_updates"]) == 1:
                assert max_rel(grads[k], g["grad/" + k]) < 4e-2, (k, "vs reference", max_rel(grads[k], g["grad/" + k]))
    img.close(); txt.close()


@pytest.mark.parametrize("name", TINY_ALL + ["b16_text_n8_k10", "b16_text_n64_k200"])
def test_text_episode(name):
    g, vcfg, tcfg, Wv, Wt, x, ids, lora0 = load_text_case(name)
    kw = episode_kwargs(g)
    N, K = x.shape[0], ids.shape[0]
    img, txt, flat, names = make(vcfg, tcfg, Wv, Wt, lora0, N, K)
    txt.set_prompts(ids)
    snap, m, v = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat)
    mode = 1 if kw["mode"] == "topk" else 0
    l1, l0 = txt.episode(img, torch.from_numpy(x).cuda(), snap, m, v, n_updates=kw["n_updates"], objective=kw["objective"],
                         mode=mode, rho=kw["rho"], margin=kw["margin"], lr=kw["lr"], want_logits0=True)
    torch.cuda.synchronize()
    l0, l1 = l0.cpu().numpy(), l1.cpu().numpy()
    # selection set of the first update: bit-exact vs the reference
    H = O.softmax_entropy(l0)
    idx = O.select_views(H, kw["mode"], N, kw["rho"])
    assert np.array_equal(np.sort(idx), np.sort(g["idx"])), "confidence-selection set differs from the reference"
    tol0 = 1.5e-2 if vcfg.width >= 768 else 3e-2        # both towers carry bf16-operand noise in this mode
    assert max_rel(l0, g["logits0"]) < tol0, ("logits0 vs reference", max_rel(l0, g["logits0"]))
    if vcfg.width < 768 or N <= 8:                       # oracle episode at full size takes minutes on the CPU
        ob = O.episode_text(vcfg, tcfg, Wv, Wt, lora0, x, ids, prec="bf16", **kw)
        assert max_rel(l0, ob["logits0"]) < 1.5e-2
        assert max_rel(l1, ob["logits1"]) < 3e-2
    assert max_rel(l1, g["logits1"]) < 5e-2, ("logits1 vs reference", max_rel(l1, g["logits1"]))
    assert int(np.argmax(l1)) == int(g["top5"][0, 0])
    if kw["n_updates"] == 1:
        lora1 = split(flat, lora0, names)
        grads = split(txt.grads, lora0, names)
        for k in names:
            gr = g["grad/" + k]
            dg = float(np.abs(grads[k] - gr).max())
            check_lora_step(lora1[k], g["lora1/" + k], gr, kw["lr"], 1e-4, k, dg=dg + 1e-12)
    img.close(); txt.close()


@pytest.mark.parametrize("name", ["b16_text_n8_k10", "b16_text_n64_k200"])
def test_text_fp16_operands_tolerance(name):
    """fp16-operand build (the reference's autocast dtype): logits within 2e-3 of the reference's fp32 result
    (measured 1.3e-3: in this mode BOTH towers carry 16-bit operand noise, the image-mode figure is 4e-4),
    same selection, same prediction."""
    g, vcfg, tcfg, Wv, Wt, x, ids, lora0 = load_text_case(name)
    kw = episode_kwargs(g)
    N, K = x.shape[0], ids.shape[0]
    img, txt, flat, names = make(vcfg, tcfg, Wv, Wt, lora0, N, K, precision="fp16")
    txt.set_prompts(ids)
    snap, m, v = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat)
    l1, l0 = txt.episode(img, torch.from_numpy(x).cuda(), snap, m, v, n_updates=kw["n_updates"], mode=0, rho=kw["rho"],
                         margin=kw["margin"], lr=kw["lr"], want_logits0=True)
    torch.cuda.synchronize()
    e0 = max_rel(l0.cpu().numpy(), g["logits0"])
    assert e0 < 2e-3, e0
    grads = split(txt.grads, lora0, names)
    for k in names:
        gr = g["grad/" + k]
        if np.abs(gr).max() > 0:
            assert max_rel(grads[k], gr) < 8e-3, (k, max_rel(grads[k], gr))      # measured up to 4e-3
    assert int(np.argmax(l1.cpu().numpy())) == int(g["top5"][0, 0])
    img.close(); txt.close()


def test_text_mode_errors_are_loud():
    from ttl_amd import _lib
    g, vcfg, tcfg, Wv, Wt, x, ids, lora0 = load_text_case("tiny_text_deyo")
    img, txt, flat, names = make(vcfg, tcfg, Wv, Wt, lora0, 8, 10)
    with pytest.raises(_lib.TtlError):
        txt.forward()                                       # no prompts yet
    bad = ids.copy(); bad[0, 3] = tcfg.vocab_size
    with pytest.raises(_lib.TtlError):
        txt.set_prompts(bad)
    with pytest.raises(_lib.TtlError):
        img.forward(torch.from_numpy(x).cuda(), save=True)  # saving for backward needs bound adapters
    img.close(); txt.close()


# ------------------------------------------------------------------ host surface (drop-in) in text mode
def build_model(name):
    """ClipTestTimeTuning(lora_encoder='text') in the fixture's state + the optimizer of ttl.py:189-220."""
    import copy
    from ttl_amd.custom_clip import ClipTestTimeTuning
    g, vcfg, tcfg, Wv, Wt, x, ids, lora0 = load_text_case(name)
    K = ids.shape[0]
    model = ClipTestTimeTuning(0, [f"c{i}" for i in range(K)], None, arch=vcfg.name, layer_range=[tcfg.layer_lo, tcfg.layer_hi],
                               init_method="xavier", lora_encoder="text", rank=16, max_views=x.shape[0], max_classes=K,
                               weight_seed=0)
    layers = model.text_encoder.text_model.encoder.layers
    with torch.no_grad():
        for i, layer in enumerate(layers):
            for pj in ("q_proj", "v_proj"):
                key = f"text_model.encoder.layers.{i}.self_attn.{pj}.lora_A.default.weight"
                getattr(layer.self_attn, pj).lora_A.default.weight.copy_(torch.from_numpy(lora0[key]))
        model.LoRA_AB.init_weights = []
        for layer in layers:
            sa = layer.self_attn
            model.LoRA_AB.init_weights.append(tuple(t.detach().clone() for t in (
                sa.q_proj.lora_A.default.weight, sa.q_proj.lora_B.default.weight,
                sa.v_proj.lora_A.default.weight, sa.v_proj.lora_B.default.weight)))
    model.prompt_learner.tokenized_prompts = torch.from_numpy(ids.astype(np.int64))     # the fixture's prompts
    model._text_dirty = True
    # ttl.py:143-163 with lora_enc = 'text_encoder' / :189-220 over model.text_encoder.text_model.encoder.layers
    for n, p in model.named_parameters():
        p.requires_grad_("text_encoder" in n and ("lora_A" in n or "lora_B" in n)
                         and any(f"layers.{i}." in n for i in range(tcfg.layer_lo, tcfg.layer_hi + 1)))
    groups = []
    for i, layer in enumerate(layers):
        if tcfg.layer_lo <= i <= tcfg.layer_hi:
            groups += [{"params": layer.self_attn.q_proj.lora_A.parameters()}, {"params": layer.self_attn.q_proj.lora_B.parameters()},
                       {"params": layer.self_attn.v_proj.lora_A.parameters()}, {"params": layer.self_attn.v_proj.lora_B.parameters()}]
    opt = torch.optim.AdamW(groups, lr=5e-3)
    return g, tcfg, model, opt, copy.deepcopy(opt.state_dict()), torch.from_numpy(x).cuda()


def text_args(**over):
    import argparse
    a = argparse.Namespace(lr=5e-3, selection_p=0.1, tta_steps=1, cocoop=False, lora_encoder="text", deyo_selection=True,
                           deyo_margin=0.5, deyo_margin_e0=0.4, filter_ent=0, filter_plpd=0, reweight_ent=1, reweight_plpd=0)
    for k, v in over.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("name", TINY_ALL)
def test_reference_shaped_loop_text_mode(name):
    """The per-image sequence of ttl.py:338-352 on this build's surface with --lora_encoder text."""
    from ttl_amd.ttl import test_time_tuning
    g, tcfg, model, opt, opt_state, x = build_model(name)
    kw = episode_kwargs(g)
    deyo = kw["objective"] == "deyo"
    args = text_args(filter_ent=1 if (kw["mode"] == "topk" and deyo) else 0, deyo_selection=deyo,
                     tta_steps=int(round(kw["n_updates"] ** 0.5)) if deyo else kw["n_updates"])
    names = [n for n, _ in model.named_parameters()]
    assert any(n.startswith("text_encoder.text_model.encoder.layers.1.self_attn.q_proj.lora_A") for n in names)
    assert not any("image_encoder" in n and "lora" in n for n in names)      # no adapters on the image tower in this mode
    model.eval()
    with torch.no_grad():
        model.LoRA_reset()
    opt.load_state_dict(opt_state)
    with torch.no_grad():
        z0 = model(x)
    assert max_rel(z0.cpu().numpy(), g["logits0"]) < 3e-2
    test_time_tuning(model, x, opt, None, args)
    with torch.no_grad():
        out = model(x[:1])
    assert max_rel(out.cpu().numpy(), g["logits1"]) < 5e-2
    assert int(out.argmax()) == int(g["top5"][0, 0])
    # text features accessor of the reference surface
    t = model.get_text_features()
    assert tuple(t.shape) == (g["ids"].shape[0], tcfg.embed) and torch.allclose(t.norm(dim=-1), torch.ones_like(t[:, 0]), atol=1e-5)
    assert max_rel(t.cpu().numpy(), g["text_features_after"]) < 5e-2


def test_autograd_formulation_text_mode():
    """The reference's own step (torch loss on model(x), loss.backward(), optimizer.step()) through the autograd hook."""
    g, tcfg, model, opt, opt_state, x = build_model("tiny_text_deyo")
    model.eval()
    with torch.no_grad():
        model.LoRA_reset()
    opt.load_state_dict(opt_state)
    out = model(x)
    ent = -(out.softmax(1) * out.log_softmax(1)).sum(1)
    idx = torch.where(ent <= math.log(1000))[0]
    e = ent[idx]
    coeff = 1 / torch.exp(e.clone().detach() - 0.4)
    loss = e.mul(coeff).mean(0)
    opt.zero_grad()
    loss.backward()
    for n, p in model.named_parameters():
        key = "grad/" + n.replace("text_encoder.", "")
        if p.requires_grad and key in g.files and np.abs(g[key]).max() > 0:
            assert max_rel(p.grad.cpu().numpy(), g[key]) < 4e-2, n
    opt.step()
    with torch.no_grad():
        o1 = model(x[:1])
    assert int(o1.argmax()) == int(g["top5"][0, 0])


def test_eval_loop_text_mode():
    """ttl_amd.eval.test_time_adapt_eval with lora_encoder='text' (fused text episodes, 2 in flight) == per-image surface."""
    from ttl_amd.eval import test_time_adapt_eval, SyntheticViews
    from ttl_amd.ttl import test_time_tuning
    from ttl_amd.driver import topk_hits
    from ttl_amd.config import VIT_TINY
    g, tcfg, model, opt, opt_state, x = build_model("tiny_text_deyo")
    args = text_args()
    data = SyntheticViews(VIT_TINY, 4, 8, 10, seed=3)
    hits = torch.zeros(2, dtype=torch.int64)
    for views, label in data:
        with torch.no_grad():
            model.LoRA_reset()
        opt.load_state_dict(opt_state)
        test_time_tuning(model, views.cuda(), opt, None, args)
        with torch.no_grad():
            out = model(views[:1].cuda())
        h1, h5 = topk_hits(out.cpu(), torch.tensor([label]))
        hits += torch.stack([h1, h5])
    with torch.no_grad():
        model.LoRA_reset()
    top1, top5 = test_time_adapt_eval(data, model, None, opt, opt_state, None, args, n_streams=2)
    assert abs(top1 - 100.0 * hits[0].item() / 4) < 1e-9 and abs(top5 - 100.0 * hits[1].item() / 4) < 1e-9


def test_plpd_filter_text_mode_matches_reference():
    """--filter_plpd 1 with --lora_encoder text (deyo.py:115-151): same destroyed views, same surviving set, same counts
    as the reference's fixture; the PLPD forward reuses the pending forward's text features."""
    from ttl_amd import deyo as D
    g, tcfg, model, opt, opt_state, x = build_model("tiny_text_plpd")
    model.precision = "fp16"          # PLPD thresholds a probability difference: use the tighter build
    args = text_args(filter_plpd=1, plpd_threshold=float(g["plpd_threshold"]), aug_type="patch", patch_len=int(g["patch_len"]))
    with torch.no_grad():
        model.LoRA_reset()
    opt.load_state_dict(opt_state)
    torch.manual_seed(int(g["rng_seed"]))
    d = D.DeYO(model, args, opt, None, steps=1, deyo_margin=args.deyo_margin, margin_e0=args.deyo_margin_e0)
    outputs, backward, final_backward = d(x)
    assert backward == len(g["idx"]) and final_backward == len(g["idx2"])
    assert max_rel(outputs.cpu().numpy(), g["logits0"]) < 5e-3
    with torch.no_grad():
        out = model(x[:1])
    assert max_rel(out.cpu().numpy(), g["logits1"]) < 3e-2
    assert int(out.argmax()) == int(g["top5"][0, 0])


def test_text_tower_k_and_out_proj_adapters_vs_oracle():
    """--lora_encoder text with adapters on q, k, v and out_proj: the end-of-text pooling makes the top layer's out_proj
    gradients run on gathered rows (one per prompt, positions differ)."""
    import numpy as np
    from oracle import ttl_oracle as O
    from helpers import max_rel
    from ttl_amd import synth
    from ttl_amd.config import get_config, get_text_config
    from ttl_amd.custom_clip import build_text_mode_engine
    tg = ("q_proj", "k_proj", "v_proj", "out_proj")
    vcfg, tcfg = get_config("tiny"), get_text_config("tiny").replace(lora_targets=tg)
    Wv, Wt = synth.vision_weights(vcfg, 0), synth.text_weights(tcfg, 0)
    lora0 = synth.lora_init(tcfg, 0, tower="text_model")
    rng = np.random.default_rng(5)
    for k in lora0:
        if "lora_B" in k:
            lora0[k] = (rng.standard_normal(lora0[k].shape) * 0.02).astype(np.float32)
    K, N = 6, 4
    ids = synth.token_ids(K, tcfg, 3)
    x = synth.views(vcfg, N, 5)
    eng = build_text_mode_engine(vcfg, tcfg, Wv, Wt, torch.from_numpy(ids), float(np.exp(Wv["logit_scale"])), "cuda:0", N, K, "bf16")
    names = O.trainable_names(tcfg, "text_model")
    flat = torch.cat([torch.from_numpy(lora0[k]).reshape(-1) for k in names]).cuda().contiguous()
    eng.bind_lora(flat)
    snap, m, v = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat)
    l1 = eng.episode(torch.from_numpy(x).cuda(), snap, m, v, n_updates=1)
    torch.cuda.synchronize()
    trace = []
    ob = O.episode_text(vcfg, tcfg, Wv, Wt, lora0, x, ids, prec="bf16", trace=trace)
    g = eng.grads.cpu().numpy()
    off = 0
    gmax = max(float(np.abs(trace[-1]["grads"][k]).max()) for k in names)
    for k in names:
        gr = trace[-1]["grads"][k]
        got = g[off:off + gr.size].reshape(gr.shape)
        off += gr.size
        assert max_rel(got, gr) < 4e-2 or np.abs(got - gr).max() < 3e-3 * gmax, (k, max_rel(got, gr))
    assert max_rel(l1.cpu().numpy(), ob["logits1"]) < 2e-2
    eng.close()
