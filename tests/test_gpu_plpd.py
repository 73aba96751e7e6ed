"""-m gpu: the PLPD filter of DeYO (--filter_plpd 1, deyo.py:115-151) as device code (csrc/plpd.hip) and as a stage of the fused
episode (ttl_episode_args.plpd; SURVEY §8f-3, round-4 review item 4).

  * ttl_plpd_views against the torch chain the reference runs (ttl_amd.deyo.plpd_views: the same einops / torchvision operations
    as deyo.py:118-134 written with view / permute / F.interpolate(antialias=True)), same host RNG draws:
    'pixel' and 'patch' with image_size % patch_len == 0 (e.g. 224 / 4) are pure gathers -> BIT-equal;
    'occ' (a mean) and 'patch' with the two antialiased resizes (the reference's default --patch_len 6: 224 -> 222 -> 224) -> within
    2e-6 absolute (summation order);
  * ttl_plpd_keep against the reference's formula (deyo.py:137-146);
  * the fused episode with a PLPD stage against the fixtures the reference wrote (tiny_plpd, tiny_plpd_occ, tiny_plpd_pixel,
    tiny_text_plpd): second-stage selection exactly the reference's, PLPD values, weights and adapted logits — on the strict
    (fp32) build at its tolerances and on the fp16 headline build at the operand tolerances."""
import argparse

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from helpers import load_case, episode_kwargs, max_rel
from test_gpu_path import make_engine, split

pytestmark = pytest.mark.gpu


def _args(**kw):
    return argparse.Namespace(**kw)


@pytest.mark.parametrize("aug,S,patch_len", [("pixel", 64, 4), ("patch", 64, 4), ("patch", 224, 4), ("occ", 64, 4), ("patch", 64, 3),
                                            ("patch", 224, 5), ("patch", 63, 4), ("patch", 224, 6)])      # (224, 6): the reference's default --patch_len
@pytest.mark.parametrize("precision", ["fp16"])
def test_plpd_views_vs_the_torch_chain(aug, S, patch_len, precision):
    from ttl_amd import deyo as D
    from ttl_amd.config import get_config
    from ttl_amd.engine import TTLEngine
    eng = TTLEngine(get_config("tiny"), 8, 10, "cuda:0", precision=precision)
    g = torch.Generator().manual_seed(S + patch_len)
    N = 7
    x = torch.randn(N, 3, S, S, generator=g).cuda()
    idx = torch.tensor([5, 0, 3, 6, 2], dtype=torch.int64).cuda()           # the first-stage list: any order, a subset
    B = idx.numel()
    a = _args(aug_type=aug, patch_len=patch_len, occlusion_size=S // 2, row_start=3, column_start=5, plpd_threshold=0.2)
    torch.manual_seed(99)
    ref = D.plpd_views(x[idx].detach(), a)                                   # the torch chain draws its permutation itself
    torch.manual_seed(99)
    spec = D.plpd_spec(a)
    perm = D.draw_plpd_perms(spec, 1, B, S, "cuda")
    n_sel = torch.tensor([B], dtype=torch.int32).cuda()
    out = eng.plpd_views(x, idx, n_sel, B, spec, perm)
    torch.cuda.synchronize()
    exact = aug == "pixel" or (aug == "patch" and S % patch_len == 0)
    if exact:
        assert torch.equal(out, ref)
    else:
        d = (out - ref).abs().max().item()
        assert d < 2e-6, d
    # the device-side count guards the launch: rows >= *n_sel stay untouched
    n2 = torch.tensor([2], dtype=torch.int32).cuda()
    out2 = eng.plpd_views(x, idx, n2, B, spec, perm)
    torch.cuda.synchronize()
    assert torch.equal(out2[:2], out[:2])
    eng.close()


def test_plpd_keep_vs_the_reference_formula():
    from ttl_amd.config import get_config
    from ttl_amd.engine import TTLEngine
    eng = TTLEngine(get_config("tiny"), 8, 10, "cuda:0", precision="fp16")
    g = torch.Generator().manual_seed(3)
    for N, K, B in ((8, 10, 8), (64, 200, 6), (64, 1000, 64)):
        z = (torch.randn(N, K, generator=g) * 3).cuda()
        z[1, 4] = z[1, 7] = z[1].max() + 1.0                 # a tie for the arg-max: torch.argmax takes the first
        idx = torch.randperm(N, generator=g)[:B].to(torch.int64).cuda()
        zp = (z[idx] + torch.randn(B, K, generator=g).cuda() * 2)
        n = torch.tensor([B], dtype=torch.int32).cuda()
        keep, val = eng.plpd_keep(z, zp, idx, n, B, 0.2)
        torch.cuda.synchronize()
        prob, prob_p = z[idx].softmax(1), zp.softmax(1)
        cls1 = prob.argmax(dim=1)
        plpd = (torch.gather(prob, 1, cls1.reshape(-1, 1)) - torch.gather(prob_p, 1, cls1.reshape(-1, 1))).reshape(-1)
        assert (val - plpd).abs().max().item() < 2e-6
        want = torch.zeros(N, dtype=torch.uint8, device="cuda")
        clear = (plpd - 0.2).abs() > 1e-5                    # (a value within rounding of the threshold may fall either way)
        want[idx] = (plpd > 0.2).to(torch.uint8)
        assert torch.equal(keep[idx][clear], want[idx][clear])
        assert int(keep.sum()) == int(keep[idx].sum())       # nothing outside the candidate list
    eng.close()


def test_plpd_keep_survives_nan_and_all_minus_inf_rows():
    """A candidate row of logits that is all -inf, or holds NaNs (a NaN entropy ranks first in the top-k selection, so such a row DOES
    become a candidate): torch.argmax answers index 0 / the first NaN and the reference's comparison `plpd > threshold` is False for
    the NaN difference (deyo.py:137-146) — the view is dropped, nothing faults, the other rows are unaffected."""
    from ttl_amd.config import get_config
    from ttl_amd.engine import TTLEngine
    eng = TTLEngine(get_config("tiny"), 8, 10, "cuda:0", precision="fp16")
    g = torch.Generator().manual_seed(5)
    for N, K in ((8, 10), (16, 1000), (6, 300)):
        z = (torch.randn(N, K, generator=g) * 3).cuda()
        z[0] = float("-inf")
        z[2, K // 2] = float("nan")
        z[3] = float("nan")
        z[4, 1] = float("inf")
        idx = torch.arange(N, dtype=torch.int64).cuda()
        zp = z.clone().nan_to_num(0.0, 0.0, 0.0) + torch.randn(N, K, generator=g).cuda() * 2
        n = torch.tensor([N], dtype=torch.int32).cuda()
        keep, val = eng.plpd_keep(z, zp, idx, n, N, 0.2)
        torch.cuda.synchronize()
        prob, prob_p = z.softmax(1), zp.softmax(1)
        cls1 = prob.argmax(dim=1)                            # (index 0 for the -inf row — softmax of it is NaN — and the first NaN otherwise)
        plpd = (torch.gather(prob, 1, cls1.reshape(-1, 1)) - torch.gather(prob_p, 1, cls1.reshape(-1, 1))).reshape(-1)
        bad = torch.isnan(plpd)
        assert bad[[0, 2, 3, 4]].all() and not bad[[1, 5]].any()
        assert torch.isnan(val[bad]).all() and not keep[bad].any()              # dropped, like `nan > threshold`
        assert (val[~bad] - plpd[~bad]).abs().max().item() < 2e-6
        clear = ~bad & ((plpd - 0.2).abs() > 1e-5)
        assert torch.equal(keep[clear], (plpd > 0.2).to(torch.uint8)[clear])
    eng.close()


def survivors(eng, n_views, n_candidates, n_expected, g):
    """The second-stage list filter_ids_1[filter_ids_2] (deyo.py:146-151) as the context holds it after a PLPD update: "idx" is the
    FIRST-stage list (the reference's order), "keep" the mask over views, "n_selected" the number of survivors."""
    n = int(eng.debug_copy("n_selected", 0, (1,), np.int32)[0])
    first = eng.debug_copy("idx", 0, (n_views,), np.int64)[:n_candidates]
    assert np.array_equal(first, np.asarray(g["idx"]).reshape(-1))                 # first stage: the reference's list, order included
    keep = eng.debug_copy("keep", 0, (n_views,), np.uint8)
    out = first[keep[first] != 0]
    assert len(out) == n
    return out


def _spec_of(g):
    aug = str(g["aug_type"]) if "aug_type" in g.files else "patch"
    return dict(aug_type=aug, threshold=float(g["plpd_threshold"]), patch_len=int(g["patch_len"]),
                occlusion_size=int(g["occlusion_size"]) if "occlusion_size" in g.files else 0,
                row_start=int(g["row_start"]) if "row_start" in g.files else 0,
                column_start=int(g["column_start"]) if "column_start" in g.files else 0)


@pytest.mark.parametrize("name", ["tiny_plpd", "tiny_plpd_occ", "tiny_plpd_pixel", "b16_n64_k200_plpd"])
@pytest.mark.parametrize("precision", ["strict", "fp16", "bf16"])
def test_fused_episode_with_a_plpd_stage_vs_reference(name, precision):
    """ttl_episode with ttl_episode_args.plpd against the fixture the reference wrote with --filter_plpd 1: the first-stage list,
    the PLPD value of every candidate, the surviving (second-stage) list, the gradients of the survivors' loss, the step and
    the adapted prediction."""
    from ttl_amd import deyo as D
    from ttl_amd.engine import TTLEngine
    g, cfg, W, x, lora0, tf = load_case(name)
    kw = episode_kwargs(g)
    eng, flat, names = make_engine(cfg, W, lora0, tf, x.shape[0], precision=precision)
    aux = TTLEngine(cfg, x.shape[0], tf.shape[0], "cuda:0", precision, share_from=eng)
    aux.set_text_features(torch.from_numpy(tf), float(np.exp(W["logit_scale"])))
    aux.bind_lora(flat)
    spec = _spec_of(g)
    B = len(np.asarray(g["idx"]).reshape(-1))
    torch.manual_seed(int(g["rng_seed"]))
    perm = D.draw_plpd_perms(spec, kw["n_updates"], B, x.shape[-1], "cuda")
    st = eng.plpd_struct(spec, perm, B, aux)
    snap, m, v = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat)
    l1, l0 = eng.episode(torch.from_numpy(x).cuda(), snap, m, v, n_updates=kw["n_updates"], objective="deyo",
                         mode=1 if kw["mode"] == "topk" else 0, rho=kw["rho"], margin=kw["margin"], lr=kw["lr"], want_logits0=True, plpd=st)
    torch.cuda.synchronize()
    tol = {"strict": 1e-5, "fp16": 1e-3, "bf16": 3e-2}[precision]
    assert max_rel(l0.cpu().numpy(), g["logits0"]) < tol
    plpd = eng.debug_copy("plpd", 0, (B,), np.float32)
    idx2 = survivors(eng, x.shape[0], B, len(np.asarray(g["idx2"]).reshape(-1)), g)
    ptol = {"strict": 2e-5, "fp16": 3e-3, "bf16": 1e-1}[precision]        # (bf16 at 64 views / K = 200: 6.1e-2 — a probability difference behind 4e-3 logit noise)
    assert np.abs(plpd - g["plpd"]).max() < ptol, np.abs(plpd - g["plpd"]).max()
    # the surviving set: exactly the reference's unless a candidate's PLPD sits within the build's noise of the threshold
    margin = np.abs(np.asarray(g["plpd"]) - spec["threshold"]).min()
    if margin > ptol:
        assert np.array_equal(np.sort(idx2), np.sort(np.asarray(g["idx2"]).reshape(-1))), (idx2, g["idx2"])
    lora1, grads = split(flat, lora0, names), split(eng.grads, lora0, names)
    gtol = {"strict": 1e-4, "fp16": 6e-3, "bf16": 6e-2}[precision]
    for k in names:
        gref = g["grad/" + k]
        if np.abs(gref).max() > 0:
            assert max_rel(grads[k], gref) < gtol, (k, max_rel(grads[k], gref))
    assert max_rel(l1.cpu().numpy(), g["logits1"]) < {"strict": 1e-4, "fp16": 5e-3, "bf16": 5e-2}[precision]
    assert int(np.argmax(l1.cpu().numpy())) == int(g["top5"][0, 0])
    # the same episode as a HIP graph replays bit-identically (the permutation buffer's address is baked in)
    if precision == "fp16":
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            xb, ob = torch.from_numpy(x).cuda(), torch.empty_like(l1)
            run = eng.episode_graph(xb, snap, m, v, ob, n_updates=kw["n_updates"], objective="deyo", mode=1 if kw["mode"] == "topk" else 0,
                                    rho=kw["rho"], margin=kw["margin"], lr=kw["lr"], plpd=st)
            run()
        s.synchronize()
        assert torch.equal(ob, l1)
    aux.close()
    eng.close()


@pytest.mark.parametrize("precision", ["strict", "fp16"])
def test_fused_text_mode_episode_with_a_plpd_stage_vs_reference(precision):
    """--lora_encoder text with --filter_plpd 1 through ttl_episode_text (driver.EpisodeRunner): the destroyed views need image
    FEATURES only, scored against the pending forward's text features; fixture tiny_text_plpd written by the reference."""
    from test_gpu_text import build_model, text_args
    from ttl_amd.driver import EpisodeRunner
    g, tcfg, model, opt, opt_state, x = build_model("tiny_text_plpd")
    model.precision = precision
    args = text_args(filter_plpd=1, plpd_threshold=float(g["plpd_threshold"]), aug_type="patch", patch_len=int(g["patch_len"]))
    with torch.no_grad():
        model.LoRA_reset()
    runner = EpisodeRunner(model, args)
    torch.manual_seed(int(g["rng_seed"]))
    out = runner(x)
    torch.cuda.synchronize()
    eng = model._ensure_engine()
    B = len(np.asarray(g["idx"]).reshape(-1))
    plpd = eng.txt.debug_copy("plpd", 0, (B,), np.float32)
    ptol = 2e-5 if precision == "strict" else 8e-3        # (fp16: both towers carry operand noise in this mode)
    assert np.abs(plpd - g["plpd"]).max() < ptol, np.abs(plpd - g["plpd"]).max()
    idx2 = survivors(eng.txt, x.shape[0], B, None, g)
    if np.abs(np.asarray(g["plpd"]) - float(g["plpd_threshold"])).min() > ptol:
        assert np.array_equal(np.sort(idx2), np.sort(np.asarray(g["idx2"]).reshape(-1)))
    assert max_rel(out.cpu().numpy(), g["logits1"]) < (1e-4 if precision == "strict" else 3e-2)
    assert int(out.argmax()) == int(g["top5"][0, 0])


def test_eval_loop_takes_the_fused_plpd_path_and_agrees_with_the_stepwise_loop(monkeypatch):
    """test_time_adapt_eval with --filter_plpd 1: the fused pipeline (3 episodes in flight, per-image permutations from the host
    generator) gives the same accuracy accumulator as the reference-shaped step-wise loop (TTL_PLPD_STEPWISE=1) on the same
    seed; 'occ' draws nothing, so the two runs see the same destroyed views whatever the draw order."""
    from test_gpu_dropin import build, ref_args
    from ttl_amd import eval as E
    g, cfg, model, opt, opt_state, x = build("tiny_plpd_occ")
    args = ref_args(filter_plpd=1, plpd_threshold=float(g["plpd_threshold"]), aug_type="occ", occlusion_size=int(g["occlusion_size"]),
                    row_start=int(g["row_start"]), column_start=int(g["column_start"]), patch_len=4)
    data = [(x.roll(i, 0).contiguous(), torch.tensor([i % 10])) for i in range(6)]
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("TTL_PLPD_STEPWISE", mode)
        with torch.no_grad():
            model.LoRA_reset()
        res[mode] = E.test_time_adapt_eval(data, model, None, opt, opt_state, None, args, n_streams=3)
    assert res["0"] == res["1"], res
