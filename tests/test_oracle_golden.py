"""Pins oracle/ttl_oracle.py (the CPU restatement) against fixtures produced by the
reference itself (tests/golden/make_golden.py).  Runs on CPU; no GPU, no /root/reference."""
import os

import numpy as np
import pytest

from oracle import ttl_oracle as O
from helpers import load_case, episode_kwargs, max_rel, check_lora_step

TINY = ["tiny_deyo", "tiny_topk", "tiny_steps2", "tiny_r32", "tiny_tpt", "tiny197_deyo", "tiny_mid_deyo", "tiny_all_deyo", "tiny_qkvo_deyo", "tiny_qkvo_steps2",
        "tiny_qkvo_steps2_b",     # 4 updates, adapters on q/k/v/out with NON-ZERO B: every gradient is signal -> the tight multi-update bounds apply
        "tiny_outliers"]          # CLIP-like activation outliers (synth.add_activation_outliers)


@pytest.fixture(scope="module")
def unit(golden_dir):
    return np.load(golden_dir + "/unit_loss_adamw.npz")


@pytest.mark.parametrize("s", ["a", "b", "c", "d"])
def test_softmax_entropy_and_selection(unit, s):
    z = unit[f"{s}/z"]
    H = O.softmax_entropy(z)
    np.testing.assert_allclose(H, unit[f"{s}/H"], rtol=2e-5, atol=2e-6)   # deyo.py:85-90
    _, idx = O.select_confident_samples(z, 0.1)                            # ttl.py:50-54
    assert np.array_equal(idx, unit[f"{s}/topk_idx"])                      # bit-exact index set + order
    if idx.size:
        assert abs(O.avg_entropy(z[idx]) - unit[f"{s}/avg_entropy"]) <= 2e-5 * max(1, abs(unit[f"{s}/avg_entropy"]))
        t = O.tpt_loss_and_grad(z, rho=0.1)
        assert max_rel(t["dz"], unit[f"{s}/tpt_dz"]) < 1e-4
    else:
        assert int(z.shape[0] * 0.1) == 0                                  # int(8*0.1)=0 -> empty


@pytest.mark.parametrize("s", ["a", "b", "c", "d"])
@pytest.mark.parametrize("mode", ["le_thresh", "topk"])
def test_deyo_loss_grad(unit, s, mode):
    z = unit[f"{s}/z"]
    L = O.deyo_loss_and_grad(z, mode=mode, rho=0.1, margin=0.4, reweight=1.0)
    if f"{s}/{mode}/idx" not in unit.files:
        assert L["loss"] is None and not L["dz"].any()                     # deyo.py:110-113 early return
        return
    assert np.array_equal(L["idx"], unit[f"{s}/{mode}/idx"])
    assert abs(L["loss"] - unit[f"{s}/{mode}/loss"]) <= 1e-5 * abs(unit[f"{s}/{mode}/loss"]) + 1e-7
    assert max_rel(L["dz"], unit[f"{s}/{mode}/dz"]) < 1e-4


@pytest.mark.parametrize("s", ["t1", "t2", "t3"])
def test_exact_entropy_ties_at_the_selection_boundary(golden_dir, s):
    """Bit-identical logit rows straddling rank int(N*rho): the reference's torch.argsort (unstable) keeps the LOWEST view
    indices of the tied group but returns them in no particular order; the oracle (stable sort) selects the same SET, and
    everything ahead of the tied group in the same order.  The loss is a mean over the set, so the order does not matter."""
    u = np.load(golden_dir + "/unit_ties.npz")
    z, rho = u[f"{s}/z"], float(u[f"{s}/rho"])
    H = O.softmax_entropy(z)
    ref = u[f"{s}/topk/idx"]
    assert np.array_equal(ref, u[f"{s}/tpt_idx"])
    idx = O.select_views(H, "topk", z.shape[0], rho)
    tied = set(u[f"{s}/tied_rows"].tolist())
    assert sorted(idx.tolist()) == sorted(ref.tolist())
    lead = [i for i in ref.tolist() if i not in tied]
    assert idx.tolist()[:len(lead)] == lead
    assert [i for i in idx.tolist() if i in tied] == sorted(tied)[:len(idx) - len(lead)]      # lowest indices, ascending


def test_adamw_three_steps(unit):
    p = unit["adamw/p0"]
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for t in range(3):
        p, m, v = O.adamw_step(p, unit[f"adamw/g{t}"], m, v, t + 1)
        np.testing.assert_allclose(p, unit[f"adamw/p{t + 1}"], rtol=1e-5, atol=1e-7)


def _run_case(name, check_taps):
    g, cfg, W, x, lora0, tf = load_case(name)
    if check_taps:
        taps = {}
        net = O.VitOracle(cfg, W, lora0, "fp32")
        z = net.logits(net.forward(x, taps=taps), tf)
        for k, v in taps.items():
            if "tap/" + k in g.files:
                assert max_rel(v, g["tap/" + k]) < 2e-5, k
    trace = []
    out = O.episode(cfg, W, lora0, x, tf, trace=trace, **episode_kwargs(g))
    assert max_rel(out["logits0"], g["logits0"]) < 2e-5
    np.testing.assert_allclose(trace[0]["H"], g["H"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(trace[0]["idx"], g["idx"])                       # selection: bit-exact
    assert abs(trace[0]["loss"] - g["loss"]) <= 2e-5 * abs(g["loss"])
    n_up = int(g["n_updates"])
    gmax = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("grad/"))
    # noise-driven episodes: several updates with k_proj adapters whose B starts at ZERO (tiny_qkvo_steps2).  With non-zero B
    # (tiny_qkvo_steps2_b) the k_proj gradients are signal and the episode takes the tight bounds of the q / v case
    noisy = n_up > 1 and any("k_proj" in k and "lora_B" in k and not np.any(lora0[k[5:]]) for k in g.files if k.startswith("grad/"))
    for k in g.files:
        if k.startswith("grad/"):
            got = trace[-1]["grads"][k[5:]]
            # the first AdamW steps are sign-like (Q11): an element whose gradient is at fp32 noise level (k_proj adapters: the
            # softmax is shift-invariant along the keys) moves by +-lr on the sign of that noise, so after several updates the two
            # runs sit at measurably different points and the LAST update's gradients agree only on the scale of the largest one
            if n_up == 1:
                assert max_rel(got, g[k]) < 1e-4, (k, max_rel(got, g[k]))
            elif noisy:
                # ONLY episodes that train k_proj adapters get the loose bound: the true k_proj gradient is ~0 by shift
                # invariance, what both runs hold is fp32 noise, the sign-like steps (Q11) move those adapters by +-lr on the
                # sign of that noise, and from the second update on EVERY tensor's gradient is taken at a measurably different
                # point; compared on the scale of the largest gradient of the episode
                assert max_rel(got, g[k]) < 2e-2 or np.abs(got - g[k]).max() < 0.25 * gmax, (k, max_rel(got, g[k]))
            else:     # q / v adapters only: four sign-like updates leave the last gradients within 2.3e-2 (b16_r32_n16_steps2)
                assert max_rel(got, g[k]) < 3e-2, (k, max_rel(got, g[k]))
    lr = float(g["lr"])
    for k in g.files:
        if k.startswith("lora1/"):
            if n_up == 1:
                check_lora_step(out["lora"][k[6:]], g[k], g["grad/" + k[6:]], lr, 1e-4, k)
            else:
                # several sign-like updates: an element whose gradient sits at fp32 noise level in one of them lands one
                # +-lr step away (Q11).  All but a handful of elements agree closely, and hardly any is off by more than ONE
                # such step (a regression in the resumed forward or the multi-step backward moves whole tensors, not a handful)
                err = np.abs(np.asarray(out["lora"][k[6:]], np.float64) - g[k])
                far = float((err > 2e-2 * np.abs(g[k]).max()).mean())
                off = float((err > 1.05 * lr).mean())
                if noisy:     # k_proj adapters in the episode: their elements are noise-driven, the others follow from update 2 on
                    assert far < (0.6 if "k_proj" in k else 0.05) and off < (0.3 if "k_proj" in k else 0.02), (k, far, off)
                else:
                    assert far < 1e-3 and off < 1e-3, (k, far, off)
    assert max_rel(trace[-1]["logits"], g["logits_last"]) < (2e-5 if n_up == 1 else 2e-3)
    assert max_rel(out["logits1"], g["logits1"]) < (1e-4 if n_up == 1 else 2e-3)
    assert np.array_equal(np.argsort(-out["logits1"], 1)[:, :1], g["top5"][:, :1])


@pytest.mark.parametrize("name", TINY)
def test_episode_tiny(name):
    _run_case(name, check_taps=True)


def test_episode_b16_n8_k10():
    """BASELINE config 1 (CPU plumbing case): ViT-B/16, r=16, 8 views, K=10."""
    _run_case("b16_n8_k10", check_taps=False)


def test_episode_b16_n8_k10_qkvo():
    """Adapters on q, k, v AND out_proj (BASELINE.json north_star) on the full ViT-B/16 geometry, every B non-zero, one update:
    the unmodified reference with the harness's LoraConfig override (clip/custom_clip.py:583-590 hard-codes q_proj, v_proj)."""
    _run_case("b16_n8_k10_qkvo", check_taps=False)


def test_episode_b16_n8_k10_outliers():
    """ViT-B/16 with CLIP-like activation outliers (residual channels 100-150x the median on the CLS token, one patch token and,
    from layer 2 on, every token; LayerNorm gains from 0.02 to 4 on them): the statistics of the checkpoint the reference loads
    (clip/custom_clip.py:581), which Gaussian weights do not show — through the reference itself, 8 views."""
    _run_case("b16_n8_k10_outliers", check_taps=False)


def test_episode_l14_n4_k10():
    """BASELINE config 4's geometry through the reference itself (ViT-L/14: patch 14, T = 257, D = 1024, 16 heads, 24 layers,
    adapters on layers 21-23) at 4 views."""
    _run_case("l14_n4_k10", check_taps=False)


def test_episode_b32_n8_k10():
    """The run script's other --arch option (ViT-B/32: patch 32, T = 50) through the reference itself."""
    _run_case("b32_n8_k10", check_taps=False)


def test_episode_b16_r32_n16_steps2():
    """BASELINE config 5's features at a CPU-sized view count, through the reference itself: rank 32, --tta_steps 2 = 4 optimizer
    updates (Q6), top-rho selection (1 of 16 views), full ViT-B/16 geometry."""
    _run_case("b16_r32_n16_steps2", check_taps=False)


@pytest.mark.slow
@pytest.mark.parametrize("name", ["b16_n64_k200_ent0", "b16_n64_k200_ent1", "b16_n64_k1000_ent1", "b16_n64_k200_outliers", "b16_n64_k200_qkvo",
                                  "b16_n64_k200_tpt", "l14_n64_k200", "b16_r32_n128_k1000_steps2"])
def test_episode_b16_n64(name):
    # ViT-L/14 at 64 views is 170 s of numpy on 8 cores, config 5 at 128 views x 4 updates several minutes (both pass, round 4):
    # opt-in, so that the default CPU suite stays at minutes
    if name in ("l14_n64_k200", "b16_r32_n128_k1000_steps2") and not os.environ.get("TTL_FULL_ORACLE"):
        pytest.skip("set TTL_FULL_ORACLE=1 to pin the oracle on the full-size ViT-L/14 and 128-view fixtures too (minutes each)")
    _run_case(name, check_taps=False)


def test_bf16_mode_is_close_to_fp32():
    """The bf16-emulating mode only moves rounding points; it must stay near fp32."""
    g, cfg, W, x, lora0, tf = load_case("tiny_deyo")
    z32 = O.VitOracle(cfg, W, lora0, "fp32")
    z16 = O.VitOracle(cfg, W, lora0, "bf16")
    a = z32.logits(z32.forward(x), tf)
    b = z16.logits(z16.forward(x), tf)
    assert max_rel(b, a) < 3e-2


@pytest.mark.parametrize("name", ["tiny_plpd", pytest.param("b16_n64_k200_plpd", marks=pytest.mark.slow)])
def test_plpd_filter_against_reference(name):
    """deyo.py:115-151 (filter_plpd=1): the second-stage mask from the reference's own destroyed-view logits,
    then loss / grads / post-step weights of the surviving views (tiny geometry; and the benched size with the reference's
    default --patch_len 6, 25 of 64 views surviving)."""
    if name != "tiny_plpd" and not os.environ.get("TTL_FULL_ORACLE"):
        pytest.skip("set TTL_FULL_ORACLE=1 to pin the oracle on the 64-view ViT-B/16 PLPD fixture too (a minute; passes, round 5)")
    g, cfg, W, x, lora0, tf = load_case(name)
    plpd, keep = O.plpd_keep(g["logits0"], g["logits_prime"], g["idx"], float(g["plpd_threshold"]))
    np.testing.assert_allclose(plpd, g["plpd"], rtol=1e-4, atol=1e-6)
    assert np.array_equal(np.nonzero(keep)[0], np.sort(g["idx2"]))
    trace = []
    out = O.episode(cfg, W, lora0, x, tf, trace=trace, keep=keep, **episode_kwargs(g))
    assert np.array_equal(trace[0]["idx"], np.sort(g["idx2"]))
    assert abs(trace[0]["loss"] - g["loss"]) <= 2e-5 * abs(g["loss"])
    for k in g.files:
        if k.startswith("grad/"):
            assert max_rel(trace[-1]["grads"][k[5:]], g[k]) < 1e-4, k
    assert max_rel(out["logits1"], g["logits1"]) < 1e-4


def test_k_and_out_proj_adapters_gradients_by_finite_differences():
    """k_proj / out_proj adapters (BASELINE.json north_star; the reference ships q_proj, v_proj only, clip/custom_clip.py:586):
    the oracle's analytic LoRA gradients for all four targets against central finite differences of its own loss, fp32."""
    from ttl_amd import synth
    from ttl_amd.config import get_config
    cfg = get_config("tiny").replace(lora_targets=("q_proj", "k_proj", "v_proj", "out_proj"))
    W = synth.vision_weights(cfg, 0)
    lora = synth.lora_init(cfg, 0)
    rng = np.random.default_rng(1)
    for k in lora:                                  # B != 0, otherwise dA == 0 (Q11) and nothing is tested
        if "lora_B" in k:
            lora[k] = (rng.standard_normal(lora[k].shape) * 0.05).astype(np.float32)
    x = synth.views(cfg, 3, 2)
    tf = synth.text_features(7, cfg.embed)
    names = O.trainable_names(cfg)
    assert len(names) == 3 * 4 * 2 and any("k_proj" in n for n in names) and any("out_proj" in n for n in names)

    def loss_of(lr):
        net = O.VitOracle(cfg, W, lr, "fp32")
        z = net.logits(net.forward(x.astype(np.float64)).astype(np.float64), tf.astype(np.float64))
        return float(O.deyo_loss_and_grad(z.astype(np.float64))["loss"])

    net = O.VitOracle(cfg, W, lora, "fp32")
    save = {}
    f = net.forward(x, save)
    L = O.deyo_loss_and_grad(net.logits(f, tf))
    grads = net.backward(L["dz"], tf, save)
    checked = 0
    for k in names:
        g = grads[k]
        idx = np.unravel_index(np.argmax(np.abs(g)), g.shape)
        eps = 1e-2
        lp = {n: v.copy() for n, v in lora.items()}; lp[k][idx] += eps
        lm = {n: v.copy() for n, v in lora.items()}; lm[k][idx] -= eps
        fd = (loss_of(lp) - loss_of(lm)) / (2 * eps)
        assert abs(fd - g[idx]) < 2e-2 * abs(g[idx]) + 1e-6, (k, fd, g[idx])
        checked += 1
    assert checked == len(names)


@pytest.mark.parametrize("name", ["tiny_deyo", "tiny_topk", "tiny_tpt", "tiny_r32", "tiny_mid_deyo", "tiny_qkvo_deyo", "tiny_steps2", "b16_n8_k10", "b16_n8_k10_qkvo",
                                  "tiny_outliers", "b16_n8_k10_outliers"])
def test_torch_restatement_vs_reference_goldens(name):
    """oracle/ttl_oracle_torch.py (torch fp32 + autograd + torch.optim.AdamW on the host cores: what bench.py's cpu_baseline leg
    times) against the fixtures the reference itself wrote: logits, selection list, every gradient, the updated
    adapters and the adapted logits."""
    torch = pytest.importorskip("torch")
    from oracle import ttl_oracle_torch as OT
    g, cfg, W, x, lora0, tf = load_case(name)
    kw = episode_kwargs(g)
    tower = OT.TorchTower(cfg, W)
    trace = []
    out = OT.episode(tower, lora0, torch.from_numpy(x), torch.from_numpy(np.asarray(tf, np.float32)), objective=kw["objective"],
                     mode=kw["mode"], rho=kw["rho"], margin=kw["margin"], n_updates=kw["n_updates"], lr=kw["lr"], trace=trace)
    assert max_rel(out["logits0"], g["logits0"]) < 2e-5
    assert np.array_equal(np.sort(out["idx"]), np.sort(np.asarray(g["idx"]).reshape(-1)))
    n_up = int(g["n_updates"])
    assert OT.trainable_names(cfg) == O.trainable_names(cfg)
    for k in OT.trainable_names(cfg):
        if n_up == 1:
            if np.abs(g["grad/" + k]).max() > 0:
                assert max_rel(trace[-1][k], g["grad/" + k]) < 1e-4, k
            check_lora_step(out["lora"][k], g["lora1/" + k], g["grad/" + k], kw["lr"], 1e-4, k)
        else:
            assert max_rel(trace[-1][k], g["grad/" + k]) < 3e-2, k
    assert max_rel(out["logits1"], g["logits1"]) < (1e-4 if n_up == 1 else 2e-3)
    assert np.array_equal(np.argsort(-out["logits1"], 1)[:, :1], g["top5"][:, :1])
