"""--lora_encoder text: the oracle's text tower / text-mode episode against fixtures the
reference itself produced (tests/golden/make_golden_text.py)."""
import numpy as np
import pytest

from oracle import ttl_oracle as O
from helpers import load_text_case, episode_kwargs, max_rel, check_lora_step

CASES = ["tiny_text_deyo", "tiny_text_topk", "tiny_text_steps2", "tiny_text_tpt"]


@pytest.mark.parametrize("name", CASES)
def test_text_episode_matches_reference(name):
    g, vcfg, tcfg, Wv, Wt, x, ids, lora0 = load_text_case(name)
    kw = episode_kwargs(g)
    trace = []
    r = O.episode_text(vcfg, tcfg, Wv, Wt, lora0, x, ids, trace=trace, **kw)
    assert max_rel(r["image_features"], g["image_features"]) < 5e-6
    assert max_rel(r["logits0"], g["logits0"]) < 1e-5
    assert max_rel(trace[0]["H"], g["H"]) < 5e-5
    assert np.array_equal(np.sort(trace[0]["idx"]), np.sort(g["idx"]))
    assert abs(float(trace[0]["loss"]) - float(g["loss"])) < 2e-6
    names = O.trainable_names(tcfg, "text_model")
    if kw["n_updates"] == 1:
        for k in names:
            gr = g["grad/" + k]
            if np.abs(gr).max() == 0:
                assert not trace[0]["grads"][k].any()
            else:
                assert max_rel(trace[0]["grads"][k], gr) < 5e-5, k
            dg = float(np.abs(trace[0]["grads"][k] - gr).max())
            check_lora_step(r["lora"][k], g["lora1/" + k], gr, kw["lr"], 1e-5, k, dg=dg + 1e-12)
    tol = 2e-3 if kw["n_updates"] == 1 else 6e-3                 # sign-like Adam steps on near-zero grads (Q11), 4 of them
    assert max_rel(trace[-1]["logits"], g["logits_last"]) < tol
    assert max_rel(r["logits1"], g["logits1"]) < tol
    assert int(np.argmax(r["logits1"])) == int(g["top5"][0, 0])


@pytest.mark.slow
def test_text_b16_forward_matches_reference():
    g, vcfg, tcfg, Wv, Wt, x, ids, lora0 = load_text_case("b16_text_n8_k10")
    net = O.TextOracle(tcfg, Wt, lora0)
    t = net.forward(ids)
    th = t / np.linalg.norm(t, axis=-1, keepdims=True)
    z = np.float32(np.exp(Wv["logit_scale"])) * g["image_features"] @ th.T
    assert max_rel(z, g["logits0"]) < 2e-5


def test_causal_mask_and_eot_pooling_are_what_matters():
    """Tokens after the end-of-text position cannot change the text features (causal mask + eot pooling)."""
    from ttl_amd import synth
    from ttl_amd.config import TEXT_TINY
    Wt = synth.text_weights(TEXT_TINY, 0)
    lora = synth.lora_init(TEXT_TINY, 1, tower="text_model")
    ids = synth.token_ids(4, TEXT_TINY, 5)
    a = O.TextOracle(TEXT_TINY, Wt, lora).forward(ids)
    ids2 = ids.copy()
    for k in range(4):
        e = ids[k].argmax()
        ids2[k, e + 1:] = 7        # garbage (smaller than eot) after the pooled position
    b = O.TextOracle(TEXT_TINY, Wt, lora).forward(ids2)
    assert np.allclose(a, b, atol=1e-6)


def test_torch_text_tower_equals_the_pinned_numpy_one():
    """oracle/ttl_oracle_torch.text_features_forward (what bench.py's reference-faithful CPU figure times: the reference
    recomputes the class-text features in every forward, clip/custom_clip.py:669-671) against TextOracle.forward — the numpy
    text tower pinned to the reference-generated text fixtures above — on the tiny and the ViT-B/16 text geometries."""
    pytest.importorskip("torch")
    from oracle import ttl_oracle_torch as OT
    from ttl_amd import synth
    from ttl_amd.config import get_text_config
    for arch, n in (("tiny", 12), ("ViT-B/16", 4)):
        tcfg = get_text_config(arch)
        Wt = synth.text_weights(tcfg, 0)
        ids = synth.token_ids(n, tcfg, 3)
        net = O.TextOracle(tcfg, Wt, synth.lora_init(tcfg, 0, tower="text_model"), "fp32")
        net.trained = lambda i: False
        want = net.forward(ids)
        got = OT.text_features_forward(tcfg, Wt, ids).numpy()
        assert max_rel(got, want) < 2e-5, (arch, max_rel(got, want))
