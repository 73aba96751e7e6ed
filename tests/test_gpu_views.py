"""View generator (§8f-2) on the GPU, through the C ABI: byte work, so the bar is bit-exactness —
against the Pillow-generated fixture and against the oracle on random boxes."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import views_oracle as VO
from ttl_amd import views as V
from ttl_amd import _lib

pytestmark = pytest.mark.gpu


def test_views_bit_exact_with_pillow_fixture():
    g = np.load(f"{GOLDEN}/views_pil.npz")
    S = int(g["size"])
    for n in g["names"]:
        img = torch.from_numpy(g[f"{n}_img"]).cuda()
        got = V.make_views(img, torch.from_numpy(g[f"{n}_boxes"]), S).cpu().numpy()
        from test_views_cpu import normalized
        assert np.array_equal(got, normalized(g[f"{n}_views_u8"])), n


@pytest.mark.parametrize("hw", [(375, 500), (64, 48), (1080, 1920), (224, 224)])
def test_views_bit_exact_with_oracle_on_sampled_boxes(hw):
    H, W = hw
    rng = np.random.default_rng(H * 7 + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)       # white noise: every rounding decision matters
    boxes = V.draw_boxes(H, W, 12, torch.Generator().manual_seed(H + W))
    got = V.make_views(torch.from_numpy(img).cuda(), boxes, 224).cpu().numpy()
    ref = VO.make_views(img, boxes.numpy(), 224, V.CLIP_MEAN, V.CLIP_STD)
    assert np.array_equal(got, ref)


def test_views_extreme_boxes_and_small_output():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (97, 131, 3), dtype=np.uint8)
    boxes = np.asarray([[0, 0, 0, 0, 2], [0, 0, 97, 131, 0], [96, 130, 1, 1, 1], [10, 0, 1, 131, 0], [0, 7, 97, 1, 1],
                        [40, 40, 16, 16, 1]], np.int32)
    for S in (16, 64):
        got = V.make_views(torch.from_numpy(img).cuda(), torch.from_numpy(boxes), S).cpu().numpy()
        assert np.array_equal(got, VO.make_views(img, boxes, S, V.CLIP_MEAN, V.CLIP_STD)), S


def test_make_views_rejects_small_workspace_and_bad_input():
    lib = _lib.load("bf16")
    img = torch.zeros(32, 32, 3, dtype=torch.uint8, device="cuda")
    boxes = torch.tensor([[0, 0, 32, 32, 0]], dtype=torch.int32, device="cuda")
    out = torch.empty(1, 3, 16, 16, device="cuda")
    import ctypes as C
    m, s = (C.c_float * 3)(*V.CLIP_MEAN), (C.c_float * 3)(*V.CLIP_STD)
    need = lib.ttl_make_views_workspace_bytes(32, 32, 1, 16)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    assert lib.ttl_make_views(img.data_ptr(), 32, 32, boxes.data_ptr(), 1, 16, m, s, out.data_ptr(), ws.data_ptr(), need - 1, None) != 0
    assert b"workspace" in lib.ttl_last_error()
    assert lib.ttl_make_views(None, 32, 32, boxes.data_ptr(), 1, 16, m, s, out.data_ptr(), ws.data_ptr(), need, None) != 0
    with pytest.raises(ValueError):
        V.make_views(torch.zeros(32, 32, 3), boxes)             # host tensor: no silent CPU path


def test_gpu_augmenter_feeds_the_episode():
    """End to end: decoded uint8 image -> GPU views -> fused episode; view 0 is the base view."""
    from ttl_amd.config import VIT_TINY
    from ttl_amd import synth
    from ttl_amd.engine import TTLEngine
    from oracle import ttl_oracle as O
    cfg = VIT_TINY
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    aug = V.GpuAugMixAugmenter(n_views=7, size=cfg.image_size, generator=torch.Generator().manual_seed(2))
    views = aug(torch.from_numpy(img).cuda())
    assert tuple(views.shape) == (8, 3, cfg.image_size, cfg.image_size)
    base = VO.make_view(img, (0, 0, 0, 0, 2), cfg.image_size, V.CLIP_MEAN, V.CLIP_STD)
    assert np.array_equal(views[0].cpu().numpy(), base)
    W = synth.vision_weights(cfg, 0)
    lora0 = synth.lora_init(cfg, 1)
    tf = synth.text_features(10, cfg.embed, 2)
    eng = TTLEngine(cfg, 8, 10, "cuda:0", precision="bf16")       # (checked against the bf16-emulating oracle below)
    eng.load_weights(W)
    eng.set_text_features(torch.from_numpy(tf), float(np.exp(W["logit_scale"])))
    names = O.trainable_names(cfg)
    flat = torch.cat([torch.from_numpy(lora0[k]).reshape(-1) for k in names]).cuda().contiguous()
    eng.bind_lora(flat)
    l1 = eng.episode(views, flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat), n_updates=1)
    ref = O.episode(cfg, W, lora0, views.cpu().numpy(), tf, prec="bf16", n_updates=1)
    err = np.abs(l1.cpu().numpy() - ref["logits1"]).max() / np.abs(ref["logits1"]).max()
    assert err < 2e-2, err
    eng.close()


def test_eval_loop_with_gpu_views_equals_host_generated_views():
    """test_time_adapt_eval fed decoded uint8 images + GpuAugMixAugmenter == the same loop fed the views
    the oracle (== Pillow) produces for the same boxes."""
    from test_gpu_dropin import build, ref_args
    from ttl_amd.eval import test_time_adapt_eval
    g, cfg, model, opt, opt_state, x = build("tiny_deyo")
    args = ref_args()
    rng = np.random.default_rng(3)
    imgs = [rng.integers(0, 256, (80 + 7 * i, 100 - 5 * i, 3), dtype=np.uint8) for i in range(4)]
    labels = [1, 3, 5, 7]

    class Decoded:
        def __iter__(self):
            for im, lb in zip(imgs, labels):
                yield torch.from_numpy(im), lb

    aug = V.GpuAugMixAugmenter(7, cfg.image_size, generator=torch.Generator().manual_seed(9))
    a = test_time_adapt_eval(Decoded(), model, None, opt, opt_state, None, args, n_streams=2, gpu_augmenter=aug)

    gen = torch.Generator().manual_seed(9)

    class HostViews:
        def __iter__(self):
            for im, lb in zip(imgs, labels):
                boxes = V.draw_boxes(im.shape[0], im.shape[1], 8, gen).numpy()
                yield torch.from_numpy(VO.make_views(im, boxes, cfg.image_size, V.CLIP_MEAN, V.CLIP_STD)), lb

    b = test_time_adapt_eval(HostViews(), model, None, opt, opt_state, None, args, n_streams=2)
    assert a == b


def test_eval_pipeline_is_reused_across_datasets():
    """Second dataset on the same model (new class names -> new text features): the cached slots are
    rebound, results equal a fresh pipeline's."""
    from test_gpu_dropin import build, ref_args
    from ttl_amd.eval import test_time_adapt_eval, SyntheticViews
    g, cfg, model, opt, opt_state, x = build("tiny_deyo")
    args = ref_args()
    data = SyntheticViews(cfg, 4, 8, 10, seed=5)
    r1 = test_time_adapt_eval(data, model, None, opt, opt_state, None, args, n_streams=2)
    pipes = dict(model._episode_pipelines)
    r2 = test_time_adapt_eval(data, model, None, opt, opt_state, None, args, n_streams=2)
    assert r1 == r2 and model._episode_pipelines == pipes          # same objects, accumulators were reset
    model.reset_classnames([f"other {i}" for i in range(7)], "ViT-tiny")
    ra = test_time_adapt_eval(data, model, None, opt, opt_state, None, args, n_streams=2)
    for p in model._episode_pipelines.values():
        p.close()
    model._episode_pipelines.clear()
    rb = test_time_adapt_eval(data, model, None, opt, opt_state, None, args, n_streams=2)
    assert ra == rb
