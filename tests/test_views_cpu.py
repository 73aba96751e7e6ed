"""View generator (§8f-2), CPU side: the oracle restatement of Pillow's 8-bit resampler is pinned
bit-exactly to Pillow outputs (tests/golden/views_pil.npz), and the host box sampler follows
torchvision's RandomResizedCrop.get_params semantics."""
import math

import numpy as np
import torch

from conftest import GOLDEN
from oracle import views_oracle as VO
from ttl_amd import views as V


def _fixture():
    return np.load(f"{GOLDEN}/views_pil.npz")


def normalized(u8):
    """ToTensor + Normalize (ttl.py:232-234) of Pillow's uint8 views [n,S,S,3] -> fp32 [n,3,S,S]."""
    a = u8.astype(np.float32) / np.float32(255.0)
    a = (a - np.asarray(V.CLIP_MEAN, np.float32)) / np.asarray(V.CLIP_STD, np.float32)
    return np.ascontiguousarray(a.transpose(0, 3, 1, 2))


def test_oracle_is_bit_exact_with_pillow_fixture():
    g = _fixture()
    S = int(g["size"])
    for n in g["names"]:
        u8 = np.stack([VO.view_u8(g[f"{n}_img"], b, S) for b in g[f"{n}_boxes"]])
        assert np.array_equal(u8, g[f"{n}_views_u8"]), n                      # the resampler, byte for byte
        got = VO.make_views(g[f"{n}_img"], g[f"{n}_boxes"], S, V.CLIP_MEAN, V.CLIP_STD)
        assert got.dtype == np.float32 and np.array_equal(got, normalized(g[f"{n}_views_u8"])), n


def test_oracle_matches_live_pillow_when_available():
    try:
        from PIL import Image
    except ImportError:                                        # pragma: no cover
        import pytest
        pytest.skip("Pillow not installed")
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (300, 211, 3), dtype=np.uint8)   # worst case for rounding: white noise
    pil = Image.fromarray(img)
    for box in [(10, 20, 250, 150, 0), (0, 0, 300, 211, 1), (100, 50, 9, 12, 0), (3, 5, 224, 200, 1)]:
        top, left, h, w, flip = box
        r = pil.crop((left, top, left + w, top + h)).resize((224, 224), Image.BILINEAR)
        ref = np.asarray(r)[:, ::-1] if flip else np.asarray(r)
        assert np.array_equal(VO.view_u8(img, box, 224), ref), box
    # base view: Resize(224, bicubic) + CenterCrop(224)
    nh, nw = int(224 * 300 / 211), 224
    r = np.asarray(pil.resize((nw, nh), Image.BICUBIC))
    oy = int(round((nh - 224) / 2.0))
    assert np.array_equal(VO.view_u8(img, (0, 0, 0, 0, 2), 224), r[oy:oy + 224])


def test_random_resized_crop_box_semantics():
    g = torch.Generator().manual_seed(0)
    H, W = 375, 500
    areas, ratios = [], []
    for _ in range(2000):
        i, j, h, w = V.random_resized_crop_box(H, W, generator=g)
        assert 0 <= i and i + h <= H and 0 <= j and j + w <= W and h > 0 and w > 0
        areas.append(h * w / (H * W))
        ratios.append(w / h)
    areas, ratios = np.asarray(areas), np.asarray(ratios)
    assert areas.min() >= 0.08 - 0.01 and areas.max() <= 1.0
    assert ratios.min() >= 3 / 4 - 0.05 and ratios.max() <= 4 / 3 + 0.05
    # log-uniform aspect: median ~ 1; area roughly uniform (rejections trim the top end a little)
    assert abs(math.log(np.median(ratios))) < 0.12      # wide image: tall boxes are rejected a bit more often
    assert 0.4 < np.median(areas) < 0.6


def test_random_resized_crop_fallback_is_central_and_ratio_clamped():
    # an extremely wide image rejects all 10 tries often; the fallback clamps to ratio 4/3 and centres
    g = torch.Generator().manual_seed(1)
    seen_fallback = False
    for _ in range(200):
        i, j, h, w = V.random_resized_crop_box(10, 1000, generator=g)
        assert h <= 10 and w <= 1000
        if h == 10 and w == int(round(10 * 4 / 3)):
            seen_fallback = True
            assert i == 0 and j == (1000 - w) // 2
    assert seen_fallback


def test_draw_boxes_layout_and_determinism():
    a = V.draw_boxes(375, 500, 64, torch.Generator().manual_seed(3))
    b = V.draw_boxes(375, 500, 64, torch.Generator().manual_seed(3))
    assert a.dtype == torch.int32 and tuple(a.shape) == (64, 5) and torch.equal(a, b)
    assert int(a[0, 4]) == V.FLAG_BASE                         # view 0 = the un-augmented base view
    flips = a[1:, 4]
    assert set(flips.tolist()) <= {0, 1} and 10 < int(flips.sum()) < 53
