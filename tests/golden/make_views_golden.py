"""Generates tests/golden/views_pil.npz: PIL outputs for the host view pipeline the reference uses
(torchvision's Resize/CenterCrop/resized_crop/hflip are thin wrappers over these PIL calls;
torchvision itself is not installed in the build container, PIL is).  Run once in the build container:
    python tests/golden/make_views_golden.py
"""
import os

import numpy as np
from PIL import Image

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def smooth_image(rng, h, w):
    """Photo-like content: low-frequency colour field + edges + mild noise."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for c in range(3):
        for _ in range(6):
            fy, fx, ph = rng.uniform(0.005, 0.08), rng.uniform(0.005, 0.08), rng.uniform(0, 6.28)
            img[..., c] += rng.uniform(10, 40) * np.sin(fy * yy + fx * xx + ph)
    img += 128 + 60 * ((xx // 37 + yy // 29) % 2)[..., None] * rng.uniform(0.2, 1.0, 3)
    img += rng.normal(0, 4, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def to_tensor_norm(pil):
    a = np.asarray(pil, dtype=np.float32) / 255.0
    a = (a - np.asarray(MEAN, np.float32)) / np.asarray(STD, np.float32)
    return np.ascontiguousarray(a.transpose(2, 0, 1))


def center_view(pil, S):
    w, h = pil.size
    if w <= h:
        nw, nh = S, int(S * h / w)
    else:
        nh, nw = S, int(S * w / h)
    r = pil.resize((nw, nh), Image.BICUBIC)                      # transforms.Resize(S, BICUBIC)
    top, left = int(round((nh - S) / 2.0)), int(round((nw - S) / 2.0))
    return r.crop((left, top, left + S, top + S))                # transforms.CenterCrop(S)


def crop_view(pil, box, S):
    top, left, h, w, flip = box
    r = pil.crop((left, top, left + w, top + h)).resize((S, S), Image.BILINEAR)   # F.resized_crop
    return r.transpose(Image.FLIP_LEFT_RIGHT) if flip & 1 else r                       # F.hflip


def main():
    rng = np.random.default_rng(20241022)
    out = {}
    S = 224
    cases = [("sq", 256, 256), ("wide", 375, 500), ("tall", 640, 427), ("small", 120, 160)]
    for name, h, w in cases:
        img = smooth_image(rng, h, w)
        pil = Image.fromarray(img)
        s = min(h, w)
        boxes = [((h - s) // 2, (w - s) // 2, s, s, 2)]      # flags bit1: the base (Resize+CenterCrop) view
        views = [np.asarray(center_view(pil, S))]
        for _ in range(5):
            bh, bw = int(rng.integers(max(8, h // 6), h + 1)), int(rng.integers(max(8, w // 6), w + 1))
            top, left = int(rng.integers(0, h - bh + 1)), int(rng.integers(0, w - bw + 1))
            b = (top, left, bh, bw, int(rng.integers(0, 2)))
            boxes.append(b)
            views.append(np.asarray(crop_view(pil, b, S)))
        out[f"{name}_img"] = img
        out[f"{name}_boxes"] = np.asarray(boxes, np.int32)
        out[f"{name}_views_u8"] = np.stack(views)          # Pillow's uint8 output [n,S,S,3]; ToTensor+Normalize: to_tensor_norm
    out["names"] = np.asarray([c[0] for c in cases])
    out["size"] = np.int32(S)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "views_pil.npz"), **out)


if __name__ == "__main__":
    main()
