"""TEST INFRASTRUCTURE ONLY — CPU restatement of the view generator.

Only tests/ may import this.  It restates what the reference's host pipeline computes for one view
(data/datautils.py:98-157 with aug_list=[] and ttl.py:225-241):
    base view:   Resize(S, bicubic) + CenterCrop(S)            (ttl.py:229-231, datautils.py:151)
    other views: RandomResizedCrop(S) [bilinear] + RandomHorizontalFlip   (datautils.py:120-121)
    then ToTensor (/255) and Normalize((x-mean)/std) in fp32   (ttl.py:232-234)
torchvision's PIL backend forwards these to Pillow's ``Image.crop`` / ``Image.resize``; the resize is
Pillow's ImagingResample (third-party, Pillow 12.2.0 in the build image, src/libImaging/Resample.c):
two separable passes (horizontal, then vertical) over uint8 with 22-bit fixed-point coefficients
computed in double precision, each pass rounding and clipping to uint8.  This file restates that
algorithm bit for bit; it is pinned by tests/golden/views_pil.npz (outputs of Pillow itself,
tests/golden/make_views_golden.py) with exact equality.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bilinear(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def _bicubic(x):
    a = -0.5
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def coeffs(in_size, out_size, bicubic, first=0, count=None):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for output pixels first..first+count-1 of an
    in_size -> out_size resize.  Returns (xmin[count], taps: list of int arrays)."""
    count = out_size if count is None else count
    filt, rad = (_bicubic, 2.0) if bicubic else (_bilinear, 1.0)
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = rad * filterscale
    ss = 1.0 / filterscale
    mins, taps = [], []
    for o in range(first, first + count):
        center = (o + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0                               # Pillow accumulates left to right in double
        for v in w:
            ww += v
        k = [v / ww if ww != 0.0 else v for v in w]
        ki = [int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in k]
        mins.append(xmin)
        taps.append(np.asarray(ki, dtype=np.int64))
    return mins, taps


def _pass(img, mins, taps, axis):
    """One 8-bit resample pass along ``axis`` of an [H,W,3] uint8 image."""
    src = np.moveaxis(img.astype(np.int64), axis, 0)                 # [n_in, other, 3]
    out = np.empty((len(mins),) + src.shape[1:], dtype=np.int64)
    for o, (m, k) in enumerate(zip(mins, taps)):
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(k, src[m:m + len(k)], axes=(0, 0))
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def view_u8(img_u8, box, size):
    """uint8 [size,size,3] view.  box = (top, left, h, w, flags); flags bit0 = horizontal flip,
    bit1 = base view (Resize(shorter side -> size, bicubic) + CenterCrop(size), box ignored)."""
    top, left, h, w, flags = (int(v) for v in box)
    flip, base = flags & 1, (flags >> 1) & 1
    if base:
        H, W = img_u8.shape[:2]
        nh, nw = (int(size * H / W), size) if W <= H else (size, int(size * W / H))   # transforms.Resize(int)
        oy, ox = int(round((nh - size) / 2.0)), int(round((nw - size) / 2.0))         # transforms.CenterCrop
        src = img_u8
        cx, cy = coeffs(W, nw, 1, ox, size), coeffs(H, nh, 1, oy, size)
    else:
        src = img_u8[top:top + h, left:left + w]
        cx, cy = coeffs(w, size, 0), coeffs(h, size, 0)
    out = _pass(_pass(src, *cx, axis=1), *cy, axis=0)               # horizontal first, then vertical
    return out[:, ::-1] if flip else out


def make_view(img_u8, box, size, mean, std):
    t = view_u8(img_u8, box, size).astype(np.float32) / np.float32(255.0)            # ToTensor
    t = (t - np.asarray(mean, np.float32)) / np.asarray(std, np.float32)             # Normalize
    return np.ascontiguousarray(t.transpose(2, 0, 1))


def make_views(img_u8, boxes, size, mean, std):
    return np.stack([make_view(img_u8, b, size, mean, std) for b in boxes])
