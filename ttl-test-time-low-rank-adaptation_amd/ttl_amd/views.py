"""GPU-side view generation — the host half.

The reference builds the N views of one test image on the CPU with PIL (data/datautils.py:98-157:
``AugMixAugmenter`` = [preprocess(base_transform(x))] + [augmix(x) for n_views-1], where augmix with
an empty aug_list (Q13) is RandomResizedCrop(224) + RandomHorizontalFlip, then ToTensor + Normalize;
base_transform = Resize(224, bicubic) + CenterCrop(224), ttl.py:225-241).  That costs tens of ms per
image on one core — more than the whole adaptation episode on an MI355X — so for throughput runs the
crop boxes are drawn here (same distribution as torchvision's ``RandomResizedCrop.get_params``,
driven by the torch RNG) and the resampling + normalisation run on the GPU (csrc/views.hip),
bit-exact with Pillow's 8-bit two-pass resampler the host pipeline ends in.
"""
import ctypes as C
import math

import torch

from . import _lib

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)       # ttl.py:225-226
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


FLAG_FLIP, FLAG_BASE = 1, 2


def center_box(height, width):
    """The base view: Resize(S, bicubic) + CenterCrop(S) over the whole image (flags bit1; the kernel
    derives the geometry from H, W like torchvision does — the box is informational)."""
    s = min(height, width)
    return ((height - s) // 2, (width - s) // 2, s, s, FLAG_BASE)


def random_resized_crop_box(height, width, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.), generator=None):
    """torchvision ``RandomResizedCrop.get_params`` semantics: 10 tries of (area, log-uniform aspect),
    central fallback clamped to the ratio range.  Returns (top, left, h, w)."""
    area = height * width
    log_ratio = torch.log(torch.tensor(ratio))           # fp32, like torchvision: same RNG stream -> same boxes
    lo, hi = float(log_ratio[0]), float(log_ratio[1])
    for _ in range(10):
        target = area * torch.empty(1).uniform_(scale[0], scale[1], generator=generator).item()
        aspect = torch.exp(torch.empty(1).uniform_(lo, hi, generator=generator)).item()
        w = int(round(math.sqrt(target * aspect)))
        h = int(round(math.sqrt(target / aspect)))
        if 0 < w <= width and 0 < h <= height:
            i = torch.randint(0, height - h + 1, (1,), generator=generator).item()
            j = torch.randint(0, width - w + 1, (1,), generator=generator).item()
            return i, j, h, w
    in_ratio = float(width) / float(height)
    if in_ratio < ratio[0]:
        w = width
        h = int(round(w / ratio[0]))
    elif in_ratio > ratio[1]:
        h = height
        w = int(round(h * ratio[1]))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def draw_boxes(height, width, n_views, generator=None):
    """[n_views,5] int32 (top, left, h, w, flags): view 0 = the base view, views 1.. = RandomResizedCrop + p=0.5 flip
    (data/datautils.py:120-121,151-157)."""
    rows = [center_box(height, width)]
    for _ in range(n_views - 1):
        i, j, h, w = random_resized_crop_box(height, width, generator=generator)
        flip = int(torch.rand(1, generator=generator).item() < 0.5)
        rows.append((i, j, h, w, flip))
    return torch.tensor(rows, dtype=torch.int32)


def make_views(image_u8_hwc, boxes, size=224, mean=CLIP_MEAN, std=CLIP_STD, out=None, precision=None):
    """image_u8_hwc: CUDA uint8 [H,W,3]; boxes: int32 [N,5] (host or device).  Returns the normalised
    fp32 batch [N,3,size,size] on the image's device, enqueued on the current stream."""
    if not image_u8_hwc.is_cuda or image_u8_hwc.dtype != torch.uint8 or image_u8_hwc.dim() != 3 or image_u8_hwc.shape[2] != 3:
        raise ValueError("make_views expects a CUDA uint8 [H,W,3] image")
    lib = _lib.load(precision)
    img = image_u8_hwc.contiguous()
    H, W = int(img.shape[0]), int(img.shape[1])
    boxes = boxes.to(device=img.device, dtype=torch.int32, non_blocking=True).contiguous()
    n = int(boxes.shape[0])
    if out is None:
        out = torch.empty(n, 3, size, size, dtype=torch.float32, device=img.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    ws_bytes = int(lib.ttl_make_views_workspace_bytes(H, W, n, size))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=img.device)   # stream-ordered via the caching allocator
    stream = torch.cuda.current_stream(img.device).cuda_stream
    _lib.check(lib.ttl_make_views(img.data_ptr(), H, W, boxes.data_ptr(), n, size, m, s, out.data_ptr(), ws.data_ptr(),
                                  ws_bytes, stream), lib)
    return out


def item_generator(seed, index):
    """The RNG of test item ``index``: a function of (seed, GLOBAL index) only, so the crop boxes an image gets do
    not depend on how many ranks share the dataset or on which rank runs it.  The reference draws from
    one global stream in dataset order (data/datautils.py:141-157 under ttl.py:321); that order is not reproducible
    once images are sharded, an index-keyed stream is."""
    g = torch.Generator()
    g.manual_seed((int(seed) * 0x9E3779B1 + int(index) * 0x85EBCA77 + 0x165667B1) % (1 << 63))
    return g


class GpuAugMixAugmenter:
    """Callable with the reference augmenter's role (data/datautils.py:141-157) for decoded images:
    ``views = aug(image_u8_hwc, index)`` -> [n_views,3,S,S] on the GPU (view 0 = the un-augmented view).
    With ``seed`` set, item ``index`` draws its boxes from ``item_generator(seed, index)`` (world-size invariant);
    without, from ``generator`` / the global torch RNG in call order like the reference's host pipeline."""

    def __init__(self, n_views=63, size=224, generator=None, precision=None, seed=None):
        self.n_views, self.size, self.generator, self.precision, self.seed = n_views, size, generator, precision, seed

    def boxes(self, height, width, index=None):
        g = item_generator(self.seed, index) if (self.seed is not None and index is not None) else self.generator
        return draw_boxes(height, width, self.n_views + 1, g)

    def __call__(self, image_u8_hwc, index=None):
        H, W = int(image_u8_hwc.shape[0]), int(image_u8_hwc.shape[1])
        return make_views(image_u8_hwc, self.boxes(H, W, index), self.size, precision=self.precision)
