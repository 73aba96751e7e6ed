"""Times the GPU view generator (64 views of one decoded image) and the views->episode pipeline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"))
import numpy as np
import torch
from ttl_amd import views as V

for (H, W) in [(375, 500), (1080, 1920)]:
    img = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
    g = torch.Generator().manual_seed(0)
    t0 = time.time()
    boxes = [V.draw_boxes(H, W, 64, g) for _ in range(20)]
    t_host = (time.time() - t0) / 20
    dboxes = [b.cuda() for b in boxes]
    out = torch.empty(64, 3, 224, 224, device="cuda")
    for b in dboxes[:3]:
        V.make_views(img, b, 224, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for b in dboxes:
        V.make_views(img, b, 224, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"{H}x{W}: host box sampling {t_host*1e3:.2f} ms/image, GPU make_views {e0.elapsed_time(e1)/20*1e3:.1f} us/image (64 views)")
