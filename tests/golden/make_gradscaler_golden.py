#!/usr/bin/env python3
"""Known-answer vectors for the GradScaler contract of the path (ttl.py:222 / deyo.py:186-188): the reference's own
objects — torch.amp.GradScaler(init_scale=1000) around torch.optim.AdamW(lr=5e-3) — driven through
scale(loss).backward() / step / update with an inf and a nan injected.  Runs in the build container (CPU GradScaler);
writes tests/golden/unit_gradscaler.npz: the gradient sequence and, after every update, params / exp_avg / exp_avg_sq /
scale / Adam step count.
    python tests/golden/make_gradscaler_golden.py
"""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
torch.manual_seed(0)
N, STEPS = 96, 9
p = torch.nn.Parameter(torch.randn(N) * 0.1)
p0 = p.detach().clone().numpy()
opt = torch.optim.AdamW([p], lr=5e-3)
scaler = torch.amp.GradScaler("cpu", init_scale=1000, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
grads = torch.randn(STEPS, N) * 1e-3
grads[2, 17] = float("inf")          # step 3: overflow -> whole step skipped, scale halves
grads[6, 40] = float("nan")          # step 7: nan -> same
rec = dict(params=[], m=[], v=[], scale=[], step=[])
for t in range(STEPS):
    opt.zero_grad()
    # a loss whose gradient w.r.t. p is grads[t]; scaler.scale multiplies it by the current scale
    loss = (p * torch.nan_to_num(grads[t], nan=0.0, posinf=0.0, neginf=0.0)).sum()
    scaler.scale(loss).backward()
    bad = ~torch.isfinite(grads[t])
    if bad.any():                      # what an overflowing fp16 backward leaves in .grad
        p.grad[bad] = grads[t][bad]
    scaler.step(opt)
    scaler.update()
    st = opt.state[p]
    rec["params"].append(p.detach().clone().numpy())
    rec["m"].append(st["exp_avg"].clone().numpy() if st else np.zeros(N, np.float32))
    rec["v"].append(st["exp_avg_sq"].clone().numpy() if st else np.zeros(N, np.float32))
    rec["scale"].append(scaler.get_scale())
    rec["step"].append(int(st["step"].item()) if st else 0)
np.savez(os.path.join(HERE, "unit_gradscaler.npz"), p0=p0, grads=grads.numpy(), lr=5e-3, init_scale=1000.0, growth_factor=2.0,
         backoff_factor=0.5, growth_interval=3, **{k: np.asarray(v) for k, v in rec.items()})
print("scale:", rec["scale"], "steps:", rec["step"])
