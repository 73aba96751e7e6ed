"""Shared helpers for the parity tests (CPU oracle side and GPU side)."""
import os

import numpy as np

from ttl_amd import synth
from ttl_amd.config import get_config

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    """-> (golden npz, cfg, W, x, lora0, tfeat) rebuilt from seeds and checked by sha256."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    cfg = get_config(str(g["arch"])).replace(rank=int(g["rank"]))
    W = synth.vision_weights(cfg, int(g["weight_seed"]))
    assert synth.checksum(W) == str(g["weights_sha256"]), "synthetic weights drifted from the fixture"
    x = synth.views(cfg, int(g["n_views"]), int(g["view_seed"]))
    assert synth.checksum([x]) == str(g["x_sha256"]), "synthetic views drifted from the fixture"
    lora0 = synth.lora_init(cfg, 0)
    for k in g.files:
        if k.startswith("lora0/"):
            lora0[k[6:]] = g[k]
    return g, cfg, W, x, lora0, g["text_features"]


def episode_kwargs(g):
    return dict(objective=str(g["objective"]), mode=str(g["mode"]), rho=float(g["rho"]),
                margin=float(g["margin"]), n_updates=int(g["n_updates"]), lr=float(g["lr"]))


def max_rel(a, b):
    """max |a-b| / max |b|  (tensor-level relative error)."""
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max()
                 / (np.abs(b).max() + 1e-30))


def check_lora_step(new, ref, grad, lr, tol, name="", gtol=None, eps=1e-8):
    """Compare post-step LoRA weights.

    The first AdamW step from zero state is sign-like, p' = p(1-lr*wd) - lr*g/(|g|+eps)
    (SURVEY Q11), so an error dg in the gradient moves p' by lr*eps*dg/(|g|+eps)^2: tiny
    where |g| >> eps, up to the full +-lr step where |g| ~ eps.  The allowed deviation is
    therefore ``tol`` (relative to the tensor's max) plus that analytic sensitivity for a
    gradient error of ``gtol`` x max|g| (default: tol), capped at the 2*lr step bound.
    With grad=None (multi-step cases) only ``tol`` applies."""
    new, ref = np.asarray(new, np.float64), np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-30
    allowed = np.full(ref.shape, tol * scale)
    if grad is not None:
        g = np.abs(np.asarray(grad, np.float64))
        dg = (tol if gtol is None else gtol) * g.max()
        allowed = allowed + np.minimum(2.0 * lr, lr * eps * dg / (g + eps) ** 2)
    err = np.abs(new - ref)
    bad = err > allowed
    assert not bad.any(), (name, int(bad.sum()), float((err - allowed).max()))
