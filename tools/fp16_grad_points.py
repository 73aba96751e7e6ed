"""CPU diagnostic (oracle-side, not on the product path): which 16-bit rounding point of the fp16-operand build's BACKWARD
decides the LoRA-gradient error against the reference's fp32 result?

The oracle's fp16-emulating mode rounds every MFMA operand where the HIP build does.  Here the backward's rounding points are
switched by name (oracle VitOracle.rb): gradient-side operands are rounded UNDER THE LOSS SCALE (x * S -> fp16 -> / S, S = 2^10 like
GradScaler(init_scale=1000), ttl.py:222), optionally with an extra per-operand power of two, and the per-tensor max|a-b|/max|b|
against the reference-generated fixture is printed for each configuration.

    python tools/fp16_grad_points.py b16_n8_k10 [b16_n64_k1000_ent0 ...]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"), os.path.join(ROOT, "tests")]
from oracle import ttl_oracle as O
from helpers import load_case, episode_kwargs, max_rel

POINTS = ("dh", "du", "dhm", "do", "P", "dS", "dqkv", "dU", "w")


def run(name):
    g, cfg, W, x, lora0, tf = load_case(name)
    kw = episode_kwargs(g)
    names = O.trainable_names(cfg)
    t0 = time.time()
    net = O.VitOracle(cfg, W, lora0, "fp16")
    save = {}
    z = net.logits(net.forward(x, save), tf)
    if kw["objective"] == "deyo":
        out = O.deyo_loss_and_grad(z, kw["mode"], kw["rho"], kw["margin"])
    else:
        out = O.tpt_loss_and_grad(z, None, kw["rho"])
    dz, idx = out["dz"], out["idx"]
    print(f"{name}: forward {time.time() - t0:.0f} s, logits vs reference {max_rel(z, g['logits0']):.2e}, selected {len(idx)}", flush=True)
    S = np.float32(1024.0)
    f16 = O.fp16_round

    def make_rb(on, extra=None):
        extra = extra or {}

        def rb(nm, a):
            if nm not in on:
                return a
            if nm in ("w", "P"):
                return f16(a)                       # not gradient-scaled: weights, probabilities
            s = S * np.float32(extra.get(nm, 1.0))
            return (f16(a * s) / s).astype(np.float32)
        return rb

    def errs(rb):
        net.rb = rb
        gr = net.backward(dz, tf, save)
        return {k: max_rel(gr[k], g["grad/" + k]) for k in names if np.abs(g["grad/" + k]).max() > 0}

    def show(tag, e):
        worst = max(e, key=e.get)
        print(f"  {tag:44s} max {e[worst]:.2e} ({worst.split('layers.')[1].replace('.default.weight', '')})  median {np.median(list(e.values())):.2e}", flush=True)

    show("no rounding in the backward (forward fp16)", errs(make_rb(())))
    show("all points, scale 2^10", errs(make_rb(POINTS)))
    for p in POINTS:
        show(f"only {p}", errs(make_rb((p,))))
    for p in POINTS:
        show(f"all but {p}", errs(make_rb(tuple(q for q in POINTS if q != p))))
    show("all, dS with 2^12 more", errs(make_rb(POINTS, {"dS": 4096.0})))
    show("all, dS and do and dqkv with 2^8 more", errs(make_rb(POINTS, {"dS": 256.0, "do": 256.0, "dqkv": 256.0})))


for n in sys.argv[1:] or ["b16_n8_k10"]:
    run(n)
