"""CPU diagnostic (oracle-side, not on the product path): what does folding LayerNorm into the consuming GEMM cost in logit error?

Fold:  LN(h) W^T + b  =  rs * (h W'^T - mu * colsum(W')) + (b + W beta),   W' = gamma (.) W
so the 16-bit operand is the RAW residual stream h (not its normalised value) and the rounded weight is gamma (.) W; the row statistics
stay fp32.  Compared with the build's rounding points (normalise in fp32, round, multiply) on the reference-generated fixtures.

    python tools/ln_fold_points.py b16_n8_k10 b16_n8_k10_outliers [b16_n64_k200_ent0 ...]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"), os.path.join(ROOT, "tests")]
from oracle import ttl_oracle as O
from helpers import load_case, max_rel


class Folded(O.VitOracle):
    fold_ln1 = True          # in layers without adapters (the adapters' down-projection needs the normalised x1)
    fold_ln2 = True

    def _folded(self, h, gname, bname, wname, biasname, i):
        r, c = self.r, self.cfg
        g, be = self._lw(i, gname), self._lw(i, bname)
        W, b = self._lw(i, wname), self._lw(i, biasname)
        mu = h.mean(-1, keepdims=True, dtype=np.float64)
        var = ((h.astype(np.float64)) ** 2).mean(-1, keepdims=True) - mu ** 2          # one-pass statistics from sum / sum of squares
        rs = (1.0 / np.sqrt(var + c.ln_eps)).astype(np.float32)
        mu = mu.astype(np.float32)
        Wp = r((W * g[None, :]).astype(np.float32))
        cs = Wp.sum(-1, dtype=np.float32)
        bp = (b + W @ be).astype(np.float32)
        return (rs * (r(h) @ Wp.T - mu * cs[None, None, :]) + bp).astype(np.float32)

    def layer_forward(self, i, h, save):
        c, r = self.cfg, self.r
        if self.trained(i) and save is not None:
            return super().layer_forward(i, h, save)
        N, T, D = h.shape
        Hh, dh = c.heads, c.head_dim
        if self.fold_ln1 and not self.trained(i):
            qkv = {pj: r(self._folded(h, "layer_norm1.weight", "layer_norm1.bias", f"self_attn.{pj}.weight", f"self_attn.{pj}.bias", i))
                   for pj in ("q_proj", "k_proj", "v_proj")}
        else:
            x1 = r(O.layer_norm(h, self._lw(i, "layer_norm1.weight"), self._lw(i, "layer_norm1.bias"), c.ln_eps)[0])
            qkv = {}
            s = np.float32(c.scaling)
            for pj in ("q_proj", "k_proj", "v_proj"):
                y = x1 @ r(self._lw(i, f"self_attn.{pj}.weight")).T + self._lw(i, f"self_attn.{pj}.bias")
                if pj in self.targets(i):
                    y = y + r(s * (x1 @ r(self._lora(i, pj, "A")).T)) @ r(self._lora(i, pj, "B")).T
                qkv[pj] = r(y.astype(np.float32))
        q, k, v = (qkv[p].reshape(N, T, Hh, dh).transpose(0, 2, 1, 3) for p in ("q_proj", "k_proj", "v_proj"))
        sc = (q @ k.transpose(0, 1, 3, 2)) * np.float32(dh ** -0.5)
        e = np.exp(sc - sc.max(-1, keepdims=True))
        o = r(e / e.sum(-1, keepdims=True)) @ v
        o = r(o.transpose(0, 2, 1, 3).reshape(N, T, D).astype(np.float32))
        hm = (h + o @ r(self._lw(i, "self_attn.out_proj.weight")).T + self._lw(i, "self_attn.out_proj.bias")).astype(np.float32)
        if self.fold_ln2:
            u = self._folded(hm, "layer_norm2.weight", "layer_norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", i)
        else:
            x2 = r(O.layer_norm(hm, self._lw(i, "layer_norm2.weight"), self._lw(i, "layer_norm2.bias"), c.ln_eps)[0])
            u = (x2 @ r(self._lw(i, "mlp.fc1.weight")).T + self._lw(i, "mlp.fc1.bias")).astype(np.float32)
        gq = r(O.quick_gelu(u).astype(np.float32))
        return (hm + gq @ r(self._lw(i, "mlp.fc2.weight")).T + self._lw(i, "mlp.fc2.bias")).astype(np.float32)


for name in sys.argv[1:] or ["b16_n8_k10"]:
    g, cfg, W, x, lora0, tf = load_case(name)
    for prec in ("fp16", "bf16"):
        t0 = time.time()
        base = O.VitOracle(cfg, W, lora0, prec)
        zb = base.logits(base.forward(x), tf)
        out = [f"{name} {prec}: build's rounding points {max_rel(zb, g['logits0']):.2e}"]
        for f1, f2, tag in ((True, True, "LN1 (frozen layers) + LN2 folded"), (False, True, "LN2 folded"), (True, False, "LN1 folded")):
            net = Folded(cfg, W, lora0, prec)
            net.fold_ln1, net.fold_ln2 = f1, f2
            z = net.logits(net.forward(x), tf)
            out.append(f"{tag} {max_rel(z, g['logits0']):.2e}")
        print("; ".join(out) + f"   ({time.time() - t0:.0f} s)", flush=True)
