"""ORACLE — test infrastructure only.  NOT the product path.

CPU (numpy fp32) restatement of TTL's per-sample hot path, used as the checker for the
HIP path in tests/, in __graft_entry__.smoke() and as bench.py's ``cpu_baseline`` leg.
Nothing under ttl_amd/ imports this module.

Parity status: the reference has no tests / golden vectors of its own, so
this restatement is pinned against outputs of the reference itself, produced by importing
/root/reference in the build container (tests/golden/make_golden.py) and committed under
tests/golden/*.npz; tests/test_oracle_golden.py checks every function here against them.

The arithmetic of the path lives in three un-vendored third-party packages:
  * transformers (unpinned by the reference; 5.15.0 in this image) — CLIP ViT:
    modeling_clip.py:138-218 (embeddings), :259-277 (attention), :346-350 (MLP),
    :362-383 (layer), :641-651 (tower), :744-751 (projection)
  * peft (<0.10, not installed here) — LoRA Linear: y = xW^T + b + (alpha/r)·B(A(x))
  * torch — AdamW, autograd
and is called from the reference at clip/custom_clip.py:62-71, :665-694, deyo.py:92-196,
ttl.py:50-61, :70-110, :189-220.

``prec="fp32"`` is the reference semantics (CPU path of the reference is pure fp32, Q14).
``prec="bf16"`` / ``prec="fp16"`` round every matrix-multiply operand to that 16-bit type at the
points where the HIP path does while keeping fp32 accumulation; they are the
tight checkers for the two library builds, and their distance to "fp32" is the price of the
16-bit MFMA operands.
"""
import math

import numpy as np

LN1000 = math.log(1000.0)  # deyo.py:107


# ----------------------------------------------------------------------------- helpers
def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 -> fp32, round-to-nearest-even (what v_cvt_pk_bf16_f32 does)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000))
    out = r.view(np.float32).copy()
    nan = np.isnan(x)
    if nan.any():
        out[nan] = np.nan
    return out


def fp16_round(x: np.ndarray) -> np.ndarray:
    """fp32 -> IEEE half -> fp32 (round to nearest even; overflow -> inf like v_cvt_f16_f32)."""
    with np.errstate(over="ignore"):
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


def _rounder(prec):
    if prec == "fp32":
        return lambda a: a
    if prec == "bf16":
        return bf16_round
    if prec == "fp16":
        return fp16_round
    raise ValueError(prec)


def layer_norm(x, w, b, eps):
    """nn.LayerNorm over the last dim; returns (y, mean, rstd)."""
    mu = x.mean(-1, keepdims=True, dtype=np.float32)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True, dtype=np.float32)
    rstd = (1.0 / np.sqrt(var + np.float32(eps))).astype(np.float32)
    return (xc * rstd * w + b).astype(np.float32), mu, rstd


def layer_norm_bwd(dy, x, mu, rstd, w):
    """dL/dx of LayerNorm (weight/bias are frozen on this path)."""
    xh = (x - mu) * rstd
    dxh = dy * w
    m1 = dxh.mean(-1, keepdims=True, dtype=np.float32)
    m2 = (dxh * xh).mean(-1, keepdims=True, dtype=np.float32)
    return (rstd * (dxh - m1 - xh * m2)).astype(np.float32)


def quick_gelu(u):
    """ACT2FN['quick_gelu']: u * sigmoid(1.702 u) (modeling_clip.py:346-350)."""
    return u * (1.0 / (1.0 + np.exp(-1.702 * u)))


def quick_gelu_grad(u):
    s = 1.0 / (1.0 + np.exp(-1.702 * u))
    return s * (1.0 + 1.702 * u * (1.0 - s))


def log_softmax(z):
    m = z.max(-1, keepdims=True)
    e = z - m
    return e - np.log(np.exp(e).sum(-1, keepdims=True))


# --------------------------------------------------------------- loss side (a4-a7, K8)
def softmax_entropy(z):
    """deyo.py:85-90: H_i = -sum_k softmax(z)_ik log_softmax(z)_ik."""
    lp = log_softmax(z.astype(np.float32))
    return -(np.exp(lp) * lp).sum(-1).astype(np.float32)


def select_views(H, mode, n_views, rho=0.1, thresh=LN1000):
    """a5.  mode 'le_thresh': torch.where(H <= ln 1000) (deyo.py:107, default);
    mode 'topk': argsort(H)[:int(N*rho)] ascending (deyo.py:105 / ttl.py:52).
    Returns int64 indices (ascending index order for le_thresh, entropy order for topk)."""
    if mode == "le_thresh":
        return np.nonzero(H <= np.float32(thresh))[0].astype(np.int64)
    if mode == "topk":
        k = int(n_views * rho)
        return np.argsort(H, kind="stable")[:k].astype(np.int64)
    raise ValueError(mode)


def plpd_keep(z, z_prime, idx, threshold):
    """deyo.py:137-151: PLPD_i = p_i[c_i] - p'_i[c_i] with c_i = argmax p_i over the first-stage
    selection ``idx``; returns (plpd [len(idx)], bool keep mask over ALL views)."""
    p = np.exp(log_softmax(z[idx].astype(np.float32)))
    pp = np.exp(log_softmax(z_prime.astype(np.float32)))
    c = p.argmax(1)
    plpd = p[np.arange(len(idx)), c] - pp[np.arange(len(idx)), c]
    keep = np.zeros(z.shape[0], bool)
    keep[idx] = plpd > np.float32(threshold)
    return plpd.astype(np.float32), keep


def deyo_loss_and_grad(z, mode="le_thresh", rho=0.1, margin=0.4, reweight=1.0, keep=None):
    """a4+a5+a6 (deyo.py:102-113,159-181) with the analytic gradient of SURVEY appendix A.
    ``keep``: optional bool mask [N], the second-stage (PLPD) filter of deyo.py:144-151.

    Returns dict(H, idx, coeff, loss, dz).  n == 0 -> loss None, dz zeros (deyo.py:110-113).
    """
    z = z.astype(np.float32)
    N, K = z.shape
    lp = log_softmax(z)
    p = np.exp(lp)
    H = -(p * lp).sum(-1).astype(np.float32)
    idx = select_views(H, mode, N, rho)
    if keep is not None:
        idx = idx[np.asarray(keep, bool)[idx]]
    n = idx.size
    dz = np.zeros_like(z)
    if n == 0:
        return dict(H=H, idx=idx, coeff=np.zeros(0, np.float32), loss=None, dz=dz)
    Hs = H[idx]
    if reweight:
        coeff = (np.float32(reweight) * (1.0 / np.exp(Hs - np.float32(margin)))).astype(np.float32)
    else:  # deyo.py:159 — no reweighting: plain mean entropy
        coeff = np.ones_like(Hs)
    loss = np.float32((Hs * coeff).mean(dtype=np.float32))
    # dH_i/dz_ik = -p_ik (log p_ik + H_i)
    dz[idx] = -(coeff / np.float32(n))[:, None] * p[idx] * (lp[idx] + Hs[:, None])
    return dict(H=H, idx=idx, coeff=coeff, loss=loss, dz=dz.astype(np.float32))


def select_confident_samples(logits, top):
    """ttl.py:50-54 -> (logits[idx], idx)."""
    H = softmax_entropy(logits)
    idx = select_views(H, "topk", logits.shape[0], top)
    return logits[idx], idx


def avg_entropy(outputs):
    """ttl.py:56-61 (TPT objective): entropy of the view-averaged distribution."""
    lp = log_softmax(outputs.astype(np.float32))
    m = lp.max(0, keepdims=True)
    avg = (m + np.log(np.exp(lp - m).sum(0, keepdims=True)))[0] - np.float32(np.log(lp.shape[0]))
    avg = np.maximum(avg, np.finfo(np.float32).min)
    return np.float32(-(avg * np.exp(avg)).sum())


def tpt_loss_and_grad(z, idx=None, rho=0.1):
    """TPT branch of test_time_tuning (ttl.py:87-108): select once, minimise avg_entropy."""
    z = z.astype(np.float32)
    H = softmax_entropy(z)
    if idx is None:
        idx = select_views(H, "topk", z.shape[0], rho)
    n = idx.size
    dz = np.zeros_like(z)
    if n == 0:  # mean over empty selection: the reference produces nan; we refuse
        return dict(H=H, idx=idx, loss=None, dz=dz)
    zs = z[idx]
    lp = log_softmax(zs)
    m = lp.max(0, keepdims=True)
    lse = (m + np.log(np.exp(lp - m).sum(0, keepdims=True)))[0]
    avg = lse - np.float32(np.log(n))
    loss = np.float32(-(avg * np.exp(avg)).sum())
    g = -(1.0 + avg) * np.exp(avg)                 # dL/davg_k
    w = np.exp(lp - lse[None, :])                  # davg_k/dlogp_ik
    gw = g[None, :] * w
    dz[idx] = gw - np.exp(lp) * gw.sum(-1, keepdims=True)
    return dict(H=H, idx=idx, loss=loss, dz=dz.astype(np.float32))


# ----------------------------------------------------------------------- optimizer (a13)
def adamw_step(p, g, m, v, t, lr=5e-3, b1=0.9, b2=0.999, eps=1e-8, wd=1e-2):
    """torch.optim.AdamW single-tensor update, step counter t >= 1 (ttl.py:218)."""
    p = p * np.float32(1.0 - lr * wd)
    m = np.float32(b1) * m + np.float32(1.0 - b1) * g
    v = np.float32(b2) * v + np.float32(1.0 - b2) * g * g
    bc1 = 1.0 - b1 ** t
    bc2 = 1.0 - b2 ** t
    denom = np.sqrt(v) / np.float32(math.sqrt(bc2)) + np.float32(eps)
    p = p - np.float32(lr / bc1) * (m / denom)
    return p.astype(np.float32), m.astype(np.float32), v.astype(np.float32)


# ------------------------------------------------------------------------ image tower
class VitOracle:
    """HF CLIP vision tower + projection with peft-style LoRA on q_proj/v_proj.

    ``W``: fp32 dict with HF names (ttl_amd.synth.vision_weights layout).
    ``lora``: dict name -> array for layers.{i}.self_attn.{q,v}_proj.lora_{A,B}.default.weight.
    """

    tower = "vision_model"      # HF sub-module the layer names hang off
    causal = False              # text tower: causal mask (modeling_clip.py CLIPTextTransformer)

    def __init__(self, cfg, W, lora, prec="fp32"):
        self.cfg, self.W, self.lora, self.prec = cfg, W, lora, prec
        self.r = _rounder(prec)
        # rounding points of the BACKWARD by name ("dh", "du", "dhm", "do", "P", "dS", "dqkv", "dU", "w"): a diagnostic
        # (tools/fp16_grad_points.py) replaces this hook to switch single points on and off; default: the build's rounder
        self.rb = lambda name, x: self.r(x)

    # -- parameter access
    def _lw(self, i, name):
        return self.W[f"{self.tower}.encoder.layers.{i}.{name}"]

    def _lora(self, i, proj, ab):
        return self.lora[f"{self.tower}.encoder.layers.{i}.self_attn.{proj}.lora_{ab}.default.weight"]

    def trained(self, i):
        return self.cfg.layer_lo <= i <= self.cfg.layer_hi

    def targets(self, i):
        """Projections of layer i that carry an adapter (peft LoraConfig.target_modules, clip/custom_clip.py:586: q_proj and
        v_proj in the reference; k_proj / out_proj when configured)."""
        return tuple(getattr(self.cfg, "lora_targets", ("q_proj", "v_proj"))) if self.trained(i) else ()

    # -- forward
    def embed(self, x):
        """modeling_clip.py:202-218: conv patch-embed (no bias) + cls + pos."""
        c, r = self.cfg, self.r
        N, P, G = x.shape[0], c.patch_size, c.grid
        pat = x.reshape(N, 3, G, P, G, P).transpose(0, 2, 4, 1, 3, 5).reshape(N * G * G, 3 * P * P)
        wp = self.W["vision_model.embeddings.patch_embedding.weight"].reshape(c.width, -1)
        e = (r(pat) @ r(wp).T).reshape(N, G * G, c.width)
        cls = np.broadcast_to(self.W["vision_model.embeddings.class_embedding"], (N, 1, c.width))
        h = np.concatenate([cls, e], 1) + self.W["vision_model.embeddings.position_embedding.weight"][None]
        return h.astype(np.float32)

    def layer_forward(self, i, h, save):
        c, r = self.cfg, self.r
        N, T, D = h.shape
        Hh, dh = c.heads, c.head_dim
        s = np.float32(c.scaling)
        x1, mu1, rs1 = layer_norm(h, self._lw(i, "layer_norm1.weight"), self._lw(i, "layer_norm1.bias"), c.ln_eps)
        x1 = r(x1)
        qkv = {}
        U = {}
        for pj in ("q_proj", "k_proj", "v_proj"):
            y = x1 @ r(self._lw(i, f"self_attn.{pj}.weight")).T + self._lw(i, f"self_attn.{pj}.bias")
            if pj in self.targets(i):
                # peft LoRA Linear (dropout inactive in eval, ttl.py:312)
                A, B = self._lora(i, pj, "A"), self._lora(i, pj, "B")
                U[pj] = r(s * (x1 @ r(A).T))          # HIP path stores s·U in bf16
                y = y + U[pj] @ r(B).T
            qkv[pj] = r(y.astype(np.float32))
        q = qkv["q_proj"].reshape(N, T, Hh, dh).transpose(0, 2, 1, 3)
        k = qkv["k_proj"].reshape(N, T, Hh, dh).transpose(0, 2, 1, 3)
        v = qkv["v_proj"].reshape(N, T, Hh, dh).transpose(0, 2, 1, 3)
        sc = (q @ k.transpose(0, 1, 3, 2)) * np.float32(dh ** -0.5)
        if self.causal:
            sc = np.where(np.tril(np.ones((T, T), bool)), sc, np.float32(-np.inf))
        mx = sc.max(-1, keepdims=True)
        e = np.exp(sc - mx)
        den = e.sum(-1, keepdims=True)
        Pm = e / den
        o = r(Pm) @ v
        o = r(o.transpose(0, 2, 1, 3).reshape(N, T, D).astype(np.float32))
        a = o @ r(self._lw(i, "self_attn.out_proj.weight")).T + self._lw(i, "self_attn.out_proj.bias")
        if "out_proj" in self.targets(i):
            U["out_proj"] = r(s * (o @ r(self._lora(i, "out_proj", "A")).T))
            a = a + U["out_proj"] @ r(self._lora(i, "out_proj", "B")).T
        hm = (h + a).astype(np.float32)
        x2, mu2, rs2 = layer_norm(hm, self._lw(i, "layer_norm2.weight"), self._lw(i, "layer_norm2.bias"), c.ln_eps)
        x2 = r(x2)
        u = (x2 @ r(self._lw(i, "mlp.fc1.weight")).T + self._lw(i, "mlp.fc1.bias")).astype(np.float32)
        g = r(quick_gelu(u).astype(np.float32))
        ho = (hm + g @ r(self._lw(i, "mlp.fc2.weight")).T + self._lw(i, "mlp.fc2.bias")).astype(np.float32)
        if save is not None and i >= c.layer_lo:     # every layer the gradient passes through, adapters or not
            save[i] = dict(h_in=h, mu1=mu1, rs1=rs1, x1=x1, U=U, q=q, k=k, v=v,
                           lse=(mx + np.log(den))[..., 0], o=o, h_mid=hm, mu2=mu2, rs2=rs2,
                           u=r(u))
        return ho

    def forward(self, x, save=None, taps=None):
        """[N,3,S,S] -> image features f [N,E] (un-normalised).  ``save`` (dict) collects
        what the truncated backward needs; ``taps`` (dict) collects per-stage outputs."""
        c = self.cfg
        h = self.embed(x.astype(np.float32))
        if taps is not None:
            taps["embed"] = h
        h, _, _ = layer_norm(h, self.W["vision_model.pre_layrnorm.weight"],
                             self.W["vision_model.pre_layrnorm.bias"], c.ln_eps)
        if taps is not None:
            taps["pre_ln"] = h
        for i in range(c.layers):
            h = self.layer_forward(i, h, save)
            if taps is not None:
                taps[f"layer{i}"] = h
        cls = h[:, 0, :]
        y, mu, rs = layer_norm(cls, self.W["vision_model.post_layernorm.weight"],
                               self.W["vision_model.post_layernorm.bias"], c.ln_eps)
        f = (y @ self.W["visual_projection.weight"].T).astype(np.float32)
        if save is not None:
            save["head"] = dict(cls=cls, mu=mu, rs=rs, y=y, f=f)
        if taps is not None:
            taps["pooled"] = y
            taps["features"] = f
        return f

    def logits(self, f, tfeat):
        """clip/custom_clip.py:680-687."""
        fh = f / np.linalg.norm(f, axis=-1, keepdims=True)
        S = np.float32(np.exp(self.W["logit_scale"]))
        return (S * fh @ tfeat.T).astype(np.float32)

    # -- backward (truncated at the first trained layer)
    def backward(self, dz, tfeat, save):
        """dL/dlogits [N,K] -> dict of LoRA grads keyed like ``lora``."""
        c = self.cfg
        hd = save["head"]
        N = dz.shape[0]
        S = np.float32(np.exp(self.W["logit_scale"]))
        f = hd["f"]
        nrm = np.linalg.norm(f, axis=-1, keepdims=True)
        fh = f / nrm
        dfh = S * (dz @ tfeat)
        df = (dfh - fh * (fh * dfh).sum(-1, keepdims=True)) / nrm
        dy = df @ self.W["visual_projection.weight"]
        dcls = layer_norm_bwd(dy, hd["cls"], hd["mu"], hd["rs"], self.W["vision_model.post_layernorm.weight"])
        dh = np.zeros((N, c.tokens, c.width), np.float32)
        dh[:, 0, :] = dcls
        return self.backward_layers(dh, save)

    def backward_layers(self, dh, save):
        """d/d(residual stream after the last layer) [N,T,D] -> LoRA grads of the trained layers."""
        c, rb = self.cfg, self.rb
        r = lambda w: rb("w", w)                     # frozen weights / adapters as operands
        N = dh.shape[0]
        T, D, Hh, dhd = c.tokens, c.width, c.heads, c.head_dim
        s = np.float32(c.scaling)
        grads = {}
        # from the LAST layer down to layer_lo: layers above layer_hi are frozen (no adapters trained, B == 0 forever,
        # Q10) but lie on the path from the loss to the adapters below them (--layer_range need not end at the top)
        for i in range(c.layers - 1, c.layer_lo - 1, -1):
            sv = save[i]
            first = (i == c.layer_lo)
            # MLP
            dg = rb("dh", dh) @ r(self._lw(i, "mlp.fc2.weight"))
            du = rb("du", (dg * quick_gelu_grad(sv["u"])).astype(np.float32))
            dx2 = du @ r(self._lw(i, "mlp.fc1.weight"))
            dhm = dh + layer_norm_bwd(dx2, sv["h_mid"], sv["mu2"], sv["rs2"], self._lw(i, "layer_norm2.weight"))
            # attention
            base = f"{self.tower}.encoder.layers.{i}.self_attn."
            tg = self.targets(i)
            do = rb("dhm", dhm) @ r(self._lw(i, "self_attn.out_proj.weight"))
            if "out_proj" in tg:
                Ao, Bo = self._lora(i, "out_proj", "A"), self._lora(i, "out_proj", "B")
                dhm16 = rb("dhm", dhm).reshape(N * T, D)
                dUo = rb("dU", s * (dhm16 @ r(Bo)))
                do = do + (dUo @ r(Ao)).reshape(N, T, D)
                grads[base + "out_proj.lora_B.default.weight"] = (dhm16.T @ sv["U"]["out_proj"].reshape(N * T, -1)).astype(np.float32)
                grads[base + "out_proj.lora_A.default.weight"] = (dUo.T @ sv["o"].reshape(N * T, D)).astype(np.float32)
            do = rb("do", do.astype(np.float32))
            dO = do.reshape(N, T, Hh, dhd).transpose(0, 2, 1, 3)
            q, k, v = sv["q"], sv["k"], sv["v"]
            sc = (q @ k.transpose(0, 1, 3, 2)) * np.float32(dhd ** -0.5)
            if self.causal:
                sc = np.where(np.tril(np.ones((T, T), bool)), sc, np.float32(-np.inf))
            Pm = np.exp(sc - sv["lse"][..., None])
            O = sv["o"].reshape(N, T, Hh, dhd).transpose(0, 2, 1, 3)
            delta = (dO * O).sum(-1, keepdims=True)
            dV = rb("P", Pm).transpose(0, 1, 3, 2) @ dO
            dP = dO @ v.transpose(0, 1, 3, 2)
            dS = rb("dS", (Pm * (dP - delta)).astype(np.float32))
            dQ = (dS @ k) * np.float32(dhd ** -0.5)
            merge = lambda a: rb("dqkv", a.transpose(0, 2, 1, 3).reshape(N * T, D).astype(np.float32))
            dq, dv = merge(dQ), merge(dV)
            x1 = sv["x1"].reshape(N * T, D)
            dk = None
            if "k_proj" in tg or not first:        # (the first trained layer needs dK only for a k_proj adapter)
                dK = (dS.transpose(0, 1, 3, 2) @ q) * np.float32(dhd ** -0.5)
                dk = merge(dK)
            dU = {}
            for pj, dproj in (("q_proj", dq), ("k_proj", dk), ("v_proj", dv)):
                if pj not in tg:
                    continue
                A, B = self._lora(i, pj, "A"), self._lora(i, pj, "B")
                Us = sv["U"][pj].reshape(N * T, -1)            # = s·x1·A^T
                grads[base + pj + ".lora_B.default.weight"] = (dproj.T @ Us).astype(np.float32)
                dU[pj] = rb("dU", s * (dproj @ r(B)))                 # [M,r]
                grads[base + pj + ".lora_A.default.weight"] = (dU[pj].T @ x1).astype(np.float32)
            if first:
                break
            dx1 = (dq @ r(self._lw(i, "self_attn.q_proj.weight"))
                   + dk @ r(self._lw(i, "self_attn.k_proj.weight"))
                   + dv @ r(self._lw(i, "self_attn.v_proj.weight")))
            for pj in dU:
                dx1 = dx1 + dU[pj] @ r(self._lora(i, pj, "A"))
            dx1 = dx1.reshape(N, T, D)
            dh = dhm + layer_norm_bwd(dx1, sv["h_in"], sv["mu1"], sv["rs1"], self._lw(i, "layer_norm1.weight"))
        return grads


# ------------------------------------------------------------------------- text tower
class TextOracle(VitOracle):
    """HF CLIP text tower + text_projection with LoRA on q_proj/v_proj (``--lora_encoder text``,
    clip/custom_clip.py:602-607,672-678; HF modeling_clip.py CLIPTextTransformer: token + position
    embeddings, causal pre-LN encoder, final_layer_norm, pooled at argmax(input_ids) — the eot token,
    legacy eos_token_id == 2 branch — then text_projection without bias).  ``cfg``: config.TextConfig."""
    tower = "text_model"
    causal = True

    def forward(self, ids, save=None, taps=None):
        """ids [K,T] int -> text features t [K,E] (un-normalised)."""
        c = self.cfg
        ids = np.asarray(ids)
        h = (self.W["text_model.embeddings.token_embedding.weight"][ids]
             + self.W["text_model.embeddings.position_embedding.weight"][None, :ids.shape[1]]).astype(np.float32)
        if taps is not None:
            taps["embed"] = h
        for i in range(c.layers):
            h = self.layer_forward(i, h, save)
            if taps is not None:
                taps[f"layer{i}"] = h
        eot = ids.argmax(-1)
        pooled = h[np.arange(ids.shape[0]), eot, :]
        y, mu, rs = layer_norm(pooled, self.W["text_model.final_layer_norm.weight"],
                               self.W["text_model.final_layer_norm.bias"], c.ln_eps)
        t = (y @ self.W["text_projection.weight"].T).astype(np.float32)
        if save is not None:
            save["head"] = dict(cls=pooled, mu=mu, rs=rs, y=y, f=t, eot=eot)
        if taps is not None:
            taps["pooled"] = y
            taps["features"] = t
        return t

    def backward(self, dz, fh_img, save, logit_scale_exp):
        """dL/dlogits [N,K] with logits = S * fh_img @ that^T  ->  text LoRA grads."""
        c = self.cfg
        hd = save["head"]
        t = hd["f"]
        nrm = np.linalg.norm(t, axis=-1, keepdims=True)
        th = t / nrm
        dth = np.float32(logit_scale_exp) * (dz.T @ fh_img)                     # [K,E]
        dt = (dth - th * (th * dth).sum(-1, keepdims=True)) / nrm
        dy = dt @ self.W["text_projection.weight"]
        dpool = layer_norm_bwd(dy, hd["cls"], hd["mu"], hd["rs"], self.W["text_model.final_layer_norm.weight"])
        K = dz.shape[1]
        dh = np.zeros((K, c.tokens, c.width), np.float32)
        dh[np.arange(K), hd["eot"], :] = dpool
        return self.backward_layers(dh, save)


# ---------------------------------------------------------------------------- episode
def trainable_names(cfg, tower="vision_model"):
    """Order of the 12 param groups at ttl.py:195-213: per layer q.A, q.B, v.A, v.B (with k_proj / out_proj adapters
    configured: q, k, v, out)."""
    out = []
    tg = getattr(cfg, "lora_targets", ("q_proj", "v_proj"))
    for i in range(cfg.layer_lo, cfg.layer_hi + 1):
        for pj in [t for t in ("q_proj", "k_proj", "v_proj", "out_proj") if t in tg]:
            for ab in ("A", "B"):
                out.append(f"{tower}.encoder.layers.{i}.self_attn.{pj}.lora_{ab}.default.weight")
    return out


def episode_text(vcfg, tcfg, Wv, Wt, lora0, x, ids, *, prec="fp32", objective="deyo", mode="le_thresh",
                 rho=0.1, margin=0.4, reweight=1.0, n_updates=1, lr=5e-3, betas=(0.9, 0.999),
                 eps=1e-8, wd=1e-2, trace=None):
    """``--lora_encoder text`` episode (clip/custom_clip.py:665-703 with lora_encoder == 'text'):
    image features of the N views without grad (no LoRA on the image tower), text features of the K
    prompts with grad through the text-tower LoRA, same loss / AdamW / adapted inference on view 0."""
    lora = {k: v.copy() for k, v in lora0.items()}
    names = trainable_names(tcfg, "text_model")
    m = {k: np.zeros_like(lora[k]) for k in names}
    v = {k: np.zeros_like(lora[k]) for k in names}
    vis = VitOracle(vcfg, Wv, {}, prec)
    vis.trained = lambda i: False                          # the image tower carries no adapters in this mode
    f = vis.forward(x)
    fh = (f / np.linalg.norm(f, axis=-1, keepdims=True)).astype(np.float32)
    S = np.float32(np.exp(Wv["logit_scale"]))
    t_step, logits0, tpt_idx = 0, None, None
    for _ in range(n_updates):
        net = TextOracle(tcfg, Wt, lora, prec)
        save = {}
        t = net.forward(ids, save)
        th = t / np.linalg.norm(t, axis=-1, keepdims=True)
        z = (S * fh @ th.T).astype(np.float32)
        if logits0 is None:
            logits0 = z
        if objective == "deyo":
            L = deyo_loss_and_grad(z, mode, rho, margin, reweight)
        else:
            L = tpt_loss_and_grad(z, tpt_idx, rho)
            tpt_idx = L["idx"]
        rec = dict(logits=z, H=L["H"], idx=L["idx"], loss=L["loss"])
        if L["loss"] is not None:
            grads = net.backward(L["dz"], fh, save, S)
            t_step += 1
            for k in names:
                lora[k], m[k], v[k] = adamw_step(lora[k], grads[k], m[k], v[k], t_step, lr, betas[0], betas[1], eps, wd)
            rec["grads"] = grads
        if trace is not None:
            trace.append(rec)
    net = TextOracle(tcfg, Wt, lora, prec)
    t = net.forward(ids)
    th = t / np.linalg.norm(t, axis=-1, keepdims=True)
    z1 = (S * fh[:1] @ th.T).astype(np.float32)
    return dict(logits0=logits0, logits1=z1, lora=lora, image_features=fh, text_features=th.astype(np.float32))


def episode(cfg, W, lora0, x, tfeat, *, prec="fp32", objective="deyo", mode="le_thresh",
            rho=0.1, margin=0.4, reweight=1.0, n_updates=1, lr=5e-3, betas=(0.9, 0.999),
            eps=1e-8, wd=1e-2, trace=None, keep=None):
    """One test image: reset -> n_updates x [N-view forward, loss, LoRA backward, AdamW]
    -> adapted inference on view 0 (ttl.py:338-352).  ``n_updates`` is the *effective*
    number of optimizer steps (tta_steps**2 on the reference's DeYO branch, Q6).

    Returns dict(logits0 [N,K] of the first forward, logits1 [1,K] after adaptation,
    lora (adapted), and per-update records in ``trace`` if given)."""
    lora = {k: v.copy() for k, v in lora0.items()}          # LoRA_reset (custom_clip.py:202-215)
    names = trainable_names(cfg)
    m = {k: np.zeros_like(lora[k]) for k in names}           # load_state_dict(optim_state), ttl.py:344
    v = {k: np.zeros_like(lora[k]) for k in names}
    t = 0
    logits0 = None
    tpt_idx = None
    for _ in range(n_updates):
        net = VitOracle(cfg, W, lora, prec)
        save = {}
        f = net.forward(x, save)
        z = net.logits(f, tfeat)
        if logits0 is None:
            logits0 = z
        if objective == "deyo":
            L = deyo_loss_and_grad(z, mode, rho, margin, reweight, keep)   # keep: PLPD mask supplied by the caller
        else:
            L = tpt_loss_and_grad(z, tpt_idx, rho)
            tpt_idx = L["idx"]
        rec = dict(logits=z, H=L["H"], idx=L["idx"], loss=L["loss"])
        if L["loss"] is not None:
            grads = net.backward(L["dz"], tfeat, save)
            t += 1
            for k in names:
                lora[k], m[k], v[k] = adamw_step(lora[k], grads[k], m[k], v[k], t, lr,
                                                 betas[0], betas[1], eps, wd)
            rec["grads"] = grads
        if trace is not None:
            trace.append(rec)
    net = VitOracle(cfg, W, lora, prec)
    z1 = net.logits(net.forward(x[:1]), tfeat)
    return dict(logits0=logits0, logits1=z1, lora=lora)
