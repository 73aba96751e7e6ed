"""ORACLE — test infrastructure only.  NOT the product path.

The per-sample hot path restated in torch fp32 on the CPU (all host cores through torch's intra-op thread pool): the second
checker beside oracle/ttl_oracle.py (numpy) and the implementation bench.py's ``cpu_baseline`` leg times — SURVEY.md §8(d) asks
for "the reference's CPU path (torch fp32, all physical cores, full 64 views)", and the reference itself cannot travel to the
GPU box.  Same software stack as the reference's CPU path: torch matmuls / softmax / layer_norm, torch autograd for the
backward, torch.optim.AdamW; the model code is restated from the un-vendored packages the reference calls into:
  * transformers CLIP ViT: modeling_clip.py:202-218 (embeddings), :259-277 (attention, eager), :346-350 (quick_gelu MLP),
    :362-383 (pre-LN layer), :641-651 (pre_layrnorm / post_layernorm on the CLS token), :744-751 (visual_projection)
  * peft (<0.10) LoRA Linear: y = xW^T + b + (alpha/r)·B(A(x)), dropout inactive in eval (ttl.py:312)
and the loop from the reference: clip/custom_clip.py:665-694 (forward: normalise, logit_scale), deyo.py:85-90,103-108,175-188
(entropy, selection, weighted loss, backward, step), ttl.py:50-61,78-84,87-108 (TPT branch, tta_steps), ttl.py:338-352 (reset,
adapt, 1-view inference).  Unlike the reference the K class-text features are an input (cached per dataset, Q12); bench.py
reports the text tower's share separately.

Parity status: pinned.  tests/test_oracle_golden.py::test_torch_restatement_vs_reference_goldens checks logits0, the
selection list, the 12 gradients, the post-step adapters and logits1 against the reference-generated fixtures
(tests/golden/*.npz, written by the unmodified reference, tests/golden/make_golden.py).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LN1000 = math.log(1000.0)      # deyo.py:107
TARGETS = ("q_proj", "k_proj", "v_proj", "out_proj")


def physical_cores():
    """Physical cores this process may run on: distinct (package, core id) pairs of the allowed CPUs (SMT siblings count once)."""
    import os
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:      # pragma: no cover
        allowed = set(range(os.cpu_count() or 1))
    cores = set()
    for cpu in allowed:
        try:
            base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
            cores.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            cores.add(("?", str(cpu)))
    return max(len(cores), 1)


def cgroup_cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unreadable.
    More threads than this are throttled, not run: on the GPU boxes 256 logical CPUs are visible behind a quota of 16."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else max(int(int(q) / int(p)), 1)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else max(int(q / p), 1)
    except (OSError, ValueError):
        return None


def usable_cores():
    """Threads worth starting: the physical cores this process may run on, capped by the container's CPU quota."""
    q = cgroup_cpu_quota()
    return min(physical_cores(), q) if q else physical_cores()


class TorchTower:
    """Frozen CLIP ViT weights as torch tensors (converted once; the timed episodes only compute)."""

    def __init__(self, cfg, W):
        self.cfg = cfg
        t = lambda k: torch.from_numpy(np.ascontiguousarray(W[k], dtype=np.float32))
        self.patch = t("vision_model.embeddings.patch_embedding.weight").reshape(cfg.width, -1)
        self.cls = t("vision_model.embeddings.class_embedding")
        self.pos = t("vision_model.embeddings.position_embedding.weight")
        self.pre = (t("vision_model.pre_layrnorm.weight"), t("vision_model.pre_layrnorm.bias"))
        self.post = (t("vision_model.post_layernorm.weight"), t("vision_model.post_layernorm.bias"))
        self.proj = t("visual_projection.weight")
        self.scale = float(np.exp(W["logit_scale"]))
        self.layers = []
        for i in range(cfg.layers):
            b = f"vision_model.encoder.layers.{i}."
            self.layers.append({k: t(b + k) for k in (
                "layer_norm1.weight", "layer_norm1.bias", "layer_norm2.weight", "layer_norm2.bias",
                "self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight", "self_attn.k_proj.bias",
                "self_attn.v_proj.weight", "self_attn.v_proj.bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")})

    def targets(self, i):
        c = self.cfg
        return tuple(getattr(c, "lora_targets", ("q_proj", "v_proj"))) if c.layer_lo <= i <= c.layer_hi else ()

    def linear(self, i, name, x, lora):
        L = self.layers[i]
        y = F.linear(x, L[f"self_attn.{name}.weight"], L[f"self_attn.{name}.bias"])
        if name in self.targets(i):
            b = f"vision_model.encoder.layers.{i}.self_attn.{name}."
            y = y + self.cfg.scaling * F.linear(F.linear(x, lora[b + "lora_A.default.weight"]), lora[b + "lora_B.default.weight"])
        return y

    def forward(self, x, lora):
        """x [N,3,S,S] -> image features [N,E] (un-normalised)."""
        c = self.cfg
        N, P, G = x.shape[0], c.patch_size, c.grid
        pat = x.reshape(N, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(N, G * G, 3 * P * P)
        h = torch.cat([self.cls.expand(N, 1, -1), pat @ self.patch.T], 1) + self.pos
        h = F.layer_norm(h, (c.width,), *self.pre, c.ln_eps)
        Hh, dh = c.heads, c.head_dim
        T = h.shape[1]
        for i, L in enumerate(self.layers):
            x1 = F.layer_norm(h, (c.width,), L["layer_norm1.weight"], L["layer_norm1.bias"], c.ln_eps)
            q, k, v = (self.linear(i, pj, x1, lora).reshape(N, T, Hh, dh).transpose(1, 2) for pj in ("q_proj", "k_proj", "v_proj"))
            att = torch.softmax((q * dh ** -0.5) @ k.transpose(-1, -2), -1) @ v        # modeling_clip.py:259-277 (eager)
            h = h + self.linear(i, "out_proj", att.transpose(1, 2).reshape(N, T, c.width), lora)
            x2 = F.layer_norm(h, (c.width,), L["layer_norm2.weight"], L["layer_norm2.bias"], c.ln_eps)
            u = F.linear(x2, L["mlp.fc1.weight"], L["mlp.fc1.bias"])
            h = h + F.linear(u * torch.sigmoid(1.702 * u), L["mlp.fc2.weight"], L["mlp.fc2.bias"])
        y = F.layer_norm(h[:, 0, :], (c.width,), *self.post, c.ln_eps)
        return y @ self.proj.T

    def logits(self, f, tfeat):
        return self.scale * (f / f.norm(dim=-1, keepdim=True)) @ tfeat.T              # clip/custom_clip.py:679-687


def text_features_forward(tcfg, Wt, ids):
    """HF CLIP text tower in torch fp32, forward only (clip/custom_clip.py:651-663 get_text_features under no_grad when the
    adapters sit on the image tower): token + position embeddings, causal pre-LN encoder, final_layer_norm, pooled at
    argmax(input_ids), text_projection.  ids [K,T] int64 -> un-normalised features [K,E].  Checked against the numpy
    TextOracle (itself pinned to reference-generated text fixtures) in tests/test_oracle_text_golden.py."""
    t = lambda k: torch.from_numpy(np.ascontiguousarray(Wt[k], dtype=np.float32))
    ids = torch.as_tensor(np.asarray(ids), dtype=torch.int64)
    K, T = ids.shape
    D, Hh = tcfg.width, tcfg.heads
    dh = D // Hh
    with torch.no_grad():
        h = t("text_model.embeddings.token_embedding.weight")[ids] + t("text_model.embeddings.position_embedding.weight")[:T]
        mask = torch.full((T, T), float("-inf")).triu(1)
        for i in range(tcfg.layers):
            b = f"text_model.encoder.layers.{i}."
            x1 = F.layer_norm(h, (D,), t(b + "layer_norm1.weight"), t(b + "layer_norm1.bias"), tcfg.ln_eps)
            q, k, v = (F.linear(x1, t(b + f"self_attn.{pj}.weight"), t(b + f"self_attn.{pj}.bias")).reshape(K, T, Hh, dh).transpose(1, 2)
                       for pj in ("q_proj", "k_proj", "v_proj"))
            att = torch.softmax((q * dh ** -0.5) @ k.transpose(-1, -2) + mask, -1) @ v
            h = h + F.linear(att.transpose(1, 2).reshape(K, T, D), t(b + "self_attn.out_proj.weight"), t(b + "self_attn.out_proj.bias"))
            x2 = F.layer_norm(h, (D,), t(b + "layer_norm2.weight"), t(b + "layer_norm2.bias"), tcfg.ln_eps)
            u = F.linear(x2, t(b + "mlp.fc1.weight"), t(b + "mlp.fc1.bias"))
            h = h + F.linear(u * torch.sigmoid(1.702 * u), t(b + "mlp.fc2.weight"), t(b + "mlp.fc2.bias"))
        pooled = h[torch.arange(K), ids.argmax(-1)]
        y = F.layer_norm(pooled, (D,), t("text_model.final_layer_norm.weight"), t("text_model.final_layer_norm.bias"), tcfg.ln_eps)
        return y @ t("text_projection.weight").T


def softmax_entropy(z):                                                                # deyo.py:85-90
    return -(z.softmax(1) * z.log_softmax(1)).sum(1)


def avg_entropy(outputs):                                                              # ttl.py:56-61
    logits = outputs - outputs.logsumexp(dim=-1, keepdim=True)
    avg = logits.logsumexp(dim=0) - math.log(logits.shape[0])
    avg = torch.clamp(avg, min=torch.finfo(avg.dtype).min)
    return -(avg * torch.exp(avg)).sum(dim=-1)


def trainable_names(cfg):
    tg = [t for t in TARGETS if t in getattr(cfg, "lora_targets", ("q_proj", "v_proj"))]
    return [f"vision_model.encoder.layers.{i}.self_attn.{pj}.lora_{ab}.default.weight"
            for i in range(cfg.layer_lo, cfg.layer_hi + 1) for pj in tg for ab in ("A", "B")]


def episode(tower, lora0, x, tfeat, *, objective="deyo", mode="le_thresh", rho=0.1, margin=0.4, reweight=1.0, n_updates=1,
            lr=5e-3, trace=None):
    """One test image (ttl.py:338-352): reset -> n_updates x [N-view forward, loss, autograd backward, AdamW] -> adapted
    1-view inference.  ``tower``: TorchTower; x, tfeat: torch fp32 tensors.  -> dict(logits0, logits1, lora, idx)."""
    cfg = tower.cfg
    names = trainable_names(cfg)
    lora = {k: torch.from_numpy(np.array(v, dtype=np.float32, copy=True)) for k, v in lora0.items()}      # LoRA_reset
    params = [lora[k].requires_grad_(True) for k in names]
    opt = torch.optim.AdamW(params, lr=lr, weight_decay=1e-2)                           # ttl.py:218 (torch defaults: wd 1e-2)
    logits0, idx0, tpt_idx = None, None, None
    for _ in range(n_updates):
        z = tower.logits(tower.forward(x, lora), tfeat)
        if logits0 is None:
            logits0 = z.detach().clone()
        if objective == "deyo":
            H = softmax_entropy(z)
            if mode == "topk":
                idx = torch.argsort(H.detach(), stable=True)[:int(z.shape[0] * rho)]                  # deyo.py:105
            else:
                idx = torch.where(H.detach() <= LN1000)[0]                                            # deyo.py:107
            if idx0 is None:
                idx0 = idx.clone()
            if idx.numel() == 0:                                                                      # deyo.py:110-113
                continue
            Hs = H[idx]
            coeff = reweight * (1.0 / torch.exp(Hs.detach() - margin)) if reweight else torch.ones_like(Hs)
            loss = (Hs * coeff).mean()                                                                # deyo.py:175-181
        else:
            if tpt_idx is None:                                                                       # ttl.py:50-54, once
                tpt_idx = torch.argsort(softmax_entropy(z.detach()), stable=True)[:int(z.shape[0] * rho)]
                idx0 = tpt_idx.clone()
            loss = avg_entropy(z[tpt_idx])
        opt.zero_grad()
        loss.backward()
        if trace is not None:
            trace.append({k: lora[k].grad.detach().clone().numpy() for k in names})
        opt.step()
    with torch.no_grad():
        z1 = tower.logits(tower.forward(x[:1], lora), tfeat)
    return dict(logits0=logits0.numpy(), logits1=z1.numpy(), idx=None if idx0 is None else idx0.numpy(),
                lora={k: v.detach().numpy() for k, v in lora.items()})
