"""Host-side mirror of the reference's deyo.py for the path TTL uses: ``DeYO`` (deyo.py:17-82),
``softmax_entropy`` (:85-90) and ``forward_and_adapt_sar`` (:92-196).

On a ``ttl_amd.custom_clip.ClipTestTimeTuning`` model the step is one enqueue of HIP kernels:
forward -> fused entropy / selection / weighted loss + analytic dlogits -> truncated LoRA
backward -> fused AdamW on the flat LoRA buffer.  The optimizer / scaler objects the reference
passes in stay consistent: hyper-parameters are read from ``optimizer.param_groups``, Adam state
lives in ``optimizer.state`` (so ``optimizer.load_state_dict(optim_state)`` at ttl.py:344 resets it
exactly like in the reference); the GradScaler contract (scale the loss, unscale the gradients, skip the WHOLE step
and halve the scale on inf/nan, double it after growth_interval clean steps, keep the scale across images) is kept by the
context on the device (include/ttl_hip.h, ttl_optimizer_step) with the hyper-parameters of the scaler object passed in.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


def softmax_entropy(x: torch.Tensor) -> torch.Tensor:
    """deyo.py:85-90."""
    return -(x.softmax(1) * x.log_softmax(1)).sum(1)


def _adam_hparams(optimizer, model):
    """Check that ``optimizer`` is the AdamW of ttl.py:218 over this model's LoRA tensors and
    return its hyper-parameters."""
    params = model.trainable_lora_parameters()
    got = [p for g in optimizer.param_groups for p in g["params"]]
    if len(got) != len(params) or any(a is not b for a, b in zip(got, params)):
        raise ValueError("optimizer must hold exactly the LoRA parameters of layer_range in the order of ttl.py:195-213")
    g0 = optimizer.param_groups[0]
    for g in optimizer.param_groups:
        for k in ("lr", "betas", "eps", "weight_decay"):
            if g[k] != g0[k]:
                raise ValueError("per-group hyper-parameters differ; the fused step needs one setting")
    if not isinstance(optimizer, torch.optim.AdamW) or g0.get("amsgrad", False) or g0.get("maximize", False):
        raise ValueError("fused step implements torch.optim.AdamW (amsgrad=False) only")
    return params, g0["lr"], tuple(g0["betas"]), g0["eps"], g0["weight_decay"]


def _adam_state(optimizer, model, params):
    """Bind optimizer.state to the model's flat exp_avg / exp_avg_sq buffers; a cleared state
    (after load_state_dict of the empty snapshot, ttl.py:344) zeroes them.  -> step count so far."""
    m, v = model._opt_m, model._opt_v
    st = optimizer.state
    p0 = params[0]
    fresh = (p0 not in st) or ("exp_avg" not in st[p0]) or (st[p0]["exp_avg"].data_ptr() != m.data_ptr())
    if fresh:
        m.zero_()
        v.zero_()
        off = 0
        for p in params:
            n = p.numel()
            st[p] = {"step": torch.tensor(0.0), "exp_avg": m[off:off + n].view(p.shape),
                     "exp_avg_sq": v[off:off + n].view(p.shape)}
            off += n
        return 0
    return int(st[p0]["step"].item())


def _scaled_engine(eng):
    """The context whose backward carries the loss scale: the text tower's in text mode, else the engine itself."""
    return getattr(eng, "txt", eng)


def forward_and_adapt_sar(x, iter_, model, args, optimizer, scaler, deyo_margin, margin, targets=None, flag=True,
                          group=None):
    """deyo.py:92-196 on the HIP path.  Returns (outputs, backward, final_backward)."""
    if targets is not None:
        raise NotImplementedError("targets / pseudo-label accounting is not part of the TTL hot path")
    if getattr(args, "reweight_plpd", 0):
        raise NotImplementedError("reweight_plpd: the term is commented out in the reference (deyo.py:176)")
    eng = model._ensure_engine()
    _scaled_engine(eng).bind_scaler(scaler)
    if not flag:
        return eng.forward(x, save=False)
    params, lr, betas, eps, wd = _adam_hparams(optimizer, model)
    step = _adam_state(optimizer, model, params)
    eng._pending_gen = None          # (this path drives the context directly: nothing is owed to an autograd node)
    outputs = eng.forward(x, save=True)                                          # deyo.py:97
    mode = _lib.TTL_SEL_TOPK if getattr(args, "filter_ent", 0) else _lib.TTL_SEL_LE_THRESH
    reweight = float(getattr(args, "reweight_ent", 1))
    backward = None
    if getattr(args, "filter_plpd", 0):
        # PLPD filter (deyo.py:115-151): second forward on destroyed views, keep the views whose confidence in the predicted class
        # drops by more than plpd_threshold.  Views, keep mask and the loss over the survivors are HIP launches (csrc/plpd.hip);
        # the host only draws the permutations, from torch's CPU generator exactly as the reference does
        L1 = eng.entropy_select_loss(outputs, mode, rho=args.selection_p, thresh=math.log(1000), margin=margin,
                                     reweight=reweight)
        backward = int(L1["n"].item())                                           # (the reference's own host sync: len(entropys))
        if backward == 0:
            return outputs, 0, 0                                                 # deyo.py:110-113
        spec = plpd_spec(args)
        perm = draw_plpd_perms(spec, 1, backward, x.shape[-1], outputs.device)
        x_prime = eng.plpd_views(x, L1["idx"], L1["n"], backward, spec, perm)    # deyo.py:116-134
        outputs_prime = model._aux_engine().forward(x_prime, save=False)         # deyo.py:135
        keep, _ = eng.plpd_keep(outputs, outputs_prime, L1["idx"], L1["n"], backward, spec["threshold"])     # deyo.py:137-146
        L = eng.entropy_select_loss(outputs, mode, rho=args.selection_p, thresh=math.log(1000), margin=margin,
                                    reweight=reweight, keep=keep)
    else:
        L = eng.entropy_select_loss(outputs, mode, rho=args.selection_p, thresh=math.log(1000), margin=margin,
                                    reweight=reweight)                           # deyo.py:102-108,159-181
    eng.backward(L["dlogits"], selection=L)                                      # deyo.py:185-186 (top-k: on the selected views only)
    for p, gslice in zip(params, _grad_views(eng, params)):
        p.grad = gslice
    # scaler.step(optimizer); scaler.update()  (deyo.py:187-188): the context's GradScaler state decides — the whole
    # step or none of it on inf/nan gradients, dynamic loss scale in the fp16 build (the state lives on the device;
    # the torch scaler object only supplies its hyper-parameters, see TTLEngine.bind_scaler)
    eng.optimizer_step(model._flat, eng.grads, model._opt_m, model._opt_v, step + 1, lr, betas, eps, wd, n_selected=L["n"])
    n = int(L["n"].item())                                                       # the host sync of the step
    if n and _scaled_engine(eng).step_was_taken():                               # taken (not skipped on inf/nan)
        for p in params:
            optimizer.state[p]["step"] += 1
    return outputs, (n if backward is None else backward), n


def plpd_spec(args):
    """The PLPD arguments of the reference's CLI (ttl.py argparse: --aug_type, --plpd_threshold, --patch_len, --occlusion_size,
    --row_start, --column_start) as the dict TTLEngine.plpd_struct takes."""
    return dict(aug_type=getattr(args, "aug_type", "patch"), threshold=float(getattr(args, "plpd_threshold", 0.2)),
                patch_len=int(getattr(args, "patch_len", 6)), occlusion_size=int(getattr(args, "occlusion_size", 0) or 0),
                row_start=int(getattr(args, "row_start", 0) or 0), column_start=int(getattr(args, "column_start", 0) or 0))


def plpd_perm_shape(spec, n_updates, n_candidates, size):
    """Shape of the int32 permutation tensor of ``n_updates`` PLPD steps (None: 'occ' draws nothing)."""
    if spec["aug_type"] == "patch":
        return (n_updates, n_candidates, spec["patch_len"] ** 2)
    if spec["aug_type"] == "pixel":
        return (n_updates, size * size)
    return None


def draw_plpd_perms(spec, n_updates, n_candidates, size, device, out=None):
    """The random permutations of ``n_updates`` PLPD steps, drawn from torch's CPU generator in the reference's call order —
    'patch': torch.argsort(torch.rand(B, patch_len**2), dim=-1) per step (deyo.py:127), 'pixel': torch.randperm(S*S) per step
    (deyo.py:133) — as ONE int32 tensor [n_updates, ...]; 'occ' draws nothing (None).  ``out``: a host int32 tensor of
    ``plpd_perm_shape`` (the pipeline's pinned staging buffer) to draw into; otherwise the result is moved to ``device``.
    The draws are the reference's (torch.randperm fills an int32 tensor with the SAME permutation it would fill an int64 one with:
    ATen's randperm_cpu draws `generator->random() % (n - i)` whatever the dtype; tests/test_host_logic_cpu.py pins it) but they are
    made on ONE intra-op thread: every draw is a sequential algorithm, and on a host whose CPU quota is far below its core count (the
    GPU boxes: 256 logical CPUs, quota 16, torch defaults to 128 threads) each OpenMP region around it costs milliseconds —
    10.4 ms per image for the 50 176-entry 'pixel' permutation before, 0.2 ms now (profiles/r06_experiments.txt r06d)."""
    shape = plpd_perm_shape(spec, n_updates, n_candidates, size)
    if shape is None:
        return None
    if out is not None and (tuple(out.shape) != shape or out.dtype != torch.int32 or out.is_cuda or not out.is_contiguous()):
        raise ValueError(f"out must be a contiguous host int32 tensor of shape {shape}")
    buf = out if out is not None else torch.empty(shape, dtype=torch.int32)
    nt = torch.get_num_threads()
    try:
        if nt > 1:
            torch.set_num_threads(1)
        for j in range(n_updates):
            if spec["aug_type"] == "patch":
                buf[j].copy_(torch.argsort(torch.rand(n_candidates, shape[2]), dim=-1))
            else:
                torch.randperm(shape[1], dtype=torch.int32, out=buf[j])
    finally:
        if nt > 1:
            torch.set_num_threads(nt)
    return buf if out is not None else buf.to(device, non_blocking=True)


def plpd_candidates(args, n_views, n_classes):
    """How many views the FIRST selection stage yields, when the host can know it without looking at the logits (the row count of
    the reference's torch.rand(B, P)): int(N * selection_p) in top-rho mode (deyo.py:105); N in threshold mode, where
    H <= ln K <= ln 1000 holds for every view as long as K <= 1000 (deyo.py:107).  None: data-dependent -> step-wise path."""
    if getattr(args, "filter_ent", 0):
        return int(n_views * args.selection_p)
    return n_views if n_classes <= 1000 else None


def plpd_views(x_prime, args):
    """The view-destroying transform of deyo.py:118-134 as host-side torch ops — the restatement the tests hold the HIP kernels
    (csrc/plpd.hip, TTLEngine.plpd_views) against, and what autograd-formulation callers may use; NOT on the product path.
    'patch': resize to a multiple of patch_len, permute the patch_len^2 patches of every view with
    torch.argsort(torch.rand(B, P)) on the CPU generator (the reference's RNG call), resize back;
    'pixel': one random pixel permutation shared by the batch; 'occ': fill a window with the view mean.
    torchvision.transforms.Resize on a tensor == bilinear F.interpolate with antialias (torchvision >= 0.17)."""
    aug = getattr(args, "aug_type", "patch")
    S = x_prime.shape[-1]
    if aug == 'occ':
        mean = x_prime.view(x_prime.shape[0], x_prime.shape[1], -1).mean(dim=2)[:, :, None, None]
        r0, c0, sz = args.row_start, args.column_start, args.occlusion_size
        x_prime = x_prime.clone()
        x_prime[:, :, r0:r0 + sz, c0:c0 + sz] = mean.expand(-1, -1, sz, sz)
        return x_prime
    if aug == 'patch':
        pl = args.patch_len
        St = (S // pl) * pl
        xp = F.interpolate(x_prime, size=(St, St), mode="bilinear", align_corners=False, antialias=True)
        B, Cc = xp.shape[:2]
        h = St // pl
        xp = xp.view(B, Cc, pl, h, pl, h).permute(0, 2, 4, 1, 3, 5).reshape(B, pl * pl, Cc, h, h)   # b (ps1 ps2) c h w
        perm = torch.argsort(torch.rand(B, pl * pl), dim=-1).to(xp.device)
        xp = xp[torch.arange(B, device=xp.device).unsqueeze(-1), perm]
        xp = xp.view(B, pl, pl, Cc, h, h).permute(0, 3, 1, 4, 2, 5).reshape(B, Cc, St, St)             # b c (ps1 h) (ps2 w)
        return F.interpolate(xp, size=(S, S), mode="bilinear", align_corners=False, antialias=True).contiguous()
    if aug == 'pixel':
        B, Cc = x_prime.shape[:2]
        xp = x_prime.reshape(B, Cc, S * S)
        xp = xp[:, :, torch.randperm(S * S).to(xp.device)]
        return xp.reshape(B, Cc, S, S).contiguous()
    raise ValueError(f"unknown aug_type {aug!r}")


def _grad_views(eng, params):
    off = 0
    for p in params:
        yield eng.grads[off:off + p.numel()].view(p.shape)
        off += p.numel()


class DeYO(nn.Module):
    """deyo.py:17-82."""

    def __init__(self, model, args, optimizer, scaler, steps=1, episodic=False, deyo_margin=0.5 * math.log(1000),
                 margin_e0=0.4 * math.log(1000)):
        super().__init__()
        self.model = model
        self.optimizer = optimizer
        self.scaler = scaler
        self.args = args
        self.steps = steps
        self.episodic = episodic
        self.deyo_margin = deyo_margin
        self.margin_e0 = margin_e0

    def forward(self, x, iter_=None, targets=None, flag=True, group=None):
        if targets is not None:
            raise NotImplementedError("targets / pseudo-label accounting is not part of the TTL hot path")
        outputs = backward = final_backward = None
        for _ in range(self.steps):
            r = forward_and_adapt_sar(x, iter_, self.model, self.args, self.optimizer, self.scaler, self.deyo_margin,
                                      self.margin_e0, targets, flag, group)
            if flag:
                outputs, backward, final_backward = r
            else:
                outputs = r
        return (outputs, backward, final_backward) if flag else outputs
