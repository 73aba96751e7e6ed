"""Host-side mirror of the reference's ttl.py hot-path functions: ``test_time_tuning`` (ttl.py:70-110),
``select_confident_samples`` (:50-54), ``avg_entropy`` (:56-61).  Same signatures, same quirks:
the DeYO branch runs ``tta_steps`` DeYO objects of ``steps=tta_steps`` each, i.e. tta_steps**2
optimizer steps; ``--deyo_selection`` is truthy for any non-empty string (Q3).
"""
import numpy as np
import torch

from . import deyo as _deyo
from .deyo import _adam_hparams, _adam_state, _grad_views


def select_confident_samples(logits, top):
    """ttl.py:50-54."""
    batch_entropy = -(logits.softmax(1) * logits.log_softmax(1)).sum(1)
    idx = torch.argsort(batch_entropy, descending=False)[:int(batch_entropy.size()[0] * top)]
    return logits[idx], idx


def avg_entropy(outputs):
    """ttl.py:56-61."""
    logits = outputs - outputs.logsumexp(dim=-1, keepdim=True)
    avg_logits = logits.logsumexp(dim=0) - np.log(logits.shape[0])
    min_real = torch.finfo(avg_logits.dtype).min
    avg_logits = torch.clamp(avg_logits, min=min_real)
    return -(avg_logits * torch.exp(avg_logits)).sum(dim=-1)


def test_time_tuning(model, inputs, optimizer, scaler, args):
    """ttl.py:70-110.  Mutates the model's LoRA parameters in place; returns None."""
    if getattr(args, "cocoop", False):
        raise NotImplementedError("--cocoop is a dead branch in the reference (ttl.py:132-133)")
    if args.deyo_selection and args.lora_encoder != 'prompt':
        for j in range(args.tta_steps):                                            # ttl.py:78
            d = _deyo.DeYO(model, args, optimizer, scaler, steps=args.tta_steps, deyo_margin=args.deyo_margin,
                           margin_e0=args.deyo_margin_e0)                          # ttl.py:80
            d(inputs)
        return
    # TPT-style objective on the LoRA parameters (ttl.py:87-108): select once, then minimise the
    # entropy of the view-averaged prediction over the cached selection.
    eng = model._ensure_engine()
    _deyo._scaled_engine(eng).bind_scaler(scaler)
    params, lr, betas, eps, wd = _adam_hparams(optimizer, model)
    if int(inputs.shape[0] * args.selection_p) == 0:
        raise ValueError("int(n_views * selection_p) == 0: the reference averages an empty selection here (nan)")
    sel = None
    for j in range(args.tta_steps):
        step = _adam_state(optimizer, model, params)
        eng._pending_gen = None
        out = eng.forward(inputs, save=True)
        sel = eng.tpt_select_loss(out, rho=args.selection_p, idx=None if sel is None else sel["idx"],
                                  n=None if sel is None else sel["n"])
        eng.backward(sel["dlogits"], selection=sel)
        for p, g in zip(params, _grad_views(eng, params)):
            p.grad = g
        eng.optimizer_step(model._flat, eng.grads, model._opt_m, model._opt_v, step + 1, lr, betas, eps, wd)   # ttl.py:106-108
        if _deyo._scaled_engine(eng).step_was_taken():               # taken (GradScaler skips the whole step on inf/nan)
            for p in params:
                optimizer.state[p]["step"] += 1
    return


test_time_tuning.__test__ = False  # not a pytest test
