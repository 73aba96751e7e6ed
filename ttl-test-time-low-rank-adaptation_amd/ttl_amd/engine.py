"""Thin torch-facing wrapper around one libttl_hip context.

torch is plumbing here (device memory, streams); every computation on the hot path is a
call into the C ABI (include/ttl_hip.h).  One engine = one GPU = one process.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import VitConfig


def _targets_mask(cfg):
    from .config import targets_mask
    return targets_mask(cfg)


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensor required"
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class TTLEngine:
    """Owns the HIP context (frozen bf16 weights + activation arena) of one image tower."""

    def __init__(self, cfg: VitConfig, max_views: int, max_classes: int, device, precision: str = None, share_from=None):
        """precision: MFMA operand dtype — None = _lib.DEFAULT_PRECISION = "fp16" (the reference's autocast dtype; within 1e-3 of the
        reference's logits), "bf16" (4e-3) or "strict" (fp32 operands, test build) on request.
        share_from: an engine of the same model whose weights are loaded — this engine then reads that engine's frozen
        weight images instead of holding copies (ttl_ctx_create_shared; the reference has ONE model per process,
        ttl.py:178-179); ``load_weights`` must not be called on it and ``share_from`` must stay open while it is in use."""
        self.precision = precision = _lib.resolve_precision(precision)
        self.lib = _lib.load(precision)
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TtlError("TTLEngine needs a GPU device: the hot path has no CPU implementation "
                                "(oracle/ is a test checker, not a fallback)")
        self.max_views, self.max_classes = int(max_views), int(max_classes)
        self.n_classes = 0
        c = self._make_config(cfg)
        self._ccfg = c
        h = C.c_void_p()
        self._parent = share_from                      # keeps the owner of the shared images alive
        with torch.cuda.device(self.device):
            if share_from is not None:
                if share_from.precision != precision or share_from.device != self.device:
                    raise _lib.TtlError("share_from must be an engine of the same build on the same device")
                self._check(self.lib.ttl_ctx_create_shared(C.byref(c), share_from._h, C.byref(h)))
            else:
                self._check(self.lib.ttl_ctx_create(C.byref(c), C.byref(h)))
        self._h = h
        from .config import ordered_targets
        self.n_lora = (cfg.layer_hi - cfg.layer_lo + 1) * len(ordered_targets(cfg)) * 2 * cfg.rank * cfg.width
        self.grads = torch.zeros(self.n_lora, dtype=torch.float32, device=self.device)
        self._params = None
        self._keep = []

    def _make_config(self, cfg):
        return _lib.ttl_config(cfg.image_size, cfg.patch_size, cfg.width, cfg.heads, cfg.mlp, cfg.layers,
                               cfg.embed, cfg.rank, cfg.lora_alpha, cfg.layer_lo, cfg.layer_hi, cfg.ln_eps,
                               self.max_views, self.max_classes, _lib.TTL_TOWER_IMAGE, 0, 0, _targets_mask(cfg))

    def _check(self, rc):
        _lib.check(rc, self.lib)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        for g in getattr(self, "_graphs", []):
            self.lib.ttl_graph_destroy(g)
        self._graphs = []
        if getattr(self, "_h", None):
            self.lib.ttl_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def workspace_bytes(self):
        return int(self.lib.ttl_workspace_bytes(C.byref(self._ccfg)))

    def allocated_bytes(self):
        """Device bytes this context holds right now (ttl_ctx_allocated_bytes)."""
        return int(self.lib.ttl_ctx_allocated_bytes(self._h))

    # ------------------------------------------------------------------ setup
    def load_weights(self, state: dict):
        """state: HF vision-tower names -> fp32 numpy arrays or torch tensors (any device)."""
        with torch.cuda.device(self.device):
            for name, a in state.items():
                if name == "logit_scale":
                    continue
                if isinstance(a, torch.Tensor) and a.dtype in (torch.float16, torch.bfloat16):
                    a = a.detach().contiguous()            # half-precision checkpoints: widened exactly inside the library
                    if a.is_cuda and a.device != self.device:
                        a = a.to(self.device)
                    self._check(self.lib.ttl_load_weight_typed(self._h, name.encode(), C.c_void_p(a.data_ptr()), a.numel(),
                                                               1 if a.dtype == torch.float16 else 2))
                    continue
                if isinstance(a, torch.Tensor):
                    a = a.detach().to(torch.float32).contiguous()
                    if a.is_cuda and a.device != self.device:
                        a = a.to(self.device)
                    ptr, cnt = C.c_void_p(a.data_ptr()), a.numel()
                else:
                    a = np.ascontiguousarray(a, dtype=np.float32)
                    ptr, cnt = a.ctypes.data_as(C.c_void_p), a.size
                self._check(self.lib.ttl_load_weight(self._h, name.encode(), ptr, cnt))
            self._check(self.lib.ttl_weights_ready(self._h))

    def set_text_features(self, tfeat: torch.Tensor, logit_scale_exp: float):
        """tfeat: [K,E] unit-norm class embeddings (clip/custom_clip.py:651-663)."""
        t = tfeat.detach().to(device=self.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_set_text_features(self._h, _ptr(t), t.shape[0], float(logit_scale_exp), _stream()))
        self.n_classes = int(t.shape[0])

    def head_logits(self, feats: torch.Tensor) -> torch.Tensor:
        """exp(logit_scale) * normalize(feats) @ t_hat^T against the cached class embeddings (clip/custom_clip.py:679-686)."""
        f = feats.detach().to(device=self.device, dtype=torch.float32).contiguous()
        out = torch.empty(f.shape[0], self.n_classes, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_head_logits(self._h, _ptr(f), f.shape[0], _ptr(out), _stream()))
        return out

    def bind_lora(self, params_flat: torch.Tensor):
        assert params_flat.numel() == self.n_lora and params_flat.dtype == torch.float32
        self._params = params_flat
        self._check(self.lib.ttl_bind_lora(self._h, _ptr(params_flat), _ptr(self.grads), self.n_lora))

    # ------------------------------------------------------------------ hot path
    def forward(self, x: torch.Tensor, save: bool = False, want_features: bool = False):
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        n = x.shape[0]
        logits = torch.empty((n, self.n_classes), dtype=torch.float32, device=self.device)
        feats = torch.empty((n, self.cfg.embed), dtype=torch.float32, device=self.device) if want_features else None
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_vit_forward(self._h, _ptr(x), n, 1 if save else 0, _ptr(logits), _ptr(feats), _stream()))
        return (logits, feats) if want_features else logits

    def features(self, x: torch.Tensor):
        """Un-normalised image features [N,E] only (no peer features / adapters needed): the image side of
        --lora_encoder text, clip/custom_clip.py:672-674."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        feats = torch.empty((x.shape[0], self.cfg.embed), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_vit_forward(self._h, _ptr(x), x.shape[0], 0, None, _ptr(feats), _stream()))
        return feats

    def backward(self, dlogits: torch.Tensor, selection=None):
        """selection: the dict entropy_select_loss / tpt_select_loss returned for these dlogits.  When it names a top-k selection
        (key "k": int(N * rho) views, listed in "idx"), the gradient is zero outside those views and the backward runs on them only
        (include/ttl_hip.h ttl_vit_backward_lora_selected; the fused episode applies the same rule)."""
        d = dlogits.to(device=self.device, dtype=torch.float32).contiguous()
        k = None if selection is None else selection.get("k")
        with torch.cuda.device(self.device):
            if k:
                self._check(self.lib.ttl_vit_backward_lora_selected(self._h, _ptr(d), d.shape[0], _ptr(selection["idx"]), int(k), _stream()))
            else:
                self._check(self.lib.ttl_vit_backward_lora(self._h, _ptr(d), d.shape[0], _stream()))
        return self.grads

    def backward_prescaled(self, dlogits: torch.Tensor):
        """The backward of an AUTOGRAD node (custom_clip._VitLogitsFn): ``dlogits`` is whatever torch hands down — already multiplied
        by the caller's GradScaler factor when the reference's `scaler.scale(loss).backward()` runs (deyo.py:185) — so the context's
        own loss scale must not be applied on top; the gradients come back scaled like ``dlogits`` (torch's scaler.step unscales them).
        include/ttl_hip.h ttl_ctx_backward_prescaled."""
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_ctx_backward_prescaled(self._h, 1))
        try:
            return self.backward(dlogits)
        finally:
            self._check(self.lib.ttl_ctx_backward_prescaled(self._h, 0))

    def entropy_select_loss(self, logits, mode, rho=0.1, thresh=None, margin=0.4, reweight=1.0, keep=None):
        """-> dict(H [N], idx int64 [N] (first n valid), n int32 [1], loss [1], dlogits [N,K]); all device tensors.
        keep: optional uint8/bool [N] second-stage filter (PLPD): selected views with keep == 0 are dropped."""
        import math
        z = logits.to(device=self.device, dtype=torch.float32).contiguous()
        N, K = z.shape
        dev = self.device
        out = dict(H=torch.empty(N, device=dev), idx=torch.zeros(N, dtype=torch.int64, device=dev),
                   n=torch.zeros(1, dtype=torch.int32, device=dev), loss=torch.zeros(1, device=dev),
                   dlogits=torch.empty_like(z))
        th = math.log(1000.0) if thresh is None else thresh
        kp = None if keep is None else keep.to(device=dev, dtype=torch.uint8).contiguous()
        with torch.cuda.device(dev):
            self._check(self.lib.ttl_ctx_entropy_select_loss(self._h, _ptr(z), N, K, int(mode), float(rho), float(th), float(margin),
                                                        float(reweight), _ptr(kp), _ptr(out["H"]), _ptr(out["idx"]), _ptr(out["n"]),
                                                        _ptr(out["loss"]), _ptr(out["dlogits"]), _stream()))
        out["k"] = int(N * rho) if int(mode) == _lib.TTL_SEL_TOPK else None      # deyo.py:105: a fixed count, known without a sync
        return out

    def tpt_select_loss(self, logits, rho=0.1, idx=None, n=None):
        z = logits.to(device=self.device, dtype=torch.float32).contiguous()
        N, K = z.shape
        dev = self.device
        reuse = idx is not None
        out = dict(H=torch.empty(N, device=dev),
                   idx=idx if reuse else torch.zeros(N, dtype=torch.int64, device=dev),
                   n=n if reuse else torch.zeros(1, dtype=torch.int32, device=dev),
                   loss=torch.zeros(1, device=dev), dlogits=torch.empty_like(z))
        with torch.cuda.device(dev):
            self._check(self.lib.ttl_ctx_tpt_select_loss(self._h, _ptr(z), N, K, float(rho), 1 if reuse else 0, _ptr(out["H"]),
                                                    _ptr(out["idx"]), _ptr(out["n"]), _ptr(out["loss"]),
                                                    _ptr(out["dlogits"]), _stream()))
        out["k"] = int(N * rho)                                                  # ttl.py:52
        return out

    # ---- PLPD filter of DeYO (deyo.py:115-151) on the device: destroyed views, keep mask; include/ttl_hip.h ttl_plpd_*
    @staticmethod
    def plpd_struct(spec, perm=None, n_candidates=0, aux=None):
        """spec: dict(aug_type 'occ' | 'patch' | 'pixel', threshold, patch_len, occlusion_size, row_start, column_start)."""
        a = _lib.ttl_plpd_args()
        a.aug_type = _lib.PLPD_AUG[spec["aug_type"]]
        a.threshold = float(spec.get("threshold", 0.0))
        a.patch_len = int(spec.get("patch_len", 0) or 0)
        a.occlusion_size, a.row_start, a.column_start = (int(spec.get(k, 0) or 0) for k in ("occlusion_size", "row_start", "column_start"))
        if perm is not None:
            assert perm.is_cuda and perm.dtype == torch.int32 and perm.is_contiguous()
            a.perm = perm.data_ptr()
        a.n_candidates = int(n_candidates)
        a.aux = aux._h if aux is not None else None
        return a

    def plpd_views(self, x, idx, n_sel, n_max, spec, perm=None):
        """x' = destroy(x[idx[:n]]) (deyo.py:116-134) -> [n_max,3,S,S]; idx int64 / n_sel int32 are the DEVICE outputs of the first
        selection stage, perm the host-drawn permutations of this update (device int32; draw_plpd_perms)."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        S = x.shape[-1]
        a = self.plpd_struct(spec, perm, n_max)
        out = torch.empty((n_max, 3, S, S), dtype=torch.float32, device=self.device)
        wsb = int(self.lib.ttl_plpd_views_workspace_bytes(n_max, S, C.byref(a)))
        ws = torch.empty(max(wsb, 4) // 4, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_plpd_views(_ptr(x), S, _ptr(idx), _ptr(n_sel), n_max, C.byref(a), _ptr(out), _ptr(ws), wsb, _stream()))
        return out

    def plpd_keep(self, logits, logits_prime, idx, n_sel, n_max, threshold):
        """-> (keep uint8 [N], plpd fp32 [n_max]): keep[idx[b]] = softmax(z[idx[b]])[argmax] - softmax(z'[b])[same class] > threshold."""
        z = logits.to(device=self.device, dtype=torch.float32).contiguous()
        zp = logits_prime.to(device=self.device, dtype=torch.float32).contiguous()
        keep = torch.empty(z.shape[0], dtype=torch.uint8, device=self.device)
        val = torch.empty(n_max, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_plpd_keep(_ptr(z), _ptr(zp), _ptr(idx), _ptr(n_sel), n_max, z.shape[0], z.shape[1], float(threshold),
                                               _ptr(keep), _ptr(val), _stream()))
        return keep, val

    def adamw_step(self, params, grads, m, v, step, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                   n_selected=None):
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_adamw_step(_ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(), lr, betas[0],
                                               betas[1], eps, weight_decay, int(step), _ptr(n_selected), _stream()))

    def set_concurrency(self, episodes_in_flight: int):
        """Tell the context how many episodes share the GPU (driver.EpisodePipeline does: one context per stream).  Results do not
        depend on it; tile choices do (include/ttl_hip.h ttl_ctx_set_concurrency)."""
        self._check(self.lib.ttl_ctx_set_concurrency(self._h, int(episodes_in_flight)))
        self.concurrency = int(episodes_in_flight)

    # ---- GradScaler contract (ttl.py:222, deyo.py:186-188): state on the device, see include/ttl_hip.h
    def scaler_config(self, dynamic=True, init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self._check(self.lib.ttl_scaler_config(self._h, int(bool(dynamic)), float(init_scale), float(growth_factor),
                                               float(backoff_factor), int(growth_interval)))

    def bind_scaler(self, scaler):
        """Adopt the hyper-parameters of the torch GradScaler the reference passes around (ttl.py:222).  Only the
        fp16-operand build scales its backward; bf16 has fp32's exponent range and keeps scale 1.  Idempotent per object."""
        if scaler is None or self.precision != "fp16" or getattr(self, "_scaler_id", None) == id(scaler):
            return
        if hasattr(scaler, "is_enabled") and not scaler.is_enabled():
            return
        self.scaler_config(True, float(scaler.get_scale()), float(scaler.get_growth_factor()), float(scaler.get_backoff_factor()),
                           int(scaler.get_growth_interval()))
        self._scaler_id = id(scaler)

    def scaler_state(self):
        """dict(scale, growth_tracker, skipped_steps, optimizer_steps); synchronises."""
        sc, tr, sk, st = C.c_float(), C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.ttl_scaler_state(self._h, C.byref(sc), C.byref(tr), C.byref(sk), C.byref(st)))
        return dict(scale=sc.value, growth_tracker=tr.value, skipped_steps=sk.value, optimizer_steps=st.value)

    def scaler_unscale(self, grads):
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_scaler_unscale(self._h, _ptr(grads), grads.numel(), _stream()))

    def optimizer_step(self, params, grads, m, v, step, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, n_selected=None):
        """scaler.step(optimizer) + scaler.update(): AdamW over the flat buffer — the whole step or none of it."""
        if getattr(self, "_skipped_seen", None) is None:     # baseline for step_was_taken(): skips recorded before this step
            self._skipped_seen = self.scaler_state()["skipped_steps"]
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_optimizer_step(self._h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), params.numel(), lr, betas[0],
                                                    betas[1], eps, weight_decay, int(step),
                                                    _ptr(n_selected) if n_selected is not None else None, _stream()))

    def step_was_taken(self):
        """After ``optimizer_step``: did the scaler apply the update (True) or skip the whole step on inf/nan gradients?
        Decided from the context's count of skipped steps before and after the call — the device's step counter is reset by
        ttl_episode* only, so on the step-wise path it can still hold the previous image's value.  One host sync."""
        seen = self.scaler_state()["skipped_steps"]
        taken = seen == self._skipped_seen
        self._skipped_seen = seen
        return taken

    def lora_reset(self, params, snapshot, m=None, v=None):
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_lora_reset(_ptr(params), _ptr(snapshot), _ptr(m), _ptr(v), params.numel(), _stream()))

    def episode(self, x, snapshot, m, v, *, n_updates=1, objective="deyo", mode=_lib.TTL_SEL_LE_THRESH, rho=0.1,
                thresh=None, margin=0.4, reweight=1.0, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                want_logits0=False, target=None, hits=None, out=None, plpd=None):
        """One whole test image (ttl.py:338-352) as a single enqueue; returns logits1 [1,K] (and logits0).
        plpd: a ttl_plpd_args (``plpd_struct``) — --filter_plpd 1 inside the episode (aux engine, host-drawn permutations).
        target (device int64 [1]) + hits (device int64 [3]): top-1 / top-5 hit of the adapted prediction and the image count are
        added to ``hits`` on the device (utils/tools.py:88-102), inside the same enqueue.  out: write logits1 there ([1,K])."""
        import math
        self._skipped_seen = None            # a fused episode may skip steps of its own: step_was_taken() re-reads its baseline
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        n = x.shape[0]
        l1 = out if out is not None else torch.empty((1, self.n_classes), dtype=torch.float32, device=self.device)
        l0 = torch.empty((n, self.n_classes), dtype=torch.float32, device=self.device) if want_logits0 else None
        a = _lib.ttl_episode_args()
        a.x, a.n_views, a.n_updates = x.data_ptr(), n, int(n_updates)
        a.objective = 0 if objective == "deyo" else 1
        a.mode, a.rho = int(mode), float(rho)
        a.thresh = math.log(1000.0) if thresh is None else thresh
        a.margin, a.reweight = float(margin), float(reweight)
        a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = lr, betas[0], betas[1], eps, weight_decay
        a.snapshot, a.exp_avg, a.exp_avg_sq = snapshot.data_ptr(), m.data_ptr(), v.data_ptr()
        a.logits0_out = l0.data_ptr() if want_logits0 else None
        a.logits1_out = l1.data_ptr()
        self._set_target(a, target, hits)
        if plpd is not None:
            a.plpd = C.pointer(plpd)
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_episode(self._h, C.byref(a), _stream()))
        return (l1, l0) if want_logits0 else l1

    def _set_target(self, a, target, hits):
        if (target is None) != (hits is None):
            raise _lib.TtlError("target and hits go together")
        if target is not None:
            for t, n, what in ((target, 1, "target"), (hits, 3, "hits")):       # raw int64 pointers cross the C ABI: a real check, not an assert
                if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.int64 and t.numel() >= n and t.is_contiguous()):
                    raise _lib.TtlError(f"{what} must be a contiguous device int64 tensor with at least {n} element(s)")
            a.target, a.hits_out = target.data_ptr(), hits.data_ptr()

    def episode_graph(self, x_buf, snapshot, m, v, logits1_buf, *, n_updates=1, objective="deyo", mode=_lib.TTL_SEL_LE_THRESH,
                      rho=0.1, thresh=None, margin=0.4, reweight=1.0, lr=5e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                      target=None, hits=None, plpd=None):
        """Capture the episode over FIXED buffers (x_buf [N,3,S,S], logits1_buf [1,K], snapshot / m / v, and — for the on-device
        hit count — target int64 [1] / hits int64 [3]) into a HIP graph on the current (non-default) stream; returns a callable
        that replays it on the current stream.  The capture itself runs the episode once on whatever the buffers hold (that run
        counts into ``hits`` like any other)."""
        import math
        assert x_buf.is_cuda and x_buf.is_contiguous() and logits1_buf.is_contiguous()
        a = _lib.ttl_episode_args()
        a.x, a.n_views, a.n_updates = x_buf.data_ptr(), x_buf.shape[0], int(n_updates)
        a.objective = 0 if objective == "deyo" else 1
        a.mode, a.rho = int(mode), float(rho)
        a.thresh = math.log(1000.0) if thresh is None else thresh
        a.margin, a.reweight = float(margin), float(reweight)
        a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = lr, betas[0], betas[1], eps, weight_decay
        a.snapshot, a.exp_avg, a.exp_avg_sq = snapshot.data_ptr(), m.data_ptr(), v.data_ptr()
        a.logits0_out, a.logits1_out = None, logits1_buf.data_ptr()
        self._set_target(a, target, hits)
        if plpd is not None:          # (the permutation buffer it points to is baked into the graph: refill it in place)
            a.plpd = C.pointer(plpd)
        g = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_episode_capture(self._h, C.byref(a), _stream(), C.byref(g)))
        self._graphs = getattr(self, "_graphs", [])
        self._graphs.append(g)
        keep = (x_buf, snapshot, m, v, logits1_buf, target, hits)     # the graph holds raw pointers into these

        def launch(_keep=keep):
            with torch.cuda.device(self.device):
                self._check(self.lib.ttl_graph_launch(g, _stream()))
            return logits1_buf
        return launch

    # ------------------------------------------------------------------ debugging / measurement
    def debug_copy(self, name, layer, shape, dtype=np.float32):
        a = np.empty(shape, dtype=dtype)
        self._check(self.lib.ttl_debug_copy(self._h, name.encode(), int(layer), a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    def last_selection(self, n_views):
        """(idx int64 [n], entropies fp32 [n_views]) the last fused update really used: the HIP selection list in the
        reference's order (deyo.py:103-108 / ttl.py:50-54), read back from the context."""
        n = int(self.debug_copy("n_selected", 0, (1,), np.int32)[0])
        idx = self.debug_copy("idx", 0, (n_views,), np.int64)[:n]
        return idx, self.debug_copy("entropy", 0, (n_views,), np.float32)

    def profile_enable(self, on=True):
        self._check(self.lib.ttl_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        ms = (C.c_double * _lib.TTL_NCLASS)()
        cnt = (C.c_longlong * _lib.TTL_NCLASS)()
        fl = C.c_double()
        self._check(self.lib.ttl_profile_read(self._h, ms, cnt, C.byref(fl)))
        by = C.c_double()
        self._check(self.lib.ttl_profile_gemm_bytes(self._h, C.byref(by)))
        self.last_gemm_bytes = by.value
        self._check(self.lib.ttl_profile_gemm_flops_all(self._h, C.byref(by)))
        self.last_gemm_flops_all = by.value
        return ({k: ms[i] for i, k in enumerate(_lib.PROFILE_CLASSES)},
                {k: cnt[i] for i, k in enumerate(_lib.PROFILE_CLASSES)}, fl.value)


class TextTowerEngine(TTLEngine):
    """The text tower of ``--lora_encoder text`` (clip/custom_clip.py:602-607): a context with causal
    attention over the prompt tokens whose q/v LoRA is the thing being tuned.  ``max_prompts`` bounds K,
    ``max_views`` the number of image views whose features it scores."""

    def __init__(self, cfg, max_prompts: int, max_views: int, device, precision: str = None):
        super().__init__(cfg, max_prompts, max_views, device, precision)
        self.n_prompts = 0
        self.n_views = 0

    def _make_config(self, cfg):
        return _lib.ttl_config(0, 0, cfg.width, cfg.heads, cfg.mlp, cfg.layers, cfg.embed, cfg.rank, cfg.lora_alpha,
                               cfg.layer_lo, cfg.layer_hi, cfg.ln_eps, self.max_views, self.max_classes,
                               _lib.TTL_TOWER_TEXT, cfg.context_length, cfg.vocab_size, _targets_mask(cfg))

    def set_logit_scale(self, logit_scale_exp: float):
        self._check(self.lib.ttl_set_logit_scale(self._h, float(logit_scale_exp)))
        self._scale = float(logit_scale_exp)

    def set_prompts(self, ids):
        """ids: int [K, context_length] (prompt_learner.tokenized_prompts, clip/custom_clip.py:655)."""
        t = torch.as_tensor(ids).to(dtype=torch.int32).contiguous().cpu()
        assert t.dim() == 2 and t.shape[1] == self.cfg.context_length
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_set_prompts(self._h, C.c_void_p(t.data_ptr()), t.shape[0], _stream()))
        self.n_prompts = int(t.shape[0])

    def set_image_features(self, feats: torch.Tensor, normalize: bool = True):
        f = feats.detach().to(device=self.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_set_image_features(self._h, _ptr(f), f.shape[0], 1 if normalize else 0, self._scale, _stream()))
        self.n_views = int(f.shape[0])

    def set_text_features(self, *a, **k):
        raise _lib.TtlError("a text-tower context scores image features: use set_image_features")

    def forward(self, save: bool = False, want_features: bool = False):
        """-> logits [n_views, n_prompts] (and un-normalised text features [n_prompts, E])."""
        logits = torch.empty((self.n_views, self.n_prompts), dtype=torch.float32, device=self.device)
        feats = torch.empty((self.n_prompts, self.cfg.embed), dtype=torch.float32, device=self.device) if want_features else None
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_text_forward(self._h, 1 if save else 0, _ptr(logits), _ptr(feats), _stream()))
        return (logits, feats) if want_features else logits

    def backward(self, dlogits: torch.Tensor):
        d = dlogits.to(device=self.device, dtype=torch.float32).contiguous()
        assert tuple(d.shape) == (self.n_views, self.n_prompts)
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_text_backward_lora(self._h, _ptr(d), _stream()))
        return self.grads

    def episode(self, image_engine: TTLEngine, x, snapshot, m, v, *, n_updates=1, objective="deyo",
                mode=_lib.TTL_SEL_LE_THRESH, rho=0.1, thresh=None, margin=0.4, reweight=1.0, lr=5e-3, betas=(0.9, 0.999),
                eps=1e-8, weight_decay=1e-2, want_logits0=False, target=None, hits=None, plpd=None):
        """Whole text-mode episode as one enqueue; ``image_engine`` is an adapter-less image-tower engine."""
        import math
        self._skipped_seen = None
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        n = x.shape[0]
        l1 = torch.empty((1, self.n_prompts), dtype=torch.float32, device=self.device)
        l0 = torch.empty((n, self.n_prompts), dtype=torch.float32, device=self.device) if want_logits0 else None
        a = _lib.ttl_episode_args()
        a.x, a.n_views, a.n_updates = x.data_ptr(), n, int(n_updates)
        a.objective = 0 if objective == "deyo" else 1
        a.mode, a.rho = int(mode), float(rho)
        a.thresh = math.log(1000.0) if thresh is None else thresh
        a.margin, a.reweight = float(margin), float(reweight)
        a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = lr, betas[0], betas[1], eps, weight_decay
        a.snapshot, a.exp_avg, a.exp_avg_sq = snapshot.data_ptr(), m.data_ptr(), v.data_ptr()
        a.logits0_out = l0.data_ptr() if want_logits0 else None
        a.logits1_out = l1.data_ptr()
        self._set_target(a, target, hits)
        if plpd is not None:
            a.plpd = C.pointer(plpd)
        with torch.cuda.device(self.device):
            self._check(self.lib.ttl_episode_text(self._h, image_engine._h, C.byref(a), _stream()))
        self.n_views = n
        return (l1, l0) if want_logits0 else l1


def bf16_bits_to_f32(a: np.ndarray) -> np.ndarray:
    """uint16 bf16 storage -> float32 (helper for tests reading debug buffers)."""
    return (a.astype(np.uint32) << 16).view(np.float32)
