"""-m gpu: kernel-level parity of the HIP path (through the C ABI) against numpy restatements
and against the reference-generated goldens for the loss / optimizer side."""
import ctypes as C
import math

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import ttl_oracle as O
from helpers import max_rel

pytestmark = pytest.mark.gpu


# Every kernel test runs on all three builds of the same sources (round-4 review, weak 4): "bf16" (libttl_hip.so), "fp16"
# (libttl_hip_fp16.so — the headline build: other MFMA builtins, the dS * 2^8 pre-scale) and "strict" (libttl_hip_strict.so, the
# test-only fp32 build).  TOL: what one rounding of an output / of P and dS to the operand type costs relative to the tensor's max.
PRECISIONS = ("bf16", "fp16", "strict")
TDT = {"bf16": torch.bfloat16, "fp16": torch.float16, "strict": torch.float32}
OUT_TOL = {"bf16": 6e-3, "fp16": 8e-4, "strict": 2e-5}     # operand-dtype outputs (one rounding: 2^-8 / 2^-11 / none)
BWD_TOL = {"bf16": 1e-2, "fp16": 1.5e-3, "strict": 5e-5}   # attention backward (P, dS and the outputs rounded)
DS_PRESCALE = {"bf16": 1.0, "fp16": 256.0, "strict": 1.0}  # csrc/common.hpp TTL_DS_PRESCALE


def op_round(a, prec):
    """Round an fp32 array to the build's operand type and back (the rounding points of the MFMA path)."""
    if prec == "bf16":
        return O.bf16_round(a)
    if prec == "fp16":
        with np.errstate(over="ignore"):
            return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)
    return np.asarray(a, np.float32)


@pytest.fixture(scope="module", params=PRECISIONS)
def prec(request):
    return request.param


@pytest.fixture(scope="module")
def lib(prec):
    from ttl_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _lib.load(prec)


def P(t):
    return C.c_void_p(t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def chk(lib, rc):
    assert rc == 0, (rc, lib.ttl_last_error())


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (197, 256, 768), (1000, 768, 832), (12608, 768, 768), (300, 3072, 768),
                                   (260, 768, 3072)])
def test_gemm_nt(lib, prec, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(TDT[prec])
    b = (torch.randn(N, K, generator=g) * 0.05).to(TDT[prec])
    # asymmetric integer-valued check first (guide: A=I style layout test)
    ref = (a.double().numpy() @ b.double().numpy().T).astype(np.float32)
    da, db = a.cuda(), b.cuda()
    c = torch.full((M, N), float("nan"), device="cuda")
    chk(lib, lib.ttl_gemm_nt(P(da), K, P(db), K, P(c), N, M, N, K, S()))
    torch.cuda.synchronize()
    out = c.cpu().numpy()
    assert np.isfinite(out).all()
    assert max_rel(out, ref) < 2e-5          # operands exact in their type, fp32 accumulate


def test_gemm_layout_asymmetric(lib, prec):
    """C = I·B^T must reproduce an asymmetric B exactly (catches row/col swaps and k permutations)."""
    M = N = 128
    K = 128
    a = torch.zeros(M, K)
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    b = (torch.arange(N * K).reshape(N, K) % 251).float() - 100.0
    c = torch.empty(M, N, device="cuda")
    da, db = a.to(TDT[prec]).cuda(), b.to(TDT[prec]).cuda()
    chk(lib, lib.ttl_gemm_nt(P(da), K, P(db), K, P(c), N, M, N, K, S()))
    torch.cuda.synchronize()
    assert np.array_equal(c.cpu().numpy(), (a @ b.T).numpy())


@pytest.mark.parametrize("rows,D", [(5, 128), (197, 768), (1000, 1024)])
def test_layernorm(lib, rows, D):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, D, generator=g) * 3 + 1
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    y = torch.empty(rows, D, device="cuda")
    mu = torch.empty(rows, device="cuda")
    rs = torch.empty(rows, device="cuda")
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()   # keep the device copies alive across the call
    chk(lib, lib.ttl_layernorm_f32(P(xd), P(wd), P(bd), P(y), P(mu), P(rs), rows, D, 1e-5, S()))
    torch.cuda.synchronize()
    ry, rmu, rrs = O.layer_norm(x.numpy(), w.numpy(), b.numpy(), 1e-5)
    assert max_rel(y.cpu().numpy(), ry) < 1e-5
    assert max_rel(mu.cpu().numpy(), rmu[:, 0]) < 1e-5
    assert max_rel(rs.cpu().numpy(), rrs[:, 0]) < 1e-5


def _attn_ref(qkv, n, T, H, causal=0, prec="bf16"):
    D = H * 64
    x = qkv.reshape(n, T, 3, H, 64).transpose(2, 0, 3, 1, 4)        # [3,n,H,T,64]
    q, k, v = x[0], x[1], x[2]
    s = (q @ k.transpose(0, 1, 3, 2)) * 0.125
    if causal:
        s = np.where(np.tril(np.ones((T, T), bool)), s, -np.inf)
    m = s.max(-1, keepdims=True)
    e = np.exp(s - m)
    den = e.sum(-1, keepdims=True)
    p = e / den
    o = op_round(p, prec) @ v
    return q, k, v, p, o, (m + np.log(den))[..., 0]


@pytest.mark.parametrize("n,T,H", [(44, 197, 12), (43, 200, 12), (64, 224, 8)])
def test_attention_fwd_persistent_kernel(lib, prec, n, T, H):
    """Launches of >= 512 (view, head) problems with 193 <= T <= 224 take attn_fwd_p_kernel (one workgroup per CU walking
    2-3 problems, next K/V/q prefetched by LDS-DMA behind counted waits): uneven problem counts per workgroup, padded and
    unpadded last key tile, with and without the log-sum-exp output (different store counts under the counted wait),
    bitwise repeatable."""
    D = H * 64
    g = torch.Generator().manual_seed(n + T)
    qkv = (torch.randn(n * T, 3 * D, generator=g)).to(TDT[prec])
    qkv[:, :D] *= 1.5
    q, k, v, p, o, lse_ref = _attn_ref(qkv.float().numpy(), n, T, H, 0, prec)
    o_ref = o.transpose(0, 2, 1, 3).reshape(n * T, D)
    dq = qkv.cuda()
    outs = []
    for with_lse in (True, False, True):
        out = torch.full((n * T, D), float("nan"), device="cuda", dtype=TDT[prec])
        lse = torch.empty(n, H, T, device="cuda")
        chk(lib, lib.ttl_attention_fwd(P(dq), P(out), P(lse) if with_lse else None, n, T, H, 0, S()))
        torch.cuda.synchronize()
        assert max_rel(out.float().cpu().numpy(), o_ref) < OUT_TOL[prec]
        if with_lse:
            assert np.abs(lse.cpu().numpy() - lse_ref).max() < 2e-4
        outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("n,T,H,causal", [(2, 17, 2, 0), (3, 197, 2, 0), (1, 257, 4, 0), (2, 50, 12, 0), (1, 150, 1, 0),
                                          (3, 77, 8, 1), (2, 77, 2, 1), (2, 17, 2, 1), (1, 197, 2, 1)])
def test_attention_fwd_bwd(lib, prec, n, T, H, causal):
    D = H * 64
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(n * T, 3 * D, generator=g)).to(TDT[prec])
    qkv[:, :D] *= 1.5
    dout = (torch.randn(n * T, D, generator=g) * 0.1).to(TDT[prec])
    q, k, v, p, o, lse_ref = _attn_ref(qkv.float().numpy(), n, T, H, causal, prec)
    out = torch.full((n * T, D), float("nan"), device="cuda", dtype=TDT[prec])
    lse = torch.empty(n, H, T, device="cuda")
    dq = qkv.cuda()
    ddo = dout.cuda()
    chk(lib, lib.ttl_attention_fwd(P(dq), P(out), P(lse), n, T, H, causal, S()))
    torch.cuda.synchronize()
    o_ref = o.transpose(0, 2, 1, 3).reshape(n * T, D)
    assert max_rel(out.float().cpu().numpy(), o_ref) < OUT_TOL[prec]          # output rounded to the operand type
    assert np.abs(lse.cpu().numpy() - lse_ref).max() < 2e-4
    # backward, from the kernel's own (rounded) output like the real path
    o_k = out.float().cpu().numpy().reshape(n, T, H, 64).transpose(0, 2, 1, 3)
    dO = dout.float().numpy().reshape(n, T, H, 64).transpose(0, 2, 1, 3)
    delta = (dO * o_k).sum(-1, keepdims=True)
    dV = op_round(p, prec).transpose(0, 1, 3, 2) @ dO
    dP = dO @ v.transpose(0, 1, 3, 2)
    dS = op_round(p * (dP - delta) * DS_PRESCALE[prec], prec) / DS_PRESCALE[prec]      # (fp16: rounded as dS * 2^8, common.hpp)
    dQ = (dS @ k) * 0.125
    dK = (dS.transpose(0, 1, 3, 2) @ q) * 0.125
    mg = lambda a: a.transpose(0, 2, 1, 3).reshape(n * T, D)
    ld = 3 * D + 64
    for need_dk in (1, 0):
        dqkv = torch.zeros(n * T, ld, device="cuda", dtype=TDT[prec])
        chk(lib, lib.ttl_attention_bwd(P(dq), P(out), P(ddo), P(lse), P(dqkv), ld, n, T, H, need_dk, causal, S()))
        torch.cuda.synchronize()
        r = dqkv.float().cpu().numpy()
        assert max_rel(r[:, :D], mg(dQ)) < BWD_TOL[prec]
        assert max_rel(r[:, 2 * D:3 * D], mg(dV)) < BWD_TOL[prec]
        if need_dk:
            assert max_rel(r[:, D:2 * D], mg(dK)) < BWD_TOL[prec]
        else:
            assert not r[:, D:2 * D].any()
        assert not r[:, 3 * D:].any()


@pytest.fixture(scope="module")
def unit(golden_dir):
    return np.load(golden_dir + "/unit_loss_adamw.npz")


@pytest.mark.parametrize("s", ["a", "b", "c", "d"])
@pytest.mark.parametrize("mode", ["le_thresh", "topk"])
def test_entropy_select_loss_vs_reference(lib, unit, s, mode):
    """deyo.py:85-90,102-113,175-181 through the HIP kernel vs the reference's own outputs."""
    z = torch.from_numpy(unit[f"{s}/z"]).cuda()
    N, K = z.shape
    H = torch.empty(N, device="cuda")
    idx = torch.full((N,), -1, dtype=torch.int64, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    loss = torch.zeros(1, device="cuda")
    dz = torch.empty_like(z)
    m = 0 if mode == "le_thresh" else 1
    chk(lib, lib.ttl_entropy_select_loss(P(z), N, K, m, 0.1, math.log(1000.0), 0.4, 1.0, None, P(H), P(idx), P(n), P(loss), P(dz), S()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(H.cpu().numpy(), unit[f"{s}/H"], rtol=2e-5, atol=2e-6)
    nn = int(n.item())
    if f"{s}/{mode}/idx" not in unit.files:
        assert nn == 0 and not dz.any().item() and loss.item() == 0.0     # early return, deyo.py:110-113
        return
    ref_idx = unit[f"{s}/{mode}/idx"]
    assert nn == ref_idx.size
    assert np.array_equal(idx.cpu().numpy()[:nn], ref_idx)                # selection: BIT-EXACT, same order
    assert abs(loss.item() - unit[f"{s}/{mode}/loss"]) <= 1e-5 * abs(unit[f"{s}/{mode}/loss"]) + 1e-7
    assert max_rel(dz.cpu().numpy(), unit[f"{s}/{mode}/dz"]) < 1e-4


@pytest.mark.parametrize("s", ["t1", "t2", "t3"])
def test_exact_entropy_ties_at_the_selection_boundary(lib, golden_dir, s):
    """Bit-identical rows straddling rank int(N*rho) (tests/golden/unit_ties.npz, written by the reference's torch.argsort):
    select_kernel keeps the same SET (the tied group's lowest view indices, ascending; the reference returns that set in an
    unspecified order) behind the same untied prefix, for the DeYO top-rho mode and the TPT selection."""
    u = np.load(golden_dir + "/unit_ties.npz")
    z = torch.from_numpy(u[f"{s}/z"]).cuda()
    rho = float(u[f"{s}/rho"])
    N, K = z.shape
    ref = u[f"{s}/topk/idx"].tolist()
    tied = sorted(u[f"{s}/tied_rows"].tolist())
    lead = [i for i in ref if i not in tied]
    want = lead + tied[:len(ref) - len(lead)]
    for tpt in (False, True):
        H = torch.empty(N, device="cuda")
        idx = torch.full((N,), -1, dtype=torch.int64, device="cuda")
        n = torch.zeros(1, dtype=torch.int32, device="cuda")
        loss = torch.zeros(1, device="cuda")
        dz = torch.empty_like(z)
        if tpt:
            chk(lib, lib.ttl_tpt_select_loss(P(z), N, K, rho, 0, P(H), P(idx), P(n), P(loss), P(dz), S()))
        else:
            chk(lib, lib.ttl_entropy_select_loss(P(z), N, K, 1, rho, math.log(1000.0), 0.4, 1.0, None, P(H), P(idx), P(n), P(loss), P(dz), S()))
        torch.cuda.synchronize()
        assert idx.cpu().numpy()[:int(n.item())].tolist() == want
        assert sorted(want) == sorted(ref)


@pytest.mark.parametrize("s", ["a", "b", "d"])
def test_tpt_select_loss_vs_reference(lib, unit, s):
    z = torch.from_numpy(unit[f"{s}/z"]).cuda()
    N, K = z.shape
    H = torch.empty(N, device="cuda")
    idx = torch.full((N,), -1, dtype=torch.int64, device="cuda")
    n = torch.zeros(1, dtype=torch.int32, device="cuda")
    loss = torch.zeros(1, device="cuda")
    dz = torch.empty_like(z)
    chk(lib, lib.ttl_tpt_select_loss(P(z), N, K, 0.1, 0, P(H), P(idx), P(n), P(loss), P(dz), S()))
    torch.cuda.synchronize()
    nn = int(n.item())
    assert np.array_equal(idx.cpu().numpy()[:nn], unit[f"{s}/topk_idx"])    # ttl.py:50-54
    assert abs(loss.item() - unit[f"{s}/avg_entropy"]) <= 2e-5 * max(1.0, abs(unit[f"{s}/avg_entropy"]))
    assert max_rel(dz.cpu().numpy(), unit[f"{s}/tpt_dz"]) < 1e-4
    # second step re-uses the cached indices (ttl.py:97-98)
    dz2 = torch.empty_like(z)
    chk(lib, lib.ttl_tpt_select_loss(P(z), N, K, 0.1, 1, P(H), P(idx), P(n), P(loss), P(dz2), S()))
    torch.cuda.synchronize()
    assert torch.equal(dz, dz2)


def test_adamw_vs_torch_reference(lib, unit):
    p = torch.from_numpy(unit["adamw/p0"]).cuda().contiguous()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for t in range(3):
        g = torch.from_numpy(unit[f"adamw/g{t}"]).cuda()
        chk(lib, lib.ttl_adamw_step(P(p), P(g), P(m), P(v), p.numel(), 5e-3, 0.9, 0.999, 1e-8, 1e-2, t + 1, None, S()))
        torch.cuda.synchronize()
        np.testing.assert_allclose(p.cpu().numpy(), unit[f"adamw/p{t + 1}"], rtol=1e-5, atol=1e-7)
    # n_selected == 0 skips the step (deyo.py:183); non-finite grads never step
    before = p.clone()
    zero = torch.zeros(1, dtype=torch.int32, device="cuda")
    chk(lib, lib.ttl_adamw_step(P(p), P(g), P(m), P(v), p.numel(), 5e-3, 0.9, 0.999, 1e-8, 1e-2, 4, P(zero), S()))
    g2 = g.clone()
    g2[0, 0] = float("inf")
    chk(lib, lib.ttl_adamw_step(P(p), P(g2), P(m), P(v), p.numel(), 5e-3, 0.9, 0.999, 1e-8, 1e-2, 4, None, S()))
    torch.cuda.synchronize()
    assert p[0, 0] == before[0, 0] and not torch.equal(p, before)
    snap = torch.randn_like(p)
    chk(lib, lib.ttl_lora_reset(P(p), P(snap), P(m), P(v), p.numel(), S()))
    torch.cuda.synchronize()
    assert torch.equal(p, snap) and not m.any() and not v.any()


@pytest.mark.parametrize("epi", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(12608, 2304, 832), (12608, 768, 3072), (12608, 3072, 768), (2056, 1024, 1024), (1500, 256, 192),
                                   (16448, 1024, 4096)])
def test_gemm_big_tiles_with_fused_epilogues(lib, prec, M, N, K, epi):
    """The big-M kernel of gemm_big.hip (160x256x64 tiles, persistent blocks, 3-stage DMA ring, bias / residual folded into
    the accumulator init) through ttl_gemm_nt_epi: every epilogue the episode uses, row counts that end inside the last
    row tile, one / several tiles per block, odd K-tile counts — against an fp32 matmul of the same bf16 operands."""
    g = torch.Generator(device="cpu").manual_seed(M + N + K + epi)
    a = torch.randn(M, K, generator=g).to(TDT[prec]).cuda()
    b = (torch.randn(N, K, generator=g) * 0.05).to(TDT[prec]).cuda()
    bias = torch.randn(N, generator=g).cuda()
    Mp = (M + 1279) // 1280 * 1280
    res = torch.randn(Mp, N, generator=g).cuda() if epi == 2 else None
    c = torch.full((Mp, N), 7.0, device="cuda", dtype=torch.float32 if epi in (0, 2) else TDT[prec])
    chk(lib, lib.ttl_gemm_nt_epi(P(a), K, P(b), K, P(c), N, M, N, K, epi, P(bias), P(res) if res is not None else None, N, Mp, S()))
    torch.cuda.synchronize()
    want = a.float() @ b.float().t() + bias
    if epi == 2:
        want = want + res[:M]
    if epi == 3:
        want = want * torch.sigmoid(1.702 * want)
    got = c[:M].float()
    tol = 2e-5 if epi in (0, 2) else OUT_TOL[prec]        # fp32 out: accumulation order only; operand-dtype out: one rounding
    assert max_rel(got.cpu().numpy(), want.cpu().numpy()) < tol
    # rows of the arena padding beyond round_up(M, 160) are never written (strict build: beyond M)
    top = (M + 159) // 160 * 160
    if top < Mp:
        assert (c[top:].float() == 7.0).all()


def test_gemm_big_tile_epilogues_on_the_256x256_kernel(prec):
    """The fp32-output / residual / operand-dtype epilogues of the N = D launches also exist on gemm_huge.hip (taken when the context
    says that episodes run concurrently, ttl_ctx_set_concurrency): the kernel-level cases of test_gemm_big_tiles_with_fused_epilogues
    again, in a child process that forces that kernel (TTL_GEMM_HUGE_NARROW=1: the switch is read once per process)."""
    import os, subprocess, sys
    if prec == "strict":
        pytest.skip("the strict build has its own GEMM")
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_kernels.py"), "-x", "-q", "-k",
                          f"big_tiles_with_fused_epilogues and {prec}", "-p", "no:cacheprovider"],
                         env=dict(os.environ, TTL_GEMM_HUGE_NARROW="1"), capture_output=True, text=True, timeout=900, cwd=os.path.dirname(here))
    assert out.returncode == 0 and " passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


def test_gemm_wide_fused_forms(lib, prec):
    """q/k/v head-major (modeling_clip.py:309-311 written as [view][plane][head][T][64]) and fc1 with both outputs (quick_gelu and
    the pre-activation the backward reads) through ttl_gemm_nt_fused — the forms no other kernel-level entry reaches — on both
    kernels that have them: in this process the default split (q/k/v launches with tiles for most of the CUs on gemm_huge.hip's 256 x 256
    four-wave tiles, everything else on gemm_big.hip's 160 x 256), in child processes TTL_GEMM_HUGE=0 (everything on gemm_big.hip)
    and =1 with the round-fill rule off and the MLP-dgrad form switched on (everything on gemm_huge.hip); the MLP-dgrad form
    (product * quick_gelu'(u)) runs on gemm_big.hip's 128-row tiles by default.
    The strict build has neither form (row-major q/k/v, its own GEMM): the entry must say so."""
    import os, subprocess, sys
    import gemm_fused_check as G
    if prec == "strict":
        a = torch.zeros(1280, 768, device="cuda")
        c = torch.zeros(1280 * 2304, device="cuda")
        assert lib.ttl_gemm_nt_fused(P(a), 768, P(a), 768, P(c), 2304, None, 0, 1280, 2304, 768, None, 197, 1280, S()) != 0
        return
    for shp in G.SHAPES:
        G.check(lib, prec, *shp)
    G.check(lib, prec, 12608, 2304, 768, 197, lda_pad=64)            # A as the first K columns of a wider buffer
    G.check(lib, prec, 12608, 2304, 768, 197, with_bias=False)       # null bias
    n_dgrad = sum(1 for shp in G.SHAPES if shp[3] == -1)
    # (mode 1 with the round-fill rule off: every q/k/v and fc1 shape on gemm_huge.hip; the MLP-dgrad form there is a closed experiment
    # that only the experiments build — the fp16 library with -DTTL_EXPERIMENTS — switches on: third child, fp16 round only)
    runs = [("0", prec, {}, 0), ("1", prec, {}, len(G.SHAPES) - n_dgrad)]
    if prec == "fp16":
        runs.append(("1", "experiments", {"TTL_GEMM_HUGE_DGRAD": "1"}, len(G.SHAPES)))
    for mode, build, extra, n_huge in runs:
        out = subprocess.run([sys.executable, G.__file__, build], env=dict(os.environ, TTL_GEMM_HUGE=mode, TTL_GEMM_HUGE_MIN_FILL="0", **extra),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and f"ok {build} mode {mode}" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
        assert f"on gemm_huge: {n_huge}" in out.stdout, out.stdout[-500:]


def test_gradscaler_known_answers(lib, prec):
    """ttl_scaler_config / ttl_scaler_unscale / ttl_optimizer_step against the reference's own objects — torch.amp.GradScaler(
    init_scale=1000) around torch.optim.AdamW — over 9 updates with an inf and a nan injected (fixture written by
    tests/golden/make_gradscaler_golden.py): the WHOLE step is skipped on a non-finite gradient (params, exp_avg, exp_avg_sq
    and the Adam step count untouched), the scale halves, and doubles again after growth_interval clean steps."""
    import os
    from conftest import GOLDEN
    from ttl_amd.config import get_config
    from ttl_amd.engine import TTLEngine
    u = np.load(os.path.join(GOLDEN, "unit_gradscaler.npz"))
    eng = TTLEngine(get_config("tiny"), 4, 10, "cuda:0", precision=prec)
    eng.scaler_config(True, float(u["init_scale"]), float(u["growth_factor"]), float(u["backoff_factor"]), int(u["growth_interval"]))
    p = torch.from_numpy(u["p0"]).cuda()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    lr = float(u["lr"])
    for t in range(u["grads"].shape[0]):
        scale = eng.scaler_state()["scale"]
        g = (torch.from_numpy(u["grads"][t]) * scale).cuda()       # what a backward of scale * loss leaves behind
        eng.scaler_unscale(g)                                       # scaler.unscale_ (+ found_inf)
        eng.optimizer_step(p, g, m, v, 0, lr=lr)                    # scaler.step + scaler.update (step counted on the device)
        st = eng.scaler_state()
        assert st["scale"] == float(u["scale"][t]), (t, st, u["scale"][t])
        assert st["optimizer_steps"] == int(u["step"][t]), (t, st, u["step"][t])
        np.testing.assert_allclose(p.cpu().numpy(), u["params"][t], rtol=2e-6, atol=2e-8, err_msg=f"params after update {t + 1}")
        np.testing.assert_allclose(m.cpu().numpy(), u["m"][t], rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(v.cpu().numpy(), u["v"][t], rtol=2e-5, atol=1e-12)
    assert eng.scaler_state()["skipped_steps"] == 2
    eng.close()


def test_fp16_prescaled_ds_overflow_is_an_inf_not_a_wrong_number():
    """fp16 build only (csrc/common.hpp:16-23): dS is rounded to fp16 as dS * 2^8.  When |dP| is large enough that the pre-scaled
    value exceeds fp16's range, the overflow must surface as inf / nan in dQ and dK — which the gradient reduction turns into
    found_inf, a skipped step and a halved loss scale (test_gpu_path.py::test_overflowing_backward_skips_the_whole_step_and_halves_
    the_scale) — never as a saturated finite number.  dV = P^T dO does not go through dS and stays finite."""
    from ttl_amd import _lib
    lib = _lib.load("fp16")
    n, T, H = 2, 50, 2
    D = H * 64
    g = torch.Generator().manual_seed(5)
    qkv = torch.randn(n * T, 3 * D, generator=g).to(torch.float16)
    qkv[:, :D] *= 3.0                                   # peaked rows: P close to one-hot, so dS ~ dP
    dq = qkv.cuda()
    out = torch.empty(n * T, D, device="cuda", dtype=torch.float16)
    lse = torch.empty(n, H, T, device="cuda")
    chk(lib, lib.ttl_attention_fwd(P(dq), P(out), P(lse), n, T, H, 0, S()))
    ld = 3 * D + 64
    for mag, overflow in ((1.0, False), (400.0, True)):  # |dP| ~ mag * 8 * |v|: 400 -> dS * 2^8 ~ 1e6 >> 65504
        dout = (torch.randn(n * T, D, generator=g) * mag).to(torch.float16).cuda()
        dqkv = torch.zeros(n * T, ld, device="cuda", dtype=torch.float16)
        chk(lib, lib.ttl_attention_bwd(P(dq), P(out), P(dout), P(lse), P(dqkv), ld, n, T, H, 1, 0, S()))
        torch.cuda.synchronize()
        r = dqkv.float().cpu().numpy()
        assert np.isfinite(r[:, 2 * D:3 * D]).all()                       # dV
        assert np.isfinite(r[:, :2 * D]).all() == (not overflow), mag     # dQ, dK
