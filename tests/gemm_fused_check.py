"""Check of ttl_gemm_nt_fused (the q/k/v head-major and fc1 GELU + pre-activation forms of the wide projections) against an fp32
matmul of the same operands.  Imported by tests/test_gpu_kernels.py; also runnable as a script so that the same check can run in a
child process under another TTL_GEMM_HUGE (the switch is read once per process): `python tests/gemm_fused_check.py <precision>`."""
import ctypes as C
import os
import sys

import numpy as np
import torch

# ("experiments": the fp16 library built with -DTTL_EXPERIMENTS, the only build that reads the closed switch TTL_GEMM_HUGE_DGRAD)
TDT = {"bf16": torch.bfloat16, "fp16": torch.float16, "experiments": torch.float16}
OUT_TOL = {"bf16": 6e-3, "fp16": 8e-4, "experiments": 8e-4}      # one rounding of the output to the operand type
# (M, N, K, T): T > 0 = head-major q/k/v of views of T tokens, 0 = fc1, -1 = the MLP dgrad (product * quick_gelu'(u)).  ViT-B/16 and L/14 episode shapes, a row count that ends
# inside a tile, the shortest K the 256 x 256 kernel takes (3 K-tiles) and its longest (16), one tile per block and several
SHAPES = [(12608, 2304, 768, 197), (12608, 3072, 768, 0), (16448, 3072, 1024, 257), (16448, 4096, 1024, 0), (5122, 2304, 192, 197),
          (5000, 3072, 192, 0), (1576, 2304, 832, 197), (12708, 2304, 768, 197),       # (this one ends inside a view)
          (12608, 3072, 768, -1), (16448, 4096, 1024, -1), (5000, 3072, 192, -1)]


def max_rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def huge_mode():
    """csrc/gemm_huge.hip TTL_GEMM_HUGE: 0 off, 1 q/k/v + fc1, 2 q/k/v only (default), 3 fc1 only."""
    return int(os.environ.get("TTL_GEMM_HUGE", "2"))


def takes_huge(M, N, T):
    """csrc/gemm_huge.hip gemm_huge_applicable, for the shapes of this file: the launch family is switched on and the launch has
    256 x 256 tiles for TTL_GEMM_HUGE_MIN_FILL (default 85) percent of the CUs."""
    mode = huge_mode()
    if T == -1:
        if mode == 0 or os.environ.get("TTL_GEMM_HUGE_DGRAD", "0") == "0":
            return False
    elif not (mode == 1 or (mode == 2 and T > 0) or (mode == 3 and T == 0)):
        return False
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    tiles = (M + 255) // 256 * (N // 256)
    return tiles * 100 >= cus * int(os.environ.get("TTL_GEMM_HUGE_MIN_FILL", "85"))


def check(lib, prec, M, N, K, T, lda_pad=0, with_bias=True):
    """lda_pad: extra columns in the A buffer behind the K the product reads (the episode's K-extended operand buffers); with_bias=False:
    null bias pointer."""
    huge_on = takes_huge(M, N, T)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(M + N + K + T)
    abuf = torch.randn(M, K + lda_pad, generator=g).to(TDT[prec]).cuda()
    a = abuf[:, :K]
    b = (torch.randn(N, K, generator=g) * 0.05).to(TDT[prec]).cuda()
    bias = torch.randn(N, generator=g).cuda() if with_bias else None
    Mp = (M + 1279) // 1280 * 1280 + 320
    c = torch.full((Mp, N), 7.0, device="cuda", dtype=TDT[prec])
    c2 = None if T > 0 else torch.full((Mp, N), 7.0, device="cuda", dtype=TDT[prec])
    if T == -1:       # the saved pre-activation of the MLP dgrad form
        with_bias, bias = False, None
        c2[:M] = (torch.randn(M, N, generator=g) * 1.5).to(TDT[prec]).cuda()
        u_in = c2.clone()
    rc = lib.ttl_gemm_nt_fused(P(abuf), K + lda_pad, P(b), K, P(c), N, P(c2), N, M, N, K, P(bias), T, Mp, stream)
    assert rc == 0, (rc, lib.ttl_last_error())
    torch.cuda.synchronize()
    want = a.float() @ b.float().t() + (bias if with_bias else 0.0)
    tol = OUT_TOL[prec]
    if T == -1:
        u = u_in[:M].float()
        sg = torch.sigmoid(1.702 * u)
        assert max_rel(c[:M].float().cpu().numpy(), (want * (sg * (1.0 + 1.702 * u * (1.0 - sg)))).cpu().numpy()) < tol
        assert torch.equal(c2, u_in)           # u is read only
    elif T:
        views, D = M // T, N // 3
        got = c.reshape(-1)[: views * T * N].float().reshape(views, 3, D // 64, T, 64)
        ref = want[: views * T].reshape(views, T, 3, D // 64, 64).permute(0, 2, 3, 1, 4)
        assert max_rel(got.cpu().numpy(), ref.cpu().numpy()) < tol
        if M % T:     # the rows of a trailing partial view sit at the same formula
            tail = want[views * T:]
            gt = c.reshape(-1)[views * T * N:(views + 1) * T * N].float().reshape(3, D // 64, T, 64)[:, :, : M - views * T]
            assert max_rel(gt.cpu().numpy(), tail.reshape(-1, 3, D // 64, 64).permute(1, 2, 0, 3).cpu().numpy()) < tol
    else:
        assert max_rel(c2[:M].float().cpu().numpy(), want.cpu().numpy()) < tol
        assert max_rel(c[:M].float().cpu().numpy(), (want * torch.sigmoid(1.702 * want)).cpu().numpy()) < tol
    # nothing behind the rows a launch may store: row M on the 256 x 256 kernel (range-checked stores), round_up(M, 160) otherwise
    top = M if huge_on else ((M + 127) // 128 * 128 if T == -1 else (M + 159) // 160 * 160)
    if T > 0:
        top = (top + T - 1) // T * T        # head-major: whole views
    assert (c[top:].float() == 7.0).all()
    if c2 is not None and T == 0:
        assert (c2[top:].float() == 7.0).all()


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ttl-test-time-low-rank-adaptation_amd"))
    from ttl_amd import _lib
    prec = sys.argv[1]
    lib = _lib.load(prec)
    if prec != "experiments" and os.environ.get("TTL_GEMM_HUGE_DGRAD", "0") != "0":
        raise SystemExit("TTL_GEMM_HUGE_DGRAD is a closed experiment: only the experiments build reads it")
    for shp in SHAPES:
        check(lib, prec, *shp)
    check(lib, prec, 12608, 2304, 768, 197, lda_pad=64)
    check(lib, prec, 12608, 2304, 768, 197, with_bias=False)
    check(lib, prec, 12608, 3072, 768, 0, lda_pad=64, with_bias=False)
    print("ok", prec, "mode", huge_mode(), len(SHAPES), "on gemm_huge:", sum(takes_huge(M, N, T) for M, N, K, T in SHAPES))
