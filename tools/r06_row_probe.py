#!/usr/bin/env python3
"""Round 6 (r06f): tools/experiments/gemm_row.hip — the N = 768 projections with LayerNorm as the GEMM's epilogue — as a STANDALONE probe library
(built here:  python tools/r06_row_probe.py --build  -> tools/_diag/libttl_row_probe.so), checked against fp32 torch and timed on cold
operands against what the product runs today for the same work: ttl_gemm_nt_epi (resid + product + bias on gemm_huge.hip's 256 x 256
tiles = the in-flight choice, or gemm_big.hip's 160 x 256 = the choice of a context alone) + the LayerNorm launch (10.5 us, rocprof).

    python tools/r06_row_probe.py            # on the GPU box
"""
import ctypes as C
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "_diag", "libttl_row_probe.so")
sys.path.insert(0, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"))


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    for diag in (0, 1, 2, 4, 6, 7):       # timing-only ablations beside the real kernel (TTL_ROW_DIAG)
        out = LIB if diag == 0 else LIB.replace(".so", f"_diag{diag}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fvisibility=hidden", "-DTTL_OPERAND_FP16",
                               "-DTTL_ROW_PROBE", f"-DTTL_ROW_DIAG={diag}", "-I", CSRC, "-shared", os.path.join(ROOT, "tools", "experiments", "gemm_row.hip"), "-o", out])
        print("built", out)


def child(mode):
    import torch
    from ttl_amd import _lib
    lib = _lib.load("fp16")
    probe = C.CDLL(os.environ.get("TTL_ROW_LIB", LIB))
    probe.ttl_row_probe.restype = C.c_int
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(5)

    def row(a, b, bias, res, c, gamma, beta, y, mean, rstd, M, K, lda):
        rc = probe.ttl_row_probe(P(a), C.c_int(lda), P(b), C.c_int(K), C.c_int(M), C.c_int(K), P(bias), P(res), C.c_int(768), P(c), C.c_int(768), P(gamma), P(beta),
                                 C.c_float(1e-5), P(y), C.c_int(768), P(mean), P(rstd), s)
        assert rc == 0, rc

    def timeit(f, n=40):
        for i in range(6):
            f(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            f(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    if mode == "check":
        for M, K, with_res in ((12608, 768, True), (12608, 3072, True), (12608, 832, True), (1000, 768, False), (200, 448, True)):
            a = torch.randn(M, K, generator=g).half().cuda()
            b = (torch.randn(768, K, generator=g) * 0.05).half().cuda()
            bias = torch.randn(768, generator=g).cuda()
            res = (torch.randn(M, 768, generator=g) * 2).cuda() if with_res else None
            gamma, beta = (1 + 0.2 * torch.randn(768, generator=g)).cuda(), (0.1 * torch.randn(768, generator=g)).cuda()
            c = torch.full((M + 128, 768), 7.0, device="cuda")
            y = torch.full((M + 128, 768), 7.0, device="cuda", dtype=torch.float16)
            mean, rstd = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
            row(a, b, bias, res, c, gamma, beta, y, mean, rstd, M, K, K)
            torch.cuda.synchronize()
            want = a.float() @ b.float().t() + bias + (res if with_res else 0.0)
            ln = torch.nn.functional.layer_norm(want, (768,), gamma, beta, 1e-5)
            ec = float((c[:M] - want).abs().max() / want.abs().max())
            ey = float((y[:M].float() - ln).abs().max() / ln.abs().max())
            em = float((mean - want.mean(1)).abs().max())
            er = float((rstd - (want.var(1, unbiased=False) + 1e-5).rsqrt()).abs().max() / rstd.abs().max())
            clean = bool((c[M:] == 7.0).all() and (y[M:] == 7.0).all())
            print(f"check M={M} K={K} resid={with_res}: h max_rel {ec:.2e}  y max_rel {ey:.2e}  mean abs {em:.2e}  rstd rel {er:.2e}  rows past M untouched: {clean}", flush=True)
            assert ec < 2e-5 and ey < 2e-3 and em < 1e-4 and er < 1e-4 and clean
        return
    # ---- timing on rotating operand sets (cold inputs)
    M = 12608
    for name, K in (("out_proj B/16", 768), ("fc2 B/16", 3072)):
        sets = []
        for _ in range(6):
            sets.append(dict(a=torch.randn(M, K, device="cuda").half(), b=(torch.randn(768, K, device="cuda") * 0.05).half(), res=torch.randn(M, 768, device="cuda"),
                             c=torch.empty(M + 1280, 768, device="cuda"), y=torch.empty(M + 1280, 768, device="cuda", dtype=torch.float16)))
        bias = torch.randn(768, device="cuda"); gamma = torch.ones(768, device="cuda"); beta = torch.zeros(768, device="cuda")
        mean, rstd = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
        Mp = (M + 1279) // 1280 * 1280

        def new(i):
            d = sets[i % 6]
            row(d["a"], d["b"], bias, d["res"], d["c"], gamma, beta, d["y"], mean, rstd, M, K, K)

        def old(i):
            d = sets[i % 6]
            assert lib.ttl_gemm_nt_epi(P(d["a"]), K, P(d["b"]), K, P(d["c"]), 768, M, 768, K, 2, P(bias), P(d["res"]), 768, Mp, s) == 0
        if mode == "row":
            t = [timeit(new) for _ in range(5)]
            print(f"{name:14s} K={K:5d}  gemm_row_ln (GEMM + residual + LayerNorm, 99 workgroups) {os.environ.get('TTL_ROW_TAG', ''):28s} {statistics.median(t):6.1f} us ({min(t):.1f}-{max(t):.1f})", flush=True)
        else:
            t = [timeit(old) for _ in range(5)]
            print(f"{name:14s} K={K:5d}  ttl_gemm_nt_epi resid form [{mode}]                       {statistics.median(t):6.1f} us ({min(t):.1f}-{max(t):.1f})   (+ LayerNorm launch 10.5 us)", flush=True)


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    diag = lambda d, tag: ("row", {"TTL_ROW_LIB": LIB.replace(".so", f"_diag{d}.so"), "TTL_ROW_TAG": tag})
    runs = [("check", {}), ("row", {}), ("gemm_big 160x256", {"TTL_GEMM_HUGE_NARROW": "0"}), ("gemm_huge 256x256", {"TTL_GEMM_HUGE_NARROW": "1"}), ("row", {})]
    if "--ablate" in sys.argv:
        runs += [diag(1, "[no residual stream]"), diag(2, "[no LayerNorm epilogue]"), diag(4, "[no h stores]"), diag(6, "[no LN, no h stores]"), diag(7, "[K loops only]")]
    for mode, env in runs:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode], env=dict(os.environ, **env))
        if r.returncode:
            sys.exit(r.returncode)
