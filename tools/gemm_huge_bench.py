#!/usr/bin/env python3
"""The wide, short-K big-M launches (q/k/v, fc1) one at a time on cold operands: csrc/gemm_huge.hip (256 x 256 tiles, four waves) against
csrc/gemm_big.hip (160 x 256, eight waves) and the vendor library, on one lease.

    python tools/gemm_huge_bench.py [fp16|bf16]

TTL_GEMM_HUGE is read once per process, so each setting runs in a child process (TTL_GEMM_HUGE=1 with the round-fill rule off: every
shape on gemm_huge.hip, 0: every shape on gemm_big.hip); six operand sets are rotated so that no launch finds its inputs in the caches."""
import ctypes as C
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"))
SHAPES = [("q/k/v  B/16", 12608, 2304, 768, 197), ("fc1    B/16", 12608, 3072, 768, 0), ("q/k/v  L/14", 16448, 3072, 1024, 257),
          ("fc1    L/14", 16448, 4096, 1024, 0), ("MLPdgr B/16", 12608, 3072, 768, -1), ("MLPdgr L/14", 16448, 4096, 1024, -1)]


def child(prec):
    import torch
    from ttl_amd import _lib
    lib = _lib.load(prec)
    dt = torch.float16 if prec == "fp16" else torch.bfloat16
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

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
    for name, M, N, K, T in SHAPES:
        Mp = (M + 1279) // 1280 * 1280 + 320
        sets = [(torch.randn(M, K, device="cuda").to(dt), (torch.randn(N, K, device="cuda") * 0.05).to(dt),
                 torch.empty(Mp, N, device="cuda", dtype=dt), None if T > 0 else torch.randn(Mp, N, device="cuda").to(dt)) for _ in range(6)]
        bias = torch.randn(N, device="cuda")
        bh = bias.to(dt)

        def ours(i):
            a, b, c, c2 = sets[i % 6]
            assert lib.ttl_gemm_nt_fused(P(a), K, P(b), K, P(c), N, P(c2), N, M, N, K, None if T == -1 else P(bias), T, Mp, s) == 0

        def vendor(i):
            a, b, _, _ = sets[i % 6]
            y = torch.nn.functional.linear(a, b, None if T == -1 else bh)
            if T <= 0:
                torch.nn.functional.gelu(y)        # (the library has no fused two-output / gelu' form: its second pass is part of what it costs)
        t_o = [timeit(ours) for _ in range(5)]
        t_v = [timeit(vendor) for _ in range(5)]
        print(f"{name}  M={M} N={N} K={K}  ttl_gemm_nt_fused {statistics.median(t_o):6.1f} us ({min(t_o):.1f}-{max(t_o):.1f})   "
              f"torch linear{'' if T > 0 else ' + gelu'} {statistics.median(t_v):6.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--child":
        child(sys.argv[1])
        sys.exit(0)
    prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    if prec == "fp16":      # TTL_GEMM_HUGE_DGRAD is a closed experiment: only the -DTTL_EXPERIMENTS build of the fp16 library reads it
        os.environ.setdefault("TTL_HIP_LIB_FP16", os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd", "ttl_amd", "libttl_hip_fp16_exp.so"))
    for mode, label in (("1", "gemm_huge.hip (256 x 256, four waves)"), ("0", "gemm_big.hip (160 x 256, eight waves)"), ("1", "gemm_huge.hip again")):
        print(f"---- TTL_GEMM_HUGE={mode}: {label}   [{prec}]", flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), prec, "--child"], env=dict(os.environ, TTL_GEMM_HUGE=mode, TTL_GEMM_HUGE_MIN_FILL="0", TTL_GEMM_HUGE_DGRAD="1"))
        if r.returncode:
            sys.exit(r.returncode)
