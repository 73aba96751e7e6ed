"""Reference point: the vendor library's 16-bit GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the episode's big shapes (M = 12 608,
the 64-view token count; L/14: 16 448), beside this build's kernel through ttl_gemm_nt_epi, both with operand-dtype output and no bias.
Inputs rotate over 6 buffer sets (~> L2, < Infinity Cache).     python tools/blas_reference_point.py [fp16|bf16]   (default fp16: the headline build)"""
import os, sys, ctypes as C
os.environ["TTL_GEMM_PADDED"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"))
from ttl_amd import _lib
PREC = sys.argv[1] if len(sys.argv) > 1 else "fp16"
DT = {"fp16": torch.float16, "bf16": torch.bfloat16}[PREC]
lib = _lib.load(PREC)
print("operand build:", PREC)
P = lambda t: C.c_void_p(t.data_ptr())
for (M, N, K) in [(12608, 2304, 768), (12608, 768, 768), (12608, 3072, 768), (12608, 768, 3072), (16448, 3072, 1024), (16448, 1024, 4096)]:
    sets = []
    for i in range(6):
        a = torch.randn(M, K, device="cuda").to(DT)
        b = (torch.randn(N, K, device="cuda") * 0.05).to(DT)
        sets.append((a, b, torch.empty(M + 320, N, device="cuda", dtype=DT), torch.empty(M, N, device="cuda", dtype=DT)))
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def ours(i):
        a, b, c, _ = sets[i % 6]; lib.ttl_gemm_nt_epi(P(a), K, P(b), K, P(c), N, M, N, K, 1, None, None, 0, M + 320, s)
    def blas(i):
        a, b, _, d = sets[i % 6]; torch.matmul(a, b.t(), out=d)
    res = []
    for f in (ours, blas):
        for i in range(6): f(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): f(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 30
        res.append(f"{us:7.1f} us {2*M*N*K/us/1e6:6.0f} TF")
    print(f"M={M} N={N} K={K}: this build ({PREC} out) {res[0]} | torch.matmul {PREC} out {res[1]}")
