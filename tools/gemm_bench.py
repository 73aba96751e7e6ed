"""Per-shape timing of the bf16 GEMM kernel through the C ABI (run on the GPU box)."""
import sys, os, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd")]
from ttl_amd import _lib
lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())
shapes = [(12608, 2304, 768), (12608, 2304, 832), (12608, 768, 768), (12608, 3072, 768), (12608, 768, 3072), (12544, 768, 768),
          (12608, 768, 2368), (197, 2304, 768), (197, 768, 3072), (16448, 3072, 1024), (16448, 1024, 4096)]
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    c = torch.empty(M, N, device="cuda")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        lib.ttl_gemm_nt(P(a), K, P(b), K, P(c), N, M, N, K, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        lib.ttl_gemm_nt(P(a), K, P(b), K, P(c), N, M, N, K, s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / it
    print(f"M={M:6d} N={N:5d} K={K:5d}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TFLOP/s  tiles={((M+127)//128)*(N//128)}")
