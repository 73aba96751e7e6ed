"""Times the product GEMM kernel and its ablations (tools/gemm_ablate.sh) on the episode's shapes."""
import sys, os, ctypes as C
os.environ["TTL_GEMM_PADDED"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {"product": os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd/ttl_amd/libttl_hip.so")}
for n, nm in ((1, "no-DMA"), (2, "no-MFMA"), (3, "no-LDS-reads"), (4, "no-barrier")):
    p = os.path.join(ROOT, f"tools/_diag/libttl_hip_diag{n}.so")
    if os.path.exists(p):
        libs[nm] = p
P = lambda t: C.c_void_p(t.data_ptr())
shapes = [(12800, 2304, 768), (12800, 768, 768), (12800, 3072, 768), (12800, 768, 3072)]
bufs = {}
for (M, N, K) in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    c = torch.empty(M + 320, N, device="cuda")
    bufs[(M, N, K)] = (a, b, c)
for name, path in libs.items():
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    f = lib.ttl_gemm_nt
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    row = []
    for (M, N, K) in shapes:
        a, b, c = bufs[(M, N, K)]
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(3):
            rc = f(P(a), K, P(b), K, P(c), N, M, N, K, s)
            assert rc == 0, rc
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 30
        e0.record()
        for _ in range(it):
            f(P(a), K, P(b), K, P(c), N, M, N, K, s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / it
        row.append(f"{us:7.1f}us {2*M*N*K/us/1e6:6.0f}TF")
    print(f"{name:14s}", " | ".join(row))
print("shapes:", shapes)
