"""Why is the QKV / fc1 GEMM slower inside the episode than in tools/gemm_shapes_bench.py?  Same kernel, same shape, with the
episode's cache state emulated piece by piece: the A operand written by a producer kernel right before, the weight matrix
rotating over 12 layers (cold), one reused output buffer, a consumer reading C afterwards.  GEMM time by HIP events."""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd/ttl_amd/libttl_hip.so"))
f = lib.ttl_gemm_nt_epi; f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)
M, Mp = 12608, 12800
def run(name, N, K, epi, producer, cold_w, one_c, consumer, flush):
    nA = 1 if producer else 4
    As = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(nA)]
    src = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    Ws = [(torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16) for _ in range(12 if cold_w else 1)]
    Cs = [torch.empty(Mp, N, device="cuda", dtype=torch.bfloat16) for _ in range(1 if one_c else 4)]
    bias = torch.randn(N, device="cuda")
    junk = torch.empty(64 << 20, device="cuda", dtype=torch.float32) if flush else None
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(36)]
    for it in range(-6, 36):
        a, w, c = As[it % nA], Ws[it % len(Ws)], Cs[it % len(Cs)]
        if flush: junk.fill_(1.0)                    # 256 MiB through the caches
        if producer: a.copy_(src)                    # the LayerNorm's role: A written right before the GEMM
        if it >= 0: ev[it][0].record()
        assert f(P(a), K, P(w), K, P(c), N, M, N, K, epi, P(bias), None, 0, Mp, s) == 0
        if it >= 0: ev[it][1].record()
        if consumer: c[:M].float().sum()             # the attention kernel's role: C read right after
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)
    print(f"{name:8s} producer={int(producer)} cold_w={int(cold_w)} one_c={int(one_c)} consumer={int(consumer)} flush={int(flush)}: "
          f"median {ts[len(ts)//2]:6.1f} us  min {ts[0]:6.1f}", flush=True)
for (name, N, K, epi) in (("qkv", 2304, 768, 1), ("fc1", 3072, 768, 3), ("fc2", 768, 3072, 0)):
    run(name, N, K, epi, 0, 0, 0, 0, 0)
    run(name, N, K, epi, 1, 0, 0, 0, 0)
    run(name, N, K, epi, 0, 1, 0, 0, 0)
    run(name, N, K, epi, 0, 0, 1, 0, 0)
    run(name, N, K, epi, 0, 0, 1, 1, 0)
    run(name, N, K, epi, 1, 1, 1, 1, 0)
    run(name, N, K, epi, 0, 0, 0, 0, 1)
