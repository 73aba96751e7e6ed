"""Per-shape timing of the big-M GEMM with the episode's epilogues and a cache regime close to the episode's:
R rotating (A, C) buffer sets, so operands come from the Infinity Cache / HBM rather than from L2 (a back-to-back loop on
one buffer ranks scheduling choices the other way round).  Libraries: the product build and whatever
tools/_diag/libttl_hip_*.so exist (ablations: timing only).   python tools/gemm_shapes_bench.py [lib-substring ...]
Environment switches of the product library (TTL_GEMM_BIG, TTL_GEMM_BIG_MT, ...) apply per process."""
import sys, os, glob, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {"product": os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd/ttl_amd/libttl_hip.so")}
for p in sorted(glob.glob(os.path.join(ROOT, "tools/_diag/libttl_hip_*.so"))):
    libs[os.path.basename(p)[len("libttl_hip_"):-3]] = p
want = sys.argv[1:]
if want:
    libs = {k: v for k, v in libs.items() if any(w in k for w in want)}
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)
M = 12608
# (name, N, K, epi): epi 1 = operand out + bias, 2 = fp32 resid + bias, 3 = quick_gelu operand out, 0 = fp32
shapes = [("qkv", 2304, 768, 1), ("qkv_lora", 2304, 832, 1), ("out_proj", 768, 768, 2), ("fc1", 3072, 768, 3), ("fc2", 768, 3072, 2),
          ("dx1", 768, 2368, 0), ("dO", 768, 768, 1)]
R = int(os.environ.get("ROTATE", 4))
Mp = (M + 1279) // 1280 * 1280
torch.manual_seed(0)
bufs = {}
for (nm, N, K, epi) in shapes:
    sets = []
    for r in range(R):
        a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        c = torch.empty(Mp, N, device="cuda", dtype=torch.float32 if epi in (0, 2) else torch.bfloat16)
        res = torch.randn(Mp, N, device="cuda") if epi == 2 else None
        sets.append((a, c, res))
    b = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda")
    bufs[nm] = (sets, b, bias)
ref = {}
for name, path in libs.items():
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    f = lib.ttl_gemm_nt_epi
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                  C.c_int, C.c_int, C.c_void_p]
    row = []
    for (nm, N, K, epi) in shapes:
        sets, b, bias = bufs[nm]
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        def call(r):
            a, c, res = sets[r % R]
            rc = f(P(a), K, P(b), K, P(c), N, M, N, K, epi, P(bias), P(res), N, Mp, s)
            assert rc == 0, rc
        for i in range(R):
            call(i)
        torch.cuda.synchronize()
        if name == "product":
            ref[nm] = sets[0][1][:M].float().clone()
        it = 40
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(it):
            call(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / it
        row.append(f"{nm} {us:6.1f}us {2*M*N*K/us/1e6:5.0f}TF")
    print(f"{name:12s}", " | ".join(row), flush=True)
if "product" in libs and os.environ.get("CHECK", "1") == "1":
    # product results against torch (fp32 matmul of the bf16 operands)
    for (nm, N, K, epi) in shapes:
        sets, b, bias = bufs[nm]
        a, c, res = sets[0]
        want_ = a.float() @ b.float().t() + (bias if epi != 0 else 0)
        if epi == 2: want_ = want_ + res[:M]
        if epi == 3: want_ = want_ * torch.sigmoid(1.702 * want_)
        err = (ref[nm] - want_).abs().max().item() / want_.abs().max().item()
        print(f"check {nm}: max rel err {err:.2e}")
