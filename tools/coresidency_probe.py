"""Which kernels of the path really run on a CU that a big-M GEMM block holds?  One stream loops the fc1-shaped big-M GEMM (one
persistent block per CU, 157 of 160 KiB LDS, 8 waves x 219 VGPRs), a second stream loops a candidate kernel; both are timed alone and
together.  together ~= max(alone) -> the candidate runs underneath the GEMM; together ~= sum(alone) -> it waits for a free CU.
   python tools/coresidency_probe.py            (on the GPU box)"""
import os, sys, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.environ.get("TTL_PROBE_LIB", os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd/ttl_amd/libttl_hip.so")), mode=C.RTLD_LOCAL)
P = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)
M, D, F, H, T, NV = 12608, 768, 3072, 12, 197, 64
Mp = (M + 1279) // 1280 * 1280
torch.manual_seed(0)
a = torch.randn(M, D, device="cuda").to(torch.bfloat16)
w = (torch.randn(F, D, device="cuda") * 0.05).to(torch.bfloat16)
cbuf = torch.empty(Mp, F, device="cuda", dtype=torch.bfloat16)
bias = torch.randn(F, device="cuda")
x = torch.randn(M, D, device="cuda"); y = torch.empty_like(x); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
qkv = (torch.randn(M, 3 * D, device="cuda") * 0.5).to(torch.bfloat16); ao = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
lse = torch.empty(NV * H * T, device="cuda")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
hA, hB = C.c_void_p(sA.cuda_stream), C.c_void_p(sB.cuda_stream)
lib.ttl_gemm_nt_epi.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_int, C.c_void_p]
lib.ttl_layernorm_f32.argtypes = [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_float, C.c_void_p]
lib.ttl_attention_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]


def gemm(h):
    assert lib.ttl_gemm_nt_epi(P(a), D, P(w), D, P(cbuf), F, M, F, D, 3, P(bias), None, F, Mp, h) == 0


cands = {
    "layernorm (ln_fwd_kernel<1>, 42 VGPRs, no LDS)": lambda h: lib.ttl_layernorm_f32(P(x), P(g), P(b), P(y), P(mean), P(rstd), M, D, 1e-5, h),
    "attention forward (140 KiB LDS)": lambda h: lib.ttl_attention_fwd(P(qkv), P(ao), P(lse), NV, T, H, 0, h),
}
spin_path = os.path.join(ROOT, "tools/_diag/libprobe_spin.so")      # hipcc -shared tools/coresidency_spin.hip
if os.path.exists(spin_path):
    spin = C.CDLL(spin_path, mode=C.RTLD_LOCAL)
    spin.ttl_probe_spin.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    for vg in (32, 64, 96):
        cands[f"ALU spin, {vg} VGPRs, no memory traffic, 1024 blocks of 256"] = (lambda v: (lambda h: spin.ttl_probe_spin(1024, v, 4000, h)))(vg)


def timed(fa, na, fb, nb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sA.wait_event(e0); sB.wait_event(e0)
    # interleave the enqueues so that neither stream runs dry while the host is still feeding the other
    for i in range(max(na, nb)):
        if i < na: fa(hA)
        if i < nb: fb(hB)
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sA); eb.record(sB)
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


NG = 200
for _ in range(2):
    timed(gemm, 20, gemm, 0)
tg = timed(gemm, NG, gemm, 0)
print(f"big-M GEMM alone: {tg / NG * 1e3:.1f} us per launch")
for name, f in cands.items():
    timed(gemm, 0, f, 20)
    t1 = timed(gemm, 0, f, 50)
    nb = max(1, int(round(tg / (t1 / 50))))          # as many candidate launches as fill the GEMM loop's time
    tb = timed(gemm, 0, f, nb)
    both = timed(gemm, NG, f, nb)
    print(f"{name}: alone {tb / nb * 1e3:.1f} us x {nb} = {tb:.2f} ms; GEMM loop {tg:.2f} ms; together {both:.2f} ms "
          f"(max {max(tg, tb):.2f}, sum {tg + tb:.2f}) -> hidden share {(tg + tb - both) / min(tg, tb):.2f}")

# ---- do two streams run at the same time at all?  Two half-chip spins (128 blocks of 256 threads: at most one block per CU on half the CUs)
if os.path.exists(spin_path):
    half = lambda h: spin.ttl_probe_spin(128, 32, 40000, h)
    timed(half, 10, half, 10)
    ta = timed(half, 40, half, 0)
    tb2 = timed(half, 40, half, 40)
    print(f"two half-chip spins (128 blocks each): one stream 40 launches {ta:.2f} ms; two streams 40 + 40 launches {tb2:.2f} ms "
          f"(concurrent streams -> {ta:.2f}, serial -> {2 * ta:.2f})")
    # a GEMM that leaves CUs free (out_proj shape: 237 tiles) beside a spin small enough for the free CUs
    w2 = (torch.randn(D, D, device="cuda") * 0.05).to(torch.bfloat16); c2 = torch.empty(Mp, D, device="cuda", dtype=torch.bfloat16); b2 = torch.randn(D, device="cuda")
    def gemm_small_n(h):
        assert lib.ttl_gemm_nt_epi(P(a), D, P(w2), D, P(c2), D, M, D, D, 1, P(b2), None, D, Mp, h) == 0
    few = lambda h: spin.ttl_probe_spin(16, 32, 12000, h)
    timed(gemm_small_n, 20, few, 20)
    t_g = timed(gemm_small_n, 200, few, 0); t_f = timed(gemm_small_n, 0, few, 200); t_b = timed(gemm_small_n, 200, few, 200)
    print(f"out_proj-shaped GEMM (237 of 256 CUs) x 200: {t_g:.2f} ms; 16-block spin x 200: {t_f:.2f} ms; together {t_b:.2f} ms")
