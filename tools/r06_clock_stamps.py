#!/usr/bin/env python3
"""In-kernel clock of the two big-M GEMM kernels (MI355X_MICROARCH.md DVFS give-back item 6): diagnostic libraries built with
-DTTL_CLOCK_STAMPS (tools/hip_variant.sh gemm_huge TTL_CLOCK_STAMPS=1; ... gemm_big TTL_CLOCK_STAMPS=1) stamp s_memtime / s_memrealtime
once per workgroup around the whole kernel and around the K loop of its first tile; clock = d s_memtime / d s_memrealtime x 100 MHz, median
over the workgroups of the LAST launch after >= 2.5 s of back-to-back launches on random operands (six rotating operand sets: cold inputs).
The product libraries execute no stamp.

    python tools/r06_clock_stamps.py            # -> stdout; tools/r06_clock_stamps.sh tees it into gpurun_out/r06/inkernel_clock.txt
"""
import ctypes as C
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd"))
DIAG = os.path.join(ROOT, "tools", "_diag")
# (label, M, N, K, T): T > 0 head-major q/k/v, 0 fc1 (gelu + pre-activation), -1 MLP dgrad
RUNS = {"huge": [("q/k/v B/16 64 views (gemm_huge<101>)", 12608, 2304, 768, 197), ("q/k/v L/14 64 views (gemm_huge<101>)", 16448, 3072, 1024, 257)],
        "big": [("fc1 B/16 64 views (gemm_big<5,3,GELU+u>)", 12608, 3072, 768, 0), ("MLP dgrad B/16 (gemm_big<4,3,105>)", 12608, 3072, 768, -1),
                ("q/k/v B/16 on gemm_big (TTL_GEMM_HUGE=0)", 12608, 2304, 768, 197)]}
SLOTS = 2048


def child(which):
    import numpy as np
    import torch
    from ttl_amd import _lib
    lib = _lib.load("fp16")
    raw = C.CDLL(_lib.LIB_PATHS["fp16"])
    fn = getattr(raw, f"ttl_diag_clock_stamps_{which}")
    fn.argtypes, fn.restype = [C.c_void_p], C.c_int
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, M, N, K, T in RUNS[which]:
        if which == "big" and T > 0 and os.environ.get("TTL_GEMM_HUGE") != "0":
            continue
        if which == "big" and T <= 0 and os.environ.get("TTL_GEMM_HUGE") == "0":
            continue
        Mp = (M + 1279) // 1280 * 1280 + 320
        dt = torch.float16
        sets = [(torch.randn(M, K, device="cuda").to(dt), (torch.randn(N, K, device="cuda") * 0.05).to(dt),
                 torch.empty(Mp, N, device="cuda", dtype=dt), None if T > 0 else torch.randn(Mp, N, device="cuda").to(dt)) for _ in range(6)]
        bias = torch.randn(N, device="cuda")

        def launch(i):
            a, b, c, c2 = sets[i % 6]
            assert lib.ttl_gemm_nt_fused(P(a), K, P(b), K, P(c), N, P(c2), N, M, N, K, None if T == -1 else P(bias), T, Mp, s) == 0
        t0, n = time.time(), 0
        while time.time() - t0 < 2.5:
            for _ in range(200):
                launch(n); n += 1
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200):
            launch(n + i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        buf = np.zeros((SLOTS, 8), dtype=np.uint64)
        assert fn(buf.ctypes.data) == 0
        live = buf[(buf[:, 3] > buf[:, 1])]
        whole = (live[:, 2] - live[:, 0]).astype(np.float64) / (live[:, 3] - live[:, 1]).astype(np.float64) * 100.0
        kl = live[(live[:, 7] > live[:, 5])]
        kloop = (kl[:, 6] - kl[:, 4]).astype(np.float64) / (kl[:, 7] - kl[:, 5]).astype(np.float64) * 100.0
        wall_us = (live[:, 3] - live[:, 1]).astype(np.float64) / 100.0
        kcyc = (kl[:, 6] - kl[:, 4]).astype(np.float64)
        fl = 2.0 * M * N * K
        print(f"{name:46s} {us:6.1f} us/launch = {fl / us / 1e6:6.0f} TFLOP/s | workgroups {len(live):4d}, resident {np.median(wall_us):5.1f} us | "
              f"clock whole kernel {np.median(whole):5.0f} MHz (p10 {np.percentile(whole, 10):.0f}, p90 {np.percentile(whole, 90):.0f}) | "
              f"first tile's K loop {np.median(kloop):5.0f} MHz, {np.median(kcyc):7.0f} cycles for {K // 64} K-steps = {np.median(kcyc) / (K // 64):5.0f} per step", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    print("In-kernel clock of the big-M GEMM kernels, fp16 operands, random data, >= 2.5 s of back-to-back launches before the stamped one\n"
          "(diagnostic -DTTL_CLOCK_STAMPS builds; 2.5 PF dense = 2.4 GHz x 256 CUs x 4 SIMDs x 1024 FLOP/cycle/CU-SIMD-quarter)", flush=True)
    for which, lib, env in (("huge", "libttl_hip_fp16_gemm_huge_TTL_CLOCK_STAMPS_1.so", {}), ("big", "libttl_hip_fp16_gemm_big_TTL_CLOCK_STAMPS_1.so", {}),
                            ("big", "libttl_hip_fp16_gemm_big_TTL_CLOCK_STAMPS_1.so", {"TTL_GEMM_HUGE": "0"})):
        e = dict(os.environ, TTL_HIP_LIB_FP16=os.path.join(DIAG, lib), TTL_GEMM_HUGE_MIN_FILL="0", **env)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", which], env=e)
        if r.returncode:
            sys.exit(r.returncode)
