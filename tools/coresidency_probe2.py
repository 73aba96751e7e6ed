"""Which resource of a big-M-GEMM-shaped block keeps other kernels off its CU?  Stand-ins with the GEMM's footprint (256 blocks of 512
threads; 32 or 224 VGPRs; 0 / 64 / 157 KiB of LDS; pure ALU) on one stream, a light spin (1024 blocks of 256 threads, 32 VGPRs, no LDS)
on another.   python tools/coresidency_probe2.py   (GPU box; needs tools/_diag/libprobe_spin.so = hipcc -shared tools/coresidency_spin.hip)"""
import os, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spin = C.CDLL(os.path.join(ROOT, "tools/_diag/libprobe_spin.so"), mode=C.RTLD_LOCAL)
spin.ttl_probe_spin.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
spin.ttl_probe_spin_big.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
torch.zeros(1, device="cuda")
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
hA, hB = C.c_void_p(sA.cuda_stream), C.c_void_p(sB.cuda_stream)


def timed(fa, na, fb, nb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); sA.wait_event(e0); sB.wait_event(e0)
    for i in range(max(na, nb)):
        if i < na: assert fa(hA) == 0
        if i < nb: assert fb(hB) == 0
    ea, eb = torch.cuda.Event(), torch.cuda.Event()
    ea.record(sA); eb.record(sB)
    torch.cuda.current_stream().wait_event(ea); torch.cuda.current_stream().wait_event(eb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


light = lambda h: spin.ttl_probe_spin(1024, 32, 2000, h)
timed(light, 20, light, 0)
tl = timed(light, 100, light, 0)
print(f"light spin (1024 x 256 threads, 32 VGPRs): {tl * 10:.1f} us per launch")
for blocks in (256, 128):
    for vg in (32, 224):
        for lds in (0, 64 * 1024, 157 * 1024):
            big = lambda h: spin.ttl_probe_spin_big(blocks, vg, lds, 4000, h)
            timed(big, 10, light, 10)
            tb = timed(big, 100, light, 0)
            nl = max(1, int(round(100 * tb / tl)))
            tl2 = timed(big, 0, light, nl)
            both = timed(big, 100, light, nl)
            print(f"big stand-in {blocks} blocks x 512 threads, {vg:3d} VGPRs, {lds // 1024:3d} KiB LDS: alone {tb:.2f} ms, light x {nl} alone {tl2:.2f} ms, "
                  f"together {both:.2f} ms  (max {max(tb, tl2):.2f}, sum {tb + tl2:.2f}) -> hidden share {(tb + tl2 - both) / min(tb, tl2):.2f}", flush=True)
