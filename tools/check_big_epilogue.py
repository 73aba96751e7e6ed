"""Static check of an assumption gemm_big.hip's counted waits rest on: between the K-tiles a block prefetches for its NEXT tile and
the waits that retire them, the epilogue of the current tile issues AT LEAST 4*MT store instructions per wave (wait_tiles1_st
allows for that many unacknowledged stores: vmcnt counts loads and stores together, in order).  If hipcc ever merged or elided
stores in some epilogue, K-tile 0/1 of the next tile could be read before it has landed — silently.  This script compiles
gemm_big.hip to ISA (or reads a .s given as argv[1]) and counts the global_store instructions of every gemm_big_kernel<MT,STAGES,EPI>.

The same holds for gemm_huge.hip: behind a tile its `s_waitcnt vmcnt(63)` retires the next tile's bias and first K-tiles only if at
least 63 - 8 = 55 store instructions (and the 8 pieces of K-tile 1) were issued after them; every epilogue has 64 (GELU + u: 128).

    python tools/check_big_epilogue.py [file.s]      -> exit code 0 when every kernel has >= 4*MT (gemm_huge: >= 64) stores
"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd", "csrc")


def store_counts(path):
    cnt, cur = collections.Counter(), None
    for l in open(path):
        m = re.match(r"^(_ZN12_GLOBAL__N_115gemm_big_kernelILi(\d+)ELi(\d+)ELi(\d+)E\w+):", l)
        h = re.match(r"^(_ZN12_GLOBAL__N_116gemm_huge_kernelILi(\d+)E\w+):", l)
        if m:
            cur = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
            cnt[cur] += 0
        elif h:
            cur = ("huge", int(h.group(2)))
            cnt[cur] += 0
        elif l.startswith(".Lfunc_end"):
            cur = None
        elif cur and re.match(r"\s+(global|buffer|flat)_store", l):
            cnt[cur] += 1
    return cnt


def main(argv):
    if len(argv) > 1:
        path = argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(), "gemm_big.s")
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", path,
                               os.path.join(CSRC, "gemm_big.hip")], cwd=CSRC, stderr=subprocess.DEVNULL)
    cnt = store_counts(path)
    if len(argv) <= 1:
        path2 = os.path.join(os.path.dirname(path), "gemm_huge.s")
        subprocess.check_call([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", path2,
                               os.path.join(CSRC, "gemm_huge.hip")], cwd=CSRC, stderr=subprocess.DEVNULL)
        cnt.update(store_counts(path2))
    need = lambda k: 64 if k[0] == "huge" else 4 * k[0]
    bad = {k: v for k, v in cnt.items() if v < need(k)}
    for k, v in sorted(cnt.items(), key=str):
        name = f"gemm_huge_kernel<{k[1]}>" if k[0] == "huge" else "gemm_big_kernel<%d,%d,%d>" % k
        print(f"{name}: {v} store instructions (needs >= {need(k)})")
    if not any(k[0] != "huge" for k in cnt) or (len(argv) <= 1 and not any(k[0] == "huge" for k in cnt)) or bad:
        print("FAILED:", bad or "kernels not found")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
