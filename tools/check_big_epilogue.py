"""Static check of an assumption gemm_big.hip's counted waits rest on: between the K-tiles a block prefetches for its NEXT tile and
the waits that retire them, the epilogue of the current tile issues AT LEAST 4*MT store instructions per wave (wait_tiles1_st
allows for that many unacknowledged stores: vmcnt counts loads and stores together, in order).  If hipcc ever merged or elided
stores in some epilogue, K-tile 0/1 of the next tile could be read before it has landed — silently.  This script compiles
gemm_big.hip to ISA (or reads a .s given as argv[1]) and counts the global_store instructions of every gemm_big_kernel<MT,STAGES,EPI>.

    python tools/check_big_epilogue.py [file.s]      -> exit code 0 when every kernel has >= 4*MT stores
"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd", "csrc")


def store_counts(path):
    cnt, cur = collections.Counter(), None
    for l in open(path):
        m = re.match(r"^(_ZN12_GLOBAL__N_115gemm_big_kernelILi(\d+)ELi(\d+)ELi(\d+)E\w+):", l)
        if m:
            cur = (int(m.group(2)), int(m.group(3)), int(m.group(4)))
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
    bad = {k: v for k, v in cnt.items() if v < 4 * k[0]}
    for (mt, st, epi), v in sorted(cnt.items()):
        print(f"gemm_big_kernel<{mt},{st},{epi}>: {v} store instructions (needs >= {4 * mt})")
    if not cnt or bad:
        print("FAILED:", bad or "no gemm_big_kernel found")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
