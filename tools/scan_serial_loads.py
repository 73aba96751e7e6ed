#!/usr/bin/env python3
"""Which kernels of the library still have load loops that run as serial memory round trips?

hipcc compiles `for (...) { v = *p; use(v); }` with a thread-dependent trip count to `global_load; s_waitcnt vmcnt(0); ...; branch`:
one load in flight, one full round trip per iteration.  This compiles every csrc/*.hip to gfx950 ISA and lists the loops (back
edges) that contain at most four global loads and a `s_waitcnt vmcnt(0)`: (loads, vmcnt(0) waits, instructions) per loop.

    python tools/scan_serial_loads.py            (CPU only; needs hipcc)
"""
import glob, os, re, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd", "csrc")
tmp = tempfile.mkdtemp()
for f in sorted(glob.glob(os.path.join(SRC, "*.hip"))):
    if os.path.basename(f) in ("api.hip",):
        continue
    out = os.path.join(tmp, os.path.basename(f)[:-4] + ".s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, f], cwd=SRC,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    s = open(out).read().split("\n")
    i = 0
    while i < len(s):
        if s[i].startswith("_Z") and ":" in s[i]:
            name = s[i].split(":")[0]
            try:
                en = next(j for j in range(i, len(s)) if s[j].startswith(".Lfunc_end"))
            except StopIteration:
                break
            body = s[i:en]
            labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
            hits = []
            for k, l in enumerate(body):
                m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
                if m and m.group(1) in labels and labels[m.group(1)] < k:
                    blk = body[labels[m.group(1)]:k]
                    nl = sum(1 for x in blk if re.search(r"\b(global_load|buffer_load)", x) and "lds" not in x)
                    w0 = sum(1 for x in blk if "s_waitcnt vmcnt(0)" in x)
                    if nl and w0 and nl <= 4:
                        hits.append((nl, w0, len(blk)))
            if hits:
                print(os.path.basename(f), re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:70], hits)
            i = en
        i += 1
