#!/usr/bin/env python3
"""Does any product kernel let a memory instruction write into registers an in-flight MFMA still owns?

Round 4 found the hazard class live (profiles/r04_experiments.txt r04s-x): in an experimental LayerNorm-fold epilogue hipcc placed a
`ds_read_b128` whose DESTINATION was the SrcC / vDst accumulator block of an MFMA issued three instructions earlier, inside the last
K-step's MFMA burst — the result differed from run to run in lanes 48-63 of those registers.  The matrix pipe reads SrcC and writes
vDst over several passes after issue; LDS / VMEM data returns asynchronously and is not ordered against those passes by the
`s_waitcnt` counters the kernels manage by hand.  The shipped kernels never do this (their epilogues read nothing from LDS and the
fragment reads target the A / B operand registers) — this scan keeps it that way.

It compiles the product kernel files of BOTH operand builds to gfx950 ISA and looks at every DS read / VMEM load (VGPR destination;
`global_load_lds` has none) whose destination registers overlap the vDst or SrcC range of an MFMA issued at most WINDOW
instructions earlier in the same straight-line stretch (labels and branches end a stretch; `s_nop` / `s_waitcnt` count as
instructions, which errs on the reporting side).  Two classes:
  * vDst overlap — the load's data and the MFMA's result race for the same registers (write after write): must be ZERO, exit 1.
  * SrcC-only overlap (vDst elsewhere) — what round 4 suspected.  The scan's own finding (round 5): hipcc's register allocator
    produces this pattern ~370 times per build in the SHIPPED gemm.hip / gemm_big.hip main loops (it rotates accumulator blocks:
    `v_mfma v[52:55], a, b, v[76:79]` frees v[76:79] at issue and a fragment read two instructions later takes them), and those
    kernels are bitwise repeatable over 2 000 episodes per test run (tests/test_gpu_race_screen.py).  A retired SrcC block is read
    in the MFMA's first passes, tens of cycles before any LDS / VMEM data can return; this class is reported, not failed.  So SrcC
    reuse alone is not what broke the round-4 experiment (its failing read sat in the EPILOGUE's last MFMA burst, where the
    accumulators are consumed by VALU right after — a different liveness situation); the count is printed so that a change of it
    is visible in review.

    python tools/scan_mfma_srcc_reuse.py            (CPU only; needs hipcc; exit 1 on a hit)      run by tests/test_cabi_cpu.py
"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "ttl-test-time-low-rank-adaptation_amd", "csrc")
FILES = ["gemm_big.hip", "gemm_huge.hip", "gemm.hip", "attention.hip", "lora.hip", "head_loss.hip", "elementwise.hip", "views.hip"]
BUILDS = {"bf16": [], "fp16": ["-DTTL_OPERAND_FP16"]}
WINDOW = 4

REG = re.compile(r"\b([av])(?:\[(\d+):(\d+)\]|(\d+))")


def regs(tok):
    """'v[4:7]' / 'a12' -> (file, lo, hi) or None"""
    m = REG.search(tok)
    if not m:
        return None
    if m.group(2) is not None:
        return m.group(1), int(m.group(2)), int(m.group(3))
    return m.group(1), int(m.group(4)), int(m.group(4))


def overlap(a, b):
    return a and b and a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2]


def operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    return parts[0], ([t.strip() for t in parts[1].split(",")] if len(parts) > 1 else [])


def scan(asm_path):
    hits, n_mfma = [], 0
    func = "?"
    recent = []          # (age, vdst, srcc, text) of MFMAs in the current straight-line stretch
    for raw in open(asm_path):
        line = raw.rstrip()
        s = line.strip()
        if not s or s.startswith((";", "//", ".")) and not s.startswith(".LBB"):
            continue
        if re.match(r"^_Z\w+:", s):
            func, recent = s[:-1], []
            continue
        if re.match(r"^\.LBB\d+_\d+:", s) or s.endswith(":"):
            recent = []
            continue
        op, ops = operands(s)
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
            recent = []
            continue
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            n_mfma += 1
            vdst = regs(ops[0]) if ops else None
            srcc = regs(ops[3]) if len(ops) > 3 else None
            # an earlier MFMA whose result block this one takes as SrcC has completed by the time this one reads it (the matrix pipe
            # orders dependent MFMAs itself): its vDst is no longer "in flight" — from here on the block is this MFMA's SrcC
            recent = [(a + 1, (None if overlap(d, srcc) else d), c, t) for a, d, c, t in recent if a + 1 <= WINDOW]
            recent.append((0, vdst, srcc, s))
            continue
        is_ds = op.startswith(("ds_read", "ds_load")) or "ds_read_tr" in op or "ds_bpermute" in op
        is_vm = op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")) and "lds" not in op
        if (is_ds or is_vm) and ops:
            dst = regs(ops[0])
            for age, d, c, t in recent:
                if overlap(dst, d):
                    hits.append(("vdst", func, s, t, age + 1))       # write-after-write against the MFMA's own result block
                elif overlap(dst, c):
                    hits.append(("srcc", func, s, t, age + 1))       # write-after-read: the accumulator-in block, dead after issue
        recent = [(a + 1, d, c, t) for a, d, c, t in recent if a + 1 <= WINDOW]
    return hits, n_mfma


def compile_one(args):
    build, flags, f, tmp = args
    out = os.path.join(tmp, f"{build}_{f[:-4]}.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, f] + flags, cwd=SRC,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    return build, f, out


def main():
    tmp = tempfile.mkdtemp()
    jobs = [(b, fl, f, tmp) for b, fl in BUILDS.items() for f in FILES]
    verbose = "-v" in sys.argv
    n_vdst = n_srcc = 0
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for build, f, out in ex.map(compile_one, jobs):
            hits, n = scan(out)
            vd = [h for h in hits if h[0] == "vdst"]
            sc = [h for h in hits if h[0] == "srcc"]
            n_vdst += len(vd); n_srcc += len(sc)
            print(f"{build:5s} {f:18s} {n:6d} MFMA instructions: {len(vd)} load(s) into an in-flight vDst, {len(sc)} into a retired SrcC block")
            for kind, func, ld, mf, dist in vd + (sc[:5] if verbose else []):
                print(f"    [{kind}] {re.sub(r'_ZN12_GLOBAL__N_1', '', func)[:90]}\n        {ld}\n        <- {dist} instruction(s) after: {mf}")
    print(f"loads into an in-flight MFMA's vDst (must be 0): {n_vdst}")
    print(f"loads into the SrcC block of an MFMA issued <= {WINDOW} instructions earlier, vDst elsewhere (informational): {n_srcc}")
    return 1 if n_vdst else 0


if __name__ == "__main__":
    sys.exit(main())
