"""Turn the scratch output of tools/collect_profiles.sh (gpurun_out/<run>/) into the tracked evidence files profiles/<tag>_*:
    python tools/assemble_profiles.py gpurun_out/r05_fp16 r05 [fp16]
(third argument: a suffix naming the operand build the files were measured on -> profiles/<tag>_<name>_<suffix>.<ext>)
bench line (pretty-printed), rocprofv3 kernel stats for 3 streams / 1 stream, per-shape GEMM durations, the other
configurations' bench lines in short form, and achieved GB/s of the HBM-bound kernels (kernel-trace durations / algorithmic bytes).
"""
import csv
import json
import os
import re
import shutil
import statistics
import sys


def last_json(path):
    try:
        lines = [l for l in open(path).read().strip().split("\n") if l.startswith("{")]
        return json.loads(lines[-1])
    except Exception:
        return None


def short(d):
    r = d["roofline"]
    return {"value": d["value"], "value_min": d.get("value_min"), "value_max": d.get("value_max"), "unit": d["unit"],
            "ms_per_step": d["ms_per_step"], "host_enqueue_ms_per_image": d.get("host_enqueue_ms_per_image"),
            "hip_graph": d["config"].get("hip_graph"), "config": d["config"]["workload"],
            "dtype": d.get("dtype"),
            "gemm_tflops_alone": r["achieved"], "gemm_frac": r["frac"], "avg_launch_us": r["avg_launch_us"],
            # the roofline figure of the whole path: FLOPs this build EXECUTES (measured sum 2MNK of every GEMM launch + the attention /
            # LoRA / head FLOPs) x images/s / peak.  Multi-update configurations resume at the first trained layer for updates 2..n
            # (identical results), so the reference's own FLOP count — which repeats layers 0-8 per update — is work the hardware never
            # does: it is given for orientation only, under a name that says so.
            "whole_path_frac_executed": d.get("whole_path_frac_executed"),
            "reference_flops_x_rate_over_peak__counts_work_this_build_skips": d.get("whole_path_frac_of_bf16_peak")}


def main(src, tag, suffix=""):
    out = "profiles"
    os.makedirs(out, exist_ok=True)
    sfx = ("_" + suffix) if suffix else ""
    _open, _copy = open, shutil.copy

    def named(path):          # profiles/r05_bench.json -> profiles/r05_bench_fp16.json
        if not sfx or not path.startswith(out + "/"):
            return path
        root, ext = os.path.splitext(path)
        return root + sfx + ext
    open_w = lambda path, mode="w": _open(named(path), mode)
    b = last_json(f"{src}/bench.json")
    if b:
        json.dump(b, open_w(f"{out}/{tag}_bench.json"), indent=1)
    # ONE kernel-stats file: rocprofv3 serialises kernels, so the regime is "kernel alone on the chip" whatever --streams was
    for sub, name in (("prof1/p1_kernel_stats.csv", "bench_kernel_stats_serialized.csv"),):
        cand = [os.path.join(dp, f) for dp, _, fs in os.walk(f"{src}/{sub.split('/')[0]}") for f in fs if f.endswith("kernel_stats.csv")]
        if cand:
            shutil.copy(cand[0], named(f"{out}/{tag}_{name}"))
    if os.path.exists(f"{src}/prof1_gemm_shapes.txt"):
        shutil.copy(f"{src}/prof1_gemm_shapes.txt", named(f"{out}/{tag}_gemm_shapes_serialized.txt"))
    other = {}
    for key, f in (("streams1", "bench_streams1.json"), ("adapters_q_k_v_out", "bench_qkvo.json"), ("plain_enqueues_no_graph", "bench_graph0.json"),
                   ("k1000", "bench_k1000.json"), ("vit_l14", "bench_l14.json"),
                   ("r32_128v_4updates", "bench_r32_128v_4up.json"), ("r32_128v_16updates", "bench_r32_128v_16up.json"),
                   ("8views_k10_hip_graph", "bench_8v_graph.json")):
        d = last_json(f"{src}/{f}")
        if d:
            other[key] = short(d)
    for key, f in (("text_mode", "text_mode.log"), ("eval_gpu_views", "eval_gpu_views.log"), ("views", "views.log")):
        if os.path.exists(f"{src}/{f}"):
            other[key] = [l.strip() for l in open(f"{src}/{f}").read().strip().split("\n")
                          if l.strip() and "amdgpu.ids" not in l][-6:]
    sweep = {}
    for n in (2, 3, 4, 6):       # stream-count sweep of the build named by the suffix (tools/r05_sweeps.sh, one lease)
        d = last_json(f"{src}/bench_streams{n}.json")
        if d:
            sweep[f"streams{n}"] = {k: d[k] for k in ("value", "value_min", "value_max", "ms_per_step", "host_enqueue_ms_per_image", "repeats", "steps")}
            sweep[f"streams{n}"]["hip_graph"] = d["protocol"]["hip_graph"]
    if sweep:
        other["streams_sweep" + sfx] = sweep
    json.dump(other, open_w(f"{out}/{tag}_other_configs.json"), indent=1)
    for f in ("pmc_summary.txt", "pmc_memory_path.txt", "gemm_traffic.json", "class_cost_in_flight.txt"):
        if os.path.exists(f"{src}/{f}"):
            shutil.copy(f"{src}/{f}", named(f"{out}/{tag}_{f}"))
    for f in ("parity_per_fixture.txt", "plpd_trace_summary.txt"):      # tables over ALL builds: no build suffix
        if os.path.exists(f"{src}/{f}"):
            with open(f"{src}/{f}") as fi, open(f"{out}/{tag}_{f}", "w") as fo:
                fo.writelines(l for l in fi if "amdgpu.ids" not in l)
    if os.path.exists(f"{src}/torch_stack.json"):
        shutil.copy(f"{src}/torch_stack.json", f"{out}/{tag}_torch_stack_reference_point.json")
    # HBM-bound kernels: per-launch durations from the 1-stream kernel trace
    trace = [os.path.join(dp, f) for dp, _, fs in os.walk(f"{src}/prof1") for f in fs if f.endswith("kernel_trace.csv")]
    if trace:
        dur = {}
        for r in csv.DictReader(open(trace[0])):
            n = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
            blocks = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) if "Grid_Size_X" in r else 0
            dur.setdefault((n.split("<")[0], blocks), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        M, D = 12608, 768
        cus8 = 256 * 8
        spec = {"ln_fwd_kernel": (3152, M * D * 4 + M * D * 2, "read fp32 [M,D] + write operand [M,D]"),
                "ln_fwd_persist_kernel": (cus8, M * D * 4 + M * D * 2, "read fp32 [M,D] + write operand [M,D] (persistent waves, next row prefetched: round 4)"),
                "ln_bwd_kernel": (3152, M * D * (4 + 4 + 4 + 4 + 2), "read dy, x fp32 + residual gradient fp32, write fp32 + operand [M,D]"),
                "im2col_kernel": (4704, 64 * 3 * 224 * 224 * 4 + 12544 * 768 * 2, "read 64x3x224x224 fp32, write operand patches [12544,768]")}
        k = {}
        for name, (blocks, nbytes, what) in spec.items():
            v = dur.get((name, blocks))
            if v:
                med = statistics.median(v)
                k[name] = {"launch_blocks": blocks, "algorithmic_bytes": nbytes, "what": what, "median_us": round(med, 2),
                           "achieved_GBps": round(nbytes / med / 1e3, 1), "frac_of_8TBps": round(nbytes / med / 1e3 / 8000, 3),
                           "launches": len(v)}
        vl = other.get("views", [])
        m = [re.search(r"375x500.*GPU make_views ([0-9.]+) us", l) for l in vl]
        m = [x for x in m if x]
        if m:
            us = float(m[0].group(1))
            nb = 64 * 3 * 224 * 224 * 4 + 375 * 500 * 3
            k["views_kernel+coef_kernel"] = {"what": "64 views of one 375x500 uint8 image: write fp32 [64,3,224,224] (38.5 MB), source crop reads served by L2",
                                             "algorithmic_bytes": nb, "median_us": us, "achieved_GBps": round(nb / us / 1e3, 1),
                                             "note": "byte work bound by the integer tap arithmetic (bit-exact Pillow fixed point), not by HBM"}
        json.dump({"source": f"rocprofv3 --kernel-trace of `bench.py --streams 1` (profiles/{tag}_bench_kernel_stats_serialized.csv), tools/views_bench.py",
                   "kernels": k}, open_w(f"{out}/{tag}_hbm_kernels.json"), indent=1)
    print("wrote", sorted(f for f in os.listdir(out) if f.startswith(tag + "_")))


if __name__ == "__main__":
    main(*sys.argv[1:4])
