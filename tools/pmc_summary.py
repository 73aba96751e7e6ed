"""Summarise rocprofv3 --pmc passes (one *_counter_collection.csv per pass) per kernel and launch geometry:
average of every counter per launch, launch duration, and the derived figures the bench line / DESIGN quote:
  * MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)   (GUI_ACTIVE sums the 8 XCDs)
  * HBM-side bytes: FETCH_SIZE x 2 x 1024 (gfx950 tallies 128-B read requests at 64 B; KB units) + WRITE_SIZE x 1024
  * L2 hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
    python tools/pmc_summary.py out.txt [out_traffic.json] pass1.csv pass2.csv ...
"""
import collections
import csv
import json
import re
import sys


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*$", "", n)
    return n


def main(argv):
    out_txt = argv[0]
    out_json = argv[1] if argv[1].endswith(".json") else None
    files = argv[2:] if out_json else argv[1:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))   # (kernel, blocks) -> counter -> values
    dur = collections.defaultdict(list)
    for f in files:
        seen = set()
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    lines = ["# rocprofv3 --pmc passes on tools/quick_bench.py (ViT-B/16, 64 views, K=200, one episode stream); per-launch averages.",
             "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (MI355X guide).",
             "# profiled passes run at lower clocks than un-profiled ones: durations here are not the bench's.", ""]
    traffic = {}
    order = sorted(agg, key=lambda k: -sum(dur[k]))
    for key in order:
        c = {n: sum(v) / len(v) for n, v in agg[key].items()}
        d = dur[key]
        lines.append(f"{key[0]}  [{key[1]} blocks]  launches/pass={len(d) // max(len(files), 1) or len(d)}  avg_us={sum(d) / len(d):.1f}")
        lines.append("    " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
        der = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("GRBM_GUI_ACTIVE"):
            der.append(f"MFMA pipe utilisation={c['SQ_VALU_MFMA_BUSY_CYCLES'] / (c['GRBM_GUI_ACTIVE'] / 8 * 1024):.3f}")
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
                if n in c:
                    der.append(f"{n}/WAVE_CYCLES={c[n] / c['SQ_WAVE_CYCLES']:.2f}")
        if "TCC_HIT_sum" in c:
            der.append(f"L2 hit rate={c['TCC_HIT_sum'] / max(c['TCC_HIT_sum'] + c.get('TCC_MISS_sum', 0), 1):.3f}")
        rd = 2.0 * 1024.0 * c["FETCH_SIZE"] if "FETCH_SIZE" in c else None
        wr = 1024.0 * c["WRITE_SIZE"] if "WRITE_SIZE" in c else None
        if rd is not None or wr is not None:
            der.append(f"HBM-side read={0 if rd is None else rd / 1e6:.1f} MB write={0 if wr is None else wr / 1e6:.1f} MB per launch")
            traffic[f"{key[0]} [{key[1]} blocks]"] = {"launches": len(agg[key].get("FETCH_SIZE", agg[key].get("WRITE_SIZE", []))),
                                                     "read_bytes_per_launch": None if rd is None else round(rd),
                                                     "write_bytes_per_launch": None if wr is None else round(wr)}
        if der:
            lines.append("    -> " + "; ".join(der))
    open(out_txt, "w").write("\n".join(lines) + "\n")
    if out_json:
        big = {k: v for k, v in traffic.items() if ("gemm_big_kernel" in k or "gemm_huge_kernel" in k or "gemm_kernel<160" in k) and v["read_bytes_per_launch"] is not None
               and v["write_bytes_per_launch"] is not None}
        n = sum(v["launches"] for v in big.values())
        tot = sum(v["launches"] * (v["read_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in big.values())
        rdt = sum(v["launches"] * v["read_bytes_per_launch"] for v in big.values())
        json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 tools/quick_bench.py",
                   "regime": "rocprofv3 --pmc, one episode stream, separate FETCH_SIZE / WRITE_SIZE passes",
                   "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported; KB = 1024 B",
                   "gemm_launches": n, "traffic_bytes_per_launch": round(tot / max(n, 1)), "read_bytes_per_launch": round(rdt / max(n, 1)),
                   "write_bytes_per_launch": round((tot - rdt) / max(n, 1)), "per_kernel": big}, open(out_json, "w"), indent=1)
    print(open(out_txt).read()[:6000])


if __name__ == "__main__":
    main(sys.argv[1:])
