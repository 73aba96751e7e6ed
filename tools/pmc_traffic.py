"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide prescribes)
into HBM-side bytes per GEMM launch.  gfx950 correction: FETCH_SIZE counts 128-B requests at 64 B for wide
coalesced reads (16 B/lane global_load / global_load_lds) -> doubled; WRITE_SIZE is exact for 16-B/lane stores.
Units of both counters: KiB... rocprofv3 reports kilobytes (1 KB = 1024 B here).  Infinity-Cache hits are counted.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch/fetch_counter_collection.csv \
                                gpurun_out/pmc_write/write_counter_collection.csv profiles/r01_gemm_traffic.json
"""
import collections
import csv
import json
import re
import sys


def per_kernel(path, counter):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            m = re.search(r"gemm_kernel<([^>]*)>", r["Kernel_Name"])
            if m and not m.group(1).startswith("160"):
                continue                      # the roofline is for the big-M 160x128 kernel; small-M launches are their own class
            by["gemm_kernel<%s>" % (m.group(1) if m else "?")].append(float(r["Counter_Value"]))
    return by


def main(fetch_csv, write_csv, out_json):
    f, w = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    rows, tot_f, tot_w, n = {}, 0.0, 0.0, 0
    for k in sorted(f):
        nf, nw = len(f[k]), len(w.get(k, []))
        rd = 2.0 * 1024.0 * sum(f[k]) / nf
        wr = 1024.0 * sum(w.get(k, [0.0])) / max(nw, 1)
        rows[k] = {"launches": nf, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr)}
        tot_f += 2.0 * 1024.0 * sum(f[k]); tot_w += 1024.0 * sum(w.get(k, [0.0])); n += nf
    out = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (two runs) --kernel-include-regex gemm_kernel -- "
                      "python3 bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline",
           "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported; KB = 1024 B",
           "gemm_launches": n, "traffic_bytes_per_launch": round((tot_f + tot_w) / n),
           "read_bytes_per_launch": round(tot_f / n), "write_bytes_per_launch": round(tot_w / n), "per_kernel": rows}
    json.dump(out, open(out_json, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("gemm_launches", "traffic_bytes_per_launch", "read_bytes_per_launch", "write_bytes_per_launch")}))


if __name__ == "__main__":
    main(*sys.argv[1:4])
