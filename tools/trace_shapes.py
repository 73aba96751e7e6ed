"""Per-(kernel, grid) duration summary of a rocprofv3 --kernel-trace CSV: python tools/trace_shapes.py trace.csv [substr]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
sub = sys.argv[2] if len(sys.argv) > 2 else "gemm"
agg = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if sub not in n: continue
    k = n[n.find(sub):][:34]
    agg[(k, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0
for (k, g, gy), d in sorted(agg.items(), key=lambda x: -sum(x[1])):
    d2 = sorted(d); tot += sum(d)
    if sum(d) < 200: continue
    print(f"  {k:36s} blocks {g:5d}x{gy} n={len(d):4d} med {d2[len(d2)//2]:7.1f} min {d2[0]:6.1f} tot {sum(d)/1e3:7.2f} ms")
print(f"  total {tot/1e3:.2f} ms")
