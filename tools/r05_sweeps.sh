#!/bin/bash
# Round 5: what each kernel class costs with three episodes in flight, on the fp16 (headline) build, + the stream-count sweep of
# that build on the same lease.   bash tools/r05_sweeps.sh   -> gpurun_out/r05_fp16/{class_cost_in_flight.txt, bench_streamsN.json}
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_fp16
TTL_PRECISION=fp16 bash tools/class_cost_ab.sh > /dev/null 2>&1
cp gpurun_out/class_cost.txt gpurun_out/r05_fp16/class_cost_in_flight.txt
for n in 2 3 4 6; do
  python3 bench.py --streams $n --no-cpu-baseline --no-parity --precision fp16 > gpurun_out/r05_fp16/bench_streams$n.json 2>/dev/null
done
cat gpurun_out/r05_fp16/class_cost_in_flight.txt
python3 - <<'PY'
import json
for n in (2, 3, 4, 6):
    d = json.loads([l for l in open(f"gpurun_out/r05_fp16/bench_streams{n}.json") if l.startswith("{")][-1])
    print("streams", n, d["value"], d["value_min"], d["value_max"], d["protocol"]["hip_graph"])
PY
