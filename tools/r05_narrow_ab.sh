#!/bin/bash
# Round 5 (experiment r05r): the N = 768 big-M launches (out_proj, fc2, dgrads: 150 tiles of 256 x 256 on 256 CUs) on csrc/gemm_huge.hip —
# slower one at a time by construction; does the smaller CU-time pay with three episodes in flight?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_fp16
Q="--no-cpu-baseline --no-parity --precision fp16 --steps 150 --repeats 3"
run() { name=$1; shift; env "$@" python3 bench.py $Q 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
r = d['roofline']
print('%-50s %7.2f images/s (%.2f-%.2f)  GEMM class one at a time %.3f ms' % ('$name', d['value'], d['value_min'], d['value_max'], r['class_ms_per_image']['gemm']))"; }
{
for rep in 1 2; do
run "default" TTL_NOP=1
run "NARROW=1 (every N = 768 big-M launch)" TTL_GEMM_HUGE_NARROW=1
run "NARROW=1, K >= 2304 (fc2, fc1 / qkv dgrad)" TTL_GEMM_HUGE_NARROW=1 TTL_GEMM_HUGE_NARROW_MINK=2304
run "NARROW=1, K <= 1024 (out_proj and its dgrad)" TTL_GEMM_HUGE_NARROW=1 TTL_GEMM_HUGE_NARROW_MAXK=1024
done
} | tee gpurun_out/r05_fp16/narrow_ab.txt
