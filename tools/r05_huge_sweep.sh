#!/bin/bash
# Round 5 (experiment r05n): the 256 x 256 four-wave kernel (csrc/gemm_huge.hip) in situ — bench.py's timed region (three episodes in
# flight) and the one-at-a-time GEMM class, per launch family and per grid size.   -> gpurun_out/r05_fp16/huge_sweep.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_fp16
Q="--no-cpu-baseline --no-parity --precision ${PREC:-fp16} --steps 150 --repeats 3"
run() { name=$1; shift; env "$@" python3 bench.py $Q 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
r = d['roofline']
print('%-40s %7.2f images/s (%.2f-%.2f)  GEMM class one at a time %.3f ms (%.0f TF), in flight %.3f ms' % ('$name', d['value'], d['value_min'], d['value_max'], r['class_ms_per_image']['gemm'], r['achieved'], r['episodes_in_flight']['class_ms_per_image']['gemm']))"; }
{
for rep in 1 2; do
run "TTL_GEMM_HUGE=0 (gemm_big only)" TTL_GEMM_HUGE=0
run "TTL_GEMM_HUGE=1 (q/k/v + fc1)" TTL_GEMM_HUGE=1
run "TTL_GEMM_HUGE=2 (q/k/v only)" TTL_GEMM_HUGE=2
run "TTL_GEMM_HUGE=3 (fc1 only)" TTL_GEMM_HUGE=3
done
run "HUGE=1 BLOCKS=240" TTL_GEMM_HUGE=1 TTL_GEMM_HUGE_BLOCKS=240
run "HUGE=1 BLOCKS=225" TTL_GEMM_HUGE=1 TTL_GEMM_HUGE_BLOCKS=225
run "HUGE=1 BLOCKS=200" TTL_GEMM_HUGE=1 TTL_GEMM_HUGE_BLOCKS=200
} | tee gpurun_out/r05_fp16/huge_sweep${TAG}.txt
