#!/bin/bash
# Round 5 (r05n): tile order of csrc/gemm_huge.hip (TTL_GEMM_HUGE_ORDER: 0 = contiguous XCD chunks of the row-major tile list, 1 = XCD row
# ranges walked column-major, 2 = XCD row ranges row-major) — one at a time through ttl_gemm_nt_fused, and in situ.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_fp16
Q="--no-cpu-baseline --no-parity --precision fp16 --steps 150 --repeats 3"
run() { name=$1; shift; env "$@" python3 bench.py $Q 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
r = d['roofline']
print('%-28s %7.2f images/s (%.2f-%.2f)  GEMM class one at a time %.3f ms' % ('$name', d['value'], d['value_min'], d['value_max'], r['class_ms_per_image']['gemm']))"; }
{
for o in 0 1 2; do echo "== TTL_GEMM_HUGE_ORDER=$o"; TTL_GEMM_HUGE_ORDER=$o TTL_GEMM_HUGE=1 TTL_GEMM_HUGE_MIN_FILL=0 python3 tools/gemm_huge_bench.py fp16 --child 2>/dev/null | head -4; done
for rep in 1 2; do for o in 0 1 2; do run "TTL_GEMM_HUGE_ORDER=$o" TTL_GEMM_HUGE_ORDER=$o; done; done
} | tee gpurun_out/r05_fp16/huge_order_ab.txt
