#!/bin/bash
# Round 5 experiment r05k: with three episodes in flight a partial last round is filled by the other episodes' kernels (r05j: 224
# blocks = +23 % GEMM time one at a time, same images/s), so what a tile shape costs is CU-TIME per tile, not rounds.  Re-judge the
# 224 x 256 / 256 x 256 two-stage tiles (TTL_GEMM_BIG_MT=7 / 8; rounds 2-4 judged them one episode at a time) in flight.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_fp16
Q="--no-cpu-baseline --no-parity --precision fp16 --steps 200 --repeats 3"
run() { env $1 python3 bench.py $Q 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('%-34s %7.2f images/s (%.2f-%.2f)  gemm one-at-a-time %.3f ms' % ('$1', d['value'], d['value_min'], d['value_max'], d['roofline']['class_ms_per_image']['gemm']))"; }
{ for v in ${SWEEP:-TTL_NOP=1 TTL_GEMM_BIG_MT=7 TTL_GEMM_BIG_MT=8 TTL_NOP=1 TTL_GEMM_BIG_STAGES=2 TTL_GEMM_BIG_MT=8 TTL_GEMM_BIG_MT=7 TTL_NOP=1}; do run $v; done; } | tee -a gpurun_out/r05_fp16/mt_sweep.txt
