#!/bin/bash
# Round 5 experiment r05j: every big-M launch of the 64-view episode is a multiple of 237 tiles (79 row tiles x 3 / 9 / 12), so a
# persistent grid of 237-240 blocks has the SAME makespan as 256 (3 / 4 tiles per block either way) and leaves 16-19 CUs free for
# the other two episodes' LayerNorm / attention / LoRA kernels.  TTL_GEMM_BIG_BLOCKS=N, fp16 build, three episodes in flight.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_fp16
Q="--no-cpu-baseline --no-parity --precision fp16 --steps 200 --repeats 3"
run() { env TTL_GEMM_BIG_BLOCKS=$1 python3 bench.py $Q 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('blocks %-4s %7.2f images/s (%.2f-%.2f)  gemm one-at-a-time %.3f ms, in flight avg launch %.1f us' % ('$1', d['value'], d['value_min'], d['value_max'], d['roofline']['class_ms_per_image']['gemm'], d['roofline']['episodes_in_flight']['avg_launch_us']))"; }
{ for b in 0 240 237 0 248 232 224 0 240; do run $b; done; } | tee gpurun_out/r05_fp16/blocks_sweep.txt
