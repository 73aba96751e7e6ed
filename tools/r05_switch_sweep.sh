#!/bin/bash
# Round 5: are the run-time defaults — all tuned on the bf16 build in rounds 1-4 — still the right ones for the fp16 (headline) build?
# One lease, bench.py's timed region (three episodes in flight), each switch against the default.   -> gpurun_out/r05_fp16/switch_sweep.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_fp16
Q="--no-cpu-baseline --no-parity --precision fp16 --steps 150 --repeats 3"
run() { name=$1; shift; env "$@" python3 bench.py $Q 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('%-44s %7.2f images/s (%.2f-%.2f)  one-at-a-time class ms: %s' % ('$name', d['value'], d['value_min'], d['value_max'], d['roofline']['class_ms_per_image']))"; }
{
run "default" TTL_NOP=1
run "TTL_ATTN_VARIANT=5 (per-tile pipeline)" TTL_ATTN_VARIANT=5
run "TTL_ATTN_VARIANT=1 (one block per problem)" TTL_ATTN_VARIANT=1
run "TTL_ATTN_XCD_MAP=1" TTL_ATTN_XCD_MAP=1
run "TTL_LN_PBLK=4" TTL_LN_PBLK=4
run "TTL_LN_PBLK=16" TTL_LN_PBLK=16
run "TTL_GEMM_BIG_DGRAD=0 (MLP dgrad on gemm.hip)" TTL_GEMM_BIG_DGRAD=0
run "TTL_GEMM_XCD2D=0 (1-D tile order, small kernel)" TTL_GEMM_XCD2D=0
run "TTL_WGRAD_MERGE=1" TTL_WGRAD_MERGE=1
run "TTL_QKV_HEAD_MAJOR=0" TTL_QKV_HEAD_MAJOR=0
run "TTL_POOLED_LAST_LAYER=0 (dense last layer)" TTL_POOLED_LAST_LAYER=0
run "default (again)" TTL_NOP=1
} | tee gpurun_out/r05_fp16/switch_sweep.txt
