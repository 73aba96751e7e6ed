#!/bin/bash
# Round 6 (r06i): tile shapes for the big-M launches of the SMALL-view settings (8 / 16 / 32 views: M = 1 576 / 3 152 / 6 304 rows) — experiments build,
# closed switches TTL_GEMM_BIG (0: gemm.hip's 160 x 128 kernel for every big-M launch) and TTL_GEMM_VARIANT (0: 128 x 128).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
EXP=$PWD/ttl-test-time-low-rank-adaptation_amd/ttl_amd/libttl_hip_fp16_exp.so
Q="--no-cpu-baseline --no-parity --precision fp16 --sustain-seconds 0 --graph 1 --variant-lib --variant-env --classes 10 --steps 400 --repeats 3"
run() { v=$1; name=$2; shift 2; env TTL_HIP_LIB_FP16=$EXP "$@" python3 bench.py $Q --views $v 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('views %2d  %-44s %8.1f images/s (%.1f-%.1f)' % ($v, '$name', d['value'], d['value_min'], d['value_max']))"; }
{
for v in 8 16 32; do
run $v "product choices" TTL_GEMM_HUGE=2
run $v "gemm_big 160x256 only (TTL_GEMM_HUGE=0)" TTL_GEMM_HUGE=0
run $v "gemm.hip 160x128 (TTL_GEMM_BIG=0)" TTL_GEMM_BIG=0 TTL_GEMM_HUGE=0
run $v "gemm.hip 128x128 (TTL_GEMM_BIG=0 VARIANT=0)" TTL_GEMM_BIG=0 TTL_GEMM_HUGE=0 TTL_GEMM_VARIANT=0
run $v "product choices again" TTL_GEMM_HUGE=2
done
} | tee gpurun_out/r06/small_views_tiles.txt
