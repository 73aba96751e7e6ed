#!/bin/bash
# Round 5 (r05n): which launches csrc/gemm_huge.hip should take, in situ — configurations whose q/k/v launches do not fill their rounds of
# 256 x 256 tiles (ViT-L/14: 780 tiles = 76 % of four rounds; 8 views: 63 tiles), on either kernel.  Run with a rule that demanded 85 % of
# ALL rounds ("rule on": L/14 and 8 views on gemm_big); the outcome (L/14 better on gemm_huge, 8 views worse) is the shipped rule: tiles for
# >= 85 % of ONE round.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_fp16
Q="--no-cpu-baseline --no-parity --precision fp16 --repeats 3"
run() { name=$1; shift; env "$@" | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('%-64s %8.2f images/s (%.2f-%.2f)' % ('$name', d['value'], d['value_min'], d['value_max']))"; }
{
for rep in 1 2; do
run "ViT-L/14, rule on (q/k/v on gemm_big)" TTL_NOP=1 python3 bench.py --arch ViT-L/14 --steps 60 $Q 2>/dev/null
run "ViT-L/14, TTL_GEMM_HUGE_MIN_FILL=0 (q/k/v on gemm_huge)" TTL_GEMM_HUGE_MIN_FILL=0 python3 bench.py --arch ViT-L/14 --steps 60 $Q 2>/dev/null
run "8 views graph, rule on" TTL_NOP=1 python3 bench.py --views 8 --classes 10 --graph 1 --steps 400 $Q 2>/dev/null
run "8 views graph, TTL_GEMM_HUGE_MIN_FILL=0" TTL_GEMM_HUGE_MIN_FILL=0 python3 bench.py --views 8 --classes 10 --graph 1 --steps 400 $Q 2>/dev/null
run "128 views r=32 4 updates, default (891 tiles: gemm_huge)" TTL_NOP=1 python3 bench.py --views 128 --classes 1000 --rank 32 --updates 4 --steps 40 $Q 2>/dev/null
run "128 views r=32 4 updates, TTL_GEMM_HUGE=0" TTL_GEMM_HUGE=0 python3 bench.py --views 128 --classes 1000 --rank 32 --updates 4 --steps 40 $Q 2>/dev/null
run "default config, default" TTL_NOP=1 python3 bench.py --steps 150 $Q 2>/dev/null
run "default config, TTL_GEMM_HUGE=0" TTL_GEMM_HUGE=0 python3 bench.py --steps 150 $Q 2>/dev/null
done
} | tee gpurun_out/r05_fp16/huge_fill_ab.txt
