#!/bin/bash
# Round 5 (r05r): the N = 768 / 1024 big-M launches on csrc/gemm_huge.hip in the other configurations (three episodes in flight)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_fp16
Q="--no-cpu-baseline --no-parity --precision fp16 --repeats 3"
run() { name=$1; shift; env "$@" 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('%-58s %8.2f images/s (%.2f-%.2f)' % ('$name', d['value'], d['value_min'], d['value_max']))"; }
{
for n in 0 1; do
run "ViT-L/14 (260 tiles), NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --arch ViT-L/14 --steps 60 $Q
run "8 views graph (21 tiles), NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --views 8 --classes 10 --graph 1 --steps 400 $Q
run "16 views (39 tiles), NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --views 16 --steps 400 $Q
run "32 views (75 tiles), NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --views 32 --steps 300 $Q
run "128 views r=32 4 updates (297 tiles), NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --views 128 --classes 1000 --rank 32 --updates 4 --steps 40 $Q
run "q/k/v/out adapters, NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --lora-targets qkvo --steps 150 $Q
run "K = 1000, NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --classes 1000 --steps 150 $Q
run "top-k selection, NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --selection topk --steps 150 $Q
run "2 streams, NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --streams 2 --steps 150 $Q
run "4 streams, NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --streams 4 --steps 150 $Q
run "bf16 build, NARROW=$n" TTL_GEMM_HUGE_NARROW=$n python3 bench.py --steps 150 --no-cpu-baseline --no-parity --precision bf16 --repeats 3
done
} | tee gpurun_out/r05_fp16/narrow_configs.txt
