#!/bin/bash
# Round evidence: bench line + rocprofv3 kernel stats of the same command (default = 2 episodes in flight, and
# --streams 1), written under gpurun_out/ (copy what should be judged into profiles/).
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$R
python3 bench.py > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err
python3 bench.py --streams 1 --no-cpu-baseline > gpurun_out/$R/bench_streams1.json 2>> gpurun_out/$R/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/$R/prof2 -o p2 --output-format csv -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline > gpurun_out/$R/prof2.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/$R/prof1 -o p1 --output-format csv -- python3 bench.py --steps 60 --warmup 10 --streams 1 --no-cpu-baseline > gpurun_out/$R/prof1.log 2>&1
python3 bench.py --classes 1000 --no-cpu-baseline > gpurun_out/$R/bench_k1000.json 2>> gpurun_out/$R/bench.err
python3 bench.py --arch ViT-L/14 --steps 60 --no-cpu-baseline > gpurun_out/$R/bench_l14.json 2>> gpurun_out/$R/bench.err
python3 bench.py --views 128 --classes 1000 --rank 32 --updates 4 --steps 40 --no-cpu-baseline > gpurun_out/$R/bench_r32_128v_4up.json 2>> gpurun_out/$R/bench.err
python3 bench.py --precision fp16 --no-cpu-baseline > gpurun_out/$R/bench_fp16.json 2>> gpurun_out/$R/bench.err
ls -la gpurun_out/$R gpurun_out/$R/prof2 gpurun_out/$R/prof1
