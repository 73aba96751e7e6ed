#!/bin/bash
# Round evidence: bench line + rocprofv3 kernel stats of the same workload, other configurations, view generator; written under
# gpurun_out/<round>/ (tools/assemble_profiles.py copies what should be judged into profiles/).
# rocprofv3 --kernel-trace SERIALISES kernels, so its durations are "the kernel alone on the chip" whatever --streams is: ONE
# kernel-stats file is collected (plain enqueues, one stream); the 3-episodes-in-flight regime exists only in bench.py's own
# event timing (roofline.episodes_in_flight) and in the end-to-end rate.
#   bash tools/collect_profiles.sh r06 [fp16|bf16]      (second argument: the operand build every profiled run uses; default fp16 =
#   the headline build since round 4.  Output directory gpurun_out/<round>_<build>/)
R=${1:-r06}
P=${2:-fp16}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${R}_$P; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --streams 1 --no-cpu-baseline --no-parity --precision $P --sustain-seconds 0 > $O/bench_streams1.json 2>> $O/bench.err
Q="--no-cpu-baseline --no-parity --precision $P --sustain-seconds 0"
# (TTL_CONCURRENCY=3: one stream, serialised by the profiler, with the tile choices of the three-stream timed region; a run-time
#  switch off its default needs --variant-env since round 6)
export TTL_CONCURRENCY=3
rocprofv3 --kernel-trace --stats -d $O/prof1 -o p1 --output-format csv -- python3 bench.py --steps 60 --warmup 10 --repeats 1 --streams 1 --graph 0 $Q --variant-env > $O/prof1.log 2>&1
unset TTL_CONCURRENCY
python3 bench.py --lora-targets qkvo $Q > $O/bench_qkvo.json 2>> $O/bench.err
python3 bench.py --graph 0 $Q > $O/bench_graph0.json 2>> $O/bench.err
python3 bench.py --classes 1000 $Q > $O/bench_k1000.json 2>> $O/bench.err
python3 bench.py --arch ViT-L/14 --steps 60 $Q > $O/bench_l14.json 2>> $O/bench.err
python3 bench.py --views 128 --classes 1000 --rank 32 --updates 4 --steps 40 $Q > $O/bench_r32_128v_4up.json 2>> $O/bench.err
python3 bench.py --views 128 --classes 1000 --rank 32 --updates 16 --steps 20 $Q > $O/bench_r32_128v_16up.json 2>> $O/bench.err
python3 bench.py --views 8 --classes 10 --graph 1 --steps 400 $Q > $O/bench_8v_graph.json 2>> $O/bench.err
python3 tools/text_mode_bench.py $P > $O/text_mode.log 2>&1
python3 tools/views_bench.py > $O/views.log 2>&1
PYTHONPATH=ttl-test-time-low-rank-adaptation_amd python3 -m ttl_amd.eval --gpu_views 1 --images 1500 --precision $P > $O/eval_gpu_views.log 2>&1
python3 tools/trace_shapes.py $O/prof1/p1_kernel_trace.csv gemm > $O/prof1_gemm_shapes.txt 2>&1
ls $O
tail -c 600 $O/bench.err
