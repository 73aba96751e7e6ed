#!/bin/bash
# One GPU call that collects a round's evidence: headline (fp16) build profiles + PMC passes, a bf16 kernel trace for comparison, the torch-stack
# reference point on the same lease, the per-fixture parity table (strict / fp16 / bf16) and the PLPD kernel trace.   bash tools/collect_round.sh
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
bash tools/collect_profiles.sh r06 fp16 > gpurun_out/collect_fp16.log 2>&1
bash tools/collect_pmc.sh r06 fp16 > gpurun_out/collect_pmc_fp16.log 2>&1
O=gpurun_out/r06_bf16; mkdir -p $O
export TTL_CONCURRENCY=3      # (the tile choices of the three-stream timed region, on one profiled stream)
rocprofv3 --kernel-trace --stats -d $O/prof1 -o p1 --output-format csv -- python3 bench.py --steps 60 --warmup 10 --repeats 1 --streams 1 --graph 0 --no-cpu-baseline --no-parity --precision bf16 --sustain-seconds 0 --variant-env > $O/prof1.log 2>&1
unset TTL_CONCURRENCY
python3 tools/trace_shapes.py $O/prof1/p1_kernel_trace.csv gemm > $O/prof1_gemm_shapes.txt 2>&1
python3 tools/torch_stack_reference_point.py > gpurun_out/r06_fp16/torch_stack.json 2> gpurun_out/r06_fp16/torch_stack.err
python3 tools/parity_per_fixture.py > gpurun_out/r06_fp16/parity_per_fixture.txt 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r06_fp16/plpd_prof -o p --output-format csv -- python3 tools/plpd_trace.py patch > gpurun_out/r06_fp16/plpd_trace.log 2>&1
python3 tools/plpd_trace.py --summarize gpurun_out/r06_fp16/plpd_prof > gpurun_out/r06_fp16/plpd_trace_summary.txt 2>&1
python3 tools/plpd_trace.py patch --graph --host-timing >> gpurun_out/r06_fp16/plpd_trace.log 2>&1
python3 tools/plpd_trace.py pixel --graph --host-timing >> gpurun_out/r06_fp16/plpd_trace.log 2>&1
python3 tools/plpd_trace.py pixel --host-timing >> gpurun_out/r06_fp16/plpd_trace.log 2>&1
python3 tools/plpd_trace.py occ >> gpurun_out/r06_fp16/plpd_trace.log 2>&1
# the bf16 leg's own HBM-side traffic (the bench line reads each leg's newest profiles/r*_gemm_traffic_<build>.json)
bash tools/collect_pmc.sh r06 bf16 traffic-only > gpurun_out/collect_pmc_bf16.log 2>&1
ls gpurun_out/r06_fp16 | head -50
cat gpurun_out/r06_fp16/plpd_trace_summary.txt | head; tail -3 gpurun_out/r06_fp16/plpd_trace.log
