#!/bin/bash
# Round 6 (experiment r06a): what would the OTHER 16-bit MFMA shape do to csrc/gemm_huge.hip?  TIMING ONLY: the probe library
# (tools/hip_variant.sh gemm_huge TTL_HUGE_SHAPE_PROBE=1) runs the shipped loop with every v_mfma_f32_32x32x16 replaced by two
# v_mfma_f32_16x16x32 on the same operand registers — equal MFMA cycles, LDS reads, DMA pieces, waits, epilogue; WRONG products —
# against the experiments build of the same sources (same compiler flags).  Kernel level, cold operands, alternating A/B/A/B.
#   -> gpurun_out/r06/shape_probe.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
BASE=$PWD/ttl-test-time-low-rank-adaptation_amd/ttl_amd/libttl_hip_fp16_exp.so
PROBE=$PWD/tools/_diag/libttl_hip_fp16_gemm_huge_TTL_HUGE_SHAPE_PROBE_1.so
{
for rep in 1 2; do
  for lib in $BASE $PROBE; do
    echo "==== $(basename $lib)  (rep $rep)"
    TTL_HIP_LIB_FP16=$lib TTL_GEMM_HUGE=1 TTL_GEMM_HUGE_MIN_FILL=0 python3 tools/gemm_huge_bench.py fp16 --child 2>&1 | grep -v amdgpu.ids
  done
done
} | tee gpurun_out/r06/shape_probe.txt
