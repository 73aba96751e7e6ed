#!/bin/bash
# Round 6: (1) in-kernel clock of gemm_huge / gemm_big (tools/r06_clock_stamps.py) and (2) the MFMA-shape timing probe (tools/r06_shape_probe.sh)
# on ONE lease.  Needs tools/_diag/ variants built here first:  tools/hip_variant.sh gemm_huge TTL_CLOCK_STAMPS=1 TTL_HUGE_SHAPE_PROBE=1;
# tools/hip_variant.sh gemm_big TTL_CLOCK_STAMPS=1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
python3 tools/r06_clock_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/inkernel_clock.txt
bash tools/r06_shape_probe.sh
