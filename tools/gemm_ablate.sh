#!/bin/bash
# Builds diagnostic variants of the GEMM translation unit (see TTL_GEMM_DIAG in csrc/gemm.hip) into
# tools/_diag/libttl_hip_diagN.so.  Timing tools only: results of N != 0 are wrong by construction.
set -e
cd "$(dirname "$0")/../ttl-test-time-low-rank-adaptation_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/_diag
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fvisibility=hidden"
for n in "$@"; do
  /opt/rocm/bin/hipcc $FL -DTTL_GEMM_DIAG=$n -c gemm.hip -o ../../tools/_diag/gemm_$n.o &
done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../../tools/_diag/libttl_hip_diag$n.so \
     ../../tools/_diag/gemm_$n.o $(ls build/bf16/*.o | grep -v gemm.o)
done
ls -la ../../tools/_diag/*.so
