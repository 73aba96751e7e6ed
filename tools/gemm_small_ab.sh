#!/bin/bash
set -e
cd "$(dirname "$0")/../ttl-test-time-low-rank-adaptation_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/_diag
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fvisibility=hidden"
for n in "$@"; do /opt/rocm/bin/hipcc $FL -DTTL_GEMM_SMALL=$n -c gemm.hip -o ../../tools/_diag/gemm_sm$n.o & done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../../tools/_diag/libttl_hip_small$n.so \
     ../../tools/_diag/gemm_sm$n.o $(ls build/bf16/*.o | grep -v gemm.o)
done
