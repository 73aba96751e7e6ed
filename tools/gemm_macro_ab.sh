#!/bin/bash
# In-situ A/B of a gemm.hip build-time macro: tools/gemm_macro_ab.sh MACRO v1 v2 ... -> tools/_diag/libttl_hip_MACRO_v.so
set -e
M=$1; shift
cd "$(dirname "$0")/../ttl-test-time-low-rank-adaptation_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/_diag
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fvisibility=hidden"
for n in "$@"; do /opt/rocm/bin/hipcc $FL -D$M=$n -c gemm.hip -o ../../tools/_diag/gemm_${M}_$n.o & done
wait
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../../tools/_diag/libttl_hip_${M}_$n.so \
     ../../tools/_diag/gemm_${M}_$n.o $(ls build/bf16/*.o | grep -v gemm.o)
done
