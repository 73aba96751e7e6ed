#!/bin/bash
# Builds diagnostic variants of gemm_big.hip (TTL_GEMM_DIAG=N, see the source) into tools/_diag/libttl_hip_bdiagN.so and
# macro variants NAME=VALUE into tools/_diag/libttl_hip_NAME_VALUE.so.  Timing tools only.
#   tools/gemm_big_ablate.sh 1 2 3 4 5   |   tools/gemm_big_ablate.sh TTL_BIG_X=1 TTL_BIG_X=2
set -e
cd "$(dirname "$0")/../ttl-test-time-low-rank-adaptation_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/_diag
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fvisibility=hidden"
for n in "$@"; do
  case $n in *=*) d="-D$n"; t=$(echo $n | tr '=' '_');; *) d="-DTTL_GEMM_DIAG=$n"; t="bdiag$n";; esac
  ( /opt/rocm/bin/hipcc $FL $d -c gemm_big.hip -o ../../tools/_diag/gemm_big_$t.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../../tools/_diag/libttl_hip_$t.so \
     ../../tools/_diag/gemm_big_$t.o $(ls build/bf16/*.o | grep -v gemm_big.o) ) &
done
wait
ls ../../tools/_diag/*.so
