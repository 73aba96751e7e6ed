#!/bin/bash
# In-situ A/B of an elementwise.hip build-time macro: tools/ew_macro_ab.sh MACRO v1 v2 ... -> tools/_diag/libttl_hip_MACRO_v.so
set -e
M=$1; shift
cd "$(dirname "$0")/../ttl-test-time-low-rank-adaptation_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/_diag
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fvisibility=hidden"
for n in "$@"; do ( /opt/rocm/bin/hipcc $FL -D$M=$n -c elementwise.hip -o ../../tools/_diag/ew_${M}_$n.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../../tools/_diag/libttl_hip_${M}_$n.so \
     ../../tools/_diag/ew_${M}_$n.o $(ls build/bf16/*.o | grep -v elementwise.o) ) & done
wait
ls ../../tools/_diag/*${M}*.so
