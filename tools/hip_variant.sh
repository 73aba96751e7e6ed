#!/bin/bash
# Builds a macro variant of ONE csrc file into tools/_diag/libttl_hip_<file>_<NAME>_<VALUE>.so, for in-situ A/B through
# TTL_HIP_LIB_BF16 / TTL_HIP_LIB_FP16.   tools/hip_variant.sh attention TTL_ATTN_DIAG=1 TTL_ATTN_DIAG=2
# TTL_PRECISION=fp16 (default since round 5: the headline build) | bf16 picks the operand build the variant is made of; fp16 variants
# are named libttl_hip_fp16_<file>_<NAME>_<VALUE>.so
set -e
cd "$(dirname "$0")/../ttl-test-time-low-rank-adaptation_amd/csrc"
f=$1; shift
make -j8 >/dev/null
mkdir -p ../../tools/_diag
P=${TTL_PRECISION:-fp16}
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fvisibility=hidden"
PFX=""
# fp16 variants are made of the EXPERIMENTS objects (build/exp: -DTTL_EXPERIMENTS, csrc/common.hpp) so that the closed A/B switches stay
# readable in an A/B library; the product builds compile them out
OBJ=$P
if [ "$P" = fp16 ]; then FL="$FL -DTTL_OPERAND_FP16 -DTTL_EXPERIMENTS"; PFX="fp16_"; OBJ=exp; fi
for n in "$@"; do
  t=${PFX}${f}_$(echo $n | tr '=' '_')
  ( /opt/rocm/bin/hipcc $FL -D$n -c $f.hip -o ../../tools/_diag/$t.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../../tools/_diag/libttl_hip_$t.so \
     ../../tools/_diag/$t.o $(ls build/$OBJ/*.o | grep -v /$f.o) ) &
done
wait
ls ../../tools/_diag/libttl_hip_${PFX}${f}_*.so
