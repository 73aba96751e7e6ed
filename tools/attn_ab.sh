#!/bin/bash
# In-situ A/B of attention-backward wave counts: tools/_diag/libttl_hip_attn_Q_K.so  (0 = one wave per 32-row block)
set -e
cd "$(dirname "$0")/../ttl-test-time-low-rank-adaptation_amd/csrc"
make -j8 >/dev/null
mkdir -p ../../tools/_diag
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fvisibility=hidden"
for v in "$@"; do q=${v%_*}; k=${v#*_}; /opt/rocm/bin/hipcc $FL -DTTL_ATTN_NW_DQ=$q -DTTL_ATTN_NW_DKV=$k -c attention.hip -o ../../tools/_diag/attn_$v.o & done
wait
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../../tools/_diag/libttl_hip_attn_$v.so \
     ../../tools/_diag/attn_$v.o $(ls build/bf16/*.o | grep -v attention.o)
done
