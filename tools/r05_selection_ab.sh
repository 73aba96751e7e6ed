#!/bin/bash
# Round 5: what the backward on the selected views only (csrc/api.hip backward_impl, TTL_BWD_COMPACT) buys for the top-k objectives —
# --filter_ent 1 (deyo.py:105) and TPT (ttl.py:87-108): 6 of 64 views carry gradient.  fp16 build, three episodes in flight, one lease.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_fp16
Q="--no-cpu-baseline --no-parity --precision fp16 --steps 150 --repeats 3"
run() { name=$1; shift; env "$@" | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
c = d['roofline']['class_ms_per_image']
print('%-62s %8.2f images/s (%.2f-%.2f)   one at a time, ms per image: %s' % ('$name', d['value'], d['value_min'], d['value_max'], ' '.join('%s %.3f' % (k, v) for k, v in c.items())))"; }
{
for rep in 1 2; do
run "all views (reference default, BASELINE config)" TTL_NOP=1 python3 bench.py $Q 2>/dev/null
run "top-k (--filter_ent 1), backward on 6 views" TTL_NOP=1 python3 bench.py --selection topk $Q 2>/dev/null
run "top-k (--filter_ent 1), TTL_BWD_COMPACT=0 (all 64 views)" TTL_BWD_COMPACT=0 python3 bench.py --selection topk $Q 2>/dev/null
run "TPT objective, backward on 6 views" TTL_NOP=1 python3 bench.py --selection tpt $Q 2>/dev/null
run "TPT objective, TTL_BWD_COMPACT=0" TTL_BWD_COMPACT=0 python3 bench.py --selection tpt $Q 2>/dev/null
done
run "ViT-L/14 top-k, backward on 6 views" TTL_NOP=1 python3 bench.py --arch ViT-L/14 --selection topk --steps 60 --no-cpu-baseline --no-parity --precision fp16 --repeats 3 2>/dev/null
run "ViT-L/14 top-k, TTL_BWD_COMPACT=0" TTL_BWD_COMPACT=0 python3 bench.py --arch ViT-L/14 --selection topk --steps 60 --no-cpu-baseline --no-parity --precision fp16 --repeats 3 2>/dev/null
run "128 views r=32 4 updates top-k (12 views), packed" TTL_NOP=1 python3 bench.py --views 128 --classes 1000 --rank 32 --updates 4 --selection topk --steps 40 --no-cpu-baseline --no-parity --precision fp16 --repeats 3 2>/dev/null
run "128 views r=32 4 updates top-k, TTL_BWD_COMPACT=0" TTL_BWD_COMPACT=0 python3 bench.py --views 128 --classes 1000 --rank 32 --updates 4 --selection topk --steps 40 --no-cpu-baseline --no-parity --precision fp16 --repeats 3 2>/dev/null
} | tee gpurun_out/r05_fp16/selection_ab.txt
