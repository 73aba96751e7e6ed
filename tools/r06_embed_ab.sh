#!/bin/bash
# Round 6 (r06h): CLS rows + pre-LayerNorm + LayerNorm 1 of layer 0 as ONE pass (elementwise.hip embed_ln2_kernel) against the three launches of
# rounds 1-5 (TTL_EMBED_FUSED=0, experiments build), in bench.py's timed region, alternating.   -> gpurun_out/r06/embed_ab.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
EXP=$PWD/ttl-test-time-low-rank-adaptation_amd/ttl_amd/libttl_hip_fp16_exp.so
Q="--no-cpu-baseline --no-parity --precision fp16 --steps 150 --repeats 5 --sustain-seconds 0 --variant-lib"
run() { name=$1; shift; env TTL_HIP_LIB_FP16=$EXP "$@" python3 bench.py $Q 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
r = d['roofline']
print('%-28s %7.2f images/s (%.2f-%.2f)  layernorm_elementwise class one at a time %.3f ms, in flight %.3f ms per 3 images' % ('$name', d['value'], d['value_min'], d['value_max'], r['class_ms_per_image']['layernorm_elementwise'], r['episodes_in_flight']['class_ms_per_image']['layernorm_elementwise']))"; }
{
for rep in 1 2 3; do
run "three launches (rounds 1-5)" TTL_EMBED_FUSED=0
run "one pass (round 6)" TTL_EMBED_FUSED=1
done
for v in 8 16; do
python3 bench.py --no-cpu-baseline --no-parity --precision fp16 --sustain-seconds 0 --variant-lib --views $v --classes 10 --graph 1 --steps 400 --repeats 3 2>/dev/null > /dev/null
for f in 0 1; do
TTL_HIP_LIB_FP16=$EXP TTL_EMBED_FUSED=$f python3 bench.py --no-cpu-baseline --no-parity --precision fp16 --sustain-seconds 0 --variant-lib --views $v --classes 10 --graph 1 --steps 400 --repeats 3 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('views %2d  TTL_EMBED_FUSED=$f  %8.1f images/s (%.1f-%.1f)' % ($v, d['value'], d['value_min'], d['value_max']))"
done
done
} | tee gpurun_out/r06/embed_ab.txt
