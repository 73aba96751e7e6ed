#!/bin/bash
# Round 6 (r06c): the MFMA-shape timing probe of r06b IN FLIGHT — bench.py's timed region (three episodes in flight), experiments build vs
# the probe library (wrong products: parity off; the probe's logits are checked for finiteness first, NaN data would flatter the clock).
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06
BASE=$PWD/ttl-test-time-low-rank-adaptation_amd/ttl_amd/libttl_hip_fp16_exp.so
PROBE=$PWD/tools/_diag/libttl_hip_fp16_gemm_huge_TTL_HUGE_SHAPE_PROBE_1.so
Q="--no-cpu-baseline --no-parity --precision fp16 --steps 150 --repeats 3 --sustain-seconds 0 --variant-lib"
run() { name=$1; lib=$2; TTL_HIP_LIB_FP16=$lib python3 bench.py $Q 2>/dev/null | python3 -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
r = d['roofline']
print('%-34s %7.2f images/s (%.2f-%.2f)  GEMM class one at a time %.3f ms, in flight %.3f ms' % ('$name', d['value'], d['value_min'], d['value_max'], r['class_ms_per_image']['gemm'], r['episodes_in_flight']['class_ms_per_image']['gemm']))"; }
{
TTL_HIP_LIB_FP16=$PROBE TTL_PRECISION=fp16 python3 - <<'PY'
import os, sys, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "ttl-test-time-low-rank-adaptation_amd")]
from ttl_amd import synth
from ttl_amd.config import get_config, trainable_names
from ttl_amd.driver import EpisodePipeline
cfg = get_config("ViT-B/16")
pipe = EpisodePipeline(cfg, synth.vision_weights(cfg, 0), trainable_names(cfg), synth.lora_init(cfg, 0), torch.from_numpy(synth.text_features(200, cfg.embed)), 100.0, "cuda:0", n_streams=3, max_views=64)
x = torch.from_numpy(synth.views(cfg, 64, 1000)).cuda()
outs = [pipe.submit(x, n_updates=1) for _ in range(3)]
pipe.synchronize()
print("probe library: adapted logits finite:", all(bool(torch.isfinite(o).all()) for o in outs), " max |logit|", float(max(o.abs().max() for o in outs)))
PY
for rep in 1 2; do
run "32x32x16 (experiments build)" $BASE
run "16x16x32 timing probe" $PROBE
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/shape_probe_inflight.txt
