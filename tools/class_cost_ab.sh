#!/bin/bash
# What does each kernel class cost WITH THREE EPISODES IN FLIGHT?  Builds timing-only variants of the library in which one class of
# launches is not issued (TTL_DIAG_SKIP, results wrong on purpose) and runs bench.py's timed region on each through TTL_HIP_LIB_BF16:
# the gain in images/s over the product build is that class's share of the step in the overlapped regime (a class whose kernels hide
# under other episodes' GEMMs gains little; a CU-exclusive class gains its full serial time).
#   bash tools/class_cost_ab.sh       (on the GPU box; -> gpurun_out/class_cost.txt)     TTL_PRECISION=fp16 (default) | bf16
cd "$(dirname "$0")/.."
export TTL_PRECISION=${TTL_PRECISION:-fp16}
P=$TTL_PRECISION
if [ "$P" = fp16 ]; then PFX="fp16_"; VAR=TTL_HIP_LIB_FP16; else PFX=""; VAR=TTL_HIP_LIB_BF16; fi
bash tools/hip_variant.sh attention TTL_DIAG_SKIP=1 TTL_DIAG_SKIP=2 > /dev/null
bash tools/hip_variant.sh elementwise TTL_DIAG_SKIP=1 TTL_DIAG_SKIP=2 > /dev/null
bash tools/hip_variant.sh head_loss TTL_DIAG_SKIP=1 > /dev/null
bash tools/hip_variant.sh lora TTL_DIAG_SKIP=3 > /dev/null
bash tools/hip_variant.sh gemm TTL_DIAG_SKIP=1 TTL_DIAG_SKIP=2 > /dev/null
mkdir -p gpurun_out
# plain launches (a captured graph would keep the launches that were issued at capture time); every variant issues its class
# normally through the warm-up and the first part of the first timed block, so the buffers downstream hold realistic data; the
# median of the three blocks is a block in which the class is not issued
Q="--no-cpu-baseline --no-parity --precision $P --steps 150 --repeats 3 --graph 0 --variant-lib"
run() { env $VAR=$2 python bench.py $Q 2>/dev/null | python -c "
import sys, json
d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('%-34s %7.2f images/s  %.3f ms/image   (1-stream class ms: %s)' % ('$1', d['value'], d['ms_per_step'], d['roofline']['class_ms_per_image']))"; }
{
echo "# operand build: $P"
run "product" ""
run "no dense attention forward" tools/_diag/libttl_hip_${PFX}attention_TTL_DIAG_SKIP_1.so
run "no dense attention backward" tools/_diag/libttl_hip_${PFX}attention_TTL_DIAG_SKIP_2.so
run "no big LayerNorm forward" tools/_diag/libttl_hip_${PFX}elementwise_TTL_DIAG_SKIP_1.so
run "no big LayerNorm backward" tools/_diag/libttl_hip_${PFX}elementwise_TTL_DIAG_SKIP_2.so
run "no head forward / backward" tools/_diag/libttl_hip_${PFX}head_loss_TTL_DIAG_SKIP_1.so
run "no big LoRA skinny / wgrad" tools/_diag/libttl_hip_${PFX}lora_TTL_DIAG_SKIP_3.so
run "no small-M GEMM launches" tools/_diag/libttl_hip_${PFX}gemm_TTL_DIAG_SKIP_1.so
run "no big-M GEMM launches" tools/_diag/libttl_hip_${PFX}gemm_TTL_DIAG_SKIP_2.so
run "product (again)" ""
} | tee gpurun_out/class_cost.txt
