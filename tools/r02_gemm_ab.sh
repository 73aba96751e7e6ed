#!/bin/bash
# Round 2: the 256-column big-M GEMM (gemm_big.hip) against the 160x128 kernel, in situ (tools/quick_bench.py = one
# episode stream + per-class event profile).  Usage on the GPU box: bash tools/r02_gemm_ab.sh [tag]
T=${1:-ab1}
O=gpurun_out/r02/$T
mkdir -p $O
cd "$GRAFT_REPO_ROOT"
python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_path.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
run() { name=$1; shift; env "$@" python3 tools/quick_bench.py > $O/qb_$name.log 2>&1; echo "== $name: $(grep -E 'episode|gemm  |sum' $O/qb_$name.log | tr '\n' ' ')"; }
run old TTL_GEMM_BIG=0
run mt5 TTL_GEMM_BIG=1
run mt5_s2 TTL_GEMM_BIG_STAGES=2
run mt7 TTL_GEMM_BIG_MT=7
run mt5_o0 TTL_GEMM_BIG_ORDER=0
run mt5_o1 TTL_GEMM_BIG_ORDER=1
run mt5_o2 TTL_GEMM_BIG_ORDER=2
run mt5_np TTL_GEMM_BIG_BLOCKS=100000
run mt7_np TTL_GEMM_BIG_MT=7 TTL_GEMM_BIG_BLOCKS=100000
tail -3 $O/pytest.log
