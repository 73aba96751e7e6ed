#!/bin/bash
# PMC evidence of a round (separate passes: the MI355X guide's HBM/rocprofv3 section; never combined with trace domains):
#   bash tools/collect_pmc.sh r02      -> gpurun_out/r02/pmc_*/ ; summary + traffic json under gpurun_out/r02/
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/$R; mkdir -p $O
pass() { name=$1; shift; rocprofv3 --pmc "$@" -d $O/pmc_$name -o $name --output-format csv -- python3 tools/quick_bench.py > $O/pmc_$name.log 2>&1; }
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $O/pmc_summary.txt $O/gemm_traffic.json $(find $O/pmc_* -name "*counter_collection.csv") > $O/pmc_summary.stdout 2>&1
head -c 5000 $O/pmc_summary.txt
