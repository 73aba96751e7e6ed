#!/bin/bash
# PMC evidence of a round (separate passes: the MI355X guide's HBM/rocprofv3 section; never combined with trace domains):
#   bash tools/collect_pmc.sh r06 [fp16|bf16] [traffic-only]   -> gpurun_out/r05_fp16/pmc_*/ ; summary + traffic json beside them
# (second argument: the operand build tools/quick_bench.py runs, through TTL_PRECISION; default fp16 = the headline build)
R=${1:-r06}
P=${2:-fp16}
export TTL_PRECISION=$P
export TTL_CONCURRENCY=3      # quick_bench.py runs one stream: with the tile choices of the three-stream timed region (ttl_ctx_set_concurrency)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${R}_$P; mkdir -p $O
# every pass under its own timeout: a counter set the hardware cannot collect makes rocprofv3 abort and then hang
pass() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" -d $O/pmc_$name -o $name --output-format csv -- python3 tools/quick_bench.py > $O/pmc_$name.log 2>&1; echo "pass $name rc=$?"; }
if [ "$3" = traffic-only ]; then      # FETCH_SIZE / WRITE_SIZE passes alone -> gemm_traffic.json (the bf16 leg's figure in the bench line)
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  python3 tools/pmc_summary.py $O/pmc_summary.txt $O/gemm_traffic.json $(find $O/pmc_fetch $O/pmc_write -name "*counter_collection.csv") > $O/pmc_summary.stdout 2>&1
  head -c 2000 $O/pmc_summary.txt
  exit 0
fi
pass sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $O/pmc_summary.txt $O/gemm_traffic.json $(find $O/pmc_sq $O/pmc_fetch $O/pmc_write $O/pmc_tcc $O/pmc_lds -name "*counter_collection.csv") > $O/pmc_summary.stdout 2>&1
# memory path of a CU (TA / TCP / UTCL1): at most two of these counters fit one pass
pass ta1 TA_TA_BUSY GRBM_GUI_ACTIVE
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES
pass tcp1 TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES
pass tcp2 TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY
pass tlb TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT
python3 tools/pmc_summary.py $O/pmc_memory_path.txt $(find $O/pmc_ta1 $O/pmc_ta2 $O/pmc_tcp1 $O/pmc_tcp2 $O/pmc_tlb -name "*counter_collection.csv") > $O/pmc_memory_path.stdout 2>&1
head -c 5000 $O/pmc_summary.txt
