#!/bin/bash
# Round-5 PMC passes of the fused aggregate -> contract kernel beyond the traffic counters: issue / wait / matrix-pipe / LDS / L1
# counters, one group per rocprofv3 run (kernel trace only beside it), on `bench.py --hbm-only --hbm-steps 1`.
# Summarised by tools/prof_summary.py -> profiles/r5_pmc_fused_pipe.csv
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5pmcpipe${1:-}; mkdir -p $O
COMMON="--no-cpu-baseline --no-verify --no-minibatch-leg"
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout -s KILL 900 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o run -- python bench.py --hbm-only --hbm-steps 1 $COMMON > $O/g$i.log 2>&1
done
python tools/prof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +8M -delete
find $O -name "*kernel_trace.csv" -delete
grep -h "agg_contract_kernel" $O/summary.txt | head -60
