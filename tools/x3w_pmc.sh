#!/bin/bash
# PMC passes over the harness (one counter group per run, kernel trace only beside it):  tools/x3w_pmc.sh TAG filter variants
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
export X3W_ONLY="${2:-square}" X3W_VARIANTS="${3:-0,10}"
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/g$i -o run -- ./tools/x3w_harness bench 2 > $O/g$i.log 2>&1
done
python tools/prof_summary.py $O 2>/dev/null | grep -E "^#|gemm_|Name" | grep -v "kernel_trace" | cut -c1-160
find $O -name "*.csv" -size +4M -delete
