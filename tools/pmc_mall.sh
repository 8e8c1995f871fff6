#!/bin/bash
# PMC of the clean streaming read that defines BW_MALL (bench.py, profiles/r4_mall_sweep.txt): buffers of 4 MB (L2), 96 MB
# (Infinity Cache) and 2048 MB (HBM); one counter group per rocprofv3 run.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/pmc_mall; rm -rf $O; mkdir -p $O
for mb in 4 96 2048; do
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout -s KILL 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/m${mb}_g$i -o run -- python tools/one_stream_read.py $mb > $O/m${mb}_g$i.log 2>&1
  done
done
python tools/prof_summary.py $O 2>/dev/null | grep -E "^# |stream_read_kernel"
