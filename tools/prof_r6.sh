#!/bin/bash
# Round-6 profile collection on the GPU box: rocprofv3 kernel traces of both bench legs (verification, mini-batch leg and
# CPU baseline off: they are not part of a step), summarised into the columns of `--stats` by tools/prof_summary.py.
# The PMC passes of the gather are unchanged from round 2 (same seg_gather.hip, sha in profiles/pmc_traffic.json); the
# round-4 PMC passes are tools/pmc_r4.sh.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6prof${1:-}; mkdir -p $O
COMMON="--no-cpu-baseline --no-verify --no-minibatch-leg"
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_main -o run -- python bench.py --steps 5 --warmup 2 --no-ceiling --no-hbm-leg $COMMON > $O/trace_main.log 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_hbm -o run -- python bench.py --hbm-only --hbm-steps 3 $COMMON > $O/trace_hbm.log 2>&1
python tools/prof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
tail -3 $O/*.log | tail -20
