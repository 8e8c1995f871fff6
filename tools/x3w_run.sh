#!/bin/bash
# Development loop for gemm_x3w.hip on the GPU box: correctness, then per-kernel times (rocprofv3 kernel trace) of the
# harness' bench shapes for the variants in $X3W_VARIANTS.   usage: tools/x3w_run.sh TAG [only-filter] [variants]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
export X3W_VARIANTS=${3:-0,10}
[ -n "$2" ] && export X3W_ONLY="$2"
if [ -z "$X3W_NOCHECK" ]; then (X3W_VARIANTS=8 X3W_ONLY= timeout 300 ./tools/x3w_harness check; echo "rc=$?") > $O/check.txt 2>&1; fi
timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- ./tools/x3w_harness bench 5 > $O/bench.txt 2>&1
python tools/prof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
grep -c ok $O/check.txt; grep -v " ok$" $O/check.txt | tail; cat $O/bench.txt | grep -v "^W2\|rocprof" | tail -20; grep "gemm_\|split" $O/summary.txt | cut -c1-150
