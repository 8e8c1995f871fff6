#!/bin/bash
# Round 6: kernel trace of a short ML-10M bench run + the ordered timeline of its last step (tools/step_timeline.py).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6tl${1:-}; mkdir -p $O
COMMON="--no-cpu-baseline --no-verify --no-minibatch-leg --no-ceiling --no-hbm-leg"
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o run -- python bench.py --steps 5 --warmup 2 $COMMON > $O/trace.log 2>&1
F=$(find $O -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $F ${2:-masked_embed} > $O/timeline.txt 2>&1
python tools/prof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
tail -2 $O/timeline.txt; tail -c 600 $O/trace.log
