#!/bin/bash
# Round-2 profile collection on the GPU box: kernel traces of both bench legs + PMC passes (one counter group per run,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r2prof${1:-}; mkdir -p $O
MAIN="--steps 5 --warmup 2 --no-cpu-baseline --no-ceiling --no-hbm-leg"
timeout -s KILL 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_main -o run -- python bench.py $MAIN > $O/trace_main.log 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_hbm -o run -- python bench.py --hbm-only --hbm-steps 3 > $O/trace_hbm.log 2>&1
if [ "${2:-pmc}" = "pmc" ]; then
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -s KILL 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_main_$tag -o run -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ceiling --no-hbm-leg > $O/pmc_main_$tag.log 2>&1
  timeout -s KILL 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_hbm_$tag -o run -- python bench.py --hbm-only --hbm-steps 1 > $O/pmc_hbm_$tag.log 2>&1
done
fi
python tools/prof_summary.py $O > $O/summary.txt 2>&1
# keep the merge small: drop the raw per-dispatch tables, the summary has what is judged
find $O -name "*counter_collection.csv" -size +8M -delete
find $O -name "*kernel_trace.csv" -size +8M -delete
tail -5 $O/*.log | tail -60
