#!/bin/bash
# Round-4 PMC passes of both bench legs on the final gather kernel (one counter group per rocprofv3 run, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes): FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum + TCC_MISS_sum.
# Summarised by tools/prof_summary.py -> profiles/r4_pmc_counters_v1.csv; profiles/pmc_traffic.json is rebuilt from it.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r4pmc${1:-}; mkdir -p $O
COMMON="--no-cpu-baseline --no-verify --no-minibatch-leg"
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -s KILL 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_main_$tag -o run -- python bench.py --steps 2 --warmup 1 --no-ceiling --no-hbm-leg $COMMON > $O/pmc_main_$tag.log 2>&1
  if [ "${2:-all}" = "all" ]; then
  timeout -s KILL 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_hbm_$tag -o run -- python bench.py --hbm-only --hbm-steps 1 $COMMON > $O/pmc_hbm_$tag.log 2>&1
  fi
done
python tools/prof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +8M -delete
find $O -name "*kernel_trace.csv" -delete
grep -h "seg_gather_kernel" $O/summary.txt | head -40
