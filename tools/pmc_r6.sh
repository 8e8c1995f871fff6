#!/bin/bash
# Round-6 (re-run after agg_fused.hip gained its GEN instantiation: same 256-wide kernel, new sha) PMC passes of the config-5 leg, whose aggregations now run in the fused aggregate -> contract kernel
# (csrc/agg_fused.hip): FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum + TCC_MISS_sum, one counter group per rocprofv3 run
# (/opt/skills/guides/MI355X_MICROARCH.md).  Summarised by tools/prof_summary.py -> profiles/r5_pmc_fused_v1.csv;
# the record "hbm-config5-shard-fused:256" of profiles/pmc_traffic.json is built from it.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6pmc${1:-}; mkdir -p $O
COMMON="--no-cpu-baseline --no-verify --no-minibatch-leg"
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -s KILL 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_hbm_$tag -o run -- python bench.py --hbm-only --hbm-steps 1 $COMMON > $O/pmc_hbm_$tag.log 2>&1
done
python tools/prof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -size +8M -delete
find $O -name "*kernel_trace.csv" -delete
grep -h "agg_contract_kernel\|seg_gather_kernel" $O/summary.txt | head -40
