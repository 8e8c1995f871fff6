#!/bin/bash
# (Ran on the development build that still had level phases -- SG_FUSED_LP no longer exists in csrc/agg_fused.hip.)
# PMC of the fused kernel under its tuning switches on the config-5 shard graph (tools/exp_r5_fused.py bench-graph):
# FETCH_SIZE and TCC_HIT_sum / TCC_MISS_sum per launch for: default, SG_FUSED_NT=1, SG_FUSED_LP=8, SG_FUSED_ABLATE=1 (no matrix work)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5pmcv; mkdir -p $O
run() {  # name, env...
  name=$1; shift
  for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    env "$@" timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/${name}_$tag -o run -- python tools/exp_r5_fused.py bench-graph > $O/${name}_$tag.log 2>&1
  done
}
run default SG_FUSED_ABLATE=4
run nt SG_FUSED_ABLATE=4 SG_FUSED_NT=1
run lp8 SG_FUSED_ABLATE=4 SG_FUSED_LP=8
run lp4 SG_FUSED_ABLATE=4 SG_FUSED_LP=4
run nomatrix SG_FUSED_ABLATE=1
python tools/prof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
grep -h "^# \|agg_contract_kernel" $O/summary.txt | grep -v "kernel_trace" | head -60
