cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r5pmcq; mkdir -p $O
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/uniform -o run -- python tools/exp_r5_fused.py time > $O/uniform.log 2>&1
SG_FUSED_ROT=1 timeout -s KILL 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/rot1 -o run -- python tools/exp_r5_fused.py bench-graph > $O/rot1.log 2>&1
SG_FUSED_ROT=0 timeout -s KILL 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/rot0 -o run -- python tools/exp_r5_fused.py bench-graph > $O/rot0.log 2>&1
python tools/prof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
grep -h "^# .*counter\|agg_contract_kernel.*TCC" $O/summary.txt
