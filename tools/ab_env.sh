#!/bin/bash
# A/B of one environment switch on the ML-10M step (bench.py main leg only): tools/ab_env.sh VAR [values...]
cd $GRAFT_REPO_ROOT
VAR=$1; shift
COMMON="--no-cpu-baseline --no-verify --no-minibatch-leg --no-ceiling --no-hbm-leg"
for v in "$@"; do
  env $VAR=$v python bench.py --steps 20 --warmup 5 $COMMON 2> gpurun_out/ab_env.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$VAR', '$v', 'ms', round(d['ms_per_step'], 3), 'median', round(d['ms_per_step_median_events'], 3), 'gemm_ms', round(d['dense_roofline']['gemm_ms_per_step'], 3), 'loss', d['config']['loss'])"
done
tail -3 gpurun_out/ab_env.err
