#!/bin/bash
# A/B of two builds of the library on the ML-10M step, alternating in one gpurun call: tools/ab_lib.sh <other .so> [reps]
cd $GRAFT_REPO_ROOT
OTHER=$1; REPS=${2:-3}
COMMON="--no-cpu-baseline --no-verify --no-minibatch-leg --no-ceiling --no-hbm-leg"
for i in $(seq $REPS); do
  for v in shipped other; do
    if [ $v = other ]; then export SG_LIB_OVERRIDE=$PWD/$OTHER; else unset SG_LIB_OVERRIDE; fi
    python bench.py --steps 30 --warmup 5 $COMMON 2> gpurun_out/ab_lib.err | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v', 'ms', round(d['ms_per_step'], 3), 'median', round(d['ms_per_step_median_events'], 3), 'gemm_ms', round(d['dense_roofline']['gemm_ms_per_step'], 3), 'loss', d['config']['loss'])"
  done
done
