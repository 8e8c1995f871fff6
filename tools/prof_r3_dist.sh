#!/bin/bash
# Round-3 multi-GPU evidence obtainable on ONE GPU: (a) kernel trace of the partitioned config-5 step with one RCCL rank
# (are there still 1 GB copies around the all-reduces?), (b) per-rank step times of rank 0's share of an N-rank ML-10M
# partition (SG_BENCH_EMULATE_WORLD: everything but the other ranks' traffic).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r3dist; mkdir -p $O
FLAGS="--no-cpu-baseline --no-verify --no-minibatch-leg --no-hbm-leg --no-ceiling"
SG_BENCH_FORCE_DIST=1 timeout -s KILL 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace_c5 -o run -- python bench.py --shape config5 --steps 2 --warmup 1 > $O/trace_c5.log 2>&1
for n in 2 4 8; do
  SG_BENCH_FORCE_DIST=1 SG_BENCH_EMULATE_WORLD=$n timeout 200 python bench.py $FLAGS --graph-replay > $O/emulate_$n.log 2>&1
done
SG_BENCH_FORCE_DIST=1 timeout 200 python bench.py $FLAGS --graph-replay > $O/emulate_1.log 2>&1
python tools/prof_summary.py $O > $O/summary.txt 2>&1
find $O -name "*kernel_trace.csv" -size +8M -delete
for n in 1 2 4 8; do python - <<PY
import json
try:
    d=json.loads(open("$O/emulate_$n.log").read().strip().splitlines()[-1])
    print("world $n: ms_per_step", round(d["ms_per_step"],3), "graph_replay", d.get("graph_replay",{}).get("ms_per_step"), "exposed", d.get("collectives",{}).get("exposed_ms_per_step_per_rank"), "coll ms", d.get("collectives",{}).get("collective_ms_per_step"))
except Exception as e:
    print("world $n: failed", e)
PY
done
