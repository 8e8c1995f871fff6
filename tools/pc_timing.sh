#!/bin/bash
# how long does the N-rank partition check take at the ML-10M shape over gloo on one GPU?  tools/pc_timing.sh N
cd $GRAFT_REPO_ROOT; T0=$(date +%s)
N=${1:-2}
env SG_BENCH_BACKEND=gloo python bench.py --gpus $N --shape ml-10m --dim 256 --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-leg --no-ceiling --no-minibatch-leg --no-verify > gpurun_out/pc_timing_$N.json 2> gpurun_out/pc_timing_$N.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/pc_timing_$N.json") if l.startswith("{")][-1])
pc = d["partition_check"]
print("partition_check seconds", pc.get("seconds"), "ok", pc["ok"])
f = pc["f64"]
print({k: f[k] for k in f if k not in ("per_tensor_rank0", "method", "activation_derivative")})
PY

echo wall $(( $(date +%s) - T0 )) s
