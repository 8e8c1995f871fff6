cd $GRAFT_REPO_ROOT
F="--shape ml-10m --dim 256 --steps 1 --warmup 0 --no-cpu-baseline --no-hbm-leg --no-ceiling --no-minibatch-leg --no-verify --gpus 8"
for v in 5 0; do
  SG_X3_VARIANT=$v SG_BENCH_BACKEND=gloo timeout 600 python bench.py $F 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('variant $v', json.dumps(d.get('partition_check'))[:600])"
done
