run() { timeout 600 python bench.py --shape config5 --hbm-shape 60000,50000,6000000,16 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
c=r.get('collectives',{})
print('$1', round(r['ms_per_step'],2), (r.get('roofline') or {}).get('kernel','')[:20], 'exposed', c.get('exposed_ms_per_step_per_rank'), 'calls', c.get('calls_per_step'), 'MB', c.get('bytes_per_step',0)/1e6 if c else None)"; }
run nodist
SG_BENCH_FORCE_DIST=1 run dist_fused
SG_FUSED=0 SG_BENCH_FORCE_DIST=1 run dist_unfused
SG_FUSED=0 run nodist_unfused
