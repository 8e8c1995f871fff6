for sh in ${MID_SHAPES:-300000,250000,30000000,16 150000,140000,20000000,10 600000,500000,60000000,5 400000,300000,40000000,8 200000,150000,16000000,16 120000,100000,12000000,16 250000,200000,25000000,6}; do
  for f in 0 1; do
    SG_FUSED=$f timeout 300 python bench.py --hbm-only --hbm-shape $sh --no-cpu-baseline --no-minibatch-leg --no-verify > gpurun_out/mid.json 2>/dev/null
    python - <<PY
import json
r=json.loads([l for l in open("gpurun_out/mid.json") if l.startswith("{")][-1])
h=r.get("hbm_bound",r)
print("$sh", "fused=$f", round(h["ms_per_step"],2), h["roofline"].get("kernel","")[:30])
PY
  done
done
