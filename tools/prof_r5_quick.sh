timeout 600 python -m pytest tests/test_gpu_agg_fused.py -x -q 2>&1 | tail -1
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5profi/trace_hbm -o run -- python bench.py --hbm-only --hbm-steps 3 --no-cpu-baseline --no-verify --no-minibatch-leg > gpurun_out/r5profi.log 2>&1
python tools/prof_summary.py gpurun_out/r5profi 2>/dev/null | grep -E "bias_grad|agg_contract|gemm_f32_kernel<true" | cut -c1-120
find gpurun_out/r5profi -name "*kernel_trace.csv" -size +8M -delete
tail -c 400 gpurun_out/r5profi.log | head -c 0
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5profi.log") if l.startswith("{")][-1]); h=d["hbm_bound"]
print("step", round(h["ms_per_step"],1))
PY
