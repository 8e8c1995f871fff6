"""f16x3 plane kernel, timing-only ablations (libraries built with -DSG_X3_ABLATE=1|2|3 under tools/ablate/):
python tools/exp_x3_ablate.py  -- results are wrong by construction, only the times mean something"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from star_gcn_amd import ops, _lib as L
from tools.microbench import timeit
L.lib().sg_gemm_backend(3)
out = []
for (M, N, K) in [(4096, 4096, 4096), (262144, 4096, 1024)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
    for v in (1, 6):
        L.lib().sg_gemm_x3_variant(v)
        t = timeit(lambda: ops.gemm(a, b, trans_b=True), n=7, warm=2)
        out.append("%%dx%%dx%%d v%%d %%.3f ms %%.0f TF" %% (M, N, K, v, t * 1e3, 2.0 * M * N * K / t / 1e12))
print(" | ".join(out))
''' % ROOT
for tag, lib in [("full", None), ("no MFMA", "x3a1"), ("no fragment reads", "x3a2"), ("no DMA", "x3a3"), ("same panel", "same")]:
    env = dict(os.environ)
    if lib:
        env["SG_LIB_OVERRIDE"] = os.path.join(ROOT, "tools", "ablate", "libstargcn_%s.so" % lib)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print("%-18s %s" % (tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
