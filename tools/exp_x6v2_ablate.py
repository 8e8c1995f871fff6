"""x6v2 timing ablations (development builds tools/ablate/N/libstargcn_hip.so, SG_LIB_OVERRIDE): which part of the
producer / consumer pipeline bounds a K tile.  Results are NOT numerically valid GEMMs."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from star_gcn_amd import ops, _lib as L
from tools.microbench import timeit
L.lib().sg_gemm_backend(2)
for (M, N, K, tb) in [(4096, 4096, 4096, True), (262144, 4096, 256, True), (262144, 4096, 32, True), (10677, 2560, 256, True), (10677, 256, 2560, False)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn((N, K) if tb else (K, N), device="cuda")
    t = timeit(lambda: ops.gemm(a, b, trans_b=tb), n=5, warm=2)
    print("  M=%%7d N=%%5d K=%%5d tb=%%d  %%8.3f ms  %%6.1f TF/s-equiv" %% (M, N, K, tb, t * 1e3, 2.0 * M * N * K / t / 1e12), flush=True)
''' % ROOT
for name, lib in (("full kernel", None), ("1: producers load only (no split, no LDS stores)", "1"), ("2: consumers read LDS only (no MFMA)", "2"),
                  ("3: consumers MFMA only (no LDS reads)", "3"),
                  ("4: two planes, three products per K step (cost model of a 3-product split scheme)", "4")):
    env = dict(os.environ)
    if lib:
        env["SG_LIB_OVERRIDE"] = os.path.join(ROOT, "tools", "ablate", lib, "libstargcn_hip.so")
    print(name, flush=True)
    subprocess.run([sys.executable, "-c", code], env=env)
