"""f16x3 plane kernel: where inside a K-tile iteration a wave issues its DMA (libraries built with
-DSG_X3_ISSUE_POS=1|2|3 [-DSG_X3_TIMING=1] under tools/ablate/):  python tools/exp_x3_issue_pos.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import ctypes, os, sys, torch
sys.path.insert(0, %r)
from star_gcn_amd import ops, _lib as L
from tools.microbench import timeit
lib = L.lib(); lib.sg_gemm_backend(3)
timing = os.environ.get("X3_TIMING") == "1"
if timing:
    raw = ctypes.CDLL(os.environ["SG_LIB_OVERRIDE"]); buf = (ctypes.c_ulonglong * 6)()
out = []
for (M, N, K) in [(4096, 4096, 4096), (262144, 4096, 1024), (262144, 256, 4160)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
    ref = (a[:256].double() @ b[:256].double().t())
    for v in (1, 6, 2):
        lib.sg_gemm_x3_variant(v)
        c = ops.gemm(a, b, trans_b=True)
        err = float((c[:256, :256].double() - ref).abs().max() / ref.abs().max())
        if timing:
            raw.sg_x3_timing_read(buf); ops.gemm(a, b, trans_b=True); raw.sg_x3_timing_read(buf)
            n = max(buf[5], 1)
            out.append("%%dx%%dx%%d v%%d wait %%.0f bar %%.0f issue %%.0f reads %%.0f mfma %%.0f" %% ((M, N, K, v) + tuple(buf[q] / n for q in range(5))))
        else:
            t = timeit(lambda: ops.gemm(a, b, trans_b=True), n=7, warm=2)
            out.append("%%dx%%dx%%d v%%d %%.3f ms %%.0f TF err %%.1e" %% (M, N, K, v, t * 1e3, 2.0 * M * N * K / t / 1e12, err))
print(" | ".join(out))
''' % ROOT
CASES = [("default", None, 0), ("pos1", "p1t0", 0), ("pos2", "p2t0", 0), ("pos3", "p3t0", 0),
         ("pos0 t", "x3timing", 1), ("pos1 t", "p1t1", 1), ("pos2 t", "p2t1", 1), ("pos3 t", "p3t1", 1),
         ("prio1", "pr1", 0), ("prio2", "pr2", 0), ("nofold", "nf", 0)]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if c[0] in sys.argv[1:]]
for tag, lib, tim in CASES:
    env = dict(os.environ)
    if lib:
        env["SG_LIB_OVERRIDE"] = os.path.join(ROOT, "tools", "ablate", "libstargcn_%s.so" % lib)
    env["X3_TIMING"] = str(tim)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print("%-8s %s" % (tag, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
