"""f16x3 plane kernel: per-phase cycle counters of wave 0 of every workgroup (library built with -DSG_X3_TIMING=1
under tools/ablate/libstargcn_x3timing.so):  python tools/exp_x3_timing.py"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import ctypes, os, sys, torch
sys.path.insert(0, %r)
from star_gcn_amd import ops, _lib as L
from tools.microbench import timeit
lib = L.lib(); lib.sg_gemm_backend(3)
raw = ctypes.CDLL(os.environ["SG_LIB_OVERRIDE"])
buf = (ctypes.c_ulonglong * 6)()
NAMES = ["vmcnt wait", "barrier", "DMA issue", "frag reads", "MFMAs", "k tiles"]
for (M, N, K) in [(4096, 4096, 4096), (262144, 4096, 1024)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
    for v in (1, 6, 2):
        lib.sg_gemm_x3_variant(v)
        ops.gemm(a, b, trans_b=True); raw.sg_x3_timing_read(buf)
        t = timeit(lambda: ops.gemm(a, b, trans_b=True), n=3, warm=1)
        raw.sg_x3_timing_read(buf)
        ops.gemm(a, b, trans_b=True); raw.sg_x3_timing_read(buf)
        n = max(buf[5], 1)
        tot = sum(buf[q] for q in range(5))
        print("%%dx%%dx%%d v%%d %%.3f ms: " %% (M, N, K, v, t * 1e3) + "  ".join("%%s %%.0f" %% (NAMES[q], buf[q] / n) for q in range(5))
              + "  | sum %%.0f per kstep-tile (counter ticks; 100 MHz s_memtime => x21 core clocks)" %% (tot / n), flush=True)
''' % ROOT
env = dict(os.environ)
env["SG_LIB_OVERRIDE"] = os.path.join(ROOT, "tools", "ablate", "libstargcn_x3timing.so")
r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
print(r.stdout); print(r.stderr[-1500:])
