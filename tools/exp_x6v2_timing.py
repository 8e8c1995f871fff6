"""Per-phase cycle counters of the x6v2 GEMM (development build with -DSG_X6V2_TIMING=1 in tools/ablate/t, loaded through
SG_LIB_OVERRIDE): where a K-tile step of a consumer wave / a producer wave goes."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SG_LIB_OVERRIDE"] = os.path.join(ROOT, "tools", "ablate", os.environ.get("SG_TIMING_VARIANT", "t"), "libstargcn_hip.so")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from star_gcn_amd import ops, _lib as L  # noqa: E402
from tools.microbench import timeit  # noqa: E402

lib = ctypes.CDLL(os.environ["SG_LIB_OVERRIDE"])
L.lib().sg_gemm_backend(2)
buf = (ctypes.c_ulonglong * 8)()
for (M, N, K, ta, tb) in [(4096, 4096, 4096, False, True), (10677, 2560, 256, False, True), (10677, 256, 2624, False, True),
                          (10677, 256, 2560, False, False), (2560, 256, 10677, True, False), (1000000, 256, 256, False, True)]:
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    t = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb), n=5, warm=2)
    lib.sg_x6v2_timing_read(buf)
    ops.gemm(a, b, trans_a=ta, trans_b=tb)
    lib.sg_x6v2_timing_read(buf)
    v = list(buf)
    cs, ps = max(v[3], 1), max(v[7], 1)
    print("M=%7d N=%5d K=%5d ta=%d tb=%d  %7.3f ms %6.1f TF/s | consumer per step: barrier wait %6.0f  lds+mfma %6.0f  "
          "epilogue %6.0f (steps/wave %.0f) | producer per step: split+store %6.0f  load issue %6.0f  barrier wait %6.0f" %
          (M, N, K, ta, tb, t * 1e3, 2.0 * M * N * K / t / 1e12, v[0] / cs, v[1] / cs, v[2] / cs, cs / 256.0,
           v[4] / ps, v[5] / ps, v[6] / ps), flush=True)
