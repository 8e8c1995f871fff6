"""K = 64 products of the rating head (dX = dP W: n x 256 x 64) and M <= 64 weight gradients: exact-fp32 kernel (the default
routing for K <= 64) against the forced f16x3 backend."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L
from tools.microbench import timeit
for (M, N, K, ta, tb) in [(69878, 256, 64, False, False), (10677, 256, 64, False, False), (69878, 64, 256, False, True),
                          (10677, 64, 256, False, True), (64, 256, 69878, True, False), (64, 256, 10677, True, False)]:
    a = torch.randn((K, M) if ta else (M, K), device="cuda"); b = torch.randn((N, K) if tb else (K, N), device="cuda")
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double())
    out = []
    for be in (-1, 0, 2, 3):
        L.lib().sg_gemm_backend(be)
        c = ops.gemm(a, b, trans_a=ta, trans_b=tb)
        err = float((c.double() - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb), n=9, warm=3)
        out.append("%s %.1f us (%.0f TF, err %.1e)" % ({-1: "default", 0: "fp32", 2: "x6v2", 3: "f16x3"}[be], t * 1e6, 2.0 * M * N * K / t / 1e12, err))
    L.lib().sg_gemm_backend(-1)
    print("%6d x %4d x %6d %s%s  " % (M, N, K, "T" if ta else "N", "T" if tb else "N") + " | ".join(out), flush=True)
