"""The step's small GEMMs (one round of tiles, latency-bound): which backend is faster?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L  # noqa: E402
from tools.microbench import timeit  # noqa: E402

for (M, N, K, ta, tb) in [(10677, 256, 256, False, True), (10677, 256, 256, False, False), (256, 256, 10677, True, False),
                          (10677, 64, 256, False, True), (10677, 256, 64, False, False), (64, 256, 10677, True, False),
                          (8735, 256, 256, False, True), (69878, 64, 256, False, True), (69878, 256, 64, False, False)]:
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    res = []
    for be, nm in ((0, "fp32"), (1, "x6"), (2, "x6v2")):
        L.lib().sg_gemm_backend(be)
        t = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb), n=20, warm=5)
        res.append("%s %6.1f us" % (nm, t * 1e6))
    L.lib().sg_gemm_backend(-1)
    tt = timeit(lambda: torch.matmul(a.t() if ta else a, b.t() if tb else b), n=20, warm=5)
    print("M=%6d N=%4d K=%6d ta=%d tb=%d  %s  torch %6.1f us" % (M, N, K, ta, tb, "  ".join(res), tt * 1e6), flush=True)
