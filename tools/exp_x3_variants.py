"""f16x3 plane-kernel geometries on the plane-path shapes:  python tools/exp_x3_variants.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L
from tools.microbench import timeit
L.lib().sg_gemm_backend(3)
for (M, N, K) in [(4096, 4096, 4096), (10677, 2560, 256), (1000000, 4096, 256), (262144, 4096, 1024)]:
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
    res = []
    for v in (1, 3, 6, 7):
        L.lib().sg_gemm_x3_variant(v)
        t = timeit(lambda: ops.gemm(a, b, trans_b=True), n=7, warm=2)
        res.append("v%d %8.3f ms %6.1f TF/s" % (v, t * 1e3, 2.0 * M * N * K / t / 1e12))
    print("M=%7d N=%5d K=%5d  %s" % (M, N, K, "  ".join(res)), flush=True)
    del a, b
L.lib().sg_gemm_x3_variant(-1)
