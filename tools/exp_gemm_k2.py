"""per-item overhead vs per-step cost of the GEMM backends: time of (M x K) . (N x K)^T at fixed M, N over K"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L
from tools.microbench import timeit
M, N = 262144, 4096
for tb in (True, False):
    for K in (32, 64, 128, 256, 512, 1024):
        a = torch.randn(M, K, device="cuda")
        b = torch.randn((N, K) if tb else (K, N), device="cuda")
        res = []
        for be, nm in ((0, "fp32"), (1, "x6"), (2, "x6v2")):
            L.lib().sg_gemm_backend(be)
            t = timeit(lambda: ops.gemm(a, b, trans_b=tb), n=5, warm=2)
            res.append("%s %7.3f ms %6.1f TF/s" % (nm, t * 1e3, 2.0 * M * N * K / t / 1e12))
        L.lib().sg_gemm_backend(-1)
        tt = timeit(lambda: torch.matmul(a, b.t() if tb else b), n=5, warm=2)
        print("tb=%d M=%d N=%d K=%5d  %s  torch %7.3f ms %6.1f TF/s   (C write alone at 5 TB/s: %.2f ms)" %
              (tb, M, N, K, "  ".join(res), tt * 1e3, 2.0 * M * N * K / tt / 1e12, M * N * 4 / 5e12 * 1e3), flush=True)
