"""Full-matrix comparison of the 256-wide hybrid kernel (variant 8) with the 128-wide one (5) and fp64 on ML-10M step shapes."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import _lib as L
from star_gcn_amd import ops

g = torch.Generator(device="cuda").manual_seed(3)
for (M, N, K, ta, tb) in [(69878, 256, 2624, False, True), (69878, 256, 256, False, True), (10677, 256, 2560, False, False),
                          (256, 2624, 69878, True, False), (2560, 256, 10677, True, False)]:
    A = torch.randn((K, M) if ta else (M, K), generator=g, device="cuda")
    B = torch.randn((N, K) if tb else (K, N), generator=g, device="cuda")
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    mag = (A.double().abs().t() if ta else A.double().abs()) @ (B.double().abs().t() if tb else B.double().abs())
    res = {}
    for v in (5, 8, 0):
        L.lib().sg_gemm_backend(3)
        L.lib().sg_gemm_x3_variant(v)
        out = ops.gemm(A, B, trans_a=ta, trans_b=tb).double()
        e = (out - ref).abs() / mag
        res[v] = out
        bad = (e > 1e-6).nonzero()
        print((M, N, K, ta, tb), "variant", v, "max err/mag %.3e" % float(e.max()), "mean signed err/mag %.3e" % float(((out - ref) / mag).mean()),
              "bad elements", int(bad.shape[0]), bad[:5].tolist())
    L.lib().sg_gemm_x3_variant(-1)
    L.lib().sg_gemm_backend(-1)
