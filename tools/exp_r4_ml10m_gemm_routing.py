"""Round-4 check of the GEMM routing on the ML-10M step's shapes: default (0) against "never split inside the kernel" (4: every
operand pre-split into planes) and "always split inside the kernel" (5), and hipBLASLt fp32 (torch.matmul) for scale.
python tools/exp_r4_ml10m_gemm_routing.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L
from tools.microbench import timeit
L.lib().sg_gemm_backend(3)
SHAPES = [  # (M, N, K, trans_a, trans_b, what)
    (10677, 2560, 256, False, True, "TF fwd H"), (10677, 256, 2624, False, True, "AF fwd"), (69878, 256, 256, False, True, "out_fc fwd (user)"),
    (69878, 256, 256, False, False, "out_fc dX (user)"), (256, 256, 69878, True, False, "out_fc dW (user)"),
    (10677, 2624, 256, False, False, "AF bwd dZ"), (256, 2624, 10677, True, False, "AF bwd dWext"),
    (10677, 256, 2560, False, False, "TF bwd dX"), (2560, 256, 10677, True, False, "TF bwd dWcat"),
    (69878, 64, 256, False, True, "rating proj fwd"), (10677, 256, 256, False, True, "out_fc fwd (item)")]
for (M, N, K, ta, tb, what) in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    res = []
    for v in (0, 4, 5):
        L.lib().sg_gemm_x3_variant(v)
        try:
            t = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb), n=9, warm=3)
            res.append("v%d %6.1f us %6.1f TF/s" % (v, t * 1e6, 2.0 * M * N * K / t / 1e12))
        except Exception as e:
            res.append("v%d error %s" % (v, str(e)[:40]))
    A, B = (a.t() if ta else a), (b.t() if tb else b)
    t = timeit(lambda: torch.matmul(A, B), n=9, warm=3)
    print("%-20s M=%6d N=%5d K=%6d  %s   torch %6.1f us %6.1f TF/s" % (what, M, N, K, "  ".join(res), t * 1e6, 2.0 * M * N * K / t / 1e12), flush=True)
L.lib().sg_gemm_x3_variant(-1)
