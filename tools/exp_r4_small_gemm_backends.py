"""Round-4 check: the SMALL dense products of the ML-10M step (1.4 - 2.3 GFLOP each, ~0.45 ms per step together) on every backend:
0 exact fp32 MFMA (64- / 128-row tiles), 2 x6v2 (persistent bf16x6), 3 f16x3 (default from K = 96 on), and hipBLASLt (torch).
python tools/exp_r4_small_gemm_backends.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L
from tools.microbench import timeit
SHAPES = [  # (M, N, K, trans_a, trans_b, what)
    (10677, 256, 256, False, True, "out_fc fwd (item)"), (10677, 256, 256, False, False, "out_fc dX (item)"), (256, 256, 10677, True, False, "out_fc dW (item)"),
    (69878, 64, 256, False, True, "rating proj fwd (user)"), (69878, 256, 64, False, False, "rating proj dX (user)"), (64, 256, 69878, True, False, "rating proj dW (user)"),
    (10677, 64, 256, False, True, "rating proj fwd (item)"), (10677, 256, 64, False, False, "rating proj dX (item)"), (64, 256, 10677, True, False, "rating proj dW (item)"),
    (69878, 256, 256, False, True, "out_fc fwd (user)"), (256, 256, 69878, True, False, "out_fc dW (user)")]
for (M, N, K, ta, tb, what) in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    res = []
    for be in (-1, 0, 2, 3):
        L.lib().sg_gemm_backend(be)
        for tm in ((None,) if be != 0 else ("64", "128")):
            if tm: os.environ["SG_GEMM_TM"] = tm
            else: os.environ.pop("SG_GEMM_TM", None)
            try:
                t = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb), n=15, warm=3)
                res.append("%s%s %5.1f us" % ({-1: "default", 0: "fp32/", 2: "x6v2", 3: "f16x3"}[be], tm or "", t * 1e6))
            except Exception as e:
                res.append("be%d err" % be)
    os.environ.pop("SG_GEMM_TM", None)
    A, B = (a.t() if ta else a), (b.t() if tb else b)
    t = timeit(lambda: torch.matmul(A, B), n=15, warm=3)
    print("%-24s M=%6d N=%4d K=%6d  %s   torch %5.1f us" % (what, M, N, K, "  ".join(res), t * 1e6), flush=True)
L.lib().sg_gemm_backend(-1)
