"""Round 6: would the user-side output layers (69 878 x 256 x 256: 273 items of 256 rows on 256 CUs) gain from running the
first 256 items on the persistent 256-wide kernel (gemm_x3w.hip) and the 4 342 remaining rows on the 128-wide kernel?
Times, per form (NT = forward, NN = data gradient), events around 50 back-to-back calls:
  whole product, default routing | rows [0, 65536) on the 256-wide kernel (variant 8) | rows [65536, 69878) default | sum"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n * 1e3


M, N, K, CUT = 69878, 256, 256, 65536
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda")
out = torch.empty(M, N, device="cuda")
lib = L.lib()
for tb, name in ((True, "NT (forward)"), (False, "NN (data gradient)")):
    b = w if tb else w.t().contiguous()
    lib.sg_gemm_backend(3)
    lib.sg_gemm_x3_variant(-1)
    whole = t(lambda: ops.gemm(x, b, trans_b=tb, out=out))
    tail = t(lambda: ops.gemm(x[CUT:], b, trans_b=tb, out=out[CUT:]))
    lib.sg_gemm_x3_variant(8)
    head = t(lambda: ops.gemm(x[:CUT], b, trans_b=tb, out=out[:CUT]))
    whole8 = t(lambda: ops.gemm(x, b, trans_b=tb, out=out))
    lib.sg_gemm_x3_variant(-1)

    def both():
        lib.sg_gemm_x3_variant(8)
        ops.gemm(x[:CUT], b, trans_b=tb, out=out[:CUT])
        lib.sg_gemm_x3_variant(-1)
        ops.gemm(x[CUT:], b, trans_b=tb, out=out[CUT:])
    pair = t(both)
    print("%-20s whole %.1f us | whole on the 256-wide kernel %.1f | head 65536 rows 256-wide %.1f | tail 4342 rows %.1f | head + tail back to back %.1f" % (
        name, whole, whole8, head, tail, pair), flush=True)
lib.sg_gemm_backend(-1)
