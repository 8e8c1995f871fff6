"""one GEMM shape, a few launches (profiling target):  python tools/one_gemm.py M N K ta tb backend"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L
M, N, K, ta, tb, be = (int(x) for x in sys.argv[1:7])
L.lib().sg_gemm_backend(be)
a = torch.randn((K, M) if ta else (M, K), device="cuda")
b = torch.randn((N, K) if tb else (K, N), device="cuda")
for _ in range(4):
    ops.gemm(a, b, trans_a=bool(ta), trans_b=bool(tb))
torch.cuda.synchronize()
