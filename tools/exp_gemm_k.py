#!/usr/bin/env python3
"""GEMM efficiency vs K at the step's tile counts: separates per-workgroup fixed cost (prologue / epilogue) from the K loop."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3

for M, N in ((10677, 2560), (69878, 256)):
    for K in (64, 128, 256, 512, 1024, 2048):
        a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda")
        t = timeit(lambda: ops.gemm(a, b, trans_b=True))
        tt = timeit(lambda: torch.matmul(a, b.t()))
        print("M=%6d N=%5d K=%5d  %7.3f ms %6.1f TF/s   (rocBLAS %7.3f ms %6.1f TF/s)" % (M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e12, tt * 1e3, 2.0 * M * N * K / tt / 1e12))
