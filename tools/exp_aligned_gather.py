#!/usr/bin/env python3
"""Experiment: what would segment-aligned chunks buy?  Same edges / sources, (a) every segment exactly 128 edges so that no
segment crosses a 256-edge chunk boundary (no partial rows, fix-up has nothing to do) vs (b) random segment lengths with
the same mean."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

rng = np.random.default_rng(0)
S, T, C = 78125, 106770, 256
nnz = S * 128
src = torch.randn(T, C, device="cuda")
dst = torch.empty(S, C, device="cuda")
w = torch.rand(nnz, device="cuda")
for name, lens in (("aligned: every segment 128 edges", np.full(S, 128)),
                   ("random lengths, mean 128", rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 2.0)))):
    indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    # ascending source rows inside each segment, like a CSR neighbour list
    idx = np.sort(rng.integers(0, T, (S, 128)), axis=1).reshape(-1) if lens.min() == 128 else None
    if idx is None:
        idx = rng.integers(0, T, nnz)
        seg = np.repeat(np.arange(S), lens)
        order = np.lexsort((idx, seg))
        idx = idx[order]
    idx = torch.from_numpy(idx.astype(np.int32)).cuda()
    t = timeit(lambda: ops.gather_sum(dst, src, idx, indptr, w, S, C))
    print("%-40s %7.3f ms   (%d edges, source %.0f MB)" % (name, t, nnz, T * C * 4 / 2**20))
