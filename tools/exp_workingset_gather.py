"""How does the (column-sliced) gather respond to the per-XCD working set?  Same launch (10 M edges, 1 KiB rows, skewed
row popularity as in the ML-10M-shaped step), but the source rows are drawn from the first 1/1, 1/2, 1/4, 1/8 of the
matrix: what a split of the XCDs over source-row ranges would give each XCD."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops  # noqa: E402
from tools.microbench import timeit  # noqa: E402

rng = np.random.default_rng(0)
nnz, C = 10_000_000, 256
for name, S, T, sigma, group in (("users<-(item,level) rows, sigma 1.5", 69878, 106770, 1.5, 1),
                                 ("(item,level)<-user rows, sigma 1.0", 106770, 69878, 1.0, 1)):
    lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 2.0))
    indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    seg = np.repeat(np.arange(S), lens)
    w = torch.rand(nnz).cuda()
    x = torch.randn(T, C, device="cuda")
    out = torch.empty(S, C, device="cuda")
    pop = rng.lognormal(0.0, sigma, T)
    rng.shuffle(pop)
    for frac in (1, 2, 4, 8):
        Tp = T // frac
        p = pop[:Tp] / pop[:Tp].sum()
        idx = rng.choice(Tp, size=nnz, p=p).astype(np.int64)
        order = np.lexsort((idx, seg))                     # ascending source row inside every segment
        idx_d = torch.from_numpy(idx[order].astype(np.int32)).cuda()
        for sl in (1, 4):
            from star_gcn_amd import _lib as L
            L.lib().sg_gather_tuning(-1, sl)
            t = timeit(lambda: ops.gather_sum(out, x, idx_d, indptr, w, S, C))
            print("%-38s rows drawn from first 1/%d (%5.1f MB)  slices %d  %7.3f ms  %7.1f GB/s" %
                  (name, frac, Tp * C * 4 / 2 ** 20, sl, t * 1e3, (8 + 4 * C) * nnz / t / 1e9), flush=True)
        L.lib().sg_gather_tuning(-1, 0)
