"""Column-slice count per launch class of the ML-10M-shaped step (68 MB plain user rows / 104 MB grouped (item, level) rows)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L  # noqa: E402
from tools.microbench import timeit  # noqa: E402

rng = np.random.default_rng(0)
nnz, C = 10_000_000, 256
for name, S, T, sigma in (("users<-(item,level) rows (104 MB)", 69878, 106770, 1.5), ("(item,level)<-user rows (68 MB)", 106770, 69878, 1.0)):
    lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 2.0))
    indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    seg = np.repeat(np.arange(S), lens)
    w = torch.rand(nnz).cuda()
    x = torch.randn(T, C, device="cuda")
    out = torch.empty(S, C, device="cuda")
    pop = rng.lognormal(0.0, sigma, T)
    idx = rng.choice(T, size=nnz, p=pop / pop.sum()).astype(np.int64)
    order = np.lexsort((idx, seg))
    idx_d = torch.from_numpy(idx[order].astype(np.int32)).cuda()
    for sl in (1, 2, 4, 8):
        L.lib().sg_gather_tuning(-1, sl)
        t = timeit(lambda: ops.gather_sum(out, x, idx_d, indptr, w, S, C))
        print("%-36s slices %d  %7.3f ms" % (name, sl, t * 1e3), flush=True)
    L.lib().sg_gather_tuning(-1, 0)
