"""one gather launch configuration, a few launches (profiling target): python tools/one_gather.py FRAC [slices]
10 M edges, 1 KiB rows, skewed popularity (sigma 1.0, 69878-row source), rows drawn from the first 1/FRAC of the matrix"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L
frac = int(sys.argv[1]); sl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(0)
nnz, C, S, T = 10_000_000, 256, 106770, 69878
lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 2.0))
indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
seg = np.repeat(np.arange(S), lens)
pop = rng.lognormal(0.0, 1.0, T); rng.shuffle(pop)
Tp = T // frac
idx = rng.choice(Tp, size=nnz, p=pop[:Tp] / pop[:Tp].sum()).astype(np.int64)
idx_d = torch.from_numpy(idx[np.lexsort((idx, seg))].astype(np.int32)).cuda()
w = torch.rand(nnz).cuda(); x = torch.randn(T, C, device="cuda"); out = torch.empty(S, C, device="cuda")
L.lib().sg_gather_tuning(-1, sl)
for _ in range(4):
    ops.gather_sum(out, x, idx_d, indptr, w, S, C)
torch.cuda.synchronize()
