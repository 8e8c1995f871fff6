#!/usr/bin/env python
"""Differential fuzz of the multi-link aggregator's orders on heavy-tailed random graphs: transform-first against aggregate-first
(and the fused order where it applies), accum 'sum' / 'stack', random widths incl. the reference's 64 / 75 / 250, forward and
all gradients (no activation: orders that round a pre-activation to different sides of LeakyReLU's kink legitimately differ by
1e-5 on a hub row).  Prints every case; exit status 1 on a mismatch beyond 2e-5 of a tensor's scale."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import functional as F            # noqa: E402
from star_gcn_amd.plan import MultiLinkPlan          # noqa: E402


def main(n_cases):
    rng = np.random.default_rng(4242)
    bad = 0
    for case in range(n_cases):
        if os.environ.get("FUZZ_BIG") == "1":     # source matrices of 24 .. 256 MB: the gathers run as two source-range phases, XCD-sliced
            n_dst, n_src = int(rng.integers(2000, 30000)), int(rng.integers(2000, 30000))
            R = int(rng.integers(4, 17))
            nnz = int(10 ** rng.uniform(4.5, 6.4)) + 1
            D = U = 256 if rng.random() < 0.7 else 64
        else:
            n_dst, n_src = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
            R = int(rng.integers(1, 17))
            nnz = int(10 ** rng.uniform(0, 5.3)) + 1
            D = int(rng.choice([8, 64, 75, 250, 256, 100]))
            U = int(rng.choice([8, 64, 75, 250, 256, 33]))
        accum = "stack" if rng.random() < 0.3 else "sum"
        zipf = lambda n, a: (np.arange(1, n + 1) ** -a)[rng.permutation(n)]
        pd, ps, pl = zipf(n_dst, rng.uniform(0.3, 2.5)), zipf(n_src, rng.uniform(0.0, 1.5)), zipf(R, rng.uniform(0.0, 2.5))
        dst = rng.choice(n_dst, nnz, p=pd / pd.sum())
        src = rng.choice(n_src, nnz, p=ps / ps.sum()).astype(np.int32)
        lev = rng.choice(R, nnz, p=pl / pl.sum())
        sup = rng.uniform(0.05, 1.0, nnz).astype(np.float32)
        eps, ips, sps = [], [], []
        for r in range(R):
            sel = np.flatnonzero(lev == r)
            sel = sel[np.argsort(dst[sel], kind="stable")]
            ips.append(np.concatenate([[0], np.cumsum(np.bincount(dst[sel], minlength=n_dst))]).astype(np.int32))
            e, sp = src[sel], sup[sel]
            if e.size == 0:
                e, sp = np.zeros(1, np.int32), np.zeros(1, np.float32)
            eps.append(e); sps.append(sp)
        plan = MultiLinkPlan(eps, ips, sps, n_src, "cuda")
        g = torch.Generator().manual_seed(case)
        x = torch.randn(n_src, D, generator=g)
        ws = [torch.randn(U, D, generator=g) * (3.0 / D) ** 0.5 for _ in range(R)]
        bs = [torch.randn(U, generator=g) * 0.1 for _ in range(R)]
        gy = torch.randn(n_dst, U * (R if accum == "stack" else 1), generator=g).cuda()
        orders = ["transform_first", "aggregate_first"] + (["fused"] if accum == "sum" and D == 256 and U == 256 else [])
        res = {}
        for order in orders:
            xd = x.cuda().requires_grad_(True)
            wd = [w.cuda().requires_grad_(True) for w in ws]
            bd = [b.cuda().requires_grad_(True) for b in bs]
            out = F.multilink_aggregate(xd, wd, bd, plan, accum=accum, act=None, order=order)   # no kink: see tools/dbg_case2.py
            out.backward(gy)
            res[order] = (out.detach().double(), xd.grad.double(), torch.stack([w.grad for w in wd]).double(),
                          torch.stack([b.grad for b in bd]).double())
        worst = 0.0
        for order in orders[1:]:
            for a, b in zip(res[order], res[orders[0]]):
                worst = max(worst, float((a - b).abs().max()) / max(1.0, float(b.abs().max())))
        flag = "" if worst <= 2e-5 else "   <-- MISMATCH"
        bad += bool(flag)
        print("case %3d: %4d x %4d, %2d levels, %6d edges, in %3d units %3d, %-5s orders %d: worst %.2e%s" % (
            case, n_dst, n_src, R, nnz, D, U, accum, len(orders), worst, flag), flush=True)
    print("MISMATCHES %d of %d" % (bad, n_cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 100))
