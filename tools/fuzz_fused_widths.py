#!/usr/bin/env python3
"""Randomised differential check of the fused aggregate -> contract kernel at GENERAL widths (round 6, sg_agg_fused2_hip behind
order='fused'): rows of 4 .. 256 floats (multiple of 4), 1 .. 256 units per level, accum 'sum' / 'stack', 1 .. 12 levels,
heavy-tailed graphs of either orientation (n_dst < n_src: the forward saves the aggregates; n_dst > n_src: the data gradient
writes the R-expanded gradient), activation on / off -- forward, data gradient, weight and bias gradients through autograd
against the float64 layer oracle in the reference's operation order (oracle/model.py; aggregators.py:111-163).  Test
infrastructure:  python tools/fuzz_fused_widths.py [cases]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import star_gcn_amd  # noqa: E402,F401
from oracle import model as OM  # noqa: E402
from star_gcn_amd import functional as F  # noqa: E402
from star_gcn_amd.plan import MultiLinkPlan  # noqa: E402


def rel(got, ref):
    s = float(ref.abs().max())
    return float((got.double().cpu() - ref).abs().max()) / max(s, 1e-30)


def main(n_cases):
    rng = np.random.default_rng(606)
    worst = 0.0
    for case in range(n_cases):
        n_dst, n_src = int(rng.integers(1, 500)), int(rng.integers(1, 500))
        R = int(rng.integers(1, 13))
        d_in = 4 * int(rng.integers(1, 65))
        units = int(rng.integers(1, 257)) if case % 3 else int(rng.choice([50, 64, 75, 250, 256]))
        accum = "stack" if case & 1 else "sum"
        act = "leaky" if case & 2 else None
        eps, ips, sps = [], [], []
        a_row = 0.3 + 2.0 * rng.random()
        pr = np.arange(1, n_dst + 1, dtype=np.float64) ** -a_row
        pr = rng.permutation(pr / pr.sum())
        for r in range(R):
            n = int(10 ** (rng.random() * 3.7)) if rng.random() > 0.15 else 0
            cnt = rng.multinomial(n, pr) if n else np.zeros(n_dst, np.int64)
            ip = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
            e = rng.integers(0, n_src, n).astype(np.int32)
            sp = rng.uniform(0.05, 1.0, n).astype(np.float32)
            if n == 0:
                e, sp = np.zeros(1, np.int32), np.zeros(1, np.float32)      # reference graph.py:221-222 empty_as_zero
            eps.append(e); ips.append(ip); sps.append(sp)
        if sum(int(ip[-1]) for ip in ips) == 0:
            continue                                                        # the fused order needs at least one edge
        g = torch.Generator().manual_seed(case)
        x = torch.randn(n_src, d_in, generator=g) * 0.1 * torch.exp(torch.randn(n_src, 1, generator=g))
        ws = [torch.randn(units, d_in, generator=g) * (3.0 / d_in) ** 0.5 for _ in range(R)]
        bs = [torch.randn(units, generator=g) * 0.1 for _ in range(R)]
        gy = torch.randn(n_dst, units * (R if accum == "stack" else 1), generator=g)
        xr = x.double().requires_grad_(True)
        wr = [w.double().requires_grad_(True) for w in ws]
        br = [b.double().requires_grad_(True) for b in bs]
        ref = OM.multilink_aggregator(xr, wr, br, eps, ips, sps, accum=accum, act=act)
        ref.backward(gy.double())
        plan = MultiLinkPlan(eps, ips, sps, n_src, "cuda")
        xd = x.cuda().requires_grad_(True)
        wd = [w.cuda().requires_grad_(True) for w in ws]
        bd = [b.cuda().requires_grad_(True) for b in bs]
        out = F.multilink_aggregate(xd, wd, bd, plan, accum=accum, act=act, slope=0.1, order="fused")
        out.backward(gy.cuda())
        errs = {"out": rel(out.detach(), ref.detach()), "dx": rel(xd.grad, xr.grad),
                "dW": max(rel(torch.stack([w.grad for w in wd]), torch.stack([w.grad for w in wr])), 0.0),
                "db": rel(torch.stack([b.grad for b in bd]), torch.stack([b.grad for b in br]))}
        e = max(errs.values())
        worst = max(worst, e)
        print("case %3d: %3d x %3d rows, %2d levels, %4d -> %3d %-5s act %-5s: %s%s" % (
            case, n_dst, n_src, R, d_in, units, accum, act, " ".join("%s %.1e" % kv for kv in errs.items()),
            "   <-- FAIL" if e > 1e-5 else ""), flush=True)
    print("WORST %.3g (%s; tolerance 1e-5 of each tensor's scale)" % (worst, "ok" if worst <= 1e-5 else "FAIL"))
    return 0 if worst <= 1e-5 else 1


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 200))
