"""CPU baseline of the benchmark step -- TEST INFRASTRUCTURE ONLY (bench.py `cpu_baseline`, kind "port").

The same 2-layer multi-link GCN forward+backward as bench.py, executed the way the reference executes it on
`mx.cpu()`: per rating level one dense FullyConnected (here torch-CPU `linear`, i.e. the host BLAS, standing in
for MXNet's MKL/OpenBLAS FullyConnected) followed by one `seg_weighted_pool` call (reference
aggregators.py:141-149), whose forward/backward are the C restatement of the reference CPU kernels with the
reference's own OpenMP placement (forward parallel over rows, seg_op.cc:198; backward w.r.t. data serial for a
batch of 1, seg_op.cc:232-233).  `fair=True` swaps in the row-parallel backward (clearly not the reference).
"""
import os
import time

import numpy as np
import torch

from . import seg as O


class _SegWeightedPoolCPU(torch.autograd.Function):
    fair = False

    @staticmethod
    def forward(ctx, data, weights, indices, indptr):
        ctx.meta = (weights, indices, indptr, data.shape[0])
        out = O.seg_weighted_pool(data.detach().numpy()[None], weights[None], indices, indptr)[0]
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, og):
        weights, indices, indptr, T = ctx.meta
        g = O.seg_weighted_pool_bwd_data(weights[None], og.contiguous().numpy()[None], indices, indptr, T,
                                         fair=_SegWeightedPoolCPU.fair)[0]
        return torch.from_numpy(g), None, None, None


def leaky(x):
    return torch.where(x > 0, x, 0.1 * x)


def run_cpu_step(levels, n_user, n_item, D, steps=1, fair=False, seed=0):
    """levels: dict direction -> (end_points_l, indptr_l, support_l) for ('user','item') [dst=user] and
    ('item','user') [dst=item]; pairs: the user->item CSR (all ratings).  Returns seconds per step."""
    _SegWeightedPoolCPU.fair = fair
    g = torch.Generator().manual_seed(seed)
    R = len(levels["user"][0])
    emb = {"user": (torch.rand(n_user, D, generator=g) * 0.2 - 0.1).requires_grad_(True),
           "item": (torch.rand(n_item, D, generator=g) * 0.2 - 0.1).requires_grad_(True)}
    s = (3.0 / D) ** 0.5
    params = []

    def mk(*shape):
        p = ((torch.rand(*shape, generator=g) * 2 - 1) * s).requires_grad_(True)
        params.append(p)
        return p

    layers = []
    for _ in range(2):
        layers.append({k: dict(w=[mk(D, D) for _ in range(R)], b=[torch.zeros(D, requires_grad=True) for _ in range(R)],
                               ow=mk(D, D), ob=torch.zeros(D, requires_grad=True)) for k in ("user", "item")})
    pu_w, pi_w = mk(64, D), mk(64, D)
    ep_u, ip_u, _ = levels["pairs"]
    y = torch.randn(ep_u.shape[0], generator=g)
    seg_of = torch.from_numpy(np.repeat(np.arange(n_user), np.diff(ip_u))).long()
    items_of = torch.from_numpy(ep_u.astype(np.int64))

    def step():
        x = dict(emb)
        for lay in layers:
            nxt = dict()
            for dst, src in (("user", "item"), ("item", "user")):
                eps, ips, sps = levels[dst]
                p = lay[dst]
                acc = None
                for r in range(R):   # reference order: FullyConnected, then seg_weighted_pool, per level
                    h = torch.nn.functional.linear(x[src], p["w"][r], p["b"][r])
                    o = _SegWeightedPoolCPU.apply(h, sps[r], eps[r], ips[r])
                    acc = o if acc is None else acc + o
                nxt[dst] = leaky(torch.nn.functional.linear(leaky(acc), p["ow"], p["ob"]))
            x = nxt
        pu = torch.nn.functional.linear(x["user"], pu_w)
        pi = torch.nn.functional.linear(x["item"], pi_w)
        pred = (pu[seg_of] * pi[items_of]).sum(dim=1)
        loss = (0.5 * (pred - y) ** 2).mean()
        loss.backward()
        return float(loss.detach())

    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    return (time.perf_counter() - t0) / steps


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "logical_cores": os.cpu_count(), "torch_threads": torch.get_num_threads()}
