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


TIMES = {"seg_fwd": 0.0, "seg_bwd": 0.0}


class _SegWeightedPoolCPU(torch.autograd.Function):
    fair = False

    @staticmethod
    def forward(ctx, data, weights, indices, indptr):
        ctx.meta = (weights, indices, indptr, data.shape[0])
        t0 = time.perf_counter()
        out = O.seg_weighted_pool(data.detach().numpy()[None], weights[None], indices, indptr)[0]
        TIMES["seg_fwd"] += time.perf_counter() - t0
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, og):
        weights, indices, indptr, T = ctx.meta
        t0 = time.perf_counter()
        g = O.seg_weighted_pool_bwd_data(weights[None], og.contiguous().numpy()[None], indices, indptr, T,
                                         fair=_SegWeightedPoolCPU.fair)[0]
        TIMES["seg_bwd"] += time.perf_counter() - t0
        return torch.from_numpy(g), None, None, None


def leaky(x):
    return torch.where(x > 0, x, 0.1 * x)


def run_cpu_step(levels, n_user, n_item, D, steps=1, fair=False, seed=0, phases=None):
    """levels: dict direction -> (end_points_l, indptr_l, support_l) for ('user','item') [dst=user] and
    ('item','user') [dst=item]; pairs: the user->item CSR (all ratings).  Returns seconds per step.
    phases (dict, optional): filled with seconds per step of {forward, backward, seg_fwd, seg_bwd, dense_fwd,
    dense_bwd} -- seg_* are the seg_weighted_pool kernels, dense_* everything else (BLAS, activations, rating head)."""
    _SegWeightedPoolCPU.fair = fair
    TIMES["seg_fwd"] = TIMES["seg_bwd"] = 0.0
    t_fwd = [0.0]
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
        t_fwd[0] += time.perf_counter() - t_step[0]
        loss.backward()
        return float(loss.detach())

    t_step = [0.0]
    t0 = time.perf_counter()
    for _ in range(steps):
        t_step[0] = time.perf_counter()
        step()
    total = (time.perf_counter() - t0) / steps
    if phases is not None:
        fwd = t_fwd[0] / steps
        sf, sb = TIMES["seg_fwd"] / steps, TIMES["seg_bwd"] / steps
        phases.update(forward=fwd, backward=total - fwd, seg_fwd=sf, seg_bwd=sb, dense_fwd=fwd - sf,
                      dense_bwd=total - fwd - sb)
    return total


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
    return {"cpu_model": model, "logical_cores": os.cpu_count(), "physical_cores": physical_cores(),
            "torch_threads": torch.get_num_threads(), "omp_num_threads": os.environ.get("OMP_NUM_THREADS"),
            "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "omp_places": os.environ.get("OMP_PLACES")}


def physical_cores():
    """number of distinct (package, core) pairs in /proc/cpuinfo (SMT siblings counted once)"""
    cores, phys, core = set(), None, None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return len(cores) or (os.cpu_count() or 1)
