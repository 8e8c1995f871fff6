import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import functional as F
from star_gcn_amd.plan import MultiLinkPlan
from oracle import model as OM
target = int(sys.argv[1])
rng = np.random.default_rng(4242)
for case in range(target + 1):
    n_dst, n_src = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
    R = int(rng.integers(1, 17))
    nnz = int(10 ** rng.uniform(0, 5.3)) + 1
    D = int(rng.choice([8, 64, 75, 250, 256, 100]))
    U = int(rng.choice([8, 64, 75, 250, 256, 33]))
    accum = "stack" if rng.random() < 0.3 else "sum"
    accum = os.environ.get("DBG_ACCUM", accum) if case == target else accum
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
print(n_dst, n_src, R, nnz, D, U, accum, "largest row", np.bincount(dst).max())
plan = MultiLinkPlan(eps, ips, sps, n_src, "cuda")
g = torch.Generator().manual_seed(target)
x = torch.randn(n_src, D, generator=g)
ws = [torch.randn(U, D, generator=g) * (3.0 / D) ** 0.5 for _ in range(R)]
bs = [torch.randn(U, generator=g) * 0.1 for _ in range(R)]
gy = torch.randn(n_dst, U * (R if accum == "stack" else 1), generator=g)
xr = x.double().requires_grad_(True); wr = [w.double().requires_grad_(True) for w in ws]; br = [b.double().requires_grad_(True) for b in bs]
ref = OM.multilink_aggregator(xr, wr, br, eps, ips, sps, accum=accum, act="leaky")
ref.backward(gy.double())
refs = (ref.detach(), xr.grad, torch.stack([w.grad for w in wr]), torch.stack([b.grad for b in br]))
for order in ("transform_first", "aggregate_first"):
    xd = x.cuda().requires_grad_(True); wd = [w.cuda().requires_grad_(True) for w in ws]; bd = [b.cuda().requires_grad_(True) for b in bs]
    out = F.multilink_aggregate(xd, wd, bd, plan, accum=accum, act="leaky", slope=0.1, order=order)
    out.backward(gy.cuda())
    got = (out.detach(), xd.grad, torch.stack([w.grad for w in wd]), torch.stack([b.grad for b in bd]))
    for name, a, b in zip(("out", "dx", "dW", "db"), got, refs):
        a = a.double().cpu()
        err = float((a - b).abs().max()); sc = float(b.abs().max())
        print("%-16s %-3s err %.3e scale %.3e ratio %.2e" % (order, name, err, sc, err / max(sc, 1e-300)))
# the same network in plain torch fp32 (the oracle's operation order, float32 tensors on the CPU): what fp32 gives without this library
x3 = x.clone().requires_grad_(True); w3 = [w.clone().requires_grad_(True) for w in ws]; b3 = [b.clone().requires_grad_(True) for b in bs]
o3 = OM.multilink_aggregator(x3, w3, b3, eps, ips, sps, accum=accum, act="leaky")
o3.backward(gy)
got = (o3.detach(), x3.grad, torch.stack([w.grad for w in w3]), torch.stack([b.grad for b in b3]))
for name, a, b in zip(("out", "dx", "dW", "db"), got, refs):
    a = a.double().cpu()
    err = float((a - b).abs().max()); sc = float(b.abs().max())
    print("%-16s %-3s err %.3e scale %.3e ratio %.2e" % ("torch fp32", name, err, sc, err / max(sc, 1e-300)))
