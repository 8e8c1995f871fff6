"""The adjoint identity <A x, y> == <x, A^T y> of the fused aggregation at ML-10M size (tests/test_gpu_dense_multilink.py,
property 3) under the 128-wide (variant 5) and the routed (0) hybrid kernels: how far apart are the two sides?"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import star_gcn_amd.synthetic as S
from star_gcn_amd import functional as F
from star_gcn_amd import _lib as L
from star_gcn_amd.plan import MultiLinkPlan
graph, eu, ei, vals = S.make_graph("ml-10m")
m = graph["user", "movie"]
eps, _, ips, sps = m.sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
plan = MultiLinkPlan(eps, ips, sps, m.shape[1], "cuda")
R, D, U = plan.R, 256, 256
for variant in (5, 0, 8):
    L.lib().sg_gemm_x3_variant(variant)
    g = torch.Generator(device="cuda").manual_seed(0)
    x1 = torch.randn(plan.n_src, D, device="cuda", generator=g) * 0.1
    x2 = torch.randn(plan.n_src, D, device="cuda", generator=g) * 0.1
    ws = [torch.randn(U, D, device="cuda", generator=g) * (3.0 / D) ** 0.5 for _ in range(R)]
    bs = [torch.randn(U, device="cuda", generator=g) * 0.1 for _ in range(R)]
    f = lambda x, b, order: F.multilink_aggregate(x, ws, b, plan, accum="sum", act=None, order=order)
    for order in ("transform_first", "aggregate_first"):
        xg = x1.clone().requires_grad_(True)
        y = torch.randn(plan.n_dst, U, device="cuda", generator=g)
        out = f(xg, bs, order)
        out.backward(y)
        const = f(torch.zeros_like(x1), bs, order)
        lhs = float(((out.detach() - const).double() * y.double()).sum())
        rhs = float((xg.grad.double() * x1.double()).sum())
        mag = float(((out.detach() - const).double().abs() * y.double().abs()).sum())
        print("variant", variant, order, "lhs %.7f rhs %.7f diff %.3e  (sum |terms| %.3e: diff / that %.2e)" % (lhs, rhs, lhs - rhs, mag, abs(lhs - rhs) / mag))
L.lib().sg_gemm_x3_variant(-1)
