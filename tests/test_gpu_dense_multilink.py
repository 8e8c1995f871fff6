"""GPU parity of the dense mix (fp32 MFMA GEMM) and the fused multi-link aggregation against the float64
layer oracle that follows the reference's operation order (oracle/model.py; reference aggregators.py:111-163).
fp32 tolerance 1e-5 relative to the output scale (north star)."""
import numpy as np
import pytest
import torch

from oracle import model as OM
from tests.test_abi_and_host import make_multilink

pytestmark = pytest.mark.gpu


def rel_close(got, ref, tol=1e-5, what=""):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert err <= tol * scale, "%s: max abs err %.3e > %.1e * scale %.3e" % (what, err, tol, scale)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1, 1, 1), (130, 250, 75), (257, 64, 2570), (64, 515, 64),
                                   (1000, 75, 250), (5, 300, 1027)])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("tile_rows", ["64", "128"])
def test_gemm_layouts_and_edges(M, N, K, ta, tb, tile_rows, monkeypatch):
    from star_gcn_amd import ops
    monkeypatch.setenv("SG_GEMM_TM", tile_rows)   # exercise both tile shapes (normally picked by a wave-count model)
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)   # asymmetric random operands (transpose-detecting)
    bias = torch.randn(N, generator=g)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())
    out = ops.gemm(A.cuda(), B.cuda(), trans_a=ta, trans_b=tb)
    rel_close(out, ref, 2e-6 * max(1, K ** 0.5), "plain")
    out = ops.gemm(A.cuda(), B.cuda(), trans_a=ta, trans_b=tb, bias=bias.cuda(), act="leaky", slope=0.1)
    rel_close(out, OM.leaky(ref + bias.double()), 2e-6 * max(1, K ** 0.5), "bias+leaky")
    c0 = torch.randn(M, N, generator=g)
    out = ops.gemm(A.cuda(), B.cuda(), trans_a=ta, trans_b=tb, out=c0.clone().cuda(), accumulate=True)
    rel_close(out, ref + c0.double(), 2e-6 * max(1, K ** 0.5), "accumulate")


def test_gemm_split_k_and_strided_views():
    from star_gcn_amd import ops
    g = torch.Generator().manual_seed(1)
    # weight-gradient shape: tiny M,N, huge K (split-K path), operands are column slices of wider matrices
    Kbig, M, N = 40000, 96, 200
    dY = torch.randn(Kbig, M + 8, generator=g).cuda()
    X = torch.randn(Kbig, N + 4, generator=g).cuda()
    a, b = dY[:, 4:4 + M], X[:, :N]
    out = ops.gemm(a, b, trans_a=True)
    ref = a.double().t() @ b.double()
    rel_close(out, ref, 3e-5, "split-k")
    # all activations of the epilogue
    x = torch.randn(300, 64, generator=g).cuda()
    w = torch.randn(40, 64, generator=g).cuda()
    pre = x.double() @ w.double().t()
    for act in ("relu", "sigmoid", "tanh", "leaky", None):
        rel_close(ops.gemm(x, w, trans_b=True, act=act), OM.ACTS[act](pre), 1e-5, str(act))


@pytest.mark.parametrize("act", [None, "leaky", "tanh"])
def test_linear_autograd(act):
    from star_gcn_amd import functional as F
    g = torch.Generator().manual_seed(3)
    x = torch.randn(777, 75, generator=g)
    w = torch.randn(250, 75, generator=g) * 0.1
    b = torch.randn(250, generator=g) * 0.1
    gy = torch.randn(777, 250, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    OM.dense(xr, wr, br, act).backward(gy.double())
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = F.linear(xd, wd, bd, act=act)
    y.backward(gy.cuda())
    rel_close(y, OM.dense(xr, wr, br, act), 1e-5, "y")
    rel_close(xd.grad, xr.grad, 1e-5, "dx")
    rel_close(wd.grad, wr.grad, 2e-5, "dw")
    rel_close(bd.grad, br.grad, 2e-5, "db")


CASES = [  # n_dst, n_src, nnz, R, D, U
    (60, 45, 900, 5, 64, 250),
    (45, 60, 900, 5, 32, 250),
    (300, 40, 6000, 10, 256, 256),
    (31, 500, 4000, 3, 75, 75),
    (20, 20, 50, 4, 16, 32),
]


@pytest.mark.parametrize("n_dst,n_src,nnz,R,D,U", CASES)
@pytest.mark.parametrize("accum", ["sum", "stack"])
@pytest.mark.parametrize("order", ["transform_first", "aggregate_first"])
def test_multilink_aggregate_matches_reference_order(n_dst, n_src, nnz, R, D, U, accum, order):
    from star_gcn_amd import functional as F
    from star_gcn_amd.plan import MultiLinkPlan
    if accum == "stack":
        U = (U // R) * R
    uc = U // R if accum == "stack" else U
    rng = np.random.default_rng(n_dst + nnz + R)
    eps, ips, sps = make_multilink(rng, n_dst, n_src, nnz, R)
    g = torch.Generator().manual_seed(R + D)
    x = torch.randn(n_src, D, generator=g) * 0.1
    ws = [torch.randn(uc, D, generator=g) * (3.0 / D) ** 0.5 for _ in range(R)]
    bs = [torch.randn(uc, generator=g) * 0.1 for _ in range(R)]
    gy = torch.randn(n_dst, U, generator=g)
    # float64 oracle in the reference's order
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    br = [b.double().requires_grad_(True) for b in bs]
    ref = OM.multilink_aggregator(xr, wr, br, eps, ips, sps, accum=accum, act="leaky")
    ref.backward(gy.double())
    # HIP path
    plan = MultiLinkPlan(eps, ips, sps, n_src, "cuda")
    xd = x.cuda().requires_grad_(True)
    wd = [w.cuda().requires_grad_(True) for w in ws]
    bd = [b.cuda().requires_grad_(True) for b in bs]
    out = F.multilink_aggregate(xd, wd, bd, plan, accum=accum, act="leaky", slope=0.1, order=order)
    out.backward(gy.cuda())
    rel_close(out, ref, 1e-5, "out")
    rel_close(xd.grad, xr.grad, 1e-5, "dx")
    for r in range(R):
        rel_close(wd[r].grad, wr[r].grad, 2e-5, "dW%d" % r)
        rel_close(bd[r].grad, br[r].grad, 2e-5, "db%d" % r)


def test_take_rows_and_masked_embed():
    from star_gcn_amd import functional as F
    from star_gcn_amd import ops
    from star_gcn_amd.plan import TakePlan
    rng = np.random.default_rng(2)
    n_rows, dim, n = 97, 64, 400
    table = torch.randn(n_rows, dim)
    ids = rng.integers(0, n_rows, n).astype(np.int32)
    noise = np.arange(n_rows, dtype=np.int32)
    noise[rng.random(n_rows) < 0.2] = -1                      # zero-mask
    swap = rng.random(n_rows) < 0.1
    noise[swap] = rng.integers(0, n_rows, int(swap.sum()))    # replace by another node's embedding
    got = ops.masked_embed(table.cuda(), torch.from_numpy(ids).cuda(), torch.from_numpy(noise).cuda())
    ref = OM.masked_embed(table, ids, noise)
    assert torch.equal(got.cpu(), ref)
    resolved = noise[ids]
    tp = TakePlan(resolved, n_rows, "cuda")
    t = table.cuda().requires_grad_(True)
    gy = torch.randn(n, dim)
    out = F.take_rows(t, tp)
    assert torch.equal(out.detach().cpu(), ref)
    out.backward(gy.cuda())
    tr = table.double().requires_grad_(True)
    OM.masked_embed(tr, ids, noise).backward(gy.double())
    rel_close(t.grad, tr.grad, 1e-5, "dtable")
