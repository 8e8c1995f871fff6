"""Multi-process CPU test (gloo, world_size 2 and 4 -- uneven user blocks) of the node-partition communication pattern in star-gcn_amd/dist.py:
`copy_to_local` / `reduce_from_local` crossings + `allreduce_grads` must reproduce the single-process gradients of a
2-layer bipartite GCN whose users are sharded and whose items are replicated.  The sparse/dense math here is plain
torch-CPU (the HIP kernels need a GPU); what is under test is exactly the collective placement that bench.py uses
at N > 1."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(A, x_user, x_item, params, cross_in, cross_out, leaky):
    """A: (n_user_local, n_item) dense weights.  Two layers, both directions, then a pair loss over A's edges."""
    W_ui, W_iu, O_u, O_i, P_u, P_i = params
    for l in range(2):
        h_user = leaky(A @ (cross_in(x_item) @ W_ui[l].t()))                 # local aggregation over replicated items
        h_item = leaky(cross_out(A.t() @ (x_user @ W_iu[l].t())))            # partial over local users -> all-reduce
        x_user, x_item = leaky(h_user @ O_u[l].t()), leaky(h_item @ O_i[l].t())
    pu, pi = x_user @ P_u.t(), cross_in(x_item @ P_i.t())
    score = pu @ pi.t()
    return 0.5 * (((score - 1.0) ** 2) * (A != 0)).sum()


def _make(seed=0, nu=12, ni=7, d=5):
    g = torch.Generator().manual_seed(seed)
    A = (torch.rand(nu, ni, generator=g) < 0.4).double() * torch.rand(nu, ni, generator=g).double()
    xu, xi = torch.randn(nu, d, generator=g).double(), torch.randn(ni, d, generator=g).double()
    mk = lambda *s: (torch.randn(*s, generator=g).double() * 0.5)
    params = [[mk(d, d), mk(d, d)], [mk(d, d), mk(d, d)], [mk(d, d), mk(d, d)], [mk(d, d), mk(d, d)], mk(3, d), mk(3, d)]
    return A, xu, xi, params


# contiguous user blocks per world size; 4 ranks: uneven, one rank owns a single user
BLOCKS = {2: [(0, 5), (5, 12)], 4: [(0, 2), (2, 3), (3, 9), (9, 12)]}


def _flat(params):
    return [p for grp in params for p in (grp if isinstance(grp, list) else [grp])]


def _worker(rank, world, port, out_dir, split=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import star_gcn_amd.dist as SD
    assert SD.world() == world and SD.rank() == rank
    cross_in, cross_out = SD.copy_to_local, SD.reduce_from_local
    if split:   # the overlap-capable split forms used by heter_sage: launch / wait as separate autograd nodes
        fast = split == "owned"     # the layer's promises (fresh buffers, one reader): in-place sums, in-flight gradients;
                                    # the public defaults copy and block instead -- same numbers either way
        def cross_in(x):
            xw, pend = SD.grad_wait(x, exclusive=fast)
            return SD.copy_to_local_async(xw, pend, owned=fast)

        def cross_out(x):
            y, pend = SD.reduce_start(x, owned=fast)
            return SD.reduce_wait(y, pend)
    SD.STATS.reset()
    SD.STATS.enabled = True
    A, xu, xi, params = _make()
    lo, hi = BLOCKS[world][rank]
    leaky = lambda x: torch.where(x > 0, x, 0.1 * x)
    flat = _flat(params)
    for p in flat:
        p.requires_grad_(True)
    xu_l = xu[lo:hi].clone().requires_grad_(True)     # user embeddings: row-sharded
    xi_r = xi.clone().requires_grad_(True)            # item embeddings: replicated
    loss = _model(A[lo:hi], xu_l, xi_r, params, cross_in, cross_out, leaky)
    loss.backward()
    st = SD.STATS.read()
    # 2 forward + 3 backward data-path all-reduces (the layer-0 item embedding needs no gradient crossing beyond its own)
    assert st["calls"] == 5 and st["bytes"] > 0, st
    W_ui, W_iu, O_u, O_i, P_u, P_i = params
    local_region = W_ui + W_iu + O_u + [P_u]          # item-side O_i / P_i sit in the replicated region
    SD.allreduce_grads(local_region)
    tot = loss.detach().clone()
    dist.all_reduce(tot)
    torch.save({"loss": tot, "grads": [p.grad for p in flat], "gxu": xu_l.grad, "gxi": xi_r.grad, "lo": lo, "hi": hi},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,split", [(2, False), (2, "safe"), (2, "owned"), (4, "owned")])
def test_partition_pattern_matches_single_process(tmp_path, world, split):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), split), nprocs=world, join=True)
    A, xu, xi, params = _make()
    flat = _flat(params)
    for p in flat:
        p.requires_grad_(True)
    xu.requires_grad_(True)
    xi.requires_grad_(True)
    ident = lambda x: x
    loss = _model(A, xu, xi, params, ident, ident, lambda x: torch.where(x > 0, x, 0.1 * x))
    loss.backward()
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), "r%d.pt" % r))
        assert torch.allclose(got["loss"], loss.detach(), rtol=1e-10, atol=1e-10)
        for g_ref, g in zip([p.grad for p in flat], got["grads"]):
            assert torch.allclose(g, g_ref, rtol=1e-9, atol=1e-10)           # every rank ends with the TOTAL gradient
        assert torch.allclose(got["gxu"], xu.grad[got["lo"]:got["hi"]], rtol=1e-9, atol=1e-10)
        assert torch.allclose(got["gxi"], xi.grad, rtol=1e-9, atol=1e-10)    # replicated table: total, no extra reduce


def test_single_process_helpers_are_identity():
    import star_gcn_amd.dist as SD
    x = torch.randn(3, 2, requires_grad=True)
    assert SD.copy_to_local(x) is x and SD.reduce_from_local(x) is x
    SD.allreduce_grads([x])
    assert SD.world() == 1 and SD.rank() == 0
    y, pend = SD.reduce_start(x)
    assert y is x and pend is None and SD.reduce_wait(y, pend) is x
    xw, pend = SD.grad_wait(x)
    assert xw is x and pend is None and SD.copy_to_local_async(x, None) is x
    assert SD.replicated_dropout(x, 0.5, training=False) is x
    d1 = SD.replicated_dropout(torch.ones(64, 8), 0.5, training=True)
    assert set(d1.unique().tolist()) <= {0.0, 2.0}


def _worker_two_consumers(rank, world, port, out_dir):
    """A replicated tensor that feeds TWO rank-local consumers through the asynchronous crossing: autograd adds the two
    gradient buffers before `_GradWait` runs, so the crossing must hand out completed buffers (blocking fallback)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import star_gcn_amd.dist as SD
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 4, generator=g).double().requires_grad_(True)           # replicated
    a = torch.randn(2, 5, 6, generator=g).double()[rank]                        # rank-local operators
    b = torch.randn(2, 3, 6, generator=g).double()[rank]
    xw, pend = SD.grad_wait(x, exclusive=True)
    y1 = a @ SD.copy_to_local_async(xw, pend, owned=True)
    y2 = b @ SD.copy_to_local_async(xw, pend, owned=True)
    assert pend.users == 2
    (y1.sum() * 2.0 + (y2 ** 2).sum()).backward()
    torch.save({"gx": x.grad}, os.path.join(out_dir, "c%d.pt" % rank))
    #  the public defaults: a gradient tensor object that autograd hands to several edges must not be summed
    # in place, and a replicated tensor with a reader that is NOT a crossing must get completed buffers.  x2 feeds one
    # crossing and, directly, a replicated-region term; the upstream gradient of the crossing is retained and re-read.
    x2 = x.detach().clone().requires_grad_(True)
    xw2, pend2 = SD.grad_wait(x2)                       # not exclusive
    z = SD.copy_to_local_async(xw2, pend2)              # not owned
    z.retain_grad()
    y3 = (a @ z).sum() + 0.5 * (xw2 ** 2).sum() / world        # second term: identical on every rank (replicated region)
    y3.backward()
    assert torch.allclose(z.grad, a.sum(0).unsqueeze(1).expand(6, 4))          # the LOCAL gradient, untouched by the sum
    torch.save({"gx2": x2.grad}, os.path.join(out_dir, "d%d.pt" % rank))
    # the seed of the replicated dropout stream: drawn on rank 0, identical everywhere, re-settable
    torch.manual_seed(100 + rank)
    s = SD.set_replicated_seed()
    m1 = SD.replicated_dropout(torch.ones(32, 8), 0.5, True)
    SD.set_replicated_seed(s)
    m2 = SD.replicated_dropout(torch.ones(32, 8), 0.5, True)
    assert torch.equal(m1, m2)
    torch.save({"seed": s, "mask": m1}, os.path.join(out_dir, "s%d.pt" % rank))
    dist.destroy_process_group()


def test_replicated_tensor_with_two_local_consumers(tmp_path):
    port = _free_port()
    mp.spawn(_worker_two_consumers, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(6, 4, generator=g).double().requires_grad_(True)
    a = torch.randn(2, 5, 6, generator=g).double()
    b = torch.randn(2, 3, 6, generator=g).double()
    loss = sum((a[r] @ x).sum() * 2.0 + ((b[r] @ x) ** 2).sum() for r in range(2))
    loss.backward()
    got = [torch.load(os.path.join(str(tmp_path), "c%d.pt" % r)) for r in range(2)]
    for r in range(2):
        assert torch.allclose(got[r]["gx"], x.grad, rtol=1e-10, atol=1e-12)
    x2 = x.detach().clone().requires_grad_(True)
    (sum((a[r] @ x2).sum() for r in range(2)) + 0.5 * (x2 ** 2).sum()).backward()
    for r in range(2):
        # crossing gradient summed over the ranks + each rank's replicated-region share (x / world), reduced by the caller
        d = torch.load(os.path.join(str(tmp_path), "d%d.pt" % r))["gx2"]
        want = sum(a[q].sum(0) for q in range(2)).unsqueeze(1).expand(6, 4) + x2.detach() / 2
        assert torch.allclose(d, want, rtol=1e-10, atol=1e-12)
    s = [torch.load(os.path.join(str(tmp_path), "s%d.pt" % r)) for r in range(2)]
    assert s[0]["seed"] == s[1]["seed"] and torch.equal(s[0]["mask"], s[1]["mask"])


def test_replicated_dropout_seed_follows_torch_manual_seed():
    import star_gcn_amd.dist as SD
    torch.manual_seed(7)
    s1 = SD.set_replicated_seed()
    a = SD.replicated_dropout(torch.ones(64, 8), 0.5, True)
    torch.manual_seed(7)
    s2 = SD.set_replicated_seed()
    b = SD.replicated_dropout(torch.ones(64, 8), 0.5, True)
    torch.manual_seed(8)
    s3 = SD.set_replicated_seed()
    assert s1 == s2 and torch.equal(a, b) and s3 != s1
    SD.set_replicated_seed(12345)
    assert SD._rep_seed[0] == 12345
