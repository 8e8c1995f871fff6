"""Device-side plan builders (csrc/plan_build.hip) against the host builders (csrc/graph_host.cpp) and the oracle:
integer outputs bit-exact, fp32 weights bit-exact (same expression, same rounding)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import seg as O

pytestmark = pytest.mark.gpu

U, I = "user", "movie"


def _csr(rng, S, T, nnz, pad=0, empty_every=0):
    lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 0.5))
    if empty_every:
        moved = lens[::empty_every].sum()
        lens[::empty_every] = 0
        lens[1] += moved
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, T, nnz + pad).astype(np.int32)
    return idx, indptr


@pytest.mark.parametrize("S,T,nnz,pad", [(1, 1, 0, 1), (5, 10, 30, 0), (50, 20, 500, 7), (1000, 10000, 50000, 0),
                                        (3000, 70000, 400000, 0), (17, 3, 100000, 0)])
def test_build_transpose_hip_equals_cpu(S, T, nnz, pad):
    from star_gcn_amd import _lib as L
    from star_gcn_amd.plan import TransposePlan
    rng = np.random.default_rng(S * 7 + T)
    idx, indptr = _csr(rng, S, T, nnz, pad, empty_every=5 if S > 10 else 0)
    host = TransposePlan(idx, indptr, T, "cpu")
    dev = TransposePlan(torch.from_numpy(idx).cuda(), torch.from_numpy(indptr).cuda(), T, "cuda")
    E = int(indptr[-1])
    np.testing.assert_array_equal(dev.t_indptr.cpu().numpy(), host.t_indptr.numpy())
    np.testing.assert_array_equal(dev.t_pos.cpu().numpy()[:E], host.t_pos.numpy()[:E])
    np.testing.assert_array_equal(dev.t_seg.cpu().numpy()[:E], host.t_seg.numpy()[:E])
    assert L.lib().sg_build_transpose_workspace_bytes(S, T, nnz + pad) > 0


def test_radix_sort_is_stable_on_many_digits():
    """keys up to 2^26 (4 radix passes), heavy duplicates: the device order must equal numpy's stable argsort."""
    from star_gcn_amd.plan import TransposePlan
    rng = np.random.default_rng(3)
    T, nnz = (1 << 26) - 5, 1_500_000
    idx = rng.choice(rng.integers(0, T, 5000), nnz).astype(np.int32)      # 5 000 distinct keys spread over 26 bits
    indptr = np.array([0, nnz], np.int32)
    dev = TransposePlan(torch.from_numpy(idx).cuda(), torch.from_numpy(indptr).cuda(), T, "cuda")
    order = np.argsort(idx, kind="stable").astype(np.int32)
    np.testing.assert_array_equal(dev.t_pos.cpu().numpy(), order)
    ti = dev.t_indptr.cpu().numpy()
    assert ti[0] == 0 and ti[-1] == nnz
    np.testing.assert_array_equal(np.diff(ti.astype(np.int64))[idx[order[::1000]]] > 0, True)


def test_bwd_data_dev_operator_matches_oracle():
    """the reference-shaped backward entry (device index tensors only, plan built in the caller's workspace) through
    ctypes, against the C oracle of seg_op.cc:209-240; write and add requests."""
    from star_gcn_amd import _lib as L
    rng = np.random.default_rng(5)
    for (B, S, T, nnz, C) in [(1, 5, 10, 30, 128), (10, 50, 20, 500, 4), (2, 300, 700, 20000, 75), (1, 40, 30, 3000, 256)]:
        idx, indptr = _csr(rng, S, T, nnz, pad=0, empty_every=7 if S > 10 else 0)
        g = rng.normal(size=(B, nnz)).astype(np.float32)
        og = rng.normal(size=(B, S, C)).astype(np.float32)
        ref = O.seg_weighted_pool_bwd_data(g, og, idx, indptr, T)
        lib = L.lib()
        d_g, d_og = torch.from_numpy(g).cuda(), torch.from_numpy(og).cuda()
        d_idx, d_ip = torch.from_numpy(idx).cuda(), torch.from_numpy(indptr).cuda()
        wsb = lib.sg_seg_weighted_pool_bwd_data_dev_workspace_bytes(B, S, T, nnz, C)
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        base = torch.from_numpy(rng.normal(size=(B, T, C)).astype(np.float32)).cuda()
        for req in (L.REQ_WRITE, L.REQ_ADD):
            out = base.clone()
            L.check(lib.sg_seg_weighted_pool_bwd_data_dev_hip(L.ptr(out), L.ptr(d_g), L.ptr(d_og), L.ptr(d_idx),
                                                              L.ptr(d_ip), B, S, T, nnz, C, req, L.ptr(ws), wsb, None),
                    "sg_seg_weighted_pool_bwd_data_dev_hip")
            torch.cuda.synchronize()
            want = ref + (base.cpu().numpy() if req == L.REQ_ADD else 0)
            err = np.abs(out.cpu().numpy() - want).max() / max(1.0, np.abs(want).max())
            assert err < 1e-5, (B, S, T, nnz, C, req, err)
        with pytest.raises(L.StarGCNError, match="workspace"):
            L.check(lib.sg_seg_weighted_pool_bwd_data_dev_hip(L.ptr(out), L.ptr(d_g), L.ptr(d_og), L.ptr(d_idx),
                                                              L.ptr(d_ip), B, S, T, nnz, C, 1, L.ptr(ws), 16, None), "x")


def _host_plans(graph, dev):
    from star_gcn_amd.plan import MultiLinkPlan
    out = dict()
    for dst, src in ((U, I), (I, U)):
        eps, _v, ips, sps = graph[dst, src].sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
        out[dst] = (MultiLinkPlan(eps, ips, sps, graph[dst, src].shape[1], dev), eps, ips, sps)
    return out


_FIELDS = ("c_indptr", "c_idx", "c_q", "c_w", "t_indptr", "t_idx", "t_q", "t_w", "d_indptr", "s_indptr")


def _assert_plans_equal(a, b):
    assert (a.R, a.n_dst, a.n_src, a.nnz) == (b.R, b.n_dst, b.n_src, b.nnz)
    for f in _FIELDS:
        x, y = getattr(a, f).cpu().numpy(), getattr(b, f).cpu().numpy()
        n = a.nnz if f[2:] in ("idx", "q", "w") else x.size
        np.testing.assert_array_equal(x[:n].view(np.int32), y[:n].view(np.int32), err_msg=f)


@pytest.mark.parametrize("shape", ["tiny", "ml-100k", "ml-1m"])
def test_device_fuse_builders_equal_host_fuse(shape):
    """sg_multilink_fuse_hip (per-level device lists) and sg_multilink_fuse_csr_hip (device CSR + levels, incl. the
    device-side transpose / degrees / support) reproduce the host plan bit for bit, both directions."""
    import star_gcn_amd.synthetic as S
    from star_gcn_amd.device_graph import DeviceBipartite
    from star_gcn_amd.plan import MultiLinkPlan
    dev = torch.device("cuda")
    graph, eu, ei, vals = S.make_graph(shape)
    host = _host_plans(graph, dev)
    dg = DeviceBipartite.from_host(graph, U, I, dev)
    np.testing.assert_array_equal(dg.item_degrees.cpu().numpy(), graph[U, I].col_degrees)
    np.testing.assert_array_equal(dg.edge_row.cpu().numpy()[:dg.nnz], graph[U, I].edge_row_indices)
    for dst in (U, I):
        hp, eps, ips, sps = host[dst]
        lists = MultiLinkPlan([torch.from_numpy(np.ascontiguousarray(e, np.int32)).to(dev) for e in eps],
                              [torch.from_numpy(np.ascontiguousarray(p, np.int32)).to(dev) for p in ips],
                              [torch.from_numpy(np.ascontiguousarray(s, np.float32)).to(dev) for s in sps], hp.n_src, dev)
        _assert_plans_equal(lists, hp)
        csr = dg.plan(dst, symm=True, with_from=True)
        _assert_plans_equal(csr, hp)
        # slot -> edge id maps: the slot holds the (user, item) pair of that CSR edge
        m = graph[U, I]
        eu_all, ei_all = m.edge_row_indices, m.end_points
        c_from = csr.c_from.cpu().numpy()[:csr.nnz]
        t_from = csr.t_from.cpu().numpy()[:csr.nnz]
        src_of = ei_all if dst == U else eu_all
        dst_of = eu_all if dst == U else ei_all
        np.testing.assert_array_equal(src_of[c_from], csr.c_idx.cpu().numpy()[:csr.nnz])
        np.testing.assert_array_equal(dst_of[t_from], csr.t_idx.cpu().numpy()[:csr.nnz])


def test_network_on_device_plan_equals_network_on_host_plan():
    """make_plan_device (everything built on the device) vs make_plan (host planning): same predictions, same loss and
    the same gradient for every parameter, bit for bit (identical plans -> identical kernels -> identical sums)."""
    import star_gcn_amd.model as M
    import star_gcn_amd.synthetic as S
    from star_gcn_amd.device_graph import DeviceBipartite
    dev = torch.device("cuda")
    graph, eu, ei, vals = S.make_graph("ml-100k")
    D = 64
    torch.manual_seed(0)
    net = M.Net(graph, U, I, embed_units=D, agg_units=(D, D), out_units=(D, D), nblocks=1, use_dae=False,
                agg_accum="sum").to(dev)
    y = torch.from_numpy(((vals - vals.mean()) / vals.std()).astype(np.float32)).to(dev)
    hplan = net.make_plan(graph, rating_node_pairs=np.stack([eu, ei]), device=dev,
                          full_node_ids={k: graph.node_ids_dict[k] for k in (U, I)})
    dplan = net.make_plan_device(DeviceBipartite.from_host(graph, U, I, dev))
    res = []
    for plan in (hplan, dplan):
        net.zero_grad(set_to_none=True)
        preds, _, _ = net.run(plan)
        loss = (0.5 * (preds[0].view(-1) - y) ** 2).mean()
        loss.backward()
        res.append((preds[0].detach().clone(), loss.detach().clone(),
                    {k: p.grad.detach().clone() for k, p in net.named_parameters()}))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for k in res[0][2]:
        assert torch.equal(res[0][2][k], res[1][2][k]), k


def test_synthetic_device_graph_is_a_valid_rating_graph():
    from star_gcn_amd.device_graph import synthetic_device_graph
    dg = synthetic_device_graph(3000, 2000, 200000, 16, "cuda", seed=1)
    ip, ep, lv = dg.ind_ptr.cpu().numpy().astype(np.int64), dg.end_points.cpu().numpy(), dg.level.cpu().numpy()
    assert ip[0] == 0 and ip[-1] == ep.size == dg.nnz and abs(dg.nnz - 200000) < 5000
    assert np.all(np.diff(ip) >= 1) and np.bincount(ep, minlength=2000).min() >= 1          # degree >= 1 everywhere
    rows = np.repeat(np.arange(3000), np.diff(ip))
    key = rows * 2000 + ep
    assert np.all(np.diff(key) > 0)                                                        # sorted rows, no duplicates
    assert lv.min() >= 0 and lv.max() == 15
    p = dg.plan(U)
    assert int(p.c_indptr[-1]) == dg.nnz and int(p.t_indptr[-1]) == dg.nnz


@pytest.mark.parametrize("parts", [8, 3, 1])
@pytest.mark.parametrize("C", [64, 50])
def test_source_partitioned_gather_equals_the_plain_data_gradient(parts, C):
    """plan.SourcePartition + sg_seg_gather_sum_parts_hip against sg_seg_weighted_pool_bwd_data_hip and the oracle: the
    sub-segments partition every segment's edges by source range (bit-exact bookkeeping), the sums agree to fp32
    reordering; padding edges, empty segments and sources nobody references included."""
    from star_gcn_amd import ops
    from star_gcn_amd.plan import SourcePartition, TransposePlan
    rng = np.random.default_rng(parts * 100 + C)
    S, T, nnz, pad = 700, 90, 9000, 37
    lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 0.3))          # many empty user rows
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    items = rng.integers(0, T - 5, nnz + pad).astype(np.int32)            # items T-5.. never referenced
    d_ip, d_items = torch.from_numpy(indptr).cuda(), torch.from_numpy(items).cuda()
    tp = TransposePlan(d_items, d_ip, T, d_items.device)
    sp = SourcePartition(tp.t_indptr, tp.t_seg, S, pos=tp.t_pos, parts=parts)
    # bookkeeping: same multiset of (segment, source, slot) triples, sub-segment p only holds sources of range p
    ip = sp.indptr.cpu().numpy()
    src, pos, b = sp.src.cpu().numpy(), sp.pos.cpu().numpy(), sp.bounds.cpu().numpy()
    assert ip[0] == 0 and ip[-1] == nnz and np.all(np.diff(ip) >= 0) and b[0] == 0
    t_ip, t_seg, t_pos = tp.t_indptr.cpu().numpy(), tp.t_seg.cpu().numpy(), tp.t_pos.cpu().numpy()
    for p in range(parts):
        for s in rng.choice(T, 12, replace=False):
            lo, hi = ip[p * T + s], ip[p * T + s + 1]
            whole = slice(t_ip[s], t_ip[s + 1])
            hi_b = b[p + 1] if p + 1 < parts else S
            sel = (t_seg[whole] >= b[p]) & (t_seg[whole] < hi_b) if p else (t_seg[whole] < (b[1] if parts > 1 else S))
            assert np.array_equal(src[lo:hi], t_seg[whole][sel]) and np.array_equal(pos[lo:hi], t_pos[whole][sel])
    g = rng.normal(size=(1, nnz + pad)).astype(np.float32)
    g[:, nnz:] = 0                       # padding carries weight 0 wherever the model pads (reference graph.py:221-222)
    pu = rng.normal(size=(S, C)).astype(np.float32)
    d_g, d_pu = torch.from_numpy(g).cuda(), torch.from_numpy(pu).cuda()
    ref = ops.seg_weighted_pool_bwd_data(d_g, d_pu.unsqueeze(0), tp, T)[0]
    got = torch.full((T, C), 7.0, device="cuda")
    ops.gather_sum_parts(got, d_pu, sp, d_g, C)
    want = O.seg_weighted_pool_bwd_data(g, pu[None], items, indptr, T)[0]
    scale = float(np.abs(want).max())
    assert float((got.cpu() - torch.from_numpy(want)).abs().max()) <= 1e-5 * scale
    assert float((got - ref).abs().max()) <= 1e-5 * scale
    acc = torch.from_numpy(want).cuda().clone()
    ops.gather_sum_parts(acc, d_pu, sp, d_g, C, req=ops.REQ_ADD)
    assert float((acc.cpu() - 2 * torch.from_numpy(want)).abs().max()) <= 2e-5 * scale


def test_pair_plan_uses_the_partitioned_gather_when_it_pays(monkeypatch):
    """_PairDot.backward: with the eligibility forced, the item-side gradient through the source partition equals the
    plain transposed gather."""
    import star_gcn_amd.model as M
    rng = np.random.default_rng(4)
    nu, ni, n = 500, 80, 6000
    cells = rng.choice(nu * ni, n, replace=False)
    cells.sort()
    u, i = (cells // ni).astype(np.int32), (cells % ni).astype(np.int32)
    pu = torch.randn(nu, 64, device="cuda")
    pi = torch.randn(ni, 64, device="cuda")
    gy = torch.randn(n, device="cuda")
    grads = []
    for forced in (False, True):
        pp = M.PairPlan.from_sorted_device_pairs(torch.from_numpy(u).cuda(), torch.from_numpy(i).cuda(), nu, ni)
        if forced:
            from star_gcn_amd.plan import SourcePartition
            pp._tparts = SourcePartition(pp.tplan.t_indptr, pp.tplan.t_seg, nu, pos=pp.tplan.t_pos, parts=8)
        else:
            assert pp.item_side_partition(64) is None                  # far too small to pay
        a, b = pu.clone().requires_grad_(True), pi.clone().requires_grad_(True)
        M.pair_inner_product(a, b, pp).backward(gy)
        grads.append((a.grad, b.grad))
    assert torch.equal(grads[0][0], grads[1][0])
    assert float((grads[0][1] - grads[1][1]).abs().max()) <= 1e-5 * float(grads[0][1].abs().max())


@pytest.mark.parametrize("kind", ["sorted_device", "host_unsorted", "partitioned"])
@pytest.mark.parametrize("C", [64, 20, 128])
def test_fused_rating_loss_equals_scores_plus_l2_loss(kind, C):
    """model.pair_l2_loss (two DOT-mode gather passes, sg_pair_l2_hip) against pair_inner_product + l2_loss: value and the
    gradients w.r.t. both projections, with an upstream gradient != 1; pairs with empty users / items, padding-free and
    permuted plans, and the source-partitioned item pass."""
    import star_gcn_amd.functional as SF
    import star_gcn_amd.model as M
    from star_gcn_amd.plan import SourcePartition
    rng = np.random.default_rng(C + len(kind))
    nu, ni, n = 400, 90, 7000
    cells = rng.choice(nu * ni, n, replace=False)
    if kind != "host_unsorted":
        cells.sort()
    u, i = (cells // ni).astype(np.int32), (cells % ni).astype(np.int32)
    if kind == "host_unsorted":
        pp = M.PairPlan(u, i, nu, ni, torch.device("cuda", 0))
        assert not pp.identity
    else:
        pp = M.PairPlan.from_sorted_device_pairs(torch.from_numpy(u).cuda(), torch.from_numpy(i).cuda(), nu, ni)
        if kind == "partitioned":
            pp._tparts = SourcePartition(pp.tplan.t_indptr, pp.tplan.t_seg, nu, pos=pp.tplan.t_pos, parts=8)
    pu0 = torch.randn(nu, C, device="cuda")
    pi0 = torch.randn(ni, C, device="cuda")
    y = torch.randn(n, device="cuda")
    scale = 1.0 / n
    res = []
    for fused in (False, True):
        a, b = pu0.clone().requires_grad_(True), pi0.clone().requires_grad_(True)
        if fused:
            loss = M.pair_l2_loss(a, b, pp, y, scale)
        else:
            loss = SF.l2_loss(M.pair_inner_product(a, b, pp).view(-1), y, scale)
        (loss * 3.0).backward()
        res.append((loss.detach(), a.grad, b.grad))
    (l0, ga0, gb0), (l1, ga1, gb1) = res
    assert abs(float(l0) - float(l1)) <= 2e-6 * abs(float(l0))
    assert float((ga0 - ga1).abs().max()) <= 1e-5 * float(ga0.abs().max())
    assert float((gb0 - gb1).abs().max()) <= 1e-5 * float(gb0.abs().max())


@pytest.mark.parametrize("C", [4, 36, 64])
def test_pair_l2_raw_entry_against_numpy(C):
    """sg_pair_l2_hip through ops.pair_l2 against a float64 numpy evaluation of its definition: rows[s] = sum_j g_j
    src[idx_j], g_j = scale * dev_scale * (<src[idx_j], other[s]> - y_j), loss = loss_scale * sum r_j^2; ragged segments
    with empty ones, chunk-spanning segments, AddTo, and an edge-free plan."""
    from star_gcn_amd import ops
    rng = np.random.default_rng(C)
    S, T, nnz = 300, 70, 5000
    lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 0.2))
    lens[5] += 900                                           # one segment over several 256-edge chunks
    nnz = int(lens.sum())
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    idx = rng.integers(0, T, nnz).astype(np.int32)
    src = rng.normal(size=(T, C)).astype(np.float32)
    other = rng.normal(size=(S, C)).astype(np.float32)
    y = rng.normal(size=nnz).astype(np.float32)
    seg = np.repeat(np.arange(S), lens)
    r = (src[idx].astype(np.float64) * other[seg].astype(np.float64)).sum(1) - y
    scale, dev_scale = 0.37, 1.7
    want_rows = np.zeros((S, C))
    np.add.at(want_rows, seg, (scale * dev_scale * r)[:, None] * src[idx].astype(np.float64))
    want_loss = 0.25 * (r ** 2).sum()
    ds = torch.tensor([dev_scale], device="cuda")
    rows, loss = ops.pair_l2(dev(src), dev(other), dev(y), dev(idx), dev(indptr), S, scale, scale_dev=ds, loss_scale=0.25)
    sc = float(np.abs(want_rows).max())
    assert float((rows.cpu().double() - torch.from_numpy(want_rows)).abs().max()) <= 1e-5 * sc
    assert abs(float(loss) - want_loss) <= 2e-6 * want_loss
    acc = rows.clone()
    ops.pair_l2(dev(src), dev(other), dev(y), dev(idx), dev(indptr), S, scale, scale_dev=ds, out=acc, req=ops.REQ_ADD)
    assert float((acc.cpu().double() - 2 * torch.from_numpy(want_rows)).abs().max()) <= 2e-5 * sc
    # no edges at all (one padding slot, as the plans carry): zero rows, zero loss
    rows0, loss0 = ops.pair_l2(dev(src), dev(other), dev(np.zeros(1, np.float32)), dev(np.zeros(1, np.int32)),
                               dev(np.zeros(S + 1, np.int32)), S, scale, loss_scale=1.0)
    assert float(rows0.abs().max()) == 0.0 and float(loss0) == 0.0


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()
