"""The CPU oracle (oracle/seg_oracle.c) vs the golden vectors produced by the REFERENCE's own numpy
models (tests/golden/make_golden.py; reference test_seg_ops.py:11-99).  Tolerances are the reference
tests' own: assert_almost_equal(rtol=1e-4, atol=1e-4) (test_seg_ops.py:125 etc.), tightened to 1e-5
where the summation order is identical."""
import numpy as np
import pytest

from oracle import seg as O

FLAT = [("s0", "dense"), ("s0", "empties"), ("s1", "dense"), ("s1", "empties"), ("s2", "dense"), ("s2", "empties")]
GATHER = [(t, k) for t in ("g0", "g1", "g2", "h50", "h64", "h75", "h250", "h256") for k in ("dense", "empties")
          if not (t == "g2" and k == "empties")]


def close(a, b, rtol=1e-5, atol=1e-5):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("tag,kind", FLAT)
def test_flat_ops(golden, tag, kind):
    p = "flat_%s_%s_" % (tag, kind)
    data, rhs, indptr = golden[p + "data"], golden[p + "rhs"], golden[p + "indptr"]
    close(O.seg_sum(data, indptr), golden[p + "seg_sum"], 1e-4, 1e-4)
    close(O.seg_broadcast_add(data, rhs, indptr), golden[p + "bcast_add"])
    close(O.seg_broadcast_mul(data, rhs, indptr), golden[p + "bcast_mul"])
    close(O.seg_broadcast_to(rhs, indptr, data.shape[1]), golden[p + "bcast_to"])
    if kind == "dense":
        close(O.seg_softmax(data, indptr), golden[p + "softmax"])


@pytest.mark.parametrize("tag,kind", GATHER)
def test_gather_ops(golden, tag, kind):
    p = "gather_%s_%s_" % (tag, kind)
    data, embed1, w = golden[p + "data"], golden[p + "embed1"], golden[p + "weights"]
    idx, indptr = golden[p + "indices"], golden[p + "indptr"]
    close(O.seg_weighted_pool(data, w, idx, indptr), golden[p + "weighted_pool"], 1e-4, 1e-4)
    if p + "take_k_corr" in golden:
        close(O.seg_take_k_corr(embed1, data, idx, indptr), golden[p + "take_k_corr"], 1e-4, 1e-4)
    close(O.seg_pool(data, idx, indptr, "sum"), golden[p + "pool_sum"], 1e-4, 1e-4)
    if kind == "dense":
        close(O.seg_pool(data, idx, indptr, "avg"), golden[p + "pool_avg"], 1e-4, 1e-4)
        val, arg = O.seg_pool(data, idx, indptr, "max")
        close(val, golden[p + "pool_max"], 0, 0)
        if p + "pool_max_grad" in golden:
            g = O.seg_pool_bwd(golden[p + "pool_max_ograd"], arg, idx, indptr, data.shape[1], "max")
            close(g, golden[p + "pool_max_grad"], 1e-5, 1e-5)


def test_req_semantics_and_adjoints():
    """kAddTo accumulates, kNullOp leaves dst alone (seg_op.cc:188-196); bwd_data is the adjoint of fwd."""
    rng = np.random.default_rng(1)
    B, S, T, nnz, C = 2, 7, 9, 40, 6
    data = rng.normal(size=(B, T, C)).astype(np.float32)
    w = rng.normal(size=(B, nnz)).astype(np.float32)
    idx = rng.integers(0, T, nnz).astype(np.int32)
    indptr = np.array([0, 3, 3, 10, 18, 18, 30, 40], np.int32)
    base = O.seg_weighted_pool(data, w, idx, indptr)
    pre = rng.normal(size=base.shape).astype(np.float32)
    acc = pre.copy()
    O.seg_weighted_pool(data, w, idx, indptr, out=acc, req=O.REQ_ADD)
    close(acc, pre + base)
    keep = pre.copy()
    O.seg_weighted_pool(data, w, idx, indptr, out=keep, req=O.REQ_NULL)
    assert np.array_equal(keep, pre)
    og = rng.normal(size=base.shape).astype(np.float32)
    gd = O.seg_weighted_pool_bwd_data(w, og, idx, indptr, T)
    gd_fair = O.seg_weighted_pool_bwd_data(w, og, idx, indptr, T, fair=True)
    assert np.array_equal(gd, gd_fair)  # same per-row summation order
    lhs = float((base.astype(np.float64) * og).sum())
    rhs = float((gd.astype(np.float64) * data).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))
    gw = O.seg_take_k_corr(og, data, idx, indptr)
    rhs_w = float((gw.astype(np.float64) * w).sum())
    assert abs(lhs - rhs_w) < 1e-3 * max(1.0, abs(lhs))


def test_softmax_backward_matches_finite_difference():
    rng = np.random.default_rng(2)
    B, S, nnz = 2, 4, 12
    indptr = np.array([0, 3, 3, 8, 12], np.int32)
    x = rng.normal(size=(B, nnz)).astype(np.float32)
    og = rng.normal(size=(B, nnz)).astype(np.float32)
    val = O.seg_softmax(x, indptr)
    g = O.seg_softmax_bwd(og, val, indptr)
    eps = 1e-2
    fd = np.zeros_like(x)
    for i in range(x.size):
        xp = x.copy(); xp.ravel()[i] += eps
        xm = x.copy(); xm.ravel()[i] -= eps
        fd.ravel()[i] = ((O.seg_softmax(xp, indptr) * og).sum() - (O.seg_softmax(xm, indptr) * og).sum()) / (2 * eps)
    close(g, fd, 1e-2, 1e-3)


def test_graph_helpers_hand_cases():
    # get_support (graph_sampler.cpp:393-420): symm sqrt(1/dr/dc); 0 for zero degree
    ip = np.array([0, 2, 2, 3], np.int32)
    ep = np.array([0, 1, 1], np.int32)
    rd = np.array([2, 0, 1], np.int32)
    cd = np.array([1, 2], np.int32)
    s = O.get_support(rd, cd, ep, ip, symm=True)
    close(s, np.sqrt(1.0 / np.array([2 * 1, 2 * 2, 1 * 2], np.float32)), 1e-7, 0)
    s = O.get_support(rd, cd, ep, ip, symm=False)
    close(s, np.array([0.5, 0.5, 1.0], np.float32), 0, 0)
    # multi_link_split (graph_sampler.cpp:277-376)
    vals = np.array([1, 3, 3, 2, 1, 3], np.float32)
    ip = np.array([0, 2, 2, 6], np.int32)
    pos, ips = O.multi_link_split(vals, ip, np.array([1, 2, 3], np.float32))
    assert [p.tolist() for p in pos] == [[0, 4], [3], [1, 2, 5]]
    assert [p.tolist() for p in ips] == [[0, 1, 1, 2], [0, 0, 0, 1], [0, 1, 1, 3]]


def test_layer_oracles_agree_dense_vs_segment_order():
    """oracle/model.py: the dense whole-network restatement and the per-level FullyConnected -> seg_weighted_pool
    restatement (reference aggregators.py:141-149 order) must agree; so must the C seg_weighted_pool."""
    import torch
    from oracle import model as OM
    from tests.test_abi_and_host import make_multilink
    rng = np.random.default_rng(12)
    n_dst, n_src, nnz, R, D, Uo = 17, 13, 120, 3, 6, 5
    eps, ips, sps = make_multilink(rng, n_dst, n_src, nnz, R)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n_src, D, generator=g, dtype=torch.float64)
    ws = [torch.randn(Uo, D, generator=g, dtype=torch.float64) for _ in range(R)]
    bs = [torch.randn(Uo, generator=g, dtype=torch.float64) for _ in range(R)]
    for accum in ("sum", "stack"):
        a = OM.multilink_aggregator(x, ws, bs, eps, ips, sps, accum=accum, act="leaky")
        outs = []
        for r in range(R):
            A = torch.zeros(n_dst, n_src, dtype=torch.float64)
            for i in range(n_dst):
                for j in range(ips[r][i], ips[r][i + 1]):
                    A[i, eps[r][j]] += float(sps[r][j])
            outs.append(A @ (x @ ws[r].t() + bs[r]))
        b = OM.leaky(torch.cat(outs, 1) if accum == "stack" else sum(outs))
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)
    h = (x @ ws[0].t() + bs[0]).float().numpy()
    c = OM.c_seg_weighted_pool(h, sps[0][:ips[0][-1]] if ips[0][-1] else np.zeros(1, np.float32),
                               eps[0][:max(ips[0][-1], 1)], ips[0])
    t = OM.seg_weighted_pool(torch.from_numpy(h).double(), torch.from_numpy(sps[0]).double(), eps[0], ips[0])
    np.testing.assert_allclose(c, t.numpy(), rtol=1e-5, atol=1e-6)


def _bwd_golden():
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bwd_data_golden.npz"))


def test_bwd_data_oracle_matches_reference_derived_goldens():
    """The data gradient (seg_op.cc:209-240) against vectors derived from the REFERENCE's own forward models by
    linearity (tests/golden/make_bwd_golden.py): P^T . ograd with P = npy_seg_weighted_pool on the identity, and the
    embed2 gradient of seg_take_k_corr from npy_seg_take_k_corr on unit tensors.  Both OpenMP placements of the oracle
    (reference: serial; fair: row-parallel) must agree with them -- at the reference tests' own shapes incl. the
    (4, 1000, 10000, 50000, 4) one -- and with each other bit for bit."""
    g = _bwd_golden()
    tags = sorted({k[:k.rindex("_")] for k in g.files if k.startswith("wp_") and k.endswith("_ddata")})
    assert len(tags) >= 11
    for p in tags:
        w, og, idx, ip = g[p + "_weights"], g[p + "_ograd"], g[p + "_indices"], g[p + "_indptr"]
        T = int(g[p + "_total_ind_num"])
        got = O.seg_weighted_pool_bwd_data(w, og, idx, ip, T)
        fair = O.seg_weighted_pool_bwd_data(w, og, idx, ip, T, fair=True)
        assert np.array_equal(got, fair), p
        want = g[p + "_ddata"]
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), p
        base = np.random.default_rng(0).normal(size=want.shape).astype(np.float32)
        acc = base.copy()
        O.seg_weighted_pool_bwd_data(w, og, idx, ip, T, out=acc, req=O.REQ_ADD)
        assert np.abs(acc - (base + want)).max() <= 2e-5 * max(1.0, np.abs(want).max()), p
    for p in ("tk_g0", "tk_g1"):
        # _backward_seg_take_k_corr_embed2(ograd, embed1, ids, indptr): same kernel, `ograd` in the weights slot
        got = O.seg_weighted_pool_bwd_data(g[p + "_ograd"], g[p + "_embed1"], g[p + "_ids"], g[p + "_indptr"],
                                           int(g[p + "_total_ind_num"]))
        want = g[p + "_dembed2"]
        assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), p
