"""CPU tests of the host side: graph containers (counterparts of reference mxgraph/graph.py), plan construction
(reference layers.py:260-337 gen_plan) and the synthetic generator.  Integer outputs are checked exactly against
brute-force numpy restatements."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

import star_gcn_amd.synthetic as S
from star_gcn_amd.mxgraph import graph as G


def small_graph(seed=0, nu=30, ni=20, ne=200, R=4):
    return S.make_graph("custom", seed=seed, n_user=nu, n_item=ni, n_edges=ne, n_levels=R)


def test_unordered_unique_first_occurrence_order():
    data = np.array([7, 3, 7, 9, 3, 1, 9, 9], np.int32)
    u, inv = G.unordered_unique(data, return_inverse=True)
    assert u.tolist() == [7, 3, 9, 1]
    assert np.array_equal(u[inv], data)
    u2, cnt = G.unordered_unique(data, return_counts=True)
    assert u2.tolist() == [7, 3, 9, 1] and cnt.tolist() == [2, 2, 3, 1]


def test_merge_nodes_and_dicts():
    a, b = np.array([5, 2, 5], np.int32), np.array([2, 8], np.int32)
    u, inds = G.merge_nodes([a, b])
    assert u.tolist() == [5, 2, 8]
    assert np.array_equal(u[inds[0]], a) and np.array_equal(u[inds[1]], b)
    uniq, idx_l = G.merge_node_ids_dict([{"user": a, "movie": b}, {"user": b}, {}])
    assert uniq["user"].tolist() == [5, 2, 8] and uniq["movie"].tolist() == [2, 8]
    assert np.array_equal(uniq["user"][idx_l[1]["user"]], b) and idx_l[2] == {}


def test_csr_transpose_support_and_split():
    graph, eu, ei, vals = small_graph()
    m = graph["user", "movie"]
    m.check_consistency()
    assert np.all(np.diff(m.ind_ptr) >= 1) and np.all(m.col_degrees >= 1)          # degree >= 1 everywhere
    for i in range(m.shape[0]):                                                     # rows sorted by column
        assert np.all(np.diff(m.end_points[m.ind_ptr[i]:m.ind_ptr[i + 1]]) > 0)
    t = graph["movie", "user"]
    dense = np.zeros(m.shape)
    dense[eu, ei] = vals
    dt = np.zeros(t.shape)
    dt[t.edge_row_indices, t.end_points] = t.values
    assert np.array_equal(dense.T, dt)
    for j in range(t.shape[0]):
        assert np.all(np.diff(t.end_points[t.ind_ptr[j]:t.ind_ptr[j + 1]]) > 0)
    # symmetric support: sqrt(1/dr/dc) in float32 exactly as the reference computes it
    dr, dc = m.row_degrees, m.col_degrees
    exp = np.sqrt(np.float32(1.0) / dr[eu].astype(np.float32) / dc[ei].astype(np.float32)).astype(np.float32)
    assert np.array_equal(m.get_support(True), exp)
    assert np.array_equal(m.get_support(False), (np.float32(1.0) / dr[eu].astype(np.float32)).astype(np.float32))
    # under symm the reverse direction carries the same value edge for edge
    st = np.zeros(t.shape)
    st[t.edge_row_indices, t.end_points] = t.get_support(True)
    sm = np.zeros(m.shape)
    sm[eu, ei] = m.get_support(True)
    np.testing.assert_allclose(sm.T, st, rtol=2e-7, atol=0)   # (1/dr)/dc vs (1/dc)/dr: last-ulp float32 difference


def test_sample_neighbors_full_and_subset():
    graph, eu, ei, vals = small_graph(seed=3)
    m = graph["user", "movie"]
    src = np.array([4, 0, 17, 4], np.int32)
    eps, vs, ips, sps = m.sample_neighbors(src_ids=src, symm=True, use_multi_link=True, num_neighbors=-1)
    sup = m.get_support(True)
    for l, lvl in enumerate(m.multi_link):
        assert ips[l].shape[0] == src.size + 1 and ips[l][0] == 0
        for k, s in enumerate(src):
            row = slice(m.ind_ptr[s], m.ind_ptr[s + 1])
            sel = m.values[row] == lvl
            seg = slice(ips[l][k], ips[l][k + 1])
            assert np.array_equal(eps[l][seg], m.col_ids[m.end_points[row][sel]])   # CSR order kept inside a level
            assert np.array_equal(sps[l][seg], sup[row][sel])
            assert np.all(vs[l][seg] == lvl)
    ep, v, ip, sp = m.sample_neighbors(src_ids=src, use_multi_link=False)
    assert ip[-1] == sum(m.row_degrees[s] for s in src)
    # fixed-size sampling: at most k per row, without replacement, a subset of the row
    rng = np.random.default_rng(0)
    ep, v, ip, sp = m.sample_neighbors(src_ids=src, use_multi_link=False, num_neighbors=3, rng=rng)
    for k, s in enumerate(src):
        got = ep[ip[k]:ip[k + 1]]
        assert got.size == min(3, m.row_degrees[s]) and np.unique(got).size == got.size
        assert np.all(np.isin(got, m.end_points[m.ind_ptr[s]:m.ind_ptr[s + 1]]))


def test_native_fix_neighbor_sampler_properties():
    """sg_sample_fix_neighbor_cpu (reference random_sample_fix_neighbor, graph_sampler.cpp:742-779): sizes
    min(k, degree), no repeats, increasing positions inside the row, deterministic in (seed, row) and independent of
    the other rows, uniform over the row's edges."""
    import ctypes
    import star_gcn_amd._lib as L
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(1)
    lens = rng.integers(0, 40, 300)
    lens[7] = 1000
    ind_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    sel = rng.permutation(300).astype(np.int32)[:200]
    sel[3] = 7

    def draw(sel, k, seed):
        ptr = np.empty(sel.size + 1, np.int32)
        L.check(L.lib().sg_sample_fix_neighbor_cpu(None, vp(ptr), vp(ind_ptr), vp(sel), sel.size, k, seed))
        out = np.empty(max(int(ptr[-1]), 1), np.int32)
        L.check(L.lib().sg_sample_fix_neighbor_cpu(vp(out), vp(ptr), vp(ind_ptr), vp(sel), sel.size, k, seed))
        return out[:ptr[-1]], ptr

    out, ptr = draw(sel, 10, 42)
    for i, r in enumerate(sel):
        got = out[ptr[i]:ptr[i + 1]]
        assert got.size == min(10, lens[r])
        assert np.all(np.diff(got) > 0)
        assert got.size == 0 or (got[0] >= ind_ptr[r] and got[-1] < ind_ptr[r + 1])
    out2, _ = draw(sel, 10, 42)
    assert np.array_equal(out, out2)
    out3, _ = draw(sel, 10, 43)
    assert not np.array_equal(out, out3)
    full, fptr = draw(sel, -1, 0)
    assert fptr[-1] == lens[sel].sum()
    assert np.array_equal(full[fptr[3]:fptr[4]], np.arange(ind_ptr[7], ind_ptr[8]))
    # uniformity on the 1000-edge row: every edge is kept with probability k/len (chi-square-ish bound)
    one = np.array([7], np.int32)
    hits = np.zeros(1000)
    for seed in range(2000):
        o, _ = draw(one, 100, seed)
        hits[o - ind_ptr[7]] += 1
    assert abs(hits.mean() - 200) < 1e-9 and hits.std() < 3 * np.sqrt(200 * 0.9) and hits.min() > 130 and hits.max() < 270
    # COO row indices
    rows = np.empty(int(ind_ptr[-1]), np.int32)
    L.check(L.lib().sg_gen_row_indices_cpu(vp(rows), vp(ind_ptr), 300, int(ind_ptr[-1])))
    assert np.array_equal(rows, np.repeat(np.arange(300), lens))
    assert L.lib().sg_gen_row_indices_cpu(vp(rows), vp(ind_ptr), 300, 5) == -4


def test_remove_edges_both_directions():
    graph, eu, ei, vals = small_graph(seed=5)
    drop = np.stack([eu[::7], ei[::7]])
    g2 = graph.remove_edges_by_id("user", "movie", drop)
    m, t = g2["user", "movie"], g2["movie", "user"]
    assert m.nnz == graph["user", "movie"].nnz - drop.shape[1] == t.nnz
    keys = set(zip(m.edge_row_indices.tolist(), m.end_points.tolist()))
    assert not any((u, i) in keys for u, i in zip(*drop))
    assert keys == set(zip(t.end_points.tolist(), t.edge_row_indices.tolist()))
    # degrees (hence support) follow the CURRENT matrix
    assert np.array_equal(m.row_degrees, np.bincount(m.edge_row_indices, minlength=m.shape[0]))


def test_save_load_roundtrip(tmp_path):
    graph, *_ = small_graph(seed=6)
    graph.save(str(tmp_path / "g"))
    g2 = G.HeterGraph.load(str(tmp_path / "g"))
    for key, m in graph.csr_mat_dict.items():
        n = g2[key]
        assert np.array_equal(m.end_points, n.end_points) and np.array_equal(m.ind_ptr, n.ind_ptr)
        assert np.array_equal(m.values, n.values) and np.array_equal(m.multi_link, n.multi_link)


def test_gen_plan_full_graph_structure():
    """gen_plan on the whole graph: every level CSR of the plan reproduces A_r exactly (dense check) and the
    unique-node maps are consistent."""
    import torch
    from star_gcn_amd.mxgraph.layers import HeterGCNLayer, StackedHeterGCNLayers
    graph, eu, ei, vals = small_graph(seed=7, nu=25, ni=18, ne=160, R=3)
    enc = StackedHeterGCNLayers()
    for _ in range(2):
        enc.add(HeterGCNLayer(graph.meta_graph, graph.get_multi_link_structure(), 8, 8, agg_accum="sum",
                              agg_act="leaky", out_act="leaky"))
    sel = {"user": np.array([3, 3, 9, 0], np.int32), "movie": np.array([5, 1], np.int32)}
    req, plan = enc.gen_plan(graph, sel, symm=True, device="cpu")
    assert len(plan) == 2
    for depth in (1, 0):
        prev_ids, agg = plan[depth]
        for src_key, (base_take, sel_take, plans) in agg.items():
            for dst_key, mp in plans.items():
                m = graph[src_key, dst_key]
                rows = prev_ids_for(plan, depth, src_key, sel)          # ids whose outputs this depth produces
                assert mp.n_dst == rows.size and mp.n_src == prev_ids[dst_key].size
                R = mp.R
                dense = np.zeros((rows.size, R, m.shape[1]))
                sup = m.get_support(True)
                for k, rid in enumerate(rows):
                    r = m.row_id_to_ind(np.array([rid]))[0]
                    for j in range(m.ind_ptr[r], m.ind_ptr[r + 1]):
                        lvl = int(np.nonzero(m.multi_link == m.values[j])[0][0])
                        dense[k, lvl, m.end_points[j]] += sup[j]
                got = np.zeros_like(dense)
                ci, cx, cw = mp.c_indptr.numpy(), mp.c_idx.numpy(), mp.c_w.numpy()
                for s in range(rows.size * R):
                    for j in range(ci[s], ci[s + 1]):
                        got[s // R, s % R, prev_ids[dst_key][cx[j]]] += cw[j]
                np.testing.assert_allclose(got, dense, rtol=0, atol=1e-7)
    assert set(req) == {"user", "movie"}


def prev_ids_for(plan, depth, key, sel):
    if depth == len(plan) - 1:
        return G.unordered_unique(sel[key], return_inverse=True)[0]
    return plan[depth + 1][0][key]


def test_balanced_blocks_and_user_block_support():
    from star_gcn_amd.dist import balanced_row_blocks
    graph, eu, ei, vals = small_graph(seed=8, nu=40, ni=15, ne=300)
    m = graph["user", "movie"]
    blocks = balanced_row_blocks(m.ind_ptr, 4)
    assert blocks[0][0] == 0 and blocks[-1][1] == m.shape[0]
    assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
    sup = m.get_support(True)
    tot = 0
    for lo, hi in blocks:
        sub = S.user_block(graph, "user", "movie", lo, hi)
        a, b = m.ind_ptr[lo], m.ind_ptr[hi]
        assert np.array_equal(sub.get_support(True), sup[a:b])             # GLOBAL item degrees in the block
        st, gt = sub.T, graph["movie", "user"]
        # transposed block: same support values as the global reverse CSR restricted to these users
        mask = (gt.end_points >= lo) & (gt.end_points < hi)
        assert np.array_equal(st.get_support(True), gt.get_support(True)[mask])
        tot += sub.nnz
    assert tot == m.nnz


def test_plan_order_is_independent_of_hash_randomisation():
    """Two ranks of a node-partitioned run are separate processes with different string-hash seeds; the order in
    which gen_plan visits node types (hence the order of collectives and of lazy parameter creation) must not
    depend on it."""
    import subprocess
    import sys
    code = (
        "import numpy as np, star_gcn_amd.synthetic as S\n"
        "from star_gcn_amd.mxgraph.layers import HeterGCNLayer, StackedHeterGCNLayers\n"
        "g, eu, ei, v = S.make_graph('custom', seed=7, n_user=25, n_item=18, n_edges=160, n_levels=3)\n"
        "enc = StackedHeterGCNLayers()\n"
        "[enc.add(HeterGCNLayer(g.meta_graph, g.get_multi_link_structure(), 9, 8)) for _ in range(2)]\n"
        "req, plan = enc.gen_plan(g, {'user': eu, 'movie': ei}, device='cpu', full_node_ids={'movie': g.node_ids_dict['movie']})\n"
        "print([(list(p[0].keys()), list(p[1].keys())) for p in plan], list(req.keys()))\n")
    outs = set()
    for seed in ("1", "2", "3", "77"):
        env = dict(os.environ, PYTHONHASHSEED=seed, PYTHONPATH=ROOT)
        outs.add(subprocess.check_output([sys.executable, "-c", code], env=env, cwd=ROOT).decode())
    assert len(outs) == 1, outs


def test_native_batch_plan_builders_match_numpy():
    """sg_edge_positions_cpu / sg_pair_plan_cpu / sg_take_plan_cpu (host side of the per-batch plans) vs their numpy
    definitions: stable grouping, stable transpose, inverse index, flags."""
    import torch
    from star_gcn_amd.model import PairPlan
    from star_gcn_amd.plan import TakePlan
    graph, eu, ei, vals = small_graph(seed=9)
    m = graph["user", "movie"]
    rng = np.random.default_rng(3)
    sel = rng.choice(eu.size, 60, replace=False)
    pairs = np.stack([np.concatenate([eu[sel], [0, 3]]), np.concatenate([ei[sel], [10 ** 3 % m.shape[1], 1]])])
    pos = m.edge_positions(pairs)
    key = m.edge_row_indices.astype(np.int64) * m.shape[1] + m.end_points
    for k in range(pairs.shape[1]):
        hit = np.nonzero(key == pairs[0, k].astype(np.int64) * m.shape[1] + pairs[1, k])[0]
        assert pos[k] == (hit[0] if hit.size else -1)
    # pair plan
    u, i = rng.integers(0, 17, 200), rng.integers(0, 11, 200)
    pp = PairPlan(u, i, 17, 11, "cpu")
    order = np.argsort(u, kind="stable")
    assert np.array_equal(pp.order.numpy(), order) and np.array_equal(pp.items.numpy(), i[order])
    assert np.array_equal(pp.inv_order.numpy()[order], np.arange(200))
    assert np.array_equal(np.diff(pp.indptr.numpy()), np.bincount(u, minlength=17))
    t_order = np.argsort(i[order], kind="stable")
    assert np.array_equal(pp.tplan.t_pos.numpy(), t_order)
    assert np.array_equal(pp.tplan.t_seg.numpy(), u[order][t_order])
    assert np.array_equal(np.diff(pp.tplan.t_indptr.numpy()), np.bincount(i, minlength=11))
    srt = PairPlan(np.sort(u), i, 17, 11, "cpu")
    assert srt.identity and srt.order is None and not pp.identity
    empty = PairPlan(np.zeros(0, np.int32), np.zeros(0, np.int32), 4, 3, "cpu")
    assert empty.n_pairs == 0 and empty.indptr.numpy().tolist() == [0] * 5
    # take plan
    ids = np.array([4, -1, 2, 4, 0, 7, 2, 4], np.int32)
    tp = TakePlan(ids, 9, "cpu")
    assert not tp.identity and tp.inv_ids is None and tp.covered == 7
    assert np.array_equal(np.diff(tp.t_indptr.numpy()), np.bincount(ids[ids >= 0], minlength=9))
    valid = np.nonzero(ids >= 0)[0]
    assert np.array_equal(tp.t_pos.numpy()[:7], valid[np.argsort(ids[valid], kind="stable")])
    perm = TakePlan(np.array([3, 0, -1, 2], np.int32), 5, "cpu")
    assert perm.inv_ids.numpy().tolist() == [1, 3, 3 - 0, 0, -1][:0] + [1, -1, 3, 0, -1] and perm.t_pos is None
    assert TakePlan(np.arange(6, dtype=np.int32), 6, "cpu").identity
    assert not TakePlan(np.arange(6, dtype=np.int32), 7, "cpu").identity
    with pytest.raises(Exception):
        TakePlan(np.array([9], np.int32), 9, "cpu")


def test_override_degree_blocks_removal_and_slicing():
    """Rank-local user blocks carry GLOBAL item degrees for the support (CSRMat(support_col_degrees=...)).  They must
    survive edge removal (without a reducer in a single-process run, with one when several ranks remove edges) and
    slicing -- never a silent fall-back to the block's own degrees."""
    graph, eu, ei, vals = small_graph(seed=11, nu=40, ni=15, ne=300)
    m = graph["user", "movie"]
    lo, hi = 7, 29
    sub = S.user_block(graph, "user", "movie", lo, hi)
    a, b = int(m.ind_ptr[lo]), int(m.ind_ptr[hi])
    # remove every third edge of the block: ids of a block are LOCAL user indices / global item ids
    pos = np.arange(0, sub.nnz, 3)
    pairs = np.stack([sub.row_ids[sub.edge_row_indices[pos]], sub.col_ids[sub.end_points[pos]]])
    # reference result: the same removal on the whole graph (global user ids), restricted to the block
    gpairs = np.stack([pairs[0] + lo, pairs[1]])
    whole = m.remove_edges_by_id(gpairs)
    wa, wb = int(whole.ind_ptr[lo]), int(whole.ind_ptr[hi])
    # (1) single process, no reducer: the override degrees drop by the removed edges
    cut = sub.remove_edges_by_id(pairs)
    assert np.array_equal(cut.end_points, whole.end_points[wa:wb])
    assert np.array_equal(cut.get_support(True), whole.get_support(True)[wa:wb])
    # (2) with a reducer (here: another "rank" removed nothing, the sum is the identity) -- same result
    calls = []
    cut2 = sub.remove_edges_by_id(pairs, degree_reducer=lambda d: (calls.append(d.sum()), d)[1])
    assert calls and np.array_equal(cut2.get_support(True), cut.get_support(True))
    # (3) the transposed block (item -> local users) carries the override as ROW degrees
    tcut = sub.T.remove_edges_by_id(pairs[::-1])
    gt = whole.T
    mask = (gt.end_points >= lo) & (gt.end_points < hi)
    assert np.array_equal(tcut.get_support(True), gt.get_support(True)[mask])
    # (4) slicing: a column selection keeps the override of the kept columns; a row selection would need the other
    # ranks' counts and is refused (for the transposed block the roles swap)
    cols = np.array([14, 0, 5, 9, 2], np.int32)
    sl = sub.submat(None, cols)
    gsl = m.submat(None, cols)                      # same slice of the global graph: its own degrees ARE the global ones
    ga, gb = int(gsl.ind_ptr[lo]), int(gsl.ind_ptr[hi])
    assert np.array_equal(sl.end_points, gsl.end_points[ga:gb])
    # the override is the column degree of the UNSLICED global graph for the kept columns
    want = np.sqrt(np.float32(1.0) / sl.row_degrees[sl.edge_row_indices].astype(np.float32)
                   / m.col_degrees[cols][sl.end_points].astype(np.float32))
    assert np.array_equal(sl.get_support(True), want.astype(np.float32))
    with pytest.raises(ValueError):
        sub.submat(np.arange(3, 10, dtype=np.int32), None)
    with pytest.raises(ValueError):
        sub.T.submat(None, np.arange(3, 10, dtype=np.int32))
    assert np.array_equal(sub.submat().get_support(True), sub.get_support(True))


def test_override_degree_block_needs_reducer_in_multi_rank_run(tmp_path):
    """remove_edges_by_id on a rank-local block without a degree_reducer is an ERROR when more than one rank runs."""
    import subprocess
    import sys
    code = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, %r)
import star_gcn_amd.synthetic as S
rank = int(sys.argv[1])
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", sys.argv[2]
dist.init_process_group("gloo", rank=rank, world_size=2)
graph, eu, ei, vals = S.make_graph("custom", seed=3, n_user=20, n_item=9, n_edges=80, n_levels=3)
sub = S.user_block(graph, "user", "movie", rank * 10, rank * 10 + 10)
pairs = np.stack([sub.row_ids[sub.edge_row_indices[:2]], sub.col_ids[sub.end_points[:2]]])
try:
    sub.remove_edges_by_id(pairs)
    print("NOERROR")
except ValueError as e:
    print("RAISED", "degree_reducer" in str(e))
def reducer(d):
    import torch
    t = torch.from_numpy(d.copy())
    dist.all_reduce(t)
    return t.numpy()
cut = sub.remove_edges_by_id(pairs, degree_reducer=reducer)
print("NNZ", cut.nnz, sub.nnz)
dist.destroy_process_group()
''' % ROOT
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(port)], stdout=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "RAISED True" in o and "NOERROR" not in o, o
        nnz = [int(x) for x in o.split("NNZ")[1].split()]
        assert nnz[0] == nnz[1] - 2


def test_balanced_blocks_never_leave_a_rank_without_rows():
    """A hub row carrying most of the edges used to produce EMPTY neighbouring blocks (searchsorted returns the same cut for
    several parts): every block now has at least one row whenever there are at least as many rows as parts."""
    from star_gcn_amd.dist import balanced_row_blocks
    deg = np.array([1, 1, 5000, 1, 1, 1, 1, 1, 1, 1], np.int64)            # row 2 holds 99.8 % of the edges
    ind_ptr = np.concatenate([[0], np.cumsum(deg)])
    for n in (2, 3, 4, 8, 10):
        blocks = balanced_row_blocks(ind_ptr, n)
        assert blocks[0][0] == 0 and blocks[-1][1] == 10 and all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        assert all(hi > lo for lo, hi in blocks), (n, blocks)
    blocks = balanced_row_blocks(ind_ptr, 12)                                # more parts than rows: the tail is empty
    assert sum(hi - lo for lo, hi in blocks) == 10 and all(hi >= lo for lo, hi in blocks)
    even = balanced_row_blocks(np.arange(0, 1001, 10), 4)                    # uniform degrees: untouched by the fix
    assert even == [(0, 25), (25, 50), (50, 75), (75, 100)]
    hub_first = balanced_row_blocks(np.concatenate([[0], np.cumsum([9000, 1, 1, 1])]), 4)
    assert hub_first == [(0, 1), (1, 2), (2, 3), (3, 4)]
