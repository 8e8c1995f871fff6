"""The native graph primitives behind plan construction (csrc/graph_host.cpp), pinned two ways that need
nothing of the reference at run time (round 3; since round 4 the primary pin is the reference's own compiled C++:
tests/test_graph_primitives_ref.py / oracle/_ref):

  (1) hand-derived cases: inputs and expected outputs written out by hand from the cited reference lines
      (/root/reference/GraphSampler/graph_sampler.cpp, graph_sampler.h) -- every branch the C++ has;
  (2) independent libraries: scipy.sparse / pandas / numpy evaluate the same mathematical object on random graphs
      (the reference's own Python builds its CSRs with scipy: mxgraph/graph.py:585-593, datasets.py:116-121).

The device twins are held to the same cases in tests/test_gpu_device_twins.py.  Integer outputs bit-exact; support
values to 1e-6 (the reference builds with -ffast-math, GraphSampler/CMakeLists.txt:4) and bit-exact against the
correctly rounded fp32 evaluation of its expression."""
import numpy as np
import pandas as pd
import pytest
import scipy.sparse as sp

import star_gcn_amd.synthetic as S
from star_gcn_amd.mxgraph import graph as G

# ---- the hand-derived cases (shared with the device-twin test) ------------------------------------------------------
# a 4 x 5 rating matrix, rows sorted by column as scipy's tocsr() leaves them:
#   row 0: (c0, 2.0) (c3, 1.0)          row 1: --            row 2: (c1, 3.0) (c3, 3.0) (c4, 1.0)       row 3: (c0, 2.0)
HAND = dict(
    ind_ptr=np.array([0, 2, 2, 5, 6], np.int32),
    end_points=np.array([0, 3, 1, 3, 4, 0], np.int32),
    values=np.array([2.0, 1.0, 3.0, 3.0, 1.0, 2.0], np.float32),
    multi_link=np.array([1.0, 2.0, 3.0], np.float32),
    n_col=5,
)


def hand_csr():
    h = HAND
    return G.CSRMat(h["end_points"], h["ind_ptr"], np.arange(4, dtype=np.int32), np.arange(5, dtype=np.int32), h["values"],
                    h["multi_link"])


def f32(x):
    return np.float32(x)


def test_hand_get_support():
    """graph_sampler.cpp:393-420: symm: sqrt(1.0f / float(r_deg) / float(c_deg)), zero where a degree is zero;
    non-symm: 1.0f / float(r_deg)."""
    m = hand_csr()
    rd, cd = [2, 0, 3, 1], [2, 1, 0, 2, 1]                   # by hand from the matrix above
    assert m.row_degrees.tolist() == rd and m.col_degrees.tolist() == cd
    want = [np.sqrt(f32(1) / f32(2) / f32(2)), np.sqrt(f32(1) / f32(2) / f32(2)),          # row 0: c0 (deg 2), c3 (deg 2)
            np.sqrt(f32(1) / f32(3) / f32(1)), np.sqrt(f32(1) / f32(3) / f32(2)), np.sqrt(f32(1) / f32(3) / f32(1)),
            np.sqrt(f32(1) / f32(1) / f32(2))]
    assert np.array_equal(m.get_support(True), np.asarray(want, np.float32))
    assert np.array_equal(m.get_support(False), np.asarray([0.5, 0.5, f32(1) / f32(3), f32(1) / f32(3), f32(1) / f32(3), 1.0],
                                                           np.float32))
    # override degrees with a zero: the `r_deg != 0 && c_deg != 0` guard leaves 0 (cpp:408-411)
    z = G.CSRMat(HAND["end_points"], HAND["ind_ptr"], np.arange(4, dtype=np.int32), np.arange(5, dtype=np.int32),
                 HAND["values"], HAND["multi_link"], support_col_degrees=np.array([2, 1, 0, 0, 1], np.int32))
    s = z.get_support(True)
    assert s[1] == 0.0 and s[3] == 0.0 and s[0] == want[0]


def test_hand_multi_link_split():
    """graph_sampler.cpp:277-311: per level, edge positions in CSR order and a FULL-length ind_ptr (node_num + 1);
    levels matched by exact float equality against `multi_link`."""
    m = hand_csr()
    split, ptrs = m.multi_link_split(HAND["values"], HAND["ind_ptr"])
    # level 1.0: positions 1 (row 0) and 4 (row 2); level 2.0: 0 (row 0), 5 (row 3); level 3.0: 2, 3 (row 2)
    assert [s.tolist() for s in split] == [[1, 4], [0, 5], [2, 3]]
    assert [p.tolist() for p in ptrs] == [[0, 1, 1, 2, 2], [0, 1, 1, 1, 2], [0, 0, 0, 2, 2]]


def test_hand_remove_edges():
    """graph_sampler.cpp:154-201: listed (row, col) pairs are dropped; pairs that are not edges and repeated pairs change
    nothing; untouched rows are copied."""
    m = hand_csr()
    pairs = np.array([[0, 2, 2, 1, 3, 0], [3, 1, 1, 0, 4, 2]])     # (0,3) edge, (2,1) twice, (1,0) no edge, (3,4) no edge, (0,2) no edge
    r = m.remove_edges_by_id(pairs)
    assert r.ind_ptr.tolist() == [0, 1, 1, 3, 4]
    assert r.end_points.tolist() == [0, 3, 4, 0]
    assert r.values.tolist() == [2.0, 3.0, 1.0, 2.0]
    assert r.row_degrees.tolist() == [1, 0, 2, 1] and r.col_degrees.tolist() == [2, 0, 0, 1, 1]    # fresh degree caches


def test_hand_unique_inverse_and_cnt():
    """graph_sampler.h:441-534: unique values in first-occurrence order (n <= 10000), inverse, counts."""
    data = np.array([7, 3, 7, 9, 3, 1, 9, 9, 0], np.int32)
    u, inv = G.unordered_unique(data, return_inverse=True)
    assert u.tolist() == [7, 3, 9, 1, 0] and inv.tolist() == [0, 1, 0, 2, 1, 3, 2, 2, 4]
    u2, cnt = G.unordered_unique(data, return_counts=True)
    assert u2.tolist() == [7, 3, 9, 1, 0] and cnt.tolist() == [2, 2, 3, 1, 1]


def test_hand_csr_submat_three_branches():
    """graph_sampler.cpp:31-152 (slice_csr_mat): (a) no selection: copy; (b) rows only: the rows in the given order,
    each copied whole; (c) columns (with or without rows): entries keep their order inside a row, the new column index is
    the POSITION of the old one in the selection (cpp:108-126)."""
    m = hand_csr()
    a = m.submat()
    assert a.ind_ptr.tolist() == HAND["ind_ptr"].tolist() and a.end_points.tolist() == HAND["end_points"].tolist()
    b = m.submat(np.array([2, 0, 2], np.int32), None)               # a row may be taken twice
    assert b.ind_ptr.tolist() == [0, 3, 5, 8] and b.end_points.tolist() == [1, 3, 4, 0, 3, 1, 3, 4]
    assert b.values.tolist() == [3.0, 3.0, 1.0, 2.0, 1.0, 3.0, 3.0, 1.0] and b.row_ids.tolist() == [2, 0, 2]
    c = m.submat(np.array([3, 2], np.int32), np.array([4, 0, 3], np.int32))      # columns 4 -> 0, 0 -> 1, 3 -> 2
    assert c.ind_ptr.tolist() == [0, 1, 3]
    assert c.end_points.tolist() == [1, 2, 0]                       # row 3: c0 -> 1; row 2: c3 -> 2 then c4 -> 0 (row order kept)
    assert c.values.tolist() == [2.0, 3.0, 1.0] and c.col_ids.tolist() == [4, 0, 3]
    d = m.submat(None, np.array([3], np.int32))
    assert d.ind_ptr.tolist() == [0, 1, 1, 2, 2] and d.end_points.tolist() == [0, 0]


def test_hand_gen_row_indices_and_fix_neighbor_copy_branch():
    """gen_row_indices_by_indptr (graph_sampler.cpp:378-391) and the RNG-free branches of random_sample_fix_neighbor
    (cpp:742-779): neighbor_num < 0, or a row with <= neighbor_num edges, copies the row's positions in order."""
    m = hand_csr()
    assert m.edge_row_indices.tolist() == [0, 0, 2, 2, 2, 3]
    ep, vals, ptr, sup = m.sample_neighbors(src_ids=np.array([2, 0, 1], np.int32), use_multi_link=False, num_neighbors=-1)
    assert ep.tolist() == [1, 3, 4, 0, 3] and ptr.tolist() == [0, 3, 5, 5] and vals.tolist() == [3.0, 3.0, 1.0, 2.0, 1.0]
    ep3, _v, ptr3, _s = m.sample_neighbors(src_ids=np.array([2, 0, 1], np.int32), use_multi_link=False, num_neighbors=3,
                                           rng=np.random.default_rng(0))
    assert ep3.tolist() == [1, 3, 4, 0, 3] and ptr3.tolist() == [0, 3, 5, 5]        # every row has <= 3 edges
    ep2, _v, ptr2, _s = m.sample_neighbors(src_ids=np.array([2, 0], np.int32), use_multi_link=False, num_neighbors=2,
                                           rng=np.random.default_rng(0))
    assert ptr2.tolist() == [0, 2, 4] and ep2[2:].tolist() == [0, 3]               # row 0 copied, row 2: 2 of its 3
    assert set(ep2[:2].tolist()) < {1, 3, 4} and len(set(ep2[:2].tolist())) == 2


# ---- independent libraries on random graphs -------------------------------------------------------------------------
def _random(seed, nu=60, ni=37, ne=700, R=5):
    graph, eu, ei, vals = S.make_graph("custom", seed=seed, n_user=nu, n_item=ni, n_edges=ne, n_levels=R)
    m = graph["user", "movie"]
    A = sp.csr_matrix((m.values.astype(np.float64), m.end_points, m.ind_ptr), shape=m.shape)
    return m, A


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_scipy_support_split_transpose(seed):
    m, A = _random(seed)
    ones = sp.csr_matrix((np.ones(m.nnz), m.end_points, m.ind_ptr), shape=m.shape)
    dr, dc = np.asarray(ones.sum(1)).ravel(), np.asarray(ones.sum(0)).ravel()
    # D_r^-1/2 A D_c^-1/2 with scipy, float64
    norm = sp.diags(1.0 / np.sqrt(dr)) @ ones @ sp.diags(1.0 / np.sqrt(dc))
    norm = sp.csr_matrix(norm)
    norm.sort_indices()
    assert np.array_equal(norm.indices, m.end_points)
    assert np.abs(norm.data - m.get_support(True)).max() <= 1e-6
    assert np.abs((sp.diags(1.0 / dr) @ ones).tocsr().data - m.get_support(False)).max() <= 1e-6
    # per-level split == the CSR of the entries holding that rating value
    split, ptrs = m.multi_link_split(m.values, m.ind_ptr)
    for lv, pos, ptr in zip(m.multi_link, split, ptrs):
        sel = A.multiply(A == float(lv)).tocsr()
        sel.eliminate_zeros()
        sel.sort_indices()
        assert np.array_equal(sel.indptr, ptr) and np.array_equal(sel.indices, m.end_points[pos])
        assert np.all(m.values[pos] == lv)
    # transpose as the reference builds it (graph.py:585-593: scipy transpose -> CSR, rows sorted by column)
    T = A.T.tocsr()
    T.sort_indices()
    assert np.array_equal(T.indptr, m.T.ind_ptr) and np.array_equal(T.indices, m.T.end_points)
    assert np.array_equal(T.data.astype(np.float32), m.T.values)
    # COO row index
    assert np.array_equal(A.tocoo().row, m.edge_row_indices)


@pytest.mark.parametrize("seed", [4, 5])
def test_scipy_remove_edges_and_submat(seed):
    m, A = _random(seed)
    rng = np.random.default_rng(seed)
    pos = rng.choice(m.nnz, 90, replace=False)
    rows, cols = m.edge_row_indices[pos], m.end_points[pos]
    junk_r, junk_c = rng.integers(0, m.shape[0], 30), rng.integers(0, m.shape[1], 30)      # mostly non-edges
    pairs = np.stack([np.concatenate([rows, junk_r, rows[:10]]), np.concatenate([cols, junk_c, cols[:10]])])
    mask = sp.csr_matrix((np.ones(pairs.shape[1]), (pairs[0], pairs[1])), shape=m.shape)
    keep = A - A.multiply(mask > 0)
    keep = keep.tocsr()
    keep.eliminate_zeros()
    keep.sort_indices()
    r = m.remove_edges_by_id(pairs)
    assert np.array_equal(keep.indptr, r.ind_ptr) and np.array_equal(keep.indices, r.end_points)
    assert np.array_equal(keep.data.astype(np.float32), r.values)
    # sub-matrix: scipy fancy indexing gives the same entries; the order inside a row is pinned by the hand case
    sel_r = rng.permutation(m.shape[0])[:25].astype(np.int32)
    sel_c = rng.permutation(m.shape[1])[:20].astype(np.int32)
    sub = m.submat(sel_r, sel_c)
    ref = A[sel_r][:, sel_c].tocsr()
    got = sp.csr_matrix((sub.values.astype(np.float64), sub.end_points, sub.ind_ptr), shape=(25, 20))
    assert (abs(ref - got)).nnz == 0 and ref.nnz == got.nnz
    only_rows = m.submat(sel_r, None)
    ref_r = A[sel_r].tocsr()
    ref_r.sort_indices()
    assert np.array_equal(ref_r.indptr, only_rows.ind_ptr) and np.array_equal(ref_r.indices, only_rows.end_points)


@pytest.mark.parametrize("n,hi", [(1, 5), (5000, 300), (9999, 20000), (200000, 5000)])
def test_pandas_unique_first_occurrence(n, hi):
    """pandas.unique keeps first-occurrence order (a hash-table unique, like the reference's dense_hash_map one)."""
    data = np.random.default_rng(n).integers(0, hi, n).astype(np.int32)
    want = pd.unique(data)
    u, inv = G.unordered_unique(data, return_inverse=True)
    assert np.array_equal(u, want) and np.array_equal(u[inv], data)
    u2, cnt = G.unordered_unique(data, return_counts=True)
    vc = pd.Series(data).value_counts(sort=False)
    assert np.array_equal(u2, want) and np.array_equal(cnt, vc.loc[want].to_numpy())


def test_fix_neighbor_sampler_is_uniform_without_replacement():
    """random branch of random_sample_fix_neighbor (cpp:742-779 -> uniform_choice_range, replace = false, :703-731):
    k distinct positions of the row, every k-subset equally likely.  Chi-square over the 10 two-subsets of a 5-edge row."""
    ind_ptr = np.array([0, 5], np.int32)
    m = G.CSRMat(np.arange(5, dtype=np.int32), ind_ptr, np.zeros(1, np.int32), np.arange(5, dtype=np.int32))
    rng = np.random.default_rng(0)
    counts = dict()
    n_draw = 20000
    for _ in range(n_draw):
        ep, _v, ptr, _s = m.sample_neighbors(use_multi_link=False, num_neighbors=2, rng=rng)
        assert ptr.tolist() == [0, 2] and ep[0] < ep[1]
        counts[tuple(ep.tolist())] = counts.get(tuple(ep.tolist()), 0) + 1
    assert len(counts) == 10
    exp = n_draw / 10.0
    chi2 = sum((c - exp) ** 2 / exp for c in counts.values())
    assert chi2 < 27.9                                     # 99.9 % quantile of chi-square with 9 degrees of freedom
