"""Device twins of the plan primitives against the SAME pins as the host builders
(tests/test_graph_primitives_pinned.py): the hand-derived cases of the cited reference lines, independent libraries
(pandas first-occurrence unique), and bit equality with the host primitive on random inputs."""
import ctypes

import numpy as np
import pandas as pd
import pytest
import torch

from tests.test_graph_primitives_pinned import HAND, f32, hand_csr

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_device_unique_inverse_hand_case_and_pandas():
    from star_gcn_amd.device_graph import unique_inverse_device
    data = np.array([7, 3, 7, 9, 3, 1, 9, 9, 0], np.int32)              # graph_sampler.h:441-534, first-occurrence order
    u, inv, cnt = unique_inverse_device(_dev(data), 9, return_counts=True)
    assert u.tolist() == [7, 3, 9, 1, 0] and inv.tolist() == [0, 1, 0, 2, 1, 3, 2, 2, 4] and cnt.tolist() == [2, 2, 3, 1, 1]
    from star_gcn_amd.mxgraph import graph as G
    for n, hi in [(1, 5), (5000, 300), (9999, 20000), (3000000, 69878), (2000000, 5)]:
        d = np.random.default_rng(n).integers(0, hi, n).astype(np.int32)
        u, inv, cnt = unique_inverse_device(_dev(d), hi - 1, return_counts=True)
        want = pd.unique(d)
        assert np.array_equal(u.cpu().numpy(), want)
        hu, hinv = G.unordered_unique(d, return_inverse=True)             # host primitive: bit equality
        assert np.array_equal(hu, u.cpu().numpy()) and np.array_equal(hinv, inv.cpu().numpy())
        assert np.array_equal(cnt.cpu().numpy(), np.bincount(d, minlength=hi)[want])
    u, inv = unique_inverse_device(torch.zeros(0, dtype=torch.int32, device="cuda"), 3)
    assert u.numel() == 0 and inv.numel() == 0
    from star_gcn_amd._lib import StarGCNError
    with pytest.raises(StarGCNError):
        unique_inverse_device(_dev(np.array([1, 12, 2], np.int32)), 9)


def test_device_fix_neighbor_sampler_equals_host_bit_for_bit():
    from star_gcn_amd import _lib as L
    from star_gcn_amd.device_graph import sample_fix_neighbor_device
    import star_gcn_amd.synthetic as S
    graph, eu, ei, vals = S.make_graph("custom", seed=3, n_user=4000, n_item=900, n_edges=200000, n_levels=5)
    m = graph["user", "movie"]
    rng = np.random.default_rng(0)
    sel = rng.permutation(m.shape[0])[:2500].astype(np.int32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for k, seed in [(-1, 1), (0, 2), (1, 3), (10, 4), (64, 5), (100000, 6)]:
        dptr = np.empty(sel.size + 1, np.int32)
        L.check(L.lib().sg_sample_fix_neighbor_cpu(None, vp(dptr), vp(m.ind_ptr), vp(sel), sel.size, k, seed), "cpu")
        host = np.empty(max(int(dptr[-1]), 1), np.int32)
        L.check(L.lib().sg_sample_fix_neighbor_cpu(vp(host), vp(dptr), vp(m.ind_ptr), vp(sel), sel.size, k, seed), "cpu")
        pos, dp = sample_fix_neighbor_device(_dev(m.ind_ptr), _dev(sel), k, seed)
        assert np.array_equal(dp.cpu().numpy(), dptr)
        assert np.array_equal(pos.cpu().numpy(), host[:int(dptr[-1])])
    # hand case: the RNG-free branches (graph_sampler.cpp:742-779) copy the rows' positions in order
    h = HAND
    pos, dp = sample_fix_neighbor_device(_dev(h["ind_ptr"]), _dev(np.array([2, 0, 1], np.int32)), -1, 0)
    assert pos.tolist() == [2, 3, 4, 0, 1] and dp.tolist() == [0, 3, 5, 5]
    pos, dp = sample_fix_neighbor_device(_dev(h["ind_ptr"]), _dev(np.array([2, 0, 1], np.int32)), 3, 0)
    assert pos.tolist() == [2, 3, 4, 0, 1]
    pos, dp = sample_fix_neighbor_device(_dev(h["ind_ptr"]), _dev(np.array([2, 0], np.int32)), 2, 7)
    assert dp.tolist() == [0, 2, 4] and pos[2:].tolist() == [0, 1] and set(pos[:2].tolist()) < {2, 3, 4}


def test_device_support_and_level_split_on_the_hand_case():
    """get_support (graph_sampler.cpp:393-420) and multi_link_split (:277-311) through their device twins
    (sg_get_support_hip, sg_level_index_hip + sg_multilink_fuse_csr_hip) on the hand-derived matrix; the device edge
    removal is held bit for bit to the (pinned) host removal in tests/test_gpu_resident.py."""
    from star_gcn_amd import _lib as L
    from star_gcn_amd.device_graph import DeviceBipartite
    from star_gcn_amd.mxgraph.graph import HeterGraph
    m = hand_csr()
    g = HeterGraph({"user": np.arange(4, dtype=np.int32), "movie": np.arange(5, dtype=np.int32)}, {("user", "movie"): m})
    dg = DeviceBipartite.from_host(g, "user", "movie", "cuda")
    want = [np.sqrt(f32(1) / f32(2) / f32(2)), np.sqrt(f32(1) / f32(2) / f32(2)), np.sqrt(f32(1) / f32(3) / f32(1)),
            np.sqrt(f32(1) / f32(3) / f32(2)), np.sqrt(f32(1) / f32(3) / f32(1)), np.sqrt(f32(1) / f32(1) / f32(2))]
    assert np.array_equal(dg.support(False, True)[:6].cpu().numpy(), np.asarray(want, np.float32))
    assert dg.level.tolist() == [1, 0, 2, 2, 0, 1]                      # exact float match against multi_link [1, 2, 3]
    p = dg.plan("user")                                                 # fused per-level CSR over (row, level) segments
    R = 3
    ip = p.c_indptr.cpu().numpy()
    per_level_ptr = [[0] + [int(ip[i * R + r + 1] - ip[i * R + r]) for i in range(4)] for r in range(R)]
    assert [np.cumsum(x).tolist() for x in per_level_ptr] == [[0, 1, 1, 2, 2], [0, 1, 1, 1, 2], [0, 0, 0, 2, 2]]
    cols = p.c_idx.cpu().numpy()
    seg_cols = lambda i, r: cols[ip[i * R + r]:ip[i * R + r + 1]].tolist()
    assert seg_cols(0, 0) == [3] and seg_cols(0, 1) == [0] and seg_cols(2, 2) == [1, 3] and seg_cols(2, 0) == [4]
    assert seg_cols(3, 1) == [0] and seg_cols(1, 0) == [] and seg_cols(1, 2) == []
