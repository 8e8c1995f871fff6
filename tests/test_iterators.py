"""CPU tests of the mini-batch samplers (counterpart of reference mxgraph/iterators.py:264-370)."""
import numpy as np

import star_gcn_amd.synthetic as S
from star_gcn_amd.mxgraph.iterators import DataIterator


def make_iter(seed=0, p_zero=0.5):
    graph, eu, ei, vals = S.make_graph("custom", seed=3, n_user=50, n_item=30, n_edges=600, n_levels=5, signal=True)
    rng = np.random.default_rng(seed)
    perm = rng.permutation(eu.size)
    test = np.stack([eu[perm[:100]], ei[perm[:100]]])
    valid = np.stack([eu[perm[100:150]], ei[perm[100:150]]])
    it = DataIterator(graph, "user", "movie", test, valid, embed_P_mask=0.2, embed_p_zero=p_zero, embed_p_self=1 - p_zero,
                      seed=seed)
    return graph, it, test, valid


def test_graph_splits_and_rating_batches():
    graph, it, test, valid = make_iter()
    full = graph["user", "movie"]
    assert it.test_graph["user", "movie"].nnz == full.nnz - 100
    assert it.train_graph["user", "movie"].nnz == full.nnz - 150 == it.train_graph["movie", "user"].nnz
    np.testing.assert_array_equal(it._test_ratings, full.fetch_edges_by_id(test))
    pairs, ratings = next(it.rating_sampler(64, "train"))
    assert pairs.shape == (2, 64) and ratings.shape == (64,)
    np.testing.assert_array_equal(ratings, it.train_graph.fetch_edges_by_id("user", "movie", pairs))
    keys = set(zip(*pairs))
    assert len(keys) == 64                                   # without replacement inside a batch
    assert not (keys & set(zip(*test))) and not (keys & set(zip(*valid)))
    seen = [p.shape[1] for p, _ in it.rating_sampler(30, "valid")]
    assert seen == [30, 20]                                  # sequential sweep, once
    removed = it.train_graph.remove_edges_by_id("user", "movie", pairs)
    assert removed["user", "movie"].nnz == it.train_graph["user", "movie"].nnz - 64


def test_recon_sampler_noise_convention():
    graph, it, _, _ = make_iter(p_zero=0.5)
    noise, batch, allrec = next(it.recon_nodes_sampler(1000))
    for key, n in (("user", 50), ("movie", 30)):
        k = int(np.ceil(0.2 * n))
        assert allrec[key].size == k and np.array_equal(batch[key], allrec[key])
        nz = noise[key]
        assert nz.shape == (n,)
        rest = np.setdiff1d(np.arange(n), allrec[key])
        assert np.array_equal(nz[rest], rest)                # untouched nodes keep their own embedding
        assert np.all((nz[allrec[key]] == -1) | (nz[allrec[key]] == allrec[key]))   # zero-mask or keep-self
    _, it0, _, _ = make_iter(p_zero=0.0)
    noise0, _, rec0 = next(it0.recon_nodes_sampler(1000))
    assert all(np.array_equal(noise0[k][rec0[k]], rec0[k]) for k in rec0)       # shipped transductive setting
    assert all(np.array_equal(v, np.arange(v.size)) for v in it0.evaluate_embed_noise_dict.values())
