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


def test_movielens_file_loader(tmp_path):
    """star_gcn_amd.datasets.LoadData on files written in the two MovieLens layouts (tab-separated u1.base/u1.test,
    '::'-separated ratings.dat): id maps, CSR, rating levels, splits feed the DataIterator."""
    from star_gcn_amd.datasets import LoadData
    from star_gcn_amd.mxgraph.iterators import DataIterator
    rng = np.random.default_rng(0)
    users = rng.choice(np.arange(1, 400), 40, replace=False)           # raw ids with gaps
    movies = rng.choice(np.arange(1, 900), 30, replace=False)
    cells = rng.choice(users.size * movies.size, 500, replace=False)
    u, m = users[cells // movies.size], movies[cells % movies.size]
    r = rng.choice([1, 2, 3, 4, 5], 500)
    d = tmp_path / "ml-100k"
    d.mkdir()
    with open(d / "u1.base", "w") as f:
        f.writelines("%d\t%d\t%d\t88%d\n" % (a, b, c, k) for k, (a, b, c) in enumerate(zip(u[:400], m[:400], r[:400])))
    with open(d / "u1.test", "w") as f:
        f.writelines("%d\t%d\t%d\t99\n" % (a, b, c) for a, b, c in zip(u[400:], m[400:], r[400:]))
    data = LoadData("ml-100k", str(tmp_path), val_ratio=0.1, seed=1)
    g = data.graph
    csr = g["user", "movie"]
    assert csr.nnz == 500 and data.num_user == np.unique(u).size and data.num_item == np.unique(m).size
    assert np.array_equal(data.num_links, np.arange(1, 6, dtype=np.float32))
    tp, tv = data.test_data
    assert tp.shape == (2, 100) and np.array_equal(tv, r[400:].astype(np.float32))
    assert np.array_equal(data.raw_user_ids[tp[0]], u[400:]) and np.array_equal(data.raw_movie_ids[tp[1]], m[400:])
    assert np.array_equal(g.fetch_edges_by_id("user", "movie", tp), tv)
    vp, vv = data.valid_data
    assert vp.shape == (2, 40) and np.array_equal(g.fetch_edges_by_id("user", "movie", vp), vv)
    it = DataIterator(g, "user", "movie", tp, vp, seed=0)
    assert it.train_graph["user", "movie"].nnz == 500 - 100 - 40
    # ml-1m layout: one '::' file, random split
    d2 = tmp_path / "ml-1m"
    d2.mkdir()
    with open(d2 / "ratings.dat", "w") as f:
        f.writelines("%d::%d::%.1f::7\n" % (a, b, c / 2.0) for a, b, c in zip(u, m, r))
    data2 = LoadData("ml-1m", str(tmp_path), test_ratio=0.2, val_ratio=0.1, seed=3)
    assert data2.test_data[0].shape == (2, 100) and data2.valid_data[0].shape == (2, 40)
    assert np.array_equal(data2.num_links, np.array([0.5, 1, 1.5, 2, 2.5], np.float32))
    both = np.concatenate([data2.test_data[0], data2.valid_data[0]], axis=1)
    assert np.unique(both[0].astype(np.int64) * 10 ** 6 + both[1]).size == 140     # test and validation are disjoint
