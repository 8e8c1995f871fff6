"""MovieLens ETL + on-disk graph layout (reference mxgraph/datasets.py:39-171, :404-574 and
graph.py:465-491, :898-915, :1066-1100).  No network here, so the files are written by the test IN the GroupLens layouts
(separators, column order, latin-1 titles, `Children's`, a title without a year, info rows nobody rated)."""
import json
import os

import numpy as np
import pytest

from star_gcn_amd.datasets import GENRES, LoadData, read_ratings
from star_gcn_amd.mxgraph import graph as G
from star_gcn_amd.mxgraph.iterators import DataIterator

OCC = ["artist", "doctor", "engineer", "student", "writer"]


def _ratings(rng, n_user_ids=60, n_movie_ids=45, n=900, half_steps=False):
    users = np.sort(rng.choice(np.arange(1, 400), n_user_ids, replace=False))          # raw ids with gaps
    movies = np.sort(rng.choice(np.arange(1, 900), n_movie_ids, replace=False))
    cells = rng.choice(users.size * movies.size, n, replace=False)
    u, m = users[cells // movies.size], movies[cells % movies.size]
    r = rng.choice([1, 2, 3, 4, 5], n).astype(np.float64)
    if half_steps:
        r = r - 0.5 * rng.integers(0, 2, n)
    return users, movies, u, m, r


def _write_ml100k(root, rng):
    users, movies, u, m, r = _ratings(rng)
    d = os.path.join(root, "ml-100k")
    os.makedirs(d)
    with open(os.path.join(d, "u1.base"), "w") as f:
        f.writelines("%d\t%d\t%d\t88%d\n" % (a, b, c, k) for k, (a, b, c) in enumerate(zip(u[:700], m[:700], r[:700])))
    with open(os.path.join(d, "u1.test"), "w") as f:
        f.writelines("%d\t%d\t%d\t99\n" % (a, b, c) for a, b, c in zip(u[700:], m[700:], r[700:]))
    all_users = np.union1d(users, [401, 402])                     # two users / one movie nobody rated
    ages = rng.integers(10, 70, all_users.size)
    gender = rng.choice(["M", "F"], all_users.size)
    occ = rng.choice(OCC, all_users.size)
    with open(os.path.join(d, "u.user"), "w") as f:
        f.writelines("%d|%d|%s|%s|%05d\n" % (i, a, g, o, 90000 + i) for i, a, g, o in zip(all_users, ages, gender, occ))
    all_movies = np.union1d(movies, [901])
    flags = rng.integers(0, 2, (all_movies.size, 19))
    titles = ["Movie %d (19%02d)" % (i, 50 + k % 50) for k, i in enumerate(all_movies)]
    titles[0] = "Am\xe9lie (2001)"                                # latin-1 byte, as in the real u.item
    titles[1] = "unknown"                                         # no year -> 1950 (reference :543-545)
    with open(os.path.join(d, "u.item"), "w", encoding="latin-1") as f:
        for i, t, fl in zip(all_movies, titles, flags):
            f.write("%d|%s|01-Jan-1995||http://x/%d|%s\n" % (i, t, i, "|".join(str(x) for x in fl)))
    seen_u = np.isin(all_users, u)
    seen_m = np.isin(all_movies, m)
    return dict(u=u, m=m, r=r, users=all_users[seen_u], ages=ages[seen_u], gender=gender[seen_u], occ=occ[seen_u],
                movies=all_movies[seen_m], flags=flags[seen_m], titles=[t for t, k in zip(titles, seen_m) if k])


def test_ml100k_layout_features_and_fixed_split(tmp_path):
    rng = np.random.default_rng(0)
    w = _write_ml100k(str(tmp_path), rng)
    calls = []

    def embedder(texts):
        calls.append(list(texts))
        return np.arange(len(texts) * 300, dtype=np.float32).reshape(len(texts), 300) / 1e4

    data = LoadData("ml-100k", str(tmp_path), val_ratio=0.1, seed=1, title_embedder=embedder)
    g, csr = data.graph, data.graph["user", "movie"]
    assert csr.nnz == 900 and data.num_user == w["users"].size and data.num_item == w["movies"].size
    assert np.array_equal(data.raw_user_ids, w["users"]) and np.array_equal(data.raw_movie_ids, w["movies"])
    assert np.array_equal(data.num_links, np.arange(1, 6, dtype=np.float32))
    # fixed split: u1.test is the test set, in file order
    tp, tv = data.test_data
    assert tp.shape == (2, 200) and np.array_equal(tv, w["r"][700:].astype(np.float32))
    assert np.array_equal(data.raw_user_ids[tp[0]], w["u"][700:]) and np.array_equal(data.raw_movie_ids[tp[1]], w["m"][700:])
    assert np.array_equal(g.fetch_edges_by_id("user", "movie", tp), tv)
    vp, vv = data.valid_data
    assert vp.shape == (2, 70) and np.array_equal(g.fetch_edges_by_id("user", "movie", vp), vv)
    it = DataIterator(g, "user", "movie", tp, vp, seed=0)
    assert it.train_graph["user", "movie"].nnz == 900 - 200 - 70
    # user features: age / 50, gender == F, one-hot occupation (sorted columns)
    uf = data.user_features
    assert uf.shape == (w["users"].size, 2 + len(OCC)) and uf.dtype == np.float32
    np.testing.assert_allclose(uf[:, 0], w["ages"] / 50.0, rtol=1e-6)
    assert np.array_equal(uf[:, 1], (w["gender"] == "F").astype(np.float32))
    assert list(data.occupations) == OCC
    assert np.array_equal(uf[:, 2:].argmax(1), np.searchsorted(OCC, w["occ"])) and np.all(uf[:, 2:].sum(1) == 1)
    # movie features: title embedding | (year - 1950) / 100 | 19 genre flags
    mf = data.item_features
    assert mf.shape == (w["movies"].size, 300 + 1 + 19)
    assert np.array_equal(mf[:, 301:], w["flags"].astype(np.float32))
    assert calls[0][0] == "Am\xe9lie " and calls[0][1] == "unknown"       # the text handed to the embedder (:541-551)
    np.testing.assert_allclose(mf[0, 300], (2001 - 1950) / 100.0, rtol=1e-6)
    assert mf[1, 300] == 0.0
    np.testing.assert_allclose(mf[:, :300], np.arange(mf.shape[0] * 300).reshape(-1, 300) / 1e4, rtol=1e-6)
    assert g.features["user"] is uf and g.features["movie"] is mf
    assert "#Val/Test edges: 70/200" in repr(data)


@pytest.mark.parametrize("name", ["ml-1m", "ml-10m"])
def test_double_colon_layouts_random_split(tmp_path, name):
    rng = np.random.default_rng(1)
    users, movies, u, m, r = _ratings(rng, half_steps=(name == "ml-10m"))
    d = tmp_path / ("ml-1m" if name == "ml-1m" else "ml-10M100K")
    d.mkdir()
    with open(d / "ratings.dat", "w") as f:
        f.writelines(("%d::%d::%g::97830%d\n" % (a, b, c, k)) for k, (a, b, c) in enumerate(zip(u, m, r)))
    genres = ["Children's|Comedy", "Action|Sci-Fi", "Drama", "Film-Noir|Mystery"]
    if name == "ml-10m":
        genres += ["IMAX|Action", "(no genres listed)"]
    with open(d / "movies.dat", "w", encoding="latin-1") as f:
        for k, i in enumerate(np.union1d(movies, [950])):
            f.write("%d::Title %d: Part::II (19%02d)::%s\n" % (i, i, 60 + k % 40, genres[k % len(genres)]))
    if name == "ml-1m":
        with open(d / "users.dat", "w") as f:
            f.writelines("%d::%s::%d::%d::%05d\n" % (i, "FM"[k % 2], [1, 18, 25, 35][k % 4], k % 7, i)
                         for k, i in enumerate(users))
    data = LoadData(name, str(tmp_path), test_ratio=0.2, val_ratio=0.1, seed=3)
    n = u.size
    assert data.graph["user", "movie"].nnz == n
    assert data.test_data[0].shape == (2, int(np.ceil(0.2 * n)))
    n_train_all = n - data.test_data[1].size
    assert data.valid_data[0].shape == (2, int(np.ceil(0.1 * n_train_all)))
    assert np.array_equal(data.num_links, np.unique(r).astype(np.float32))
    both = np.concatenate([data.test_data[0], data.valid_data[0]], axis=1)
    assert np.unique(both[0].astype(np.int64) * 10 ** 6 + both[1]).size == both.shape[1]   # test / validation disjoint
    for pairs, vals in (data.test_data, data.valid_data):
        assert np.array_equal(data.graph.fetch_edges_by_id("user", "movie", pairs), vals)
    G_ = GENRES[name]
    mf = data.item_features
    assert mf.shape == (data.num_item, 300 + 1 + len(G_)) and not mf[:, :300].any()
    k0 = int(np.flatnonzero(np.union1d(movies, [950]) == data.raw_movie_ids[0])[0])
    expect = np.zeros(len(G_), np.float32)
    for gname in genres[k0 % len(genres)].split("|"):
        gname = "Children" if gname.startswith("Children") else gname
        expect[G_.index(gname) if gname in G_ else G_.index("unknown")] = 1
    assert np.array_equal(mf[0, 301:], expect)
    np.testing.assert_allclose(mf[0, 300], (1960 + k0 % 40 - 1950) / 100.0, rtol=1e-6)
    if name == "ml-1m":
        uf = data.user_features
        assert uf.shape == (data.num_user, 2 + 7)
        np.testing.assert_allclose(uf[:, 0], np.array([1, 18, 25, 35])[np.arange(users.size) % 4] / 50.0, rtol=1e-6)
        assert np.array_equal(uf[:, 1], (np.arange(users.size) % 2 == 0).astype(np.float32))
    else:
        assert data.user_features.shape == (data.num_user, 1) and not data.user_features.any()


def test_unknown_genre_without_unknown_column_and_missing_info_are_errors(tmp_path):
    d = tmp_path / "ml-1m"
    d.mkdir()
    (d / "ratings.dat").write_text("1::10::5::1\n2::11::3::2\n")
    (d / "users.dat").write_text("1::F::25::3::12345\n2::M::35::4::12345\n")
    (d / "movies.dat").write_text("10::A (1990)::Drama\n11::B (1991)::Telenovela\n")
    with pytest.raises(ValueError, match="Telenovela"):
        LoadData("ml-1m", str(tmp_path), seed=0)
    (d / "movies.dat").write_text("10::A (1990)::Drama\n")
    with pytest.raises(ValueError, match="rated movie ids have no row"):
        LoadData("ml-1m", str(tmp_path), seed=0)


@pytest.mark.parametrize("key", ["item", "user"])
def test_inductive_split(tmp_path, key):
    rng = np.random.default_rng(5)
    _write_ml100k(str(tmp_path), rng)
    data = LoadData("ml-100k", str(tmp_path), use_inductive=True, inductive_key=key, inductive_node_frac=10,
                    inductive_edge_frac=80, seed=7)
    name = "movie" if key == "item" else "user"
    n = data.graph.node_ids_dict[name].size
    tr, va, te = data.inductive_train_ids, data.inductive_valid_ids, data.inductive_test_ids
    assert te.size == int(np.ceil(n / 10.0)) and va.size == int(np.ceil((n - te.size) / 10.0))
    assert np.array_equal(np.sort(np.concatenate([tr, va, te])), np.arange(n))              # a partition of the nodes
    csr = data.graph["user", "movie"]
    deg = csr.row_degrees if key == "user" else csr.col_degrees
    for ids, (pairs, vals) in ((te, data.test_data), (va, data.valid_data)):
        own = pairs[0] if key == "user" else pairs[1]
        assert set(np.unique(own)) == set(ids)                                              # only held-out nodes' ratings
        assert np.all(deg[ids] > 10)
        cnt = np.bincount(own, minlength=n)[ids]
        assert np.array_equal(cnt, np.floor(deg[ids] / 100.0 * 80).astype(np.int64))
        assert np.unique(pairs[0].astype(np.int64) * 10 ** 6 + pairs[1]).size == pairs.shape[1]
        assert np.array_equal(data.graph.fetch_edges_by_id("user", "movie", pairs), vals)


def test_read_ratings_empty_and_malformed(tmp_path):
    p = tmp_path / "r.dat"
    p.write_text("")
    assert all(a.size == 0 for a in read_ratings(str(p), "::"))
    p.write_text("1::2::3::4\n5::6\n")
    with pytest.raises(ValueError):
        read_ratings(str(p), "::")


def test_graph_directory_layout_matches_the_reference(tmp_path):
    """Files written here are what the reference's HeterGraph.load expects, and a directory written the way the
    reference's HeterGraph.save writes it loads here."""
    rng = np.random.default_rng(2)
    w = _write_ml100k(str(tmp_path), rng)
    data = LoadData("ml-100k", str(tmp_path), seed=0)
    out = str(tmp_path / "saved")
    data.graph.save(out)
    assert sorted(os.listdir(out)) == ["meta_graph.json", "movie.npz", "user.npz", "user_movie_csr.npz"]
    assert json.load(open(os.path.join(out, "meta_graph.json"))) == {"user": {"movie": 1}, "movie": {"user": 1}}
    u = np.load(os.path.join(out, "user.npz"))
    assert sorted(u.files) == ["features", "node_ids"] and u["features"].dtype == np.float32
    c = np.load(os.path.join(out, "user_movie_csr.npz"))
    assert sorted(c.files) == ["col_ids", "end_points", "ind_ptr", "multi_link", "row_ids", "values"]
    g2 = G.HeterGraph.load(out)
    for k in (("user", "movie"), ("movie", "user")):
        a, b = data.graph[k], g2[k]
        for f in ("end_points", "ind_ptr", "values", "row_ids", "col_ids", "multi_link"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), (k, f)
    assert np.array_equal(g2.features["movie"], data.item_features)
    g3 = G.HeterGraph.load(out, fea_normalize=True)
    f3 = g3.features["user"]
    np.testing.assert_allclose(f3.mean(0), 0, atol=1e-6)
    sd = data.user_features.std(0)
    np.testing.assert_allclose(f3.std(0)[sd > 0], 1, rtol=1e-6)

    # the reference's writer: movie->user direction only, no multi_link key for a plain matrix, features always present
    ref = str(tmp_path / "ref_style")
    os.makedirs(ref)
    json.dump({"user": {"movie": 1}, "movie": {"user": 1}}, open(os.path.join(ref, "meta_graph.json"), "w"))
    mu = data.graph["movie", "user"]
    np.savez_compressed(os.path.join(ref, "movie_user_csr.npz"), row_ids=mu.row_ids, col_ids=mu.col_ids, values=mu.values,
                        end_points=mu.end_points, ind_ptr=mu.ind_ptr)
    np.savez_compressed(os.path.join(ref, "user.npz"), node_ids=data.graph.node_ids_dict["user"],
                        features=data.user_features)
    np.savez_compressed(os.path.join(ref, "movie.npz"), node_ids=data.graph.node_ids_dict["movie"],
                        features=data.item_features)
    g4 = G.HeterGraph.load(ref)
    assert g4["user", "movie"].multi_link is None
    assert np.array_equal(g4["user", "movie"].end_points, data.graph["user", "movie"].end_points)
    assert np.array_equal(g4["user", "movie"].values, data.graph["user", "movie"].values)
    os.remove(os.path.join(ref, "movie_user_csr.npz"))
    with pytest.raises(IOError):
        G.HeterGraph.load(ref)


@pytest.mark.gpu
def test_training_from_a_movielens_directory(tmp_path):
    """examples/train_star_gcn.py --data-root: ml-1m-layout files (low-rank ratings so there is something to learn) ->
    LoadData -> DataIterator -> resident plan + device samplers; validation RMSE falls, test RMSE is reported."""
    import re
    import subprocess
    import sys
    rng = np.random.default_rng(0)
    nu, nm, n = 600, 400, 40000
    pu, pm = rng.normal(size=(nu, 3)), rng.normal(size=(nm, 3))
    cells = rng.choice(nu * nm, n, replace=False)
    u, m = cells // nm, cells % nm
    score = (pu[u] * pm[m]).sum(1)
    r = np.clip(np.round(3 + 1.2 * score / score.std()), 1, 5).astype(int)
    d = tmp_path / "ml-1m"
    d.mkdir()
    with open(d / "ratings.dat", "w") as f:
        f.writelines("%d::%d::%d::1\n" % (a + 1, 2 * b + 3, c) for a, b, c in zip(u, m, r))
    with open(d / "users.dat", "w") as f:
        f.writelines("%d::%s::%d::%d::00000\n" % (i + 1, "FM"[i % 2], 25, i % 21) for i in range(nu))
    with open(d / "movies.dat", "w") as f:
        f.writelines("%d::Film %d (1999)::Drama|Comedy\n" % (2 * i + 3, i) for i in range(nm))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "train_star_gcn.py"), "--data-root", str(tmp_path),
                          "--dataset", "ml-1m", "--iters", "150", "--eval-every", "75", "--batch", "4000", "--resident",
                          "--device-sampler", "--features"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Dataset Name=ml-1m" in out.stdout and "#ratings 40000" in out.stdout
    rmse = [float(x) for x in re.findall(r"valid RMSE ([0-9.]+)", out.stdout)]
    assert len(rmse) >= 3 and rmse[-1] < rmse[0] - 0.05, out.stdout
    assert re.search(r"test RMSE [0-9.]+", out.stdout)


@pytest.mark.gpu
def test_inductive_training_from_a_movielens_directory(tmp_path):
    """The inductive setting end to end (reference datasets.py:153-171 + iterators.py:171-176): held-out ITEMS never
    appear in the training graph, are evaluated with a zero input embedding through the graph that contains them, and the
    model still predicts their ratings better than the rating mean (it reconstructs them from their neighbourhood)."""
    import re
    import subprocess
    import sys
    rng = np.random.default_rng(1)
    nu, nm, n = 500, 300, 30000
    pu, pm = rng.normal(size=(nu, 3)), rng.normal(size=(nm, 3))
    cells = rng.choice(nu * nm, n, replace=False)
    u, m = cells // nm, cells % nm
    score = (pu[u] * pm[m]).sum(1)
    r = np.clip(np.round(3 + 1.2 * score / score.std()), 1, 5).astype(int)
    d = tmp_path / "ml-1m"
    d.mkdir()
    with open(d / "ratings.dat", "w") as f:
        f.writelines("%d::%d::%d::1\n" % (a + 1, b + 1, c) for a, b, c in zip(u, m, r))
    with open(d / "users.dat", "w") as f:
        f.writelines("%d::%s::%d::%d::00000\n" % (i + 1, "FM"[i % 2], 25, i % 21) for i in range(nu))
    with open(d / "movies.dat", "w") as f:
        f.writelines("%d::Film %d (1999)::Drama\n" % (i + 1, i) for i in range(nm))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "train_star_gcn.py"), "--data-root", str(tmp_path),
                          "--dataset", "ml-1m", "--inductive", "item", "--iters", "200", "--eval-every", "100", "--batch",
                          "3000"], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "Inductive (movie)" in out.stdout
    rmse = [float(x) for x in re.findall(r"valid RMSE ([0-9.]+)", out.stdout)]
    test = float(re.search(r"test RMSE ([0-9.]+)", out.stdout).group(1))
    assert len(rmse) >= 3 and rmse[-1] < rmse[0] - 0.03, out.stdout
    assert test < float(np.std(r)) * 0.97, (test, float(np.std(r)))
