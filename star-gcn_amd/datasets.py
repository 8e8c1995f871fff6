"""MovieLens directory -> HeterGraph + node features + transductive / inductive splits.

Counterpart of reference mxgraph/datasets.py:39-171 (LoadData), :180-214 (inductive split), :404-574 (raw file readers and
feature builders).  File formats (GroupLens'):

  ml-100k   u1.base / u1.test   user \\t movie \\t rating \\t timestamp              (fixed split shipped with the data)
            u.user              id | age | gender | occupation | zip
            u.item              id | title | release | video release | url | 19 genre flags          (latin-1)
  ml-1m     ratings.dat         UserID::MovieID::Rating::Timestamp
            users.dat           UserID::Gender::Age::Occupation::Zip
            movies.dat          MovieID::Title (Year)::Genre|Genre|...                                (latin-1)
  ml-10m    ml-10M100K/ratings.dat, movies.dat (no user file: user feature = one zero column)

Nothing here downloads anything (the reference's downloader, :290-377, needs the network): point `root` at the directory
that holds the extracted `ml-100k` / `ml-1m` / `ml-10M100K` folder.

Deliberate deviations from the reference, none of which changes the graph:
* the validation ratings are the validation pairs' own ratings (reference :151 stores the TEST ratings there);
* the splits draw from the `seed`-ed generator (the reference seeds `self._rng` but then calls the global `np.random`);
* occupation one-hot columns are in sorted order (the reference enumerates a Python `set`: hash order);
* the 300 title-embedding columns come from `title_embedder(list_of_title_strings) -> (n, 300)`; the reference averages
  GloVe-840B vectors over spaCy tokens (:24-25, :551), both of which need a download, so the default fills zeros.
"""
import io
import os
import re

import numpy as np

from .mxgraph.graph import CSRMat, HeterGraph

_DIRS = {"ml-100k": "ml-100k", "ml-1m": "ml-1m", "ml-10m": "ml-10M100K"}
GENRES_ML_100K = ['unknown', 'Action', 'Adventure', 'Animation', 'Children', 'Comedy', 'Crime', 'Documentary', 'Drama',
                  'Fantasy', 'Film-Noir', 'Horror', 'Musical', 'Mystery', 'Romance', 'Sci-Fi', 'Thriller', 'War',
                  'Western']
GENRES = {"ml-100k": GENRES_ML_100K, "ml-1m": GENRES_ML_100K[1:], "ml-10m": GENRES_ML_100K + ['IMAX']}
_TITLE = re.compile(r'(.+)\s*\((\d+)\)')
TITLE_EMBED_DIM = 300


def read_ratings(path, sep):
    """(user_id, movie_id, rating) columns of a MovieLens rating file; the timestamp column is ignored.  One pass of
    numpy's text parser over the whole file (10 M lines of ml-10m in seconds, where a per-line loop takes minutes)."""
    with open(path, "rb") as f:
        raw = f.read()
    if sep != "\t":
        raw = raw.replace(sep.encode(), b" ")
    if not raw.strip():
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32)
    tab = np.loadtxt(io.BytesIO(raw), dtype=np.float64, usecols=(0, 1, 2), ndmin=2)
    return tab[:, 0].astype(np.int64), tab[:, 1].astype(np.int64), tab[:, 2].astype(np.float32)


def _read_rows(path, sep, n_min):
    rows = []
    with open(path, "r", encoding="latin-1") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if not line:
                continue
            parts = line.split(sep)
            if len(parts) < n_min:
                raise ValueError("%s: malformed line %r" % (path, line))
            rows.append(parts)
    return rows


def _title_and_year(title):
    """reference :541-549: 'Toy Story (1995)' -> ('Toy Story ', 1995); no year -> (title, 1950)."""
    m = _TITLE.match(title)
    if m is None:
        return title, 1950.0
    text, year = m.groups()
    return text, float(year)


class LoadData(object):
    name_user, name_item = "user", "movie"

    def __init__(self, name, root, use_inductive=False, test_ratio=0.1, val_ratio=0.1, inductive_key="item",
                 inductive_node_frac=10, inductive_edge_frac=90, seed=None, title_embedder=None):
        if name not in _DIRS:
            raise NotImplementedError(name)
        self._name = name
        self._rng = np.random.RandomState(seed)
        self._title_embedder = title_embedder
        path = os.path.join(root, _DIRS[name])
        if name == "ml-100k":       # fixed split shipped with the data set (reference :84-87)
            tr = read_ratings(os.path.join(path, "u1.base"), "\t")
            te = read_ratings(os.path.join(path, "u1.test"), "\t")
            u, m, r = (np.concatenate([a, b]) for a, b in zip(tr, te))
            train_sel, test_sel = np.arange(tr[0].size), np.arange(tr[0].size, u.size)
        else:
            u, m, r = read_ratings(os.path.join(path, "ratings.dat"), "::")
            train_sel = test_sel = None
        self._load_user_info(path, u)
        self._load_movie_info(path)
        self._drop_unseen(u, m)
        self.user_features = self._user_features()
        self.item_features = self._movie_features()

        # contiguous indices in info-file order (reference :108-109)
        self.raw_user_ids, self.raw_movie_ids = self.user_info["id"], self.movie_info["id"]
        uidx, midx = self._index_of(self.raw_user_ids, u, "user"), self._index_of(self.raw_movie_ids, m, "movie")
        self.uniq_ratings = np.unique(r)
        csr = CSRMat.from_edges(uidx, midx, r, self.num_user, self.num_item, multi_link=self.uniq_ratings)
        csr.check_consistency()
        self._graph = HeterGraph({self.name_user: np.arange(self.num_user, dtype=np.int32),
                                  self.name_item: np.arange(self.num_item, dtype=np.int32)},
                                 {(self.name_user, self.name_item): csr},
                                 features={self.name_user: self.user_features, self.name_item: self.item_features})

        self._use_inductive = bool(use_inductive)
        if not use_inductive:       # reference :133-152
            if test_sel is None:
                perm = self._rng.permutation(u.size)
                n_test = int(np.ceil(u.size * test_ratio))
                test_sel, train_sel = perm[:n_test], perm[n_test:]
            n_valid = int(np.ceil(train_sel.size * val_ratio))
            valid_sel = train_sel[self._rng.permutation(train_sel.size)[:n_valid]]
            self._test_data = (np.stack([uidx[test_sel], midx[test_sel]]), r[test_sel])
            self._valid_data = (np.stack([uidx[valid_sel], midx[valid_sel]]), r[valid_sel])
        else:                       # reference :153-171
            if inductive_key not in ("item", "user"):
                raise NotImplementedError(inductive_key)
            self._inductive_key = self.name_item if inductive_key == "item" else self.name_user
            self._inductive_node_frac, self._inductive_edge_frac = inductive_node_frac, inductive_edge_frac
            all_ids = self._graph.node_ids_dict[self._inductive_key]
            train_val_ids, self._inductive_test_ids, self._test_data = self._gen_inductive_data(all_ids)
            self._inductive_train_ids, self._inductive_valid_ids, self._valid_data = \
                self._gen_inductive_data(train_val_ids)
            assert (np.unique(self._inductive_train_ids).size + np.unique(self._inductive_valid_ids).size
                    + np.unique(self._inductive_test_ids).size) == all_ids.size

    # ---- raw files -------------------------------------------------------------------------------------------------
    def _load_user_info(self, path, rating_users):
        """reference :422-456"""
        if self._name == "ml-100k":
            rows = _read_rows(os.path.join(path, "u.user"), "|", 5)
            cols = dict(id=0, age=1, gender=2, occupation=3)
        elif self._name == "ml-1m":
            rows = _read_rows(os.path.join(path, "users.dat"), "::", 5)
            cols = dict(id=0, gender=1, age=2, occupation=3)
        else:
            self.user_info = {"id": np.unique(rating_users)}
            return
        self.user_info = {"id": np.array([int(p[cols["id"]]) for p in rows], np.int64),
                          "age": np.array([float(p[cols["age"]]) for p in rows], np.float32),
                          "gender": np.array([p[cols["gender"]] for p in rows]),
                          "occupation": np.array([p[cols["occupation"]] for p in rows])}

    def _load_movie_info(self, path):
        """reference :490-533"""
        genres = GENRES[self._name]
        if self._name == "ml-100k":
            rows = _read_rows(os.path.join(path, "u.item"), "|", 5 + len(genres))
            flags = np.array([[float(x) for x in p[5:5 + len(genres)]] for p in rows], np.float32)
        else:
            rows = _read_rows(os.path.join(path, "movies.dat"), "::", 3)
            rows = [[p[0], "::".join(p[1:-1]), p[-1]] for p in rows]       # a title may itself contain '::'
            gmap = {g: i for i, g in enumerate(genres)}
            gmap["Children's"] = gmap["Childrens"] = gmap["Children"]
            flags = np.zeros((len(rows), len(genres)), np.float32)
            for i, p in enumerate(rows):
                for g in p[2].split("|"):
                    if g in gmap:
                        flags[i, gmap[g]] = 1.0
                    elif "unknown" in gmap:
                        flags[i, gmap["unknown"]] = 1.0
                    else:
                        raise ValueError("%s: genre %r of movie %s has no column" % (self._name, g, p[0]))
        self.movie_info = {"id": np.array([int(p[0]) for p in rows], np.int64),
                           "title": np.array([p[1] for p in rows], dtype=object), "genres": flags}

    def _drop_unseen(self, rating_users, rating_movies):
        """reference :381-397: keep the info rows whose id occurs in the ratings, in file order."""
        for info, ids, what in ((self.user_info, rating_users, "user"), (self.movie_info, rating_movies, "movie")):
            seen = np.unique(ids)
            keep = np.isin(info["id"], seen)
            if np.unique(info["id"][keep]).size != seen.size:
                missing = np.setdiff1d(seen, info["id"])
                raise ValueError("%d rated %s ids have no row in the info file (e.g. %s)" % (missing.size, what,
                                                                                             missing[:5]))
            for k in info:
                info[k] = info[k][keep]

    @staticmethod
    def _index_of(table_ids, ids, what):
        order = np.argsort(table_ids, kind="stable")
        pos = np.searchsorted(table_ids[order], ids)
        if np.any(table_ids[order][np.minimum(pos, order.size - 1)] != ids):
            raise ValueError("unknown %s id in the ratings" % what)
        return order[pos].astype(np.int32)

    # ---- features --------------------------------------------------------------------------------------------------
    def _user_features(self):
        """reference :458-488: [age / 50, gender == 'F', one-hot occupation]; ml-10m: a single zero column."""
        info = self.user_info
        n = info["id"].size
        if self._name == "ml-10m":
            return np.zeros((n, 1), np.float32)
        occ, occ_idx = np.unique(info["occupation"], return_inverse=True)
        one_hot = np.zeros((n, occ.size), np.float32)
        one_hot[np.arange(n), occ_idx] = 1.0
        self.occupations = occ
        return np.concatenate([(info["age"] / 50.0).reshape(n, 1), (info["gender"] == "F").astype(np.float32).reshape(n, 1),
                               one_hot], axis=1).astype(np.float32)

    def _movie_features(self):
        """reference :535-563: [title embedding (300), (year - 1950) / 100, genre flags]."""
        info = self.movie_info
        n = info["id"].size
        parsed = [_title_and_year(t) for t in info["title"]]
        years = np.array([y for _, y in parsed], np.float32).reshape(n, 1)
        if self._title_embedder is None:
            emb = np.zeros((n, TITLE_EMBED_DIM), np.float32)
        else:
            emb = np.asarray(self._title_embedder([t for t, _ in parsed]), np.float32)
            if emb.shape != (n, TITLE_EMBED_DIM):
                raise ValueError("title_embedder must return (%d, %d), got %s" % (n, TITLE_EMBED_DIM, emb.shape))
        return np.concatenate([emb, (years - 1950.0) / 100.0, info["genres"]], axis=1).astype(np.float32)

    # ---- inductive split ---------------------------------------------------------------------------------------------
    def _gen_inductive_data(self, node_ids):
        """reference :180-214.  Walk the nodes in random order; a node with > 10 ratings becomes a held-out node until
        ceil(#nodes * node_frac %) are found, and floor(#ratings * edge_frac %) of its ratings, chosen at random, are
        its held-out pairs; everything else trains."""
        by_user = self._inductive_key == self.name_user
        csr = self._graph[self.name_user, self.name_item] if by_user else self._graph[self.name_item, self.name_user]
        shuffled = self._rng.permutation(node_ids)
        n_test = int(np.ceil(node_ids.size / 100.0 * self._inductive_node_frac))
        test_ids, train_ids, pairs_l, stop = [], [], [], None
        for pos, nid in enumerate(shuffled):
            row = int(csr.row_id_to_ind(np.array([nid]))[0])
            nbrs = csr.col_ids[csr.end_points[csr.ind_ptr[row]:csr.ind_ptr[row + 1]]]
            assert nbrs.size != 0
            if nbrs.size <= 10:
                train_ids.append(nid)
            else:
                test_ids.append(nid)
                chosen = self._rng.permutation(nbrs.size)[:int(np.floor(nbrs.size / 100.0 * self._inductive_edge_frac))]
                own = np.full(chosen.size, nid, np.int32)
                pairs_l.append(np.stack([own, nbrs[chosen]]) if by_user else np.stack([nbrs[chosen], own]))
            if len(test_ids) == n_test:
                stop = pos
                break
        if stop is None or stop + 1 >= node_ids.size:
            raise ValueError("not enough nodes with more than 10 ratings for a %d %% inductive split"
                             % self._inductive_node_frac)
        test_ids = np.array(test_ids, np.int32)
        train_ids = np.concatenate([np.array(train_ids, np.int32), shuffled[stop + 1:].astype(np.int32)])
        pairs = np.hstack(pairs_l).astype(np.int32)
        return train_ids, test_ids, (pairs, self._graph.fetch_edges_by_id(self.name_user, self.name_item, pairs))

    graph = property(lambda self: self._graph)
    test_data = property(lambda self: self._test_data)
    valid_data = property(lambda self: self._valid_data)
    num_user = property(lambda self: int(self.user_info["id"].size))
    num_item = property(lambda self: int(self.movie_info["id"].size))
    num_links = property(lambda self: self.uniq_ratings)
    inductive_test_ids = property(lambda self: self._inductive_test_ids)
    inductive_train_ids = property(lambda self: self._inductive_train_ids)
    inductive_valid_ids = property(lambda self: self._inductive_valid_ids)

    def __repr__(self):
        s = "Dataset Name=%s\n#users %d (features %s)  #movies %d (features %s)  #ratings %d  levels %s\n" % (
            self._name, self.num_user, self.user_features.shape, self.num_item, self.item_features.shape,
            self._graph[self.name_user, self.name_item].nnz, self.uniq_ratings)
        s += "#Val/Test edges: %d/%d\n" % (self.valid_data[1].size, self.test_data[1].size)
        if self._use_inductive:
            s += "Inductive (%s): node ratio %d %%, #train/valid/test nodes %d/%d/%d; edge ratio %d %%\n" % (
                self._inductive_key, self._inductive_node_frac, self.inductive_train_ids.size,
                self.inductive_valid_ids.size, self.inductive_test_ids.size, self._inductive_edge_frac)
        return s
