"""MovieLens rating files -> HeterGraph + transductive splits (reference mxgraph/datasets.py:56-160).

Only the part that feeds the hot path: the rating files (`u1.base` / `u1.test`, tab separated, for ml-100k;
`ratings.dat`, '::' separated, for ml-1m / ml-10m -- reference datasets.py:84-90, 312-330), the contiguous user /
movie index maps (raw ids in increasing order among the ids that occur in the ratings, as the reference obtains from
its id-sorted info files after `_drop_unseen_nodes`), the user->movie CSR with `multi_link` = the distinct rating
values (datasets.py:116-123), and the test / validation pair splits (datasets.py:134-152).  Node features (user/movie
attributes, GloVe title embeddings) and the download helper need the network and are out of scope; the inductive
split is not implemented.  Nothing here downloads anything: point `root` at an extracted MovieLens directory.
"""
import os

import numpy as np

from .mxgraph.graph import CSRMat, HeterGraph

_FILES = {"ml-100k": ("ml-100k", "\t"), "ml-1m": ("ml-1m", "::"), "ml-10m": ("ml-10M100K", "::")}


def read_ratings(path, sep):
    """(user_id, movie_id, rating) columns of a MovieLens rating file (the 4th column, the timestamp, is ignored)."""
    users, movies, ratings = [], [], []
    with open(path, "r") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            parts = line.split(sep)
            users.append(int(parts[0]))
            movies.append(int(parts[1]))
            ratings.append(float(parts[2]))
    return np.array(users, np.int64), np.array(movies, np.int64), np.array(ratings, np.float32)


class LoadData(object):
    name_user, name_item = "user", "movie"

    def __init__(self, name, root, test_ratio=0.1, val_ratio=0.1, seed=None):
        if name not in _FILES:
            raise NotImplementedError(name)
        sub, sep = _FILES[name]
        data_path = os.path.join(root, sub)
        rng = np.random.RandomState(seed)
        if name == "ml-100k":       # fixed split shipped with the data set (reference datasets.py:84-87)
            tr = read_ratings(os.path.join(data_path, "u1.base"), sep)
            te = read_ratings(os.path.join(data_path, "u1.test"), sep)
            u, m, r = (np.concatenate([a, b]) for a, b in zip(tr, te))
            n_train_all = tr[0].size
            test_sel = np.arange(n_train_all, u.size)
            train_sel = np.arange(n_train_all)
        else:                       # random split (reference datasets.py:135-139)
            u, m, r = read_ratings(os.path.join(data_path, "ratings.dat"), sep)
            perm = rng.permutation(u.size)
            n_test = int(np.ceil(u.size * test_ratio))
            test_sel, train_sel = perm[:n_test], perm[n_test:]
        n_valid = int(np.ceil(train_sel.size * val_ratio))
        valid_sel = train_sel[rng.permutation(train_sel.size)[:n_valid]]
        self.raw_user_ids, uidx = np.unique(u, return_inverse=True)
        self.raw_movie_ids, midx = np.unique(m, return_inverse=True)
        uidx, midx = uidx.astype(np.int32), midx.astype(np.int32)
        self.uniq_ratings = np.unique(r)
        csr = CSRMat.from_edges(uidx, midx, r, self.raw_user_ids.size, self.raw_movie_ids.size,
                                multi_link=self.uniq_ratings)
        csr.check_consistency()
        self._graph = HeterGraph({self.name_user: np.arange(self.raw_user_ids.size, dtype=np.int32),
                                  self.name_item: np.arange(self.raw_movie_ids.size, dtype=np.int32)},
                                 {(self.name_user, self.name_item): csr})
        self._test_data = (np.stack([uidx[test_sel], midx[test_sel]]), r[test_sel])
        self._valid_data = (np.stack([uidx[valid_sel], midx[valid_sel]]), r[valid_sel])

    graph = property(lambda self: self._graph)
    test_data = property(lambda self: self._test_data)
    valid_data = property(lambda self: self._valid_data)
    num_user = property(lambda self: int(self.raw_user_ids.size))
    num_item = property(lambda self: int(self.raw_movie_ids.size))
    num_links = property(lambda self: self.uniq_ratings)
