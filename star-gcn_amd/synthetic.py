"""MovieLens-SHAPED synthetic bipartite rating graphs (no dataset files, no network): SURVEY.md section 8(d).

Users and items get log-normal propensities (sigma 1.0 / 1.5), edges are drawn without duplicate (user, item)
pairs, every node keeps degree >= 1, rating levels follow an ML-like skew, CSR rows are sorted by column index as
scipy `tocsr()` leaves them in the reference ETL (datasets.py:116-121)."""
import numpy as np

from .mxgraph.graph import CSRMat, HeterGraph

SHAPES = {   # name: (n_user, n_item, n_edges, n_levels)   -- SURVEY.md section 8 table
    "ml-100k": (943, 1682, 100000, 5),
    "ml-1m": (6040, 3706, 1000209, 5),
    "ml-10m": (69878, 10677, 10000054, 10),
    "tiny": (60, 45, 900, 5),
    # HBM-bound stress in the spirit of BASELINE config 5 (10M x 1M nodes, 1B edges, 16 levels, 8 GPUs), scaled to what
    # one GPU's host can plan in about a minute: every gathered matrix (0.5-8 GB) is far beyond the 256 MB Infinity Cache
    "hbm-stress": (600000, 500000, 60000000, 16),
}


def level_values(R):
    if R == 10:
        return np.arange(1, 11, dtype=np.float32) * 0.5      # 0.5 ... 5.0 (ml-10M100K half-star ratings)
    return np.arange(1, R + 1, dtype=np.float32)


def level_probs(R):
    base = np.array([1, 2, 5, 7, 4], dtype=np.float64)
    p = np.interp(np.linspace(0, 4, R), np.arange(5), base)
    return p / p.sum()


def bipartite_edges(n_user, n_item, n_edges, rng):
    pu = rng.lognormal(0.0, 1.0, n_user)
    pi = rng.lognormal(0.0, 1.5, n_item)
    pu, pi = pu / pu.sum(), pi / pi.sum()
    n_edges = int(min(n_edges, n_user * n_item // 2))
    keys = np.zeros(0, np.int64)
    while keys.size < n_edges:
        m = int((n_edges - keys.size) * 1.3) + 1024
        u = rng.choice(n_user, size=m, p=pu)
        i = rng.choice(n_item, size=m, p=pi)
        keys = np.unique(np.concatenate([keys, u.astype(np.int64) * n_item + i]))
    if keys.size > n_edges:
        keys = np.sort(rng.choice(keys, size=n_edges, replace=False))
    u, i = keys // n_item, keys % n_item
    # degree >= 1 everywhere: give isolated nodes one edge to a random partner
    miss_u = np.setdiff1d(np.arange(n_user), u)
    miss_i = np.setdiff1d(np.arange(n_item), i)
    if miss_u.size or miss_i.size:
        eu = np.concatenate([miss_u, rng.integers(0, n_user, miss_i.size)])
        ei = np.concatenate([rng.integers(0, n_item, miss_u.size), miss_i])
        keys = np.unique(np.concatenate([keys, eu.astype(np.int64) * n_item + ei]))
        u, i = keys // n_item, keys % n_item
    return u.astype(np.int32), i.astype(np.int32)


def make_graph(shape="ml-10m", seed=None, n_user=None, n_item=None, n_edges=None, n_levels=None,
               name_user="user", name_item="movie", signal=False):
    """-> (HeterGraph, user_idx, item_idx, rating_values) with edges in user-major / item-minor (CSR) order.
    signal=True makes the ratings learnable (quantised low-rank user x item affinity + noise, same level histogram)
    for training demos; the benchmark keeps the i.i.d. levels of SURVEY 8(d)."""
    nu, ni, ne, R = SHAPES[shape] if shape in SHAPES else (n_user, n_item, n_edges, n_levels)
    nu, ni, ne, R = (n_user or nu), (n_item or ni), (n_edges or ne), (n_levels or R)
    cfg_id = list(SHAPES).index(shape) if shape in SHAPES else 99
    rng = np.random.default_rng(20240917 + cfg_id if seed is None else seed)
    u, i = bipartite_edges(nu, ni, ne, rng)
    levels = level_values(R)
    if signal:
        pu, qi = rng.normal(size=(nu, 4)), rng.normal(size=(ni, 4))
        score = (pu[u] * qi[i]).sum(axis=1) + 0.5 * rng.normal(size=u.size)
        cuts = np.quantile(score, np.cumsum(level_probs(R))[:-1])
        vals = levels[np.searchsorted(cuts, score)]
    else:
        vals = levels[rng.choice(R, size=u.size, p=level_probs(R))]
    csr = CSRMat.from_edges(u, i, vals, nu, ni, multi_link=levels)
    graph = HeterGraph({name_user: np.arange(nu, dtype=np.int32), name_item: np.arange(ni, dtype=np.int32)},
                       {(name_user, name_item): csr})
    return graph, csr.edge_row_indices, csr.end_points, csr.values


def user_block(graph, name_user, name_item, lo, hi):
    """1-D node partition: the sub-graph of users [lo, hi) against ALL items (items are replicated)."""
    csr = graph[name_user, name_item]
    a, b = int(csr.ind_ptr[lo]), int(csr.ind_ptr[hi])
    sub = CSRMat(csr.end_points[a:b], csr.ind_ptr[lo:hi + 1] - csr.ind_ptr[lo], np.arange(hi - lo, dtype=np.int32),
                 csr.col_ids, csr.values[a:b], csr.multi_link, support_col_degrees=csr.col_degrees)
    return sub
