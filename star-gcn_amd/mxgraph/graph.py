"""Graph containers that produce the integer inputs of the hot path: a minimal counterpart of reference
mxgraph/graph.py (CSRMat :262-755, HeterGraph :758-1100, merge_nodes :142-164, merge_node_ids_dict :166-219,
unordered_unique :63-68, empty_as_zero :221-222), numpy on the host with the heavy loops in native code
(libstargcn_hip `_cpu` entry points instead of the reference's `mxgraph._graph_sampler` CPython extension).

Semantics kept: support uses the degrees of the CURRENT matrix over all rating levels;
multi_link levels are matched by exact float equality against the sorted unique values; each level keeps CSR
order and a full-length indptr; the reverse direction of a HeterGraph is the transpose with rows sorted by
column index; `num_neighbors < 0` copies whole rows (deterministic, RNG-free).
"""
import ctypes
import json
import os

import numpy as np

from ._native import _lib as L


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_SEED = [0x5EED]


def set_seed(seed):
    """reference graph.py:70-81: seed of the native neighbour sampler (every draw advances it)."""
    _SEED[0] = int(seed) & 0xFFFFFFFFFFFFFFFF


def _next_seed():
    _SEED[0] = (_SEED[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
    return _SEED[0]


def unordered_unique(data, return_counts=False, return_inverse=False):
    """Unique values in FIRST-OCCURRENCE order (what the reference's hash-based unique yields for n <= 10000,
    graph_sampler.h:465-534; above that its order depends on the OpenMP schedule)."""
    data = _i32(data).ravel()
    if not (return_counts or return_inverse):
        raise NotImplementedError
    n = data.size
    lo, hi = (int(data.min()), int(data.max())) if n else (0, -1)
    if n and lo >= 0 and hi <= 16 * n + (1 << 20):   # native O(n) direct-address path (sg_unique_inverse_cpu)
        uniq = np.empty(n, np.int32)
        other = np.empty(n, np.int32)
        m = ctypes.c_int64(0)
        L.check(L.lib().sg_unique_inverse_cpu(_vp(uniq), None if return_counts else _vp(other),
                                              _vp(other) if return_counts else None, ctypes.byref(m), _vp(data), n, hi),
                "sg_unique_inverse_cpu")
        return (uniq[:m.value].copy(), other[:m.value].copy()) if return_counts else (uniq[:m.value].copy(), other)
    uniq_sorted, first, inverse, counts = np.unique(data, return_index=True, return_inverse=True, return_counts=True)
    order = np.argsort(first, kind="stable")
    uniq = uniq_sorted[order].astype(np.int32)
    if return_counts:
        return uniq, counts[order].astype(np.int32)
    if return_inverse:
        rank = np.empty_like(order)
        rank[order] = np.arange(order.size)
        return uniq, rank[inverse].astype(np.int32)
    raise NotImplementedError


def merge_nodes(node_ids):
    if isinstance(node_ids, np.ndarray):
        return unordered_unique(node_ids, return_inverse=True)
    sizes = [int(np.asarray(a).size) for a in node_ids]
    flat = np.concatenate([_i32(a).ravel() for a in node_ids]) if node_ids else np.zeros(0, np.int32)
    uniq, inv = unordered_unique(flat, return_inverse=True)
    out, pos = [], 0
    for s in sizes:
        out.append(inv[pos:pos + s])
        pos += s
    return uniq, out


def merge_node_ids_dict(data):
    """reference graph.py:166-219.  data: sequence of dicts holding either {key: ids (n,)} or
    {(src_key, dst_key): ids (1 + K, n)} (row 0 = src ids, rows 1.. = dst ids).  Returns ({key: unique ids},
    [same-shaped dicts with every id replaced by its index in the unique list of its node type])."""
    per_key = dict()
    for d in data:
        for key, ids in d.items():
            ids = _i32(ids)
            if isinstance(key, tuple):
                assert ids.ndim == 2
                per_key.setdefault(key[0], []).append(ids[0, :])
                per_key.setdefault(key[1], []).append(ids[1:, :].reshape(-1))
            else:
                per_key.setdefault(key, []).append(ids)
    uniq, inds = dict(), dict()
    for key, lst in per_key.items():
        uniq[key], inds[key] = merge_nodes(lst)
    counter = {key: 0 for key in per_key}

    def pop(key):
        counter[key] += 1
        return inds[key][counter[key] - 1]

    out = []
    for d in data:
        nd = dict()
        for key, ids in d.items():
            if isinstance(key, tuple):
                shape = np.asarray(ids).shape
                src = pop(key[0]).reshape((1, shape[1]))
                dst = pop(key[1]).reshape((shape[0] - 1, shape[1]))
                nd[key] = np.concatenate([src, dst], axis=0)
            else:
                nd[key] = pop(key)
        out.append(nd)
    return uniq, out


def empty_as_zero(l, dtype):
    return [np.asarray(e, dtype=dtype) if np.asarray(e).size > 0 else np.zeros((1,), dtype=dtype) for e in l]


class _IdMap(object):
    """node id -> row/column index; ids that are not in the map give -1 (never another node's index)."""

    def __init__(self, ids):
        ids = _i32(ids)
        self._lo = int(ids.min()) if ids.size else 0
        hi = int(ids.max()) if ids.size else -1
        self._map = -np.ones(hi - self._lo + 1, dtype=np.int32)
        self._map[ids - self._lo] = np.arange(ids.size, dtype=np.int32)

    def __getitem__(self, ids):
        rel = _i32(ids).astype(np.int64) - self._lo
        ok = (rel >= 0) & (rel < self._map.size)
        if ok.all():
            return self._map[rel]
        out = -np.ones(rel.shape, dtype=np.int32)
        out[ok] = self._map[rel[ok]]
        return out


class CSRMat(object):
    def __init__(self, end_points, ind_ptr, row_ids, col_ids, values=None, multi_link=None, support_row_degrees=None,
                 support_col_degrees=None):
        """support_*_degrees: degrees to use in `get_support` instead of this matrix's own (a rank-local block of a
        node-partitioned graph must normalise with the GLOBAL degrees of the replicated side)."""
        self._sup_rd = None if support_row_degrees is None else _i32(support_row_degrees)
        self._sup_cd = None if support_col_degrees is None else _i32(support_col_degrees)
        self.end_points, self.ind_ptr = _i32(end_points), _i32(ind_ptr)
        assert self.ind_ptr[0] == 0 and self.ind_ptr[-1] == self.end_points.shape[0]
        self.values = (np.ones(self.end_points.shape, np.float32) if values is None
                       else np.ascontiguousarray(values, dtype=np.float32))
        self.multi_link = np.sort(np.asarray(multi_link, dtype=np.float32)) if multi_link is not None else None
        self.row_ids, self.col_ids = _i32(row_ids), _i32(col_ids)
        assert self.ind_ptr.size == self.row_ids.size + 1
        self._row_map, self._col_map = _IdMap(self.row_ids), _IdMap(self.col_ids)
        self._support = dict()
        self._col_deg = None
        self._rows_sorted = None

    @classmethod
    def from_edges(cls, row_ind, col_ind, values, n_rows, n_cols, row_ids=None, col_ids=None, multi_link=None):
        """COO -> CSR with rows sorted by column index (what scipy `tocsr()` of a duplicate-free COO gives)."""
        row_ind, col_ind = _i32(row_ind), _i32(col_ind)
        order = np.lexsort((col_ind, row_ind))
        ind_ptr = np.zeros(n_rows + 1, np.int64)
        np.cumsum(np.bincount(row_ind, minlength=n_rows), out=ind_ptr[1:])
        return cls(col_ind[order], ind_ptr.astype(np.int32),
                   np.arange(n_rows, dtype=np.int32) if row_ids is None else row_ids,
                   np.arange(n_cols, dtype=np.int32) if col_ids is None else col_ids,
                   None if values is None else np.asarray(values)[order], multi_link)

    @property
    def shape(self):
        return self.row_ids.size, self.col_ids.size

    @property
    def nnz(self):
        return int(self.end_points.size)

    @property
    def row_degrees(self):
        return np.ascontiguousarray(np.diff(self.ind_ptr).astype(np.int32))

    @property
    def col_degrees(self):
        if self._col_deg is None:
            self._col_deg = np.bincount(self.end_points, minlength=self.col_ids.size).astype(np.int32)
        return self._col_deg

    @property
    def edge_row_indices(self):
        out = np.empty(self.nnz, np.int32)
        L.check(L.lib().sg_gen_row_indices_cpu(_vp(out), _vp(self.ind_ptr), self.shape[0], self.nnz), "sg_gen_row_indices_cpu")
        return out

    @property
    def node_pair_ids(self):
        """(2, nnz): row id and column id of every stored edge, in CSR order."""
        return np.stack([self.row_ids[self.edge_row_indices], self.col_ids[self.end_points]], axis=0)

    @property
    def rows_sorted(self):
        """True when every row lists its columns in increasing order (what `from_edges`, `.T`, row selection and edge
        removal give).  `submat` with a non-monotone column selection does not -- the reference's slice_csr_mat keeps
        the original order inside a row and so does this -- and the binary-search helpers then take the sort-based path."""
        if self._rows_sorted is None:
            if self.nnz < 2:
                self._rows_sorted = True
            else:
                inc = self.end_points[1:] > self.end_points[:-1]
                inc[self.ind_ptr[1:-1][(self.ind_ptr[1:-1] > 0) & (self.ind_ptr[1:-1] < self.nnz)] - 1] = True   # row starts
                self._rows_sorted = bool(inc.all())
        return self._rows_sorted

    def _positions(self, r, c):
        """CSR position of every (row index, col index) pair, -1 where it is not an edge; any order inside the rows."""
        r, c = _i32(r), _i32(c)
        pos = np.empty(r.size, np.int32)
        if self.rows_sorted:
            L.check(L.lib().sg_edge_positions_cpu(_vp(pos), _vp(self.end_points), _vp(self.ind_ptr), self.shape[0],
                                                  _vp(r), _vp(c), r.size), "sg_edge_positions_cpu")
            return pos
        ncol = max(self.shape[1], 1)
        key = self.edge_row_indices.astype(np.int64) * ncol + self.end_points
        order = np.argsort(key, kind="stable")
        want = r.astype(np.int64) * ncol + c
        at = np.searchsorted(key[order], want)
        ok = (r >= 0) & (c >= 0) & (at < key.size)
        ok[ok] = key[order[at[ok]]] == want[ok]
        pos[:] = -1
        pos[ok] = order[at[ok]]
        return pos

    def fetch_edges_by_id(self, node_pair_ids):
        """Edge values of the given (row id, col id) pairs (every pair must exist)."""
        pos = self._positions(self.row_id_to_ind(node_pair_ids[0]), self.col_id_to_ind(node_pair_ids[1]))
        if np.any(pos < 0):
            raise ValueError("fetch_edges_by_id: some node pairs are not edges of this matrix")
        return self.values[pos]

    def edge_positions(self, node_pair_ids):
        """Position in CSR order (= edge id) of every (row id, col id) pair; -1 where the pair is not an edge."""
        return self._positions(self.row_id_to_ind(node_pair_ids[0]), self.col_id_to_ind(node_pair_ids[1]))

    def submat(self, row_indices=None, col_indices=None):
        """reference graph.py:493-531 -> slice_csr_mat (graph_sampler.cpp:31-152; native: sg_csr_submat_cpu): the rows
        `row_indices` in the given order and the columns `col_indices`, re-indexed by their position in that list;
        entries keep their order inside a row.  None = everything."""
        rows = None if row_indices is None else _i32(np.atleast_1d(row_indices))
        cols = None if col_indices is None else _i32(np.atleast_1d(col_indices))
        if rows is None and cols is None:
            return CSRMat(self.end_points.copy(), self.ind_ptr.copy(), self.row_ids.copy(), self.col_ids.copy(),
                          self.values.copy(), self.multi_link, support_row_degrees=self._sup_rd,
                          support_col_degrees=self._sup_cd)
        # A rank-local block normalises with override (GLOBAL) degrees.  Selecting columns keeps the override of the
        # kept columns exact; selecting ROWS changes the column degrees by an amount only the other ranks know (and vice
        # versa) -- that is an error, never a silent fall-back to rank-local degrees (same rule as remove_edges_by_id /
        # ResidentPlan).
        if (rows is not None and self._sup_cd is not None) or (cols is not None and self._sup_rd is not None):
            raise ValueError("submat: slicing a rank-local block across its override support degrees needs the degrees "
                             "of the sliced graph from all ranks; slice the global graph, then take the block")
        col_map = None
        if cols is not None:
            if np.unique(cols).size != cols.size:
                raise ValueError("submat: duplicate column indices")
            col_map = -np.ones(self.shape[1], np.int32)
            col_map[cols] = np.arange(cols.size, dtype=np.int32)
        n = self.shape[0] if rows is None else rows.size
        bound = self.nnz if rows is None else int((self.ind_ptr[rows + 1] - self.ind_ptr[rows]).sum())
        ep, vals = np.empty(max(bound, 1), np.int32), np.empty(max(bound, 1), np.float32)
        ind_ptr = np.empty(n + 1, np.int32)
        m = ctypes.c_int64(0)
        L.check(L.lib().sg_csr_submat_cpu(_vp(ep), _vp(vals), _vp(ind_ptr), ctypes.byref(m), _vp(self.end_points),
                                          _vp(self.values), _vp(self.ind_ptr), self.shape[0],
                                          None if rows is None else _vp(rows), n, None if col_map is None else _vp(col_map)),
                "sg_csr_submat_cpu")
        sup_rd = None if self._sup_rd is None else (self._sup_rd if rows is None else self._sup_rd[rows])
        sup_cd = None if self._sup_cd is None else (self._sup_cd if cols is None else self._sup_cd[cols])
        return CSRMat(ep[:m.value], ind_ptr, self.row_ids if rows is None else self.row_ids[rows],
                      self.col_ids if cols is None else self.col_ids[cols], vals[:m.value], self.multi_link,
                      support_row_degrees=sup_rd, support_col_degrees=sup_cd)

    def submat_by_id(self, row_ids=None, col_ids=None):
        """reference graph.py:535-538"""
        r = None if row_ids is None else self.row_id_to_ind(np.atleast_1d(row_ids))
        c = None if col_ids is None else self.col_id_to_ind(np.atleast_1d(col_ids))
        if (r is not None and np.any(r < 0)) or (c is not None and np.any(c < 0)):
            raise ValueError("submat_by_id: unknown row / column id")
        return self.submat(r, c)

    def row_id_to_ind(self, ids):
        return self._row_map[ids]

    def col_id_to_ind(self, ids):
        return self._col_map[ids]

    def get_support(self, symm=True):
        """reference graph.py:414-429 -> graph_sampler.cpp:393-420 (native: sg_get_support_cpu)."""
        if symm not in self._support:
            out = np.empty(max(self.nnz, 1), np.float32)
            cd = (self.col_degrees if self._sup_cd is None else self._sup_cd) if symm \
                else np.zeros(self.col_ids.size, np.int32)
            rd = self.row_degrees if self._sup_rd is None else self._sup_rd
            L.check(L.lib().sg_get_support_cpu(_vp(out), _vp(rd), _vp(_i32(cd)), _vp(self.end_points),
                                               _vp(self.ind_ptr), self.shape[0], int(bool(symm))), "sg_get_support_cpu")
            self._support[symm] = out[:self.nnz]
        return self._support[symm]

    @property
    def T(self):
        """Transpose with rows sorted by column index (reference graph.py:585-593 via scipy)."""
        order = np.argsort(self.end_points, kind="stable")
        ind_ptr = np.zeros(self.shape[1] + 1, np.int64)
        np.cumsum(self.col_degrees, out=ind_ptr[1:])
        return CSRMat(self.edge_row_indices[order], ind_ptr.astype(np.int32), self.col_ids, self.row_ids,
                      self.values[order], self.multi_link, support_row_degrees=self._sup_cd,
                      support_col_degrees=self._sup_rd)

    def multi_link_split(self, values, ind_ptr):
        """reference graph_sampler.cpp:277-376 (native: sg_multi_link_split_cpu)."""
        values, ind_ptr = np.ascontiguousarray(values, np.float32), _i32(ind_ptr)
        n, nl = ind_ptr.size - 1, self.multi_link.size
        pos = np.empty(max(values.size, 1), np.int32)
        ips = np.empty((nl, n + 1), np.int32)
        off = np.empty(nl + 1, np.int64)
        L.check(L.lib().sg_multi_link_split_cpu(_vp(pos), _vp(ips), _vp(off), _vp(values), _vp(ind_ptr),
                                                _vp(self.multi_link), n, nl), "sg_multi_link_split_cpu")
        return [pos[off[l]:off[l + 1]] for l in range(nl)], [np.ascontiguousarray(ips[l]) for l in range(nl)]

    def sample_neighbors(self, src_ids=None, symm=True, use_multi_link=True, num_neighbors=None, rng=None):
        """reference graph.py:677-748.  num_neighbors None / < 0: whole rows, deterministic."""
        src = np.arange(self.shape[0], dtype=np.int32) if src_ids is None else self.row_id_to_ind(src_ids)
        if src.size and src.min() < 0:
            raise ValueError("sample_neighbors: unknown source node id")
        beg, end = self.ind_ptr[src].astype(np.int64), self.ind_ptr[src + 1].astype(np.int64)
        lens = end - beg
        if num_neighbors is not None and num_neighbors >= 0:
            # native sampler (sg_sample_fix_neighbor_cpu == reference random_sample_fix_neighbor)
            seed = int(rng.integers(0, 2 ** 63)) if rng is not None else _next_seed()
            src32 = _i32(src)
            dst_ptr = np.empty(src32.size + 1, np.int32)
            L.check(L.lib().sg_sample_fix_neighbor_cpu(None, _vp(dst_ptr), _vp(self.ind_ptr), _vp(src32), src32.size,
                                                       int(num_neighbors), seed), "sg_sample_fix_neighbor_cpu")
            sampled = np.empty(max(int(dst_ptr[-1]), 1), np.int32)
            L.check(L.lib().sg_sample_fix_neighbor_cpu(_vp(sampled), _vp(dst_ptr), _vp(self.ind_ptr), _vp(src32),
                                                       src32.size, int(num_neighbors), seed), "sg_sample_fix_neighbor_cpu")
            sampled = sampled[:int(dst_ptr[-1])]
            lens = np.diff(dst_ptr).astype(np.int64)
        else:
            dst_ptr = np.concatenate([[0], np.cumsum(lens)])
            sampled = np.repeat(beg - dst_ptr[:-1], lens) + np.arange(int(dst_ptr[-1]))
        dst_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        sampled = sampled.astype(np.int64)
        ep_ids = self.col_ids[self.end_points[sampled]]
        values = self.values[sampled]
        support = self.get_support(symm)[sampled]
        if not use_multi_link:
            return ep_ids, values, dst_ptr, support
        assert self.multi_link is not None
        split, ptr_l = self.multi_link_split(values, dst_ptr)
        return [ep_ids[s] for s in split], [values[s] for s in split], ptr_l, [support[s] for s in split]

    def remove_edges_by_id(self, node_pair_ids, degree_reducer=None):
        """reference graph.py:660-675: drop the listed (row id, col id) pairs; a NEW CSRMat (fresh degree caches).

        A rank-local block of a node-partitioned graph carries override degrees for `get_support` (the GLOBAL degrees
        of the replicated side).  They are carried over minus the removed edges; `degree_reducer(int64 array) ->
        array summed over ranks` must be given when several ranks remove edges (each rank only sees its own), and its
        absence in a multi-rank run is an error rather than a silent fall-back to rank-local degrees."""
        r, c = _i32(self.row_id_to_ind(node_pair_ids[0])), _i32(self.col_id_to_ind(node_pair_ids[1]))
        if np.any(r < 0) or np.any(c < 0):
            raise ValueError("remove_edges_by_id: unknown row / column id")
        ep = np.empty(max(self.nnz, 1), np.int32)
        vals = np.empty(max(self.nnz, 1), np.float32)
        ind_ptr = np.empty(self.shape[0] + 1, np.int32)
        m = ctypes.c_int64(0)
        if self.rows_sorted:
            L.check(L.lib().sg_remove_edges_cpu(_vp(ep), _vp(vals), _vp(ind_ptr), ctypes.byref(m), _vp(self.end_points),
                                                _vp(self.values), _vp(self.ind_ptr), self.shape[0], _vp(r), _vp(c),
                                                r.size), "sg_remove_edges_cpu")
        else:       # rows in slice order (after a column-permuting `submat`): positions by the sort-based lookup
            hit = self._positions(r, c)
            keep = np.ones(self.nnz, bool)
            keep[hit[hit >= 0]] = False
            m.value = int(keep.sum())
            ep[:m.value], vals[:m.value] = self.end_points[keep], self.values[keep]
            ind_ptr[0] = 0
            np.cumsum(np.bincount(self.edge_row_indices[keep], minlength=self.shape[0]), out=ind_ptr[1:])
        sup_rd, sup_cd = self._sup_rd, self._sup_cd
        if sup_rd is not None or sup_cd is not None:
            if degree_reducer is None:
                from ._native import dist as D
                if D.world() > 1:
                    raise ValueError("remove_edges_by_id on a rank-local block (override support degrees) needs a "
                                     "degree_reducer in a multi-rank run")
                degree_reducer = lambda a: a
            pos = self.edge_positions(np.asarray(node_pair_ids))
            hit = np.unique(pos[pos >= 0])                              # edges that exist, each counted once
            if sup_cd is not None:
                dec = np.bincount(self.end_points[hit], minlength=self.shape[1]).astype(np.int64)
                sup_cd = (sup_cd.astype(np.int64) - degree_reducer(dec)).astype(np.int32)
            if sup_rd is not None:
                dec = np.bincount(self.edge_row_indices[hit], minlength=self.shape[0]).astype(np.int64)
                sup_rd = (sup_rd.astype(np.int64) - degree_reducer(dec)).astype(np.int32)
        return CSRMat(ep[:m.value], ind_ptr, self.row_ids, self.col_ids, vals[:m.value], self.multi_link,
                      support_row_degrees=sup_rd, support_col_degrees=sup_cd)

    def check_consistency(self):
        rows = self.edge_row_indices.astype(np.int64) * self.shape[1] + self.end_points
        if np.unique(rows).size != rows.size:
            raise ValueError('Found duplicates in end_points')


class HeterGraph(object):
    """Both directions of every edge type (reference graph.py:758-1100).

    node_ids_dict: {key: ids};  csr_mat_dict: {(src_key, dst_key): CSRMat}; the reverse (dst_key, src_key) matrix
    is added as the transpose when absent.  `features` is kept as an opaque dict (shapes only matter here)."""

    def __init__(self, node_ids_dict, csr_mat_dict, features=None):
        self.node_ids_dict = {k: _i32(v) for k, v in node_ids_dict.items()}
        self.features = features if features is not None else dict()
        self.csr_mat_dict = dict(csr_mat_dict)
        for (a, b), m in list(self.csr_mat_dict.items()):
            if (b, a) not in self.csr_mat_dict:
                self.csr_mat_dict[(b, a)] = m.T
        self.meta_graph = dict()
        for (a, b) in self.csr_mat_dict:
            self.meta_graph.setdefault(a, dict())[b] = 1
        for k in self.node_ids_dict:
            self.meta_graph.setdefault(k, dict())

    def __getitem__(self, pair_keys):
        return self.csr_mat_dict[tuple(pair_keys)]

    def get_multi_link_structure(self):
        return {k: (None if m.multi_link is None else int(m.multi_link.size)) for k, m in self.csr_mat_dict.items()}

    def remove_edges_by_id(self, src_key, dst_key, node_pair_ids, degree_reducer=None):
        """reference graph.py:952-974: remove in BOTH directions; returns a new HeterGraph."""
        node_pair_ids = np.asarray(node_pair_ids)
        new = dict(self.csr_mat_dict)
        new[(src_key, dst_key)] = self.csr_mat_dict[(src_key, dst_key)].remove_edges_by_id(node_pair_ids, degree_reducer)
        new[(dst_key, src_key)] = self.csr_mat_dict[(dst_key, src_key)].remove_edges_by_id(node_pair_ids[::-1],
                                                                                          degree_reducer)
        return HeterGraph(self.node_ids_dict, new, self.features)

    def fetch_edges_by_id(self, src_key, dst_key, node_pair_ids):
        return self.csr_mat_dict[(src_key, dst_key)].fetch_edges_by_id(np.asarray(node_pair_ids))

    def node_id_to_ind(self, key, node_ids):
        return _IdMap(self.node_ids_dict[key])[node_ids]

    def sel_subgraph_by_id(self, key, node_ids):
        """reference graph.py:1001-1030: the graph restricted to the nodes `node_ids` of type `key` (the inductive
        setting trains on the graph of the training nodes only): rows of every (key, other) matrix and columns of every
        (other, key) matrix are selected, in the order of `node_ids`; the other node types keep all their nodes."""
        node_ids = _i32(node_ids)
        new = dict(self.csr_mat_dict)
        for (a, b), m in self.csr_mat_dict.items():
            if a == key and b == key:
                new[(a, b)] = m.submat_by_id(row_ids=node_ids, col_ids=node_ids)
            elif a == key:
                new[(a, b)] = m.submat_by_id(row_ids=node_ids)
            elif b == key:
                new[(a, b)] = m.submat_by_id(col_ids=node_ids)
        ids = dict(self.node_ids_dict)
        ids[key] = node_ids
        fea = dict(self.features)
        if fea.get(key) is not None:
            fea[key] = np.take(np.asarray(fea[key]), self.node_id_to_ind(key, node_ids), axis=0)
        return HeterGraph(ids, new, fea)

    def check_continous_node_ids(self):
        for key, ids in self.node_ids_dict.items():
            np.testing.assert_array_equal(ids, np.arange(ids.size, dtype=np.int32))

    def save(self, dir_name):
        """The reference's directory layout (graph.py:898-915, CSR keys :465-481), so either side reads the other's files:
        `meta_graph.json` = {key: {neighbour key: 1}}; `<key>.npz` = node_ids + features (float32; a zero-width matrix
        when the graph carries none); ONE `<k1>_<k2>_csr.npz` per node-type pair with row_ids, col_ids, values,
        end_points, ind_ptr and -- only when the matrix is multi-link -- multi_link."""
        os.makedirs(dir_name, exist_ok=True)
        with open(os.path.join(dir_name, 'meta_graph.json'), 'w') as f:
            json.dump({k: {n: 1 for n in v} for k, v in self.meta_graph.items()}, f)
        for key, ids in self.node_ids_dict.items():
            fea = self.features.get(key)
            fea = np.zeros((ids.size, 0), np.float32) if fea is None else np.asarray(fea, np.float32)
            np.savez_compressed(os.path.join(dir_name, '{}.npz'.format(key)), node_ids=ids, features=fea)
        written = set()
        for (a, b), m in self.csr_mat_dict.items():
            if (a, b) in written:
                continue
            written.update([(a, b), (b, a)])
            arrays = dict(row_ids=m.row_ids, col_ids=m.col_ids, values=m.values, end_points=m.end_points,
                          ind_ptr=m.ind_ptr)
            if m.multi_link is not None:
                arrays['multi_link'] = m.multi_link
            np.savez_compressed(os.path.join(dir_name, '{}_{}_csr.npz'.format(a, b)), **arrays)

    @classmethod
    def load(cls, dir_name, fea_normalize=False):
        """reference graph.py:1066-1100.  `fea_normalize` standardises every feature column (zero mean, unit variance;
        constant columns are only centred -- what sklearn's StandardScaler, which the reference calls, does)."""
        with open(os.path.join(dir_name, 'meta_graph.json')) as f:
            meta = json.load(f)
        node_ids, features, mats = dict(), dict(), dict()
        for a in meta:
            dat = np.load(os.path.join(dir_name, '{}.npz'.format(a)))
            node_ids[a] = dat['node_ids']
            fea = dat['features'] if 'features' in dat else None
            if fea is not None and fea.ndim == 2 and fea.shape[1] > 0:
                if fea_normalize:
                    fea = fea.astype(np.float64)
                    std = fea.std(axis=0)
                    fea = (fea - fea.mean(axis=0)) / np.where(std == 0, 1.0, std)
                features[a] = fea
            for b in meta[a]:
                if (a, b) in mats or (b, a) in mats:
                    continue
                found = [(x, y) for x, y in ((a, b), (b, a))
                         if os.path.exists(os.path.join(dir_name, '{}_{}_csr.npz'.format(x, y)))]
                if a == b:
                    found = found[:1]
                if len(found) != 1:
                    raise IOError("expected exactly one of {0}_{1}_csr.npz / {1}_{0}_csr.npz in {2}".format(a, b, dir_name))
                d = np.load(os.path.join(dir_name, '{}_{}_csr.npz'.format(*found[0])))
                ml = d['multi_link'] if 'multi_link' in d and d['multi_link'].size else None
                mats[found[0]] = CSRMat(d['end_points'], d['ind_ptr'], d['row_ids'], d['col_ids'], d['values'], ml)
        return cls(node_ids, mats, features)
