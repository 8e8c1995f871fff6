"""User-item rating graph RESIDENT IN HBM, with its aggregation plans built on the device.

Reference flow per training iteration (experiments/STAR-GCN.py:583-600, mxgraph/graph.py:677-748, layers.py:260-377):
numpy CSR on the host -> `sample_neighbors` -> `get_support` -> `multi_link_split` -> `merge_nodes` -> upload of every
index / support array.  For graphs that live on the device (the 1-GPU shard of BASELINE config 5 has 1.25 M x 1 M nodes
and 125 M ratings: 1.5 GB of CSR) that host detour is the bottleneck, so here the same structures are produced by the
native device builders of csrc/plan_build.hip (wave64 scan + stable radix sort):

    degrees            sg_count_indices_hip / row-pointer differences
    support            sg_get_support_hip          (graph_sampler.cpp:393-420: sqrt(1/d_row/d_col) | 1/d_row)
    item->user CSR     sg_build_transpose_hip      (graph.py:585-593 transposes with scipy)
    per-level plans    sg_multilink_fuse_csr_hip   (graph_sampler.cpp:277-376 + layers.py:260-337, full neighbourhood)

The integer arrays are bit-identical to what the host path (mxgraph.graph + plan.MultiLinkPlan) builds for the same
graph (tests/test_gpu_device_plan.py).  torch is used for allocation and for the synthetic edge generator only.
"""
import numpy as np
import torch

from . import _lib as L
from .plan import MultiLinkPlan, TransposePlan


def _i32(n, dev):
    return torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)


class DeviceBipartite(object):
    """user->item CSR (rows sorted by item, as scipy `tocsr()` leaves them in the reference ETL, datasets.py:116-121):
    ind_ptr (n_user+1), end_points (nnz), level (nnz) in [0, R) -- all int32 device tensors."""

    def __init__(self, ind_ptr, end_points, level, n_item, multi_link, name_user="user", name_item="movie",
                 item_degrees=None):
        """item_degrees: override for the support normalisation (a rank-local user block of a node-partitioned graph
        must use the GLOBAL item degrees, cf. CSRMat(support_col_degrees=...))."""
        self.ind_ptr, self.end_points, self.level = L.i32c(ind_ptr), L.i32c(end_points), L.i32c(level)
        self.device = self.ind_ptr.device
        self.n_user, self.n_item, self.nnz = int(self.ind_ptr.shape[0] - 1), int(n_item), int(self.end_points.shape[0])
        self.multi_link = np.asarray(multi_link, dtype=np.float32)
        self.R = int(self.multi_link.size)
        self.U, self.I = name_user, name_item
        lib, st = L.lib(), L.stream_ptr()
        self.edge_row = _i32(self.nnz, self.device)
        L.check(lib.sg_gen_row_indices_hip(L.ptr(self.edge_row), L.ptr(self.ind_ptr), self.n_user, self.nnz, st),
                "sg_gen_row_indices_hip")
        self.user_degrees = (self.ind_ptr[1:] - self.ind_ptr[:-1]).contiguous()
        if item_degrees is None:
            self.item_degrees = _i32(self.n_item, self.device)
            L.check(lib.sg_count_indices_hip(L.ptr(self.item_degrees), L.ptr(self.end_points), self.nnz, self.n_item, st),
                    "sg_count_indices_hip")
        else:
            self.item_degrees = L.i32c(item_degrees)
        self._t = None
        self._plans = dict()
        # the attributes model.Net reads from a HeterGraph
        self.node_ids_dict = {self.U: np.arange(self.n_user, dtype=np.int32), self.I: np.arange(self.n_item, dtype=np.int32)}
        self.meta_graph = {self.U: {self.I: 1}, self.I: {self.U: 1}}

    def with_item_degrees(self, item_degrees):
        """The same user block normalised with other (= the GLOBAL) item degrees: what a rank of a node-partitioned run
        uses once the per-rank degree counts have been summed over the ranks."""
        return DeviceBipartite(self.ind_ptr, self.end_points, self.level, self.n_item, self.multi_link, self.U, self.I,
                               item_degrees=item_degrees)

    @classmethod
    def from_host(cls, graph, name_user, name_item, device):
        """Upload the user->item CSRMat of a host HeterGraph (one copy of ind_ptr / end_points / values); the level
        of every edge is matched on the device (sg_level_index_hip: exact float equality, graph_sampler.cpp:300-311)."""
        m = graph[name_user, name_item]
        dev = torch.device(device)
        ind_ptr = torch.from_numpy(np.ascontiguousarray(m.ind_ptr)).to(dev)
        ep = torch.from_numpy(np.ascontiguousarray(m.end_points)).to(dev)
        vals = torch.from_numpy(np.ascontiguousarray(m.values, dtype=np.float32)).to(dev)
        ml = torch.from_numpy(np.ascontiguousarray(m.multi_link, dtype=np.float32)).to(dev)
        level = _i32(ep.numel(), dev)
        L.check(L.lib().sg_level_index_hip(L.ptr(level), L.ptr(vals), L.ptr(ml), ep.numel(), ml.numel(), L.stream_ptr()),
                "sg_level_index_hip")
        sup_cd = None if m._sup_cd is None else torch.from_numpy(np.ascontiguousarray(m._sup_cd)).to(dev)
        return cls(ind_ptr, ep, level[:ep.numel()] if ep.numel() else level[:0], m.shape[1], m.multi_link, name_user,
                   name_item, item_degrees=sup_cd)

    def get_multi_link_structure(self):
        return {(self.U, self.I): self.R, (self.I, self.U): self.R}

    def values(self):
        """rating value of every edge (fp32, CSR order)"""
        return torch.from_numpy(self.multi_link).to(self.device)[self.level.long()]

    def transposed(self):
        """item->user CSR: (t_indptr, users, edge ids) with the users of an item in increasing order."""
        if self._t is None:
            t = TransposePlan(self.end_points, self.ind_ptr, self.n_item, self.device)
            self._t = (t.t_indptr, t.t_seg, t.t_pos)
        return self._t

    def support(self, transposed=False, symm=True):
        """reference get_support of the user->item matrix (transposed=False) or of its transpose: the two differ in the
        order of the fp32 divisions, exactly like CSRMat.get_support of the two host matrices."""
        lib, st = L.lib(), L.stream_ptr()
        out = torch.empty(max(self.nnz, 1), dtype=torch.float32, device=self.device)
        if not transposed:
            L.check(lib.sg_get_support_hip(L.ptr(out), L.ptr(self.user_degrees), L.ptr(self.item_degrees),
                                           L.ptr(self.end_points), L.ptr(self.edge_row), self.nnz, int(bool(symm)), st),
                    "sg_get_support_hip")
            return out
        t_indptr, users, _eid = self.transposed()
        t_row = _i32(self.nnz, self.device)
        L.check(lib.sg_gen_row_indices_hip(L.ptr(t_row), L.ptr(t_indptr), self.n_item, self.nnz, st),
                "sg_gen_row_indices_hip")
        L.check(lib.sg_get_support_hip(L.ptr(out), L.ptr(self.item_degrees), L.ptr(self.user_degrees), L.ptr(users),
                                       L.ptr(t_row), self.nnz, int(bool(symm)), st), "sg_get_support_hip")
        return out

    def plan(self, dst_key, symm=True, with_from=False):
        """MultiLinkPlan of the full-neighbourhood aggregation INTO `dst_key` (built once, cached)."""
        key = (dst_key, bool(symm), bool(with_from))
        if key not in self._plans:
            if dst_key == self.U:
                p = MultiLinkPlan.from_device_csr(self.ind_ptr, self.end_points, self.level, self.support(False, symm),
                                                  self.n_item, self.R, with_from)
            else:
                t_indptr, users, eid = self.transposed()
                p = MultiLinkPlan.from_device_csr(t_indptr, users, self.level[eid.long()].contiguous(),
                                                  self.support(True, symm), self.n_user, self.R, with_from)
                if with_from:    # slots of the transposed matrix hold ITS edge numbering: map back to user->item ids
                    p.c_from, p.t_from = eid[p.c_from.long()].contiguous(), eid[p.t_from.long()].contiguous()
            self._plans[key] = p
        return self._plans[key]


def synthetic_device_graph(n_user, n_item, n_edges, n_levels, device, seed=0, name_user="user", name_item="movie",
                           item_seed=None):
    """MovieLens-SHAPED synthetic graph generated ON the device (SURVEY 8(d) recipe of star_gcn_amd.synthetic at sizes
    the host cannot plan in bench time): log-normal user / item propensities (sigma 1.0 / 1.5), no duplicate (user,
    item) pairs, every node keeps degree >= 1, ML-like level skew.  Returns a DeviceBipartite.
    item_seed: draw the item propensities from their own generator -- the ranks of a node-partitioned run generate
    their user blocks with different `seed`s against the SAME item popularity."""
    from .synthetic import level_probs, level_values
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(20240917 + int(seed))
    pu = torch.exp(torch.randn(n_user, generator=g, device=dev, dtype=torch.float64))
    gi = g
    if item_seed is not None:
        gi = torch.Generator(device=dev)
        gi.manual_seed(77001 + int(item_seed))
    pi = torch.exp(1.5 * torch.randn(n_item, generator=gi, device=dev, dtype=torch.float64))
    cu, ci = torch.cumsum(pu, 0), torch.cumsum(pi, 0)
    cu, ci = cu / cu[-1], ci / ci[-1]
    n_edges = int(min(n_edges, n_user * n_item // 2))
    keys = torch.zeros(0, dtype=torch.int64, device=dev)
    while keys.numel() < n_edges:
        m = int((n_edges - keys.numel()) * 1.25) + 1024
        u = torch.searchsorted(cu, torch.rand(m, generator=g, device=dev, dtype=torch.float64)).clamp_(max=n_user - 1)
        i = torch.searchsorted(ci, torch.rand(m, generator=g, device=dev, dtype=torch.float64)).clamp_(max=n_item - 1)
        keys = torch.unique(torch.cat([keys, u * n_item + i]))
        del u, i
    if keys.numel() > n_edges:     # thin uniformly at random to the requested count (keys stay sorted)
        drop = torch.randperm(keys.numel(), generator=g, device=dev)[:keys.numel() - n_edges]
        keep = torch.ones(keys.numel(), dtype=torch.bool, device=dev)
        keep[drop] = False
        keys = keys[keep]
        del drop, keep
    u, i = keys // n_item, keys % n_item
    seen_u = torch.zeros(n_user, dtype=torch.bool, device=dev)
    seen_i = torch.zeros(n_item, dtype=torch.bool, device=dev)
    seen_u[u] = True
    seen_i[i] = True
    miss_u, miss_i = (~seen_u).nonzero().view(-1), (~seen_i).nonzero().view(-1)
    if miss_u.numel() or miss_i.numel():     # degree >= 1 everywhere: one edge to a random partner
        eu = torch.cat([miss_u, torch.randint(0, n_user, (miss_i.numel(),), generator=g, device=dev)])
        ei = torch.cat([torch.randint(0, n_item, (miss_u.numel(),), generator=g, device=dev), miss_i])
        keys = torch.unique(torch.cat([keys, eu * n_item + ei]))
        u, i = keys // n_item, keys % n_item
    del keys
    counts = torch.bincount(u, minlength=n_user)
    ind_ptr = torch.zeros(n_user + 1, dtype=torch.int64, device=dev)
    torch.cumsum(counts, 0, out=ind_ptr[1:])
    cl = torch.cumsum(torch.from_numpy(level_probs(n_levels)).to(dev), 0)
    level = torch.searchsorted(cl / cl[-1], torch.rand(u.numel(), generator=g, device=dev, dtype=torch.float64))
    level = level.clamp_(max=n_levels - 1).to(torch.int32)
    return DeviceBipartite(ind_ptr.to(torch.int32), i.to(torch.int32), level, n_item, level_values(n_levels),
                           name_user, name_item)


# ---- device twins of the remaining host-only plan primitives -----------------------------------------
def unique_inverse_device(ids, max_id, return_counts=False):
    """unique_inverse / unique_cnt of the reference (graph_sampler.h:441-534) for a DEVICE id tensor with values in
    [0, max_id]: (unique ids in first-occurrence order, inverse[, counts]).  One 4-byte read-back sizes the result
    (sg_unique_inverse_hip keeps the count on the device for callers that do not need it on the host)."""
    ids = L.i32c(ids).view(-1)
    n, dev = int(ids.numel()), ids.device
    lib, st = L.lib(), L.stream_ptr()
    uniq, inv = _i32(n, dev), _i32(n, dev)
    counts = _i32(n, dev) if return_counts else None
    meta = torch.zeros(2, dtype=torch.int32, device=dev)
    ws, wsn = L.workspace(lib.sg_unique_inverse_workspace_bytes(n, int(max_id)), dev)
    L.check(lib.sg_unique_inverse_hip(L.ptr(uniq), L.ptr(inv), L.ptr(counts), L.ptr(meta), L.ptr(meta[1:]), L.ptr(ids), n,
                                      int(max_id), L.ptr(ws), wsn, st), "sg_unique_inverse_hip")
    m, bad = (int(x) for x in meta.tolist())
    if bad:
        raise L.StarGCNError("unique_inverse_device: an id lies outside [0, %d]" % int(max_id))
    out = (uniq[:m], inv[:n]) + ((counts[:m],) if return_counts else ())
    return out


def sample_fix_neighbor_device(ind_ptr, sel_indices, neighbor_num, seed):
    """random_sample_fix_neighbor of the reference (graph_sampler.cpp:742-779) on DEVICE tensors: (edge positions,
    dst_ind_ptr) -- bit-identical to the host sampler (sg_sample_fix_neighbor_cpu) for the same seed."""
    ind_ptr, sel = L.i32c(ind_ptr), L.i32c(sel_indices).view(-1)
    n, dev = int(sel.numel()), ind_ptr.device
    lib, st = L.lib(), L.stream_ptr()
    dst_ptr = _i32(n + 1, dev)
    ws, wsn = L.workspace(lib.sg_sample_fix_neighbor_workspace_bytes(n), dev)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    L.check(lib.sg_sample_fix_neighbor_hip(None, L.ptr(dst_ptr), L.ptr(ind_ptr), L.ptr(sel), n, int(neighbor_num), seed,
                                           L.ptr(ws), wsn, st), "sg_sample_fix_neighbor_hip")
    total = int(dst_ptr[n].item())
    sampled = _i32(total, dev)
    L.check(lib.sg_sample_fix_neighbor_hip(L.ptr(sampled), L.ptr(dst_ptr), L.ptr(ind_ptr), L.ptr(sel), n, int(neighbor_num),
                                           seed, L.ptr(ws), wsn, st), "sg_sample_fix_neighbor_hip")
    return sampled[:total], dst_ptr[:n + 1]
