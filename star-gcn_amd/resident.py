"""Resident full-graph plan with per-batch edge removal ON THE DEVICE.

Reference training iteration (experiments/STAR-GCN.py:583-632): sample a rating batch, build
`train_graph.remove_edges_by_id(batch)` (two fresh CSRs, graph.py:952-974), recompute degrees and support, run
`gen_plan` top-down on the host (layers.py:260-337) and upload every plan array again (layers.py:366-377): O(E) host
work and O(E) PCIe traffic per iteration.

Here the plan of the WHOLE training graph is built once and stays in HBM.  Per batch only the following changes:
  * the weights of the resident MultiLinkPlans -- `sg_mask_edges_hip` rewrites them for the graph-minus-batch
    (new degrees, new support, removed edges weigh 0) in two passes over the edges on the device;
  * the small per-batch index plans of the heads (rating pairs, reconstruction rows, masked-input ids): O(batch + nodes).
The network then computes every node (not only the batch's 2-hop neighbourhood, which at MovieLens densities is almost
every node anyway); the rows the heads read are identical to the reference's on the reduced graph.
"""
import ctypes

import numpy as np
import torch

from . import _lib as L
from .model import PairPlan
from .plan import MultiLinkPlan, TakePlan


def _dev(a, device, dtype=np.int32):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(device)


class ResidentPlan(object):
    def __init__(self, net, graph, device="cuda", symm=None):
        self.net, self.graph, self.device = net, graph, torch.device(device)
        self.symm = net._norm_symm if symm is None else symm
        self.U, self.I = net._name_user, net._name_item
        U, I = self.U, self.I
        full = {k: graph.node_ids_dict[k] for k in graph.meta_graph}
        if getattr(net, "_recon_fea", False):
            raise NotImplementedError("resident plan with MODEL.RECON_FEA: the per-batch feature targets are not planned")
        self.plan = net.make_plan(graph, symm=self.symm, device=device, full_node_ids=full)
        self.csr = graph[U, I]
        m = self.csr
        if m._sup_rd is not None or m._sup_cd is not None:
            # a rank-local block normalises with GLOBAL degrees: the per-batch decrements of every rank would have to be
            # all-reduced inside sg_mask_edges_hip's degree pass; refuse instead of silently using rank-local degrees
            raise L.StarGCNError("ResidentPlan does not support rank-local graph blocks (override support degrees)")
        self.n_user, self.n_item, self.nnz = m.shape[0], m.shape[1], m.nnz
        eu, ei = m.edge_row_indices.astype(np.int64), m.end_points.astype(np.int64)
        R = int(m.multi_link.size) if m.multi_link is not None else 1
        level = np.searchsorted(m.multi_link, m.values) if m.multi_link is not None else np.zeros(m.nnz, np.int64)
        # slot of edge e in a (user, level, item)-ordered array and in an (item, level, user)-ordered one: the two edge
        # orders of the fused plans (c-order of users<-items = t-order of items<-users and vice versa)
        pos_a = np.empty(m.nnz, np.int32)
        pos_a[np.argsort(eu * R + level, kind="stable")] = np.arange(m.nnz, dtype=np.int32)
        pos_b = np.empty(m.nnz, np.int32)
        pos_b[np.lexsort((eu, level, ei))] = np.arange(m.nnz, dtype=np.int32)
        self._mplans, w_arrays, pos_arrays, transposed = [], [], [], []
        seen = set()
        for b in range(net._nblocks):
            for _prev, agg_args in self.plan["enc"][b]:
                for dst_key, (_bt, _st, plans) in agg_args.items():
                    for src_key, mp in plans.items():
                        if not isinstance(mp, MultiLinkPlan) or id(mp) in seen or {dst_key, src_key} != {U, I}:
                            continue
                        seen.add(id(mp))
                        if mp.R != R or mp.nnz != m.nnz:
                            raise L.StarGCNError("resident plan does not cover the whole graph")
                        user_rows = dst_key == U
                        pc, pt = (pos_a, pos_b) if user_rows else (pos_b, pos_a)
                        # the maps must reproduce the plan's own index arrays (guards the ordering argument above)
                        c_idx, t_idx = mp.c_idx.cpu().numpy(), mp.t_idx.cpu().numpy()
                        if not (np.array_equal(c_idx[pc], ei if user_rows else eu) and
                                np.array_equal(t_idx[pt], eu if user_rows else ei)):
                            raise L.StarGCNError("edge -> plan slot map is inconsistent with the plan")
                        self._mplans.append(mp)
                        w_arrays += [mp.c_w, mp.t_w]
                        pos_arrays += [pc, pt]
                        transposed += [0 if user_rows else 1] * 2
        if not w_arrays:
            raise L.StarGCNError("no multi-link plan over (%s, %s) found" % (U, I))
        self._pos_dev = {id(pos_a): _dev(pos_a, self.device), id(pos_b): _dev(pos_b, self.device)}
        self._w_arrays = w_arrays
        self._pos_arrays = [self._pos_dev[id(p)] for p in pos_arrays]
        self._transposed = transposed
        self._edge_row, self._edge_col = _dev(eu, self.device), _dev(ei, self.device)
        self._row_deg, self._col_deg = _dev(m.row_degrees, self.device), _dev(m.col_degrees, self.device)
        self._all_ids = {k: np.arange(graph.node_ids_dict[k].size, dtype=np.int32) for k in graph.meta_graph}
        def _checked(fn, what):
            def look(ids):
                ind = fn(ids)
                if ind.size and ind.min() < 0:
                    raise L.StarGCNError("unknown %s id in the batch (not a node of the resident graph)" % what)
                return ind
            return look
        self._id_maps = {U: _checked(m.row_id_to_ind, U), I: _checked(m.col_id_to_ind, I)}
        self.masked = 0

    # ---- device-side edge removal ------------------------------------------------------------------------------
    def mask_edges(self, edge_ids):
        """Remove the edges with these ids (CSR positions in graph[user, item]; tensor or array, -1 entries are ignored)
        from the resident plans; an empty list restores the full graph."""
        if not torch.is_tensor(edge_ids):
            edge_ids = _dev(np.asarray(edge_ids).reshape(-1), self.device)
        edge_ids = edge_ids.to(self.device, torch.int32).contiguous()
        lib = L.lib()
        groups = [(i, min(i + 16, len(self._w_arrays))) for i in range(0, len(self._w_arrays), 16)]
        ws, wsn = L.workspace(lib.sg_mask_edges_workspace_bytes(self.n_user, self.n_item, self.nnz), self.device)
        for lo, hi in groups:
            n = hi - lo
            wp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in self._w_arrays[lo:hi]])
            pp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in self._pos_arrays[lo:hi]])
            tr = (ctypes.c_int32 * n)(*self._transposed[lo:hi])
            L.check(lib.sg_mask_edges_hip(wp, pp, tr, n, L.ptr(self._edge_row), L.ptr(self._edge_col),
                                          L.ptr(self._row_deg), L.ptr(self._col_deg), L.ptr(edge_ids),
                                          edge_ids.numel(), self.n_user, self.n_item, self.nnz, int(bool(self.symm)),
                                          L.ptr(ws), wsn, L.stream_ptr()), "sg_mask_edges_hip")
        for mp in self._mplans:
            mp.refresh_rowsum()
        self.masked = int(edge_ids.numel())

    # ---- per-batch heads -----------------------------------------------------------------------------------------
    def set_batch(self, rating_node_pairs=None, edge_ids=None, embed_noise_dict=None, recon_node_ids_dict=None,
                  remove_batch_edges=True):
        """Prepare the resident plan for one iteration and return it (for `net.run`).  `edge_ids`: CSR positions of
        the batch's ratings when the caller knows them (the train sampler draws them directly); otherwise they are
        looked up from the pairs."""
        net, plan, dev = self.net, self.plan, self.device
        if remove_batch_edges and rating_node_pairs is not None:
            if edge_ids is None:
                edge_ids = self.csr.edge_positions(np.asarray(rating_node_pairs))
            self.mask_edges(edge_ids)
        elif self.masked:
            self.mask_edges(np.zeros(0, np.int32))
        n_out = {self.U: self.n_user, self.I: self.n_item}
        pair = recon_take = None       # every block reads the same rows: one plan, shared
        if rating_node_pairs is not None:
            pairs = np.asarray(rating_node_pairs)
            pair = PairPlan(self._id_maps[self.U](pairs[0]), self._id_maps[self.I](pairs[1]), self.n_user, self.n_item, dev)
        if recon_node_ids_dict is not None:
            recon_take = {k: TakePlan(self._id_maps[k](np.asarray(v)), n_out[k], dev)
                          for k, v in recon_node_ids_dict.items()}
        for b in range(net._nblocks):
            idx = plan["idx"][b]
            for k in ("pair", "recon_take", "rating", "recon"):
                idx.pop(k, None)
            if pair is not None:
                idx["pair"] = pair
            if recon_take is not None:
                idx["recon_take"] = recon_take
        plan["input"] = net._embed_plan({k: self.graph.node_ids_dict[k] for k in self._all_ids}, embed_noise_dict,
                                        embed_noise_dict is not None, dev)
        plan["gt"] = (net._embed_plan(recon_node_ids_dict, None, False, dev) if recon_node_ids_dict is not None else None)
        return plan

    # ---- the same, from DEVICE tensors only (device_sampler.DeviceBatchSampler) --------------------------------------
    def set_batch_device(self, batch, remove_batch_edges=True):
        """`batch` as produced by DeviceBatchSampler.next_batch(): sorted batch edge ids, their (user, item) row indices,
        noise arrays and reconstruction node indices -- all device tensors.  Masks the batch's edges, builds the pair plan
        of the rating head and the take plans of the masked input / reconstruction rows ON THE DEVICE; no host round
        trip, no synchronisation."""
        net, plan = self.net, self.plan
        if remove_batch_edges:
            self.mask_edges(batch["edge_ids"])
        elif self.masked:
            self.mask_edges(np.zeros(0, np.int32))
        n_out = {self.U: self.n_user, self.I: self.n_item}
        pair = PairPlan.from_sorted_device_pairs(batch["users"], batch["items"], self.n_user, self.n_item)
        recon_take = {k: TakePlan.from_device_unique(v, n_out[k]) for k, v in batch["recon"].items()}
        for b in range(net._nblocks):
            idx = plan["idx"][b]
            for k in ("pair", "recon_take", "rating", "recon"):
                idx.pop(k, None)
            idx["pair"] = pair
            idx["recon_take"] = recon_take
        # masked input: row i of the input is embedding noise[i] (= i) or zero (-1): every row taken at most once
        plan["input"] = {k: TakePlan.from_device_unique(v, n_out[k]) for k, v in batch["noise"].items()}
        plan["gt"] = {k: TakePlan.from_device_unique(v, n_out[k]) for k, v in batch["recon"].items()}
        return plan
