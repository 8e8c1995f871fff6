"""STAR-GCN network on the MI355X-native hot path: counterpart of `Net` in reference
experiments/STAR-GCN.py:167-461 (embedding + masked input :264-300, stacked encoder blocks :196-219, decoder
`embed_maps` :222-246 / :440-459, rating projections + InnerProductLayer :249-261 / :428-438) and of the two
losses of its training loop (:610-628).

Host work is split from device work: `make_plan()` runs the reference's top-down planning once (numpy + native
helpers) and leaves every index structure resident in HBM; `run()` is pure device work, so a training step
re-uses the plan instead of rebuilding/uploading it (the reference rebuilds everything per iteration).

The hyper-parameters that the reference reads from its global yaml config are constructor arguments here.
"""
import numpy as np
import torch
from torch import nn

from . import _lib as L
from . import dist as D
from . import functional as SF
from . import ops
from .mxgraph import graph as G
from .mxgraph.layers import (Dense, HeterGCNLayer, InnerProductLayer, LayerDictionary, StackedHeterGCNLayers,
                             get_activation)
from .plan import SourcePartition, TakePlan, TransposePlan


class _PairTranspose(object):
    """t_indptr / t_pos / t_seg of a PairPlan, in the shape ops.seg_weighted_pool_bwd_data expects."""
    __slots__ = ("t_indptr", "t_pos", "t_seg", "seg_num", "nnz", "total_ind_num", "covered")


class PairPlan(object):
    """(user, item) row-index pairs grouped by user into a CSR so the per-pair inner product of the rating head is
    ONE `seg_take_k_corr` launch (and its gradients two gather launches) instead of two (#pairs, width) takes.
    Built by native code (sg_pair_plan_cpu: two counting sorts), uploaded with one copy."""

    def __init__(self, user_idx, item_idx, n_user, n_item, device):
        import ctypes
        u = np.ascontiguousarray(user_idx, dtype=np.int32).reshape(-1)
        i = np.ascontiguousarray(item_idx, dtype=np.int32).reshape(-1)
        n = u.size
        self.n_user, self.n_item, self.n_pairs = int(n_user), int(n_item), int(n)
        m = max(n, 1)
        order, inv, items, t_pos, t_seg = (np.zeros(m, np.int32) for _ in range(5))
        indptr, t_indptr = np.empty(self.n_user + 1, np.int32), np.empty(self.n_item + 1, np.int32)
        ident = ctypes.c_int32(0)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        L.check(L.lib().sg_pair_plan_cpu(vp(order), vp(inv), vp(indptr), vp(items), vp(t_indptr), vp(t_pos), vp(t_seg),
                                         ctypes.byref(ident), vp(u), vp(i), n, self.n_user, self.n_item),
                "sg_pair_plan_cpu")
        self.identity = bool(ident.value)
        from .plan import upload_packed
        extra = [] if self.identity else [inv, order]
        views = upload_packed([indptr, items, t_indptr, t_pos, t_seg] + extra, device)
        d_indptr, d_items, d_tip, d_tpos, d_tseg = views[:5]
        d_inv, d_order = (views[5], views[6]) if extra else (None, None)
        self.indptr, self.items = d_indptr, d_items
        self.inv_order = None if self.identity else d_inv
        self.order = None if self.identity else d_order
        tp = _PairTranspose()
        tp.t_indptr, tp.t_pos, tp.t_seg = d_tip, d_tpos, d_tseg
        tp.seg_num, tp.nnz, tp.total_ind_num, tp.covered = self.n_user, m, self.n_item, n
        self.tplan = tp


def _item_side_partition(self, width):
    """SourcePartition of the transposed pair plan, or None when the plain gather is the better launch: it pays when the
    gathered table (n_user x width) is well beyond one XCD's 4 MB L2, the output (n_item x width, x8 partial copies) is
    small next to the gathered bytes, there are enough pairs to fill the chip, and the plan is one that is kept across
    steps.  Built once, on first use."""
    sp = getattr(self, "_tparts", False)
    if sp is False:
        sp = None
        table, out = self.n_user * width * 4, self.n_item * width * 4
        # `reused`: only plans that live across steps (the full-batch head over a resident CSR) amortise the extra radix
        # sort; a per-iteration batch plan would pay it every time
        if (getattr(self, "reused", False) and self.n_pairs >= (1 << 20) and table >= (6 << 20)
                and 8 * out * 8 <= self.n_pairs * width * 4):
            tp = self.tplan
            sp = SourcePartition(tp.t_indptr, tp.t_seg, self.n_user, pos=tp.t_pos, parts=8)
        self._tparts = sp
    return sp


PairPlan.item_side_partition = _item_side_partition
PairPlan.cached_targets = lambda self, tag, y, pos: _cached_targets(self, tag, y, pos)


def _pair_plan_from_device_csr(cls, indptr, end_points, n_item):
    """PairPlan over ALL edges of a device-resident user->item CSR, in CSR order (the full-batch rating head of the
    benchmark): the CSR itself is the pair grouping, its transpose comes from the device builder -- nothing touches
    the host."""
    self = cls.__new__(cls)
    indptr, end_points = L.i32c(indptr), L.i32c(end_points)
    self.n_user, self.n_item, self.n_pairs = int(indptr.shape[0] - 1), int(n_item), int(end_points.shape[0])
    self.identity, self.inv_order, self.order = True, None, None
    self.reused = True               # built once per graph, used by every step
    self.indptr, self.items = indptr, end_points
    t = TransposePlan(end_points, indptr, self.n_item, end_points.device)
    tp = _PairTranspose()
    tp.t_indptr, tp.t_pos, tp.t_seg = t.t_indptr, t.t_pos, t.t_seg
    tp.seg_num, tp.nnz, tp.total_ind_num, tp.covered = self.n_user, max(self.n_pairs, 1), self.n_item, self.n_pairs
    self.tplan = tp
    return self


PairPlan.from_device_csr = classmethod(_pair_plan_from_device_csr)


def _pair_plan_from_sorted_pairs(cls, user_idx, item_idx, n_user, n_item):
    """PairPlan of a rating batch that is already ordered by (user, item) -- e.g. batch edge ids of the user->item CSR in
    ascending order -- from DEVICE index tensors: the grouping by user is a row pointer over the sorted users
    (sg_bounds_from_sorted_hip), the transpose comes from sg_build_transpose_hip.  Scores come back in this order."""
    self = cls.__new__(cls)
    user_idx, item_idx = L.i32c(user_idx), L.i32c(item_idx)
    self.n_user, self.n_item, self.n_pairs = int(n_user), int(n_item), int(user_idx.shape[0])
    self.identity, self.inv_order, self.order = True, None, None
    self.indptr = torch.empty(self.n_user + 1, dtype=torch.int32, device=user_idx.device)
    L.check(L.lib().sg_bounds_from_sorted_hip(L.ptr(self.indptr), L.ptr(user_idx), self.n_pairs, self.n_user, L.stream_ptr()),
            "sg_bounds_from_sorted_hip")
    self.items = item_idx if self.n_pairs else torch.zeros(1, dtype=torch.int32, device=user_idx.device)
    t = TransposePlan(self.items, self.indptr, self.n_item, user_idx.device)
    tp = _PairTranspose()
    tp.t_indptr, tp.t_pos, tp.t_seg = t.t_indptr, t.t_pos, t.t_seg
    tp.seg_num, tp.nnz, tp.total_ind_num, tp.covered = self.n_user, max(self.n_pairs, 1), self.n_item, self.n_pairs
    self.tplan = tp
    return self


PairPlan.from_sorted_device_pairs = classmethod(_pair_plan_from_sorted_pairs)


class _PairDot(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pu, pi, pp):
        pu, pi = L.f32c(pu), L.f32c(pi)
        ctx.pp = pp
        ctx.save_for_backward(pu, pi)
        return ops.seg_take_k_corr(pu.unsqueeze(0), pi.unsqueeze(0), pp.items, pp.indptr).view(-1)[:pp.n_pairs]

    @staticmethod
    def backward(ctx, g):
        pu, pi = ctx.saved_tensors
        pp = ctx.pp
        g = L.f32c(g).view(1, -1)
        if g.shape[1] < pp.items.numel():
            g = torch.nn.functional.pad(g, (0, pp.items.numel() - g.shape[1]))
        d_u = ops.seg_weighted_pool(pi.unsqueeze(0), g, pp.items, pp.indptr)[0] if ctx.needs_input_grad[0] else None
        d_i = None
        if ctx.needs_input_grad[1]:
            sp = pp.item_side_partition(pu.shape[1])
            if sp is not None:      # 10 M pairs read from an 18 MB table into 10 k rows: gather from L2-sized ranges
                d_i = torch.empty((pp.n_item, pu.shape[1]), dtype=torch.float32, device=pu.device)
                ops.gather_sum_parts(d_i, pu, sp, g, pu.shape[1])
            else:
                d_i = ops.seg_weighted_pool_bwd_data(g, pu.unsqueeze(0), pp.tplan, pp.n_item)[0]
        return d_u, d_i, None


class _PairL2(torch.autograd.Function):
    """loss = scale * sum_p 0.5 (<pu[user_p], pi[item_p]> - y_p)^2 in two gather passes that form their edge weights from
    the rows they load (sg_pair_l2_hip): forward = pass over the pairs grouped by user (loss + d pu up to the upstream
    gradient), backward = pass grouped by item (d pi).  Same value and gradients as pair_inner_product + l2_loss."""

    @staticmethod
    def forward(ctx, pu, pi, pp, y, scale):
        pu, pi, y = L.f32c(pu), L.f32c(pi), L.f32c(y).view(-1)
        if not pp.identity:
            y = y[pp.order.long()]                       # plan order = pairs grouped by user
        n = pp.items.numel()
        if y.numel() < n:                                # padding slots of an empty plan
            y = torch.nn.functional.pad(y, (0, n - y.numel()))
        du, loss = ops.pair_l2(pi, pu, y, pp.items, pp.indptr, pp.n_user, scale, loss_scale=0.5 * scale)
        ctx.pp, ctx.scale = pp, scale
        ctx.save_for_backward(pu, pi, y, du)
        return loss.view(())

    @staticmethod
    def backward(ctx, gout):
        pu, pi, y, du = ctx.saved_tensors
        pp = ctx.pp
        gout = L.f32c(gout).view(1)
        d_u = du * gout if ctx.needs_input_grad[0] else None
        d_i = None
        if ctx.needs_input_grad[1]:
            sp = pp.item_side_partition(pu.shape[1])
            if sp is not None:
                yt = pp.cached_targets("parts", y, sp.pos)
                d_i, _ = ops.pair_l2(pu, pi, yt, sp.src, sp.indptr, pp.n_item, ctx.scale, parts=sp.parts, scale_dev=gout)
            else:
                tp = pp.tplan
                yt = pp.cached_targets("t", y, tp.t_pos)
                d_i, _ = ops.pair_l2(pu, pi, yt, tp.t_seg, tp.t_indptr, pp.n_item, ctx.scale, scale_dev=gout)
        return d_u, d_i, None, None, None


def _cached_targets(self, tag, y, pos):
    """y in another edge order of this plan (y[pos]); kept while the same target tensor is passed again (the full-batch
    head scores the same ratings every step)."""
    key = (tag, y.data_ptr(), y._version, y.numel())
    c = getattr(self, "_y_cache", None)
    if c is None or c[0] != key:
        # the entry holds `y` itself: while it is cached its storage cannot be recycled for another target tensor that
        # would present the same (address, version, size) key
        c = (key, y[pos.long()].contiguous(), y)
        self._y_cache = c if getattr(self, "reused", False) else None
    return c[1]


def pair_l2_supported(width):
    return width % 4 == 0 and 4 <= width <= 128


def pair_l2_loss(pu, pi, pair_plan, y, scale):
    """scale * sum 0.5 (score - y)^2 over the plan's pairs without materialising the scores (see _PairL2)."""
    return _PairL2.apply(pu, pi, pair_plan, y, scale)


def pair_inner_product(pu, pi, pair_plan):
    """score[p] = <pu[user_p], pi[item_p]> in the ORIGINAL pair order."""
    s = _PairDot.apply(pu, pi, pair_plan)
    return s if pair_plan.identity else s[pair_plan.inv_order]


class Net(nn.Module):
    def __init__(self, graph, name_user, name_item, embed_units=64, agg_units=(250,), out_units=(75,), nblocks=2,
                 use_dae=True, use_recurrent=False, gcn_recurrent=False, activation="leaky", dropout=0.0,
                 agg_accum="sum", norm_symm=True, rating_mid_map=64, agg_order="auto", use_embed=True,
                 use_fea_proj=False, recon_fea=False, fea_mid_map=16, fea_units=16, features=None):
        """use_embed / use_fea_proj / recon_fea / fea_*: reference MODEL.USE_EMBED, MODEL.USE_FEA_PROJ, MODEL.RECON_FEA,
        FEA.MID_MAP, FEA.UNITS (STAR-GCN.py:176-192, 231-234): node features (`features` {key: (n, F)} or
        graph.features, e.g. datasets.LoadData's) pass through Dense(MID_MAP) - act - Dense(UNITS) and are concatenated
        to the (masked) embeddings at the input of every block; with recon_fea the decoder reconstructs the
        concatenation instead.  All shipped yamls leave USE_FEA_PROJ off."""
        super().__init__()
        assert use_embed or use_fea_proj
        assert use_embed or not use_dae, "the reconstruction target is the embedding table (STAR-GCN.py:359-363)"
        self._name_user, self._name_item = name_user, name_item
        self._nblocks, self._use_dae, self._use_recurrent = nblocks, use_dae, use_recurrent
        self._use_embed, self._use_fea_proj, self._recon_fea = use_embed, use_fea_proj, bool(recon_fea and use_fea_proj)
        self._norm_symm = norm_symm
        self._act_name = activation
        self.embed_layers = LayerDictionary()
        self._n_nodes = dict()
        for key, ids in graph.node_ids_dict.items():
            emb = nn.Embedding(ids.size, embed_units)
            nn.init.uniform_(emb.weight, -0.1, 0.1)          # mx.init.Uniform(0.1), reference STAR-GCN.py:180
            self.embed_layers[key] = emb
            self._n_nodes[key] = int(ids.size)
        self.encoders = nn.ModuleList()
        for _ in range(1 if use_recurrent else nblocks):
            enc = StackedHeterGCNLayers(recurrent_layer_num=len(agg_units) if gcn_recurrent else None)
            for au, ou in zip(agg_units, out_units):
                enc.add(HeterGCNLayer(meta_graph=graph.meta_graph,
                                      multi_link_structure=graph.get_multi_link_structure(),
                                      dropout_rate=dropout, agg_units=au, out_units=ou,
                                      source_keys=list(graph.meta_graph.keys()), agg_accum=agg_accum,
                                      agg_act=activation, out_act=activation, agg_order=agg_order))
                if gcn_recurrent:
                    break
            self.encoders.append(enc)
        if use_fea_proj:
            features = features if features is not None else getattr(graph, "features", None)
            self.fea_mappings = LayerDictionary()
            for key in graph.node_ids_dict:
                if features is None or features.get(key) is None:
                    raise ValueError("use_fea_proj needs node features for %r" % key)
                fea = torch.as_tensor(np.ascontiguousarray(features[key], dtype=np.float32))
                assert fea.shape[0] == self._n_nodes[key]
                self.register_buffer("fea_" + key, fea)
                self.fea_mappings[key] = nn.Sequential(Dense(fea_mid_map, activation=activation), Dense(fea_units))
        if use_dae:
            out_emb = embed_units + (fea_units if self._recon_fea else 0)     # STAR-GCN.py:231-234
            self.embed_maps = nn.ModuleList()
            for _ in range(1 if use_recurrent else nblocks):
                m = LayerDictionary()
                for key in graph.meta_graph:
                    m[key] = nn.Sequential(Dense(out_emb, activation=activation), Dense(out_emb))
                self.embed_maps.append(m)
        nproj = 1 if use_recurrent else nblocks
        self.rating_user_projs = nn.ModuleList([Dense(rating_mid_map) for _ in range(nproj)])
        self.rating_item_projs = nn.ModuleList([Dense(rating_mid_map) for _ in range(nproj)])
        self.gen_ratings = InnerProductLayer()
        self.pair_partition = None   # dist.NodePartition when node-partitioned across GPUs

    def local_region_parameters(self):
        """Parameters that receive PARTIAL gradients in a node-partitioned run (see dist.py): everything that
        touches rank-local user rows or rank-local edges.  Item-side modules downstream of the all-reduce and the
        embedding tables (item: replicated with total gradients; user: row-sharded) are excluded."""
        part = self.pair_partition
        rep = set() if part is None else part.replicated_keys
        skip = set()
        for key in rep:
            for enc in self.encoders:
                for layer in enc._blocks:
                    if key in layer._out_fcs:
                        skip.update(id(p) for p in layer._out_fcs[key].parameters())
            if self._use_dae:
                for m in self.embed_maps:
                    skip.update(id(p) for p in m[key].parameters())
            if self._use_fea_proj:      # like the replicated embedding table: every rank already holds the total gradient
                skip.update(id(p) for p in self.fea_mappings[key].parameters())
        if self._name_item in rep:
            skip.update(id(p) for p in self.rating_item_projs.parameters())
        if self._name_user in rep:
            skip.update(id(p) for p in self.rating_user_projs.parameters())
        skip.update(id(p) for p in self.embed_layers.parameters())
        return [p for p in self.parameters() if id(p) not in skip]

    # ---- embeddings (reference STAR-GCN.py:264-300) ---------------------------------------------------
    def _embed_plan(self, node_ids_dict, embed_noise_dict, use_mask, device):
        plans = dict()
        for key, ids in node_ids_dict.items():
            ids = np.asarray(ids, np.int32)
            if use_mask:
                ids = np.asarray(embed_noise_dict[key], np.int32)[ids]   # -1 = zero-mask
            plans[key] = TakePlan(ids, self._n_nodes[key], device)
        return plans

    def get_embed(self, embed_plans):
        return {key: SF.take_rows(self.embed_layers[key].weight, p) for key, p in embed_plans.items()}

    def _fea_plan(self, node_ids_dict, device):
        """features are taken by node id, never masked (reference get_feature, STAR-GCN.py:302-309)"""
        return {key: TakePlan(np.asarray(ids, np.int32), self._n_nodes[key], device) for key, ids in node_ids_dict.items()}

    def get_feature(self, fea_plans):
        return {key: self.fea_mappings[key](SF.take_rows(getattr(self, "fea_" + key), p)) for key, p in fea_plans.items()}

    @staticmethod
    def _concat(a, b):
        return {key: torch.cat([a[key], b[key]], dim=1) for key in a}

    # ---- host-side planning (reference STAR-GCN.py:373-397) -------------------------------------------
    def make_plan(self, graph, rating_node_pairs=None, embed_noise_dict=None, recon_node_ids_dict=None,
                  graph_sampler_args=None, symm=None, device="cuda", full_node_ids=None):
        """full_node_ids {key: ids}: node types whose EVERY node is computed, in this order, at every block and depth
        (replicated types of a node-partitioned run; all types for a resident full-graph plan, star_gcn_amd.resident)."""
        symm = self._norm_symm if symm is None else symm
        if rating_node_pairs is None and recon_node_ids_dict is None and full_node_ids is None:
            raise NotImplementedError
        nb = self._nblocks
        plan = dict(enc=[None] * nb, idx=[None] * nb, device=torch.device(device))
        req = dict()
        full = full_node_ids
        if full is None and self.pair_partition is not None:   # replicated node types: every node, same order, on every rank
            full = {k: graph.node_ids_dict[k] for k in graph.meta_graph if k in self.pair_partition.replicated_keys}
        for b in range(nb - 1, -1, -1):
            parts, names = [], []
            if full is not None:
                parts.append(full)
                names.append("full")
            if rating_node_pairs is not None:
                parts.append({self._name_user: rating_node_pairs[0], self._name_item: rating_node_pairs[1]})
                names.append("rating")
            if recon_node_ids_dict is not None:
                parts.append(recon_node_ids_dict)
                names.append("recon")
            parts.append(req)
            names.append("req")
            uniq, idx_l = G.merge_node_ids_dict(parts)
            plan["idx"][b] = dict(zip(names, idx_l))
            if self._use_fea_proj and not self._recon_fea and b < nb - 1 and self._use_dae:
                plan["idx"][b]["req_fea"] = self._fea_plan(req, device)    # block_req_node_ids_dict, STAR-GCN.py:455-459
            enc = self.encoders[0] if self._use_recurrent else self.encoders[b]
            req, plan["enc"][b] = enc.gen_plan(graph=graph, sel_node_ids_dict=uniq,
                                               graph_sampler_args=graph_sampler_args, symm=symm, device=device,
                                               full_node_ids=full)
            plan["idx"][b]["n_out"] = {k: int(v.shape[0]) for k, v in uniq.items()}
        plan["input"] = self._embed_plan(req, embed_noise_dict, embed_noise_dict is not None, device)
        plan["gt"] = (self._embed_plan(recon_node_ids_dict, None, False, device)
                      if recon_node_ids_dict is not None else None)
        if self._use_fea_proj:
            plan["input_fea"] = self._fea_plan(req, device)
            plan["gt_fea"] = (self._fea_plan(recon_node_ids_dict, device)
                              if recon_node_ids_dict is not None and self._recon_fea else None)
        for b in range(nb):      # resident index plans for the heads
            idx, n_out = plan["idx"][b], plan["idx"][b]["n_out"]
            if "rating" in idx:
                idx["pair"] = PairPlan(idx["rating"][self._name_user], idx["rating"][self._name_item],
                                       n_out[self._name_user], n_out[self._name_item], device)
            if "recon" in idx:
                idx["recon_take"] = {k: TakePlan(v, n_out[k], device) for k, v in idx["recon"].items()}
            if b < nb - 1 and self._use_dae:
                idx["req_take"] = {k: TakePlan(v, n_out[k], device) for k, v in idx["req"].items()}
        return plan

    def make_plan_device(self, dgraph, symm=None, rating_head=True):
        """Full-graph plan for a graph that is RESIDENT ON THE DEVICE (device_graph.DeviceBipartite): every node of
        both types is computed in natural order at every depth, all index structure is built by the native device
        builders, nothing is copied to or from the host.  The rating head scores ALL ratings of the graph in CSR order
        (the benchmark's full-batch step)."""
        symm = self._norm_symm if symm is None else symm
        U, I = self._name_user, self._name_item
        n = {U: dgraph.n_user, I: dgraph.n_item}
        plans = {U: dgraph.plan(U, symm), I: dgraph.plan(I, symm)}
        ident = {k: TakePlan.identity_plan(n[k]) for k in (U, I)}
        plan = dict(enc=[None] * self._nblocks, idx=[None] * self._nblocks, device=dgraph.device)
        for b in range(self._nblocks):
            enc = self.encoders[0] if self._use_recurrent else self.encoders[b]
            cp = []
            for _depth in range(len(enc)):
                agg_args = {U: [ident[U], None, {I: plans[U]}], I: [ident[I], None, {U: plans[I]}]}
                cp.append([dict(dgraph.node_ids_dict), agg_args])
            plan["enc"][b] = cp
            idx = {"n_out": dict(n)}
            if rating_head:
                idx["pair"] = PairPlan.from_device_csr(dgraph.ind_ptr, dgraph.end_points, dgraph.n_item)
            if b < self._nblocks - 1 and self._use_dae:
                idx["req_take"] = dict(ident)
                if self._use_fea_proj and not self._recon_fea:
                    idx["req_fea"] = dict(ident)
            plan["idx"][b] = idx
        plan["input"] = dict(ident)
        plan["gt"] = None
        if self._use_fea_proj:
            plan["input_fea"], plan["gt_fea"] = dict(ident), None
        return plan

    # ---- device work (reference STAR-GCN.py:399-461) --------------------------------------------------
    def run(self, plan, rating_targets=None, rating_scale=None):
        """rating_targets (standardised ratings of the plan's pairs) + rating_scale: return per-block rating LOSSES
        scale * sum 0.5 (score - y)^2 in place of the scores (fused rating head, sg_pair_l2_hip) -- what training needs;
        evaluation keeps the scores."""
        pred_ratings, pred_embeddings = [], []
        gt = self.get_embed(plan["gt"]) if plan["gt"] is not None else dict()
        if gt and self._recon_fea:
            gt = self._concat(gt, self.get_feature(plan["gt_fea"]))
        x = self.get_embed(plan["input"]) if self._use_embed else None
        if self._use_fea_proj:      # STAR-GCN.py:405-413
            fea = self.get_feature(plan["input_fea"])
            x = self._concat(x, fea) if x is not None else fea
        for b in range(self._nblocks):
            enc = self.encoders[0] if self._use_recurrent else self.encoders[b]
            out = enc.heter_sage(x, plan["enc"][b])
            idx = plan["idx"][b]
            k = 0 if self._use_recurrent else b
            if "pair" in idx:   # Dense is row-wise, so project the unique rows first, then pair them up
                pu = self.rating_user_projs[k](out[self._name_user])
                pi = self.rating_item_projs[k](out[self._name_item])
                if self.pair_partition is not None:   # replicated projections meet rank-local rating pairs
                    if self._name_item in self.pair_partition.replicated_keys:
                        pi = D.copy_to_local(pi)
                    if self._name_user in self.pair_partition.replicated_keys:
                        pu = D.copy_to_local(pu)
                if rating_targets is not None and pair_l2_supported(pu.shape[1]):
                    pred_ratings.append(pair_l2_loss(pu, pi, idx["pair"], rating_targets, rating_scale))
                elif rating_targets is not None:
                    pred_ratings.append(SF.l2_loss(pair_inner_product(pu, pi, idx["pair"]).view(-1),
                                                   rating_targets.view(-1), rating_scale))
                else:
                    pred_ratings.append(pair_inner_product(pu, pi, idx["pair"]).view(-1, 1))
            if "recon_take" in idx and self._use_dae:
                m = self.embed_maps[k]
                pred_embeddings.append({key: m[key](SF.take_rows(out[key], tp))
                                        for key, tp in idx["recon_take"].items()})
            if b < self._nblocks - 1 and self._use_dae:
                m = self.embed_maps[k]
                x = {key: m[key](SF.take_rows(out[key], tp)) for key, tp in idx["req_take"].items()}
                if "req_fea" in idx:
                    x = self._concat(x, self.get_feature(idx["req_fea"]))
        return pred_ratings, pred_embeddings, gt

    def forward(self, graph, rating_node_pairs=None, embed_noise_dict=None, recon_node_ids_dict=None,
                graph_sampler_args=None, symm=None, device="cuda", rating_targets=None, rating_scale=None):
        return self.run(self.make_plan(graph, rating_node_pairs, embed_noise_dict, recon_node_ids_dict,
                                       graph_sampler_args, symm, device), rating_targets, rating_scale)


def star_gcn_loss(pred_ratings, pred_embeddings, gt_embeddings, gt_ratings_std, recon_lambda=0.1):
    """reference STAR-GCN.py:610-628: sum over blocks of L2Loss(pred, standardised rating).mean()  [= 0.5*(x-y)^2]
    + recon_lambda * sum over blocks and keys of mean_nodes( sum_c (gt - pred)^2 ); the target is NOT detached."""
    loss = 0.0
    for pr in pred_ratings:
        if pr.dim() == 0:       # Net.run(..., rating_targets=..., rating_scale=1 / #pairs) already returned this term
            loss = loss + pr
        else:
            loss = loss + SF.l2_loss(pr.view(-1), gt_ratings_std.view(-1), 1.0 / max(pr.numel(), 1))
    for block in pred_embeddings:
        for key, pred in block.items():
            loss = loss + recon_lambda * ((gt_embeddings[key] - pred) ** 2).sum(dim=1).mean()
    return loss


def deterministic_init(net, seed=1234, embedding_rows=None):
    """Re-initialise every (materialised) parameter from a CPU generator keyed by (seed, parameter NAME): the values
    depend neither on construction / first-use order, nor on lazily inferred shapes of OTHER parameters, nor on the
    number of rank-local rows.  Replicated parameters of a node-partitioned run are therefore bit-identical on every
    rank and equal to the single-process model's.

    embedding_rows {node key: (lo, hi, n_global)}: this rank holds rows [lo, hi) of a global embedding table.
    Weights: Xavier-in uniform (reference STAR-GCN.py:548); biases: zeros; embeddings: U(-0.1, 0.1) (:180)."""
    import zlib
    emb_names = {"embed_layers._layers.%d.weight" % i: k for k, i in net.embed_layers._key2idx.items()}
    with torch.no_grad():
        for name, p in net.named_parameters():
            if isinstance(p, nn.UninitializedParameter):
                raise RuntimeError("deterministic_init needs materialised parameters (run one forward first): " + name)
            g = torch.Generator().manual_seed((int(seed) << 32) ^ zlib.crc32(name.encode()))
            if name in emb_names:
                lo, hi, n_glob = (0, p.shape[0], p.shape[0])
                if embedding_rows and emb_names[name] in embedding_rows:
                    lo, hi, n_glob = embedding_rows[emb_names[name]]
                full = torch.rand(n_glob, p.shape[1], generator=g) * 0.2 - 0.1
                p.copy_(full[lo:hi].to(p.device))
            elif p.dim() > 1:
                s = (3.0 / max(p.shape[1], 1)) ** 0.5
                p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * s).to(p.device))
            else:
                p.zero_()


def calibrate_output_scale(net, run_forward, target=1.0, reduce=None, run_scores=None):
    """Layer-sequential scale calibration of a freshly initialised benchmark network (LSUV-style, Mishkin & Matas
    2015, without the orthogonalisation): with the reference's init (embeddings U(-0.1, 0.1), Xavier-in weights, zero
    biases; STAR-GCN.py:180, 548) the symmetric normalisation sqrt(1/d_u/d_i) shrinks every aggregation of a
    MovieLens-shaped graph by 1-2 orders of magnitude, the scores of a 2-layer network come out ~1e-6 and the loss is
    0.5 var(y) to seven digits whatever the network computes.  Going through the layers in order, the per-level
    aggregator weights of each node type are rescaled so that the layer's OUTPUT has root-mean-square `target`; last,
    the rating projections are rescaled so that the scores have unit scale (each projection to rms width^-1/4).
    Biases are zero and LeakyReLU is positively homogeneous, so one forward pass per stage gives the exact factor.

    run_forward(): one forward pass of the network (no gradients needed); run_scores() -> the score of every rating
    pair of the plan (rank-local pairs in a partitioned run).  reduce(t) -> t summed over the ranks of a
    node-partitioned run (rank-local node types need the global mean square; replicated types are unaffected because
    numerator and denominator grow alike).  Returns the measured rms values, stage by stage (before rescaling)."""
    report = []

    def rms_of(mods):
        acc = dict()
        hooks = [m.register_forward_hook(lambda _m, _i, o, k=k: acc.__setitem__(
            k, torch.stack([o.detach().double().pow(2).sum(), torch.tensor(float(o.numel()), dtype=torch.float64,
                                                                           device=o.device)])))
                 for k, m in mods.items()]
        try:
            with torch.no_grad():
                run_forward()
        finally:
            for h in hooks:
                h.remove()
        out = dict()
        for k in mods:      # the plan's key order: identical on every rank
            t = acc[k] if reduce is None else reduce(acc[k])
            out[k] = float((t[0] / t[1]).sqrt())
        return out

    with torch.no_grad():
        for enc in net.encoders:
            for depth, layer in enumerate(enc._blocks):
                keys = list(layer._out_fcs.keys())
                rms = rms_of({k: layer._out_fcs[k] for k in keys})
                report.append({"layer%d.%s" % (depth, k): v for k, v in rms.items()})
                for k in keys:
                    f = target / max(rms[k], 1e-30)
                    for nb in layer._meta_graph[k]:
                        agg = layer.aggregators[(k, nb)]
                        agg = getattr(agg, "_agg", agg)
                        for r in range(agg._num_links):
                            getattr(agg, "weight%d" % r).mul_(f)
        projs = {"user": net.rating_user_projs[0], "item": net.rating_item_projs[0]}
        rms = rms_of(projs)
        report.append({"proj.%s" % k: v for k, v in rms.items()})
        for k, m in projs.items():
            m.weight.mul_(m.weight.shape[0] ** -0.25 / max(rms[k], 1e-30))
        if run_scores is not None:      # the two projections are correlated: set the SCORES' rms to 1 directly
            sc = run_scores().detach().double()
            t = torch.stack([sc.pow(2).sum(), torch.tensor(float(sc.numel()), dtype=torch.float64, device=sc.device)])
            t = t if reduce is None else reduce(t)
            srms = float((t[0] / t[1]).sqrt())
            report.append({"scores": srms})
            for m in projs.values():
                m.weight.mul_(max(srms, 1e-30) ** -0.5)
    return report
