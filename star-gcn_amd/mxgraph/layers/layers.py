"""Heterogeneous graph-conv layers with the API of reference mxgraph/layers/layers.py:
HeterGCNLayer (:42-208), InnerProductLayer (:210-222), StackedHeterGCNLayers.gen_plan / heter_sage (:224-385).

Differences that matter for speed, not for results:
  * `gen_plan` builds, per (depth, node type, neighbour type), ONE resident `MultiLinkPlan` (fused CSR + transpose on
    the device) instead of lists of numpy arrays; `heter_sage` therefore uploads nothing (the reference re-uploads
    every index/support/edge-value array on every call, layers.py:366-377, and the edge values are never used).
  * row `take`s go through `functional.take_rows` (coalesced copy forward, atomic-free segment-sum backward).
  * the reference's debugging `print(...)`/`input()` inside gen_plan (layers.py:319-320) is not reproduced.
"""
import numpy as np
import torch
from torch import nn

from .. import graph as G
from .._native import dist as D
from .._native import functional as SF
from .._native import plan as P
from .aggregators import GCNAggregator, MultiLinkGCNAggregator
from .common import Activation, Dense, LayerDictionary, get_activation


class HeterGCNLayer(nn.Module):
    def __init__(self, meta_graph, multi_link_structure, agg_units, out_units, source_keys=None, dropout_rate=0.0,
                 agg_ordinal_sharing=False, agg_accum='stack', agg_act='relu', layer_accum='stack', accum_self=False,
                 out_act=None, agg_order='auto'):
        super().__init__()
        self._meta_graph = meta_graph
        source_keys = list(meta_graph.keys()) if source_keys is None else list(source_keys)
        self._source_keys = source_keys
        if not isinstance(out_units, dict):
            out_units = {k: out_units for k in source_keys}
        if not isinstance(agg_units, dict):
            agg_units = {k: agg_units for k in meta_graph}
        self._layer_accum, self._accum_self = layer_accum, accum_self
        self.partition = None   # set to a dist.NodePartition for multi-GPU node-partitioned execution
        self._out_act = get_activation(out_act)
        self.dropout = nn.Dropout(dropout_rate)
        self._aggregators = LayerDictionary()
        for src_key in source_keys:
            for dst_key in meta_graph[src_key]:
                nlinks = multi_link_structure[(src_key, dst_key)]
                if nlinks is None:
                    agg = GCNAggregator(units=agg_units[src_key], act=agg_act, dropout_rate=dropout_rate)
                else:
                    agg = MultiLinkGCNAggregator(units=agg_units[src_key], num_links=nlinks, act=agg_act,
                                                 dropout_rate=dropout_rate, ordinal_sharing=agg_ordinal_sharing,
                                                 accum=agg_accum, order=agg_order)
                self._aggregators[(src_key, dst_key)] = agg
        self._out_fcs = LayerDictionary()
        for key, units in out_units.items():
            if units is not None:   # output activation fused into the GEMM epilogue
                self._out_fcs[key] = Dense(units, activation=self._out_act)
        if accum_self:
            self._self_fcs = LayerDictionary()
            for key, units in out_units.items():
                if units is not None:
                    self._self_fcs[key] = nn.Sequential(nn.Dropout(dropout_rate), Dense(units),
                                                        nn.Dropout(dropout_rate))

    @property
    def aggregators(self):
        return self._aggregators

    def forward_single(self, key, base_feas, neighbor_data):
        """neighbor_data: {neighbor_key: (feas, end_points, edge_values, indptr, support)} as in the reference, or
        {neighbor_key: (feas, MultiLinkPlan)}."""
        return self.forward_finish(key, base_feas, self.forward_start(key, neighbor_data))

    def forward_start(self, key, neighbor_data, grad_pending=None):
        """First half of `forward_single`: run the aggregators of node type `key`.  In a node-partitioned run
        (dist.py) an aggregate over rank-local sources that is destined to a replicated node type is a PARTIAL sum:
        its all-reduce is launched here on the communication stream and only awaited in `forward_finish`, so the
        caller can run rank-local work in between.  grad_pending {neighbor_key: dist pending handle}: replicated
        neighbour features whose gradient all-reduce is awaited where they were produced (dist.grad_wait)."""
        started = []
        part = self.partition
        for dst_key in self._meta_graph[key]:
            item = neighbor_data[dst_key]
            agg = self._aggregators[(key, dst_key)]
            feas = item[0]
            defer = False
            if part is not None:   # node-partitioned run (dist.py): wrap the two kinds of crossing
                if part.crossing_in(key, dst_key):
                    feas = D.copy_to_local_async(feas, None if grad_pending is None else grad_pending.get(dst_key),
                                                 owned=True)   # gradient = the aggregator's fresh dx (or its dropout's)
                defer = part.crossing_out(key, dst_key)
            if len(item) == 2:
                out = agg(feas, item[1], defer_act=defer)
            else:
                _f, end_points, _edge_values, indptr, support = item
                out = agg(feas, end_points, indptr, support, defer_act=defer)
            pending = None
            if defer:   # partial sums over this rank's sources -> all-reduce (in flight), THEN the aggregator activation
                out, pending = D.reduce_start(out, owned=True)   # fresh, unsaved pre-activation partial (functional.py)
            started.append((out, pending, agg, defer))
        return started

    def forward_finish(self, key, base_feas, started):
        outs = []
        replicated = self.partition is not None and key in self.partition.replicated_keys
        for out, pending, agg, defer in started:
            if defer:
                out = D.reduce_wait(out, pending)
                act = agg.activation        # after the sum over ranks; one native pass each way for the built-in activations
                out = act(out, native=True) if isinstance(act, Activation) else act(out)
            outs.append(D.replicated_dropout(out, self.dropout.p, self.training) if replicated else self.dropout(out))
        if self._accum_self:
            outs.append(self._self_fcs[key](base_feas))
        if len(outs) == 1:
            out = outs[0]
        elif self._layer_accum == 'stack':
            out = torch.cat(outs, dim=1)
        elif self._layer_accum == 'sum':
            out = torch.stack(outs, dim=0).sum(dim=0)
        else:
            raise NotImplementedError(self._layer_accum)
        if key in self._out_fcs:
            return self._out_fcs[key](out)
        return self._out_act(out)

    def forward(self, base_feas, neighbor_data):
        return {key: self.forward_single(key, feas, neighbor_data[key]) for key, feas in base_feas.items()}


class InnerProductLayer(nn.Module):
    """score = sum_c mid(data1) * mid(data2)   (reference layers.py:210-222; the SAME mid map on both sides)."""

    def __init__(self, mid_units=None):
        super().__init__()
        self._mid_units = mid_units
        if mid_units is not None:
            self._mid_map = Dense(mid_units)

    def forward(self, data1, data2):
        if self._mid_units is not None:
            data1, data2 = self._mid_map(data1), self._mid_map(data2)
        return (data1 * data2).sum(dim=1, keepdim=True)


class StackedHeterGCNLayers(nn.Module):
    """A stack of HeterGCNLayers sharing one plan (or ONE layer applied `recurrent_layer_num` times)."""

    def __init__(self, recurrent_layer_num=None):
        super().__init__()
        self._recurrent_layer_num = recurrent_layer_num
        self._blocks = nn.ModuleList()

    def __len__(self):
        if self._recurrent_layer_num is None:
            return len(self._blocks)
        return 0 if len(self._blocks) == 0 else self._recurrent_layer_num

    def __getitem__(self, key):
        if self._recurrent_layer_num is not None:
            if key < self._recurrent_layer_num:
                return self._blocks[0]
            raise KeyError('{} is out of range. Layer number={}'.format(key, len(self)))
        return self._blocks[key]

    def add(self, *blocks):
        if self._recurrent_layer_num is not None and (len(self._blocks) == 1 or len(blocks) > 1):
            raise ValueError('Only a single block can be added when the recurrent flag is on')
        for block in blocks:
            assert isinstance(block, HeterGCNLayer)
            self._blocks.append(block)

    # ------------------------------------------------------------------------------------------------
    def gen_plan(self, graph, sel_node_ids_dict, graph_sampler_args=None, symm=True, device=None,
                 full_node_ids=None):
        """Top-down plan construction (reference layers.py:260-337).

        full_node_ids {key: all ids}: node types that are REPLICATED in a node-partitioned run keep every node, in
        this order, at every depth, so the all-reduced tensors have the same shape and row order on every rank.

        Returns (req_node_ids_dict, computing_plan); computing_plan[depth] = [prev_level_ids_dict, agg_args_dict]
        with agg_args_dict[src_key] = [uniq_sel_node_inds, sel_node_idx, {dst_key: MultiLinkPlan}] -- the
        reference keeps [end_points, edge_values, ind_ptr, support] lists in that slot."""
        device = torch.device('cuda') if device is None else torch.device(device)
        if graph_sampler_args is None:
            graph_sampler_args = {}
        computing_plan = [None] * len(self)
        for depth in range(len(self) - 1, -1, -1):
            agg_args, nbr_ids, src_ids = dict(), dict(), dict()
            raw = dict()
            for src_key, sel_node_ids in sel_node_ids_dict.items():
                if depth == len(self) - 1:
                    uniq_ids, sel_idx = G.unordered_unique(sel_node_ids, return_inverse=True)
                else:
                    uniq_ids, sel_idx = np.asarray(sel_node_ids, dtype=np.int32), None
                agg_args[src_key] = [uniq_ids, sel_idx, dict()]
                src_ids[src_key] = uniq_ids
                for dst_key in graph.meta_graph[src_key]:
                    use_ml = self[depth].aggregators[(src_key, dst_key)].use_multi_link
                    ep_ids, _vals, ind_ptr, support = graph[src_key, dst_key].sample_neighbors(
                        src_ids=uniq_ids, symm=symm, use_multi_link=use_ml,
                        num_neighbors=graph_sampler_args.get((src_key, dst_key), -1))
                    if not use_ml:
                        ep_ids, ind_ptr, support = [ep_ids], [ind_ptr], [support]
                    raw[(src_key, dst_key)] = (ind_ptr, support)
                    nbr_ids.setdefault(dst_key, dict())[src_key] = ep_ids
            prev_ids = dict()
            # map neighbour ids to row indices of the previous level.  Keys in the graph's own (deterministic) order:
            # iterating a set of strings would follow per-process hash randomisation and let two ranks of a
            # node-partitioned run build / execute their plans in different orders.
            for key in [k for k in graph.meta_graph if k in nbr_ids or k in src_ids]:
                pieces = []
                if full_node_ids is not None and key in full_node_ids:
                    pieces.append(np.asarray(full_node_ids[key], dtype=np.int32))   # fixed row order on every rank
                for _src, eps in nbr_ids.get(key, {}).items():
                    pieces.extend(eps)
                if key in src_ids:
                    pieces.append(src_ids[key])
                uniq, inds = G.merge_nodes(pieces)
                if full_node_ids is not None and key in full_node_ids:
                    inds = inds[1:]
                prev_ids[key] = uniq
                cur = 0
                for src_key, eps in nbr_ids.get(key, {}).items():
                    ind_ptr, support = raw[(src_key, key)]
                    agg_args[src_key][2][key] = P.MultiLinkPlan(inds[cur:cur + len(eps)], ind_ptr, support,
                                                                n_src=uniq.shape[0], device=device)
                    cur += len(eps)
                if key in src_ids:
                    agg_args[key][0] = inds[cur]
            for src_key in agg_args:                    # resident take plans (base rows / inverse of unique)
                a = agg_args[src_key]
                a[0] = P.TakePlan(a[0], prev_ids[src_key].shape[0], device)
                if a[1] is not None:
                    a[1] = P.TakePlan(a[1], a[0].n, device)
            computing_plan[depth] = [prev_ids, agg_args]
            sel_node_ids_dict = prev_ids
        return computing_plan[0][0], computing_plan

    def heter_sage(self, input_dict, computing_plan):
        """Bottom-up execution of the plan (reference layers.py:339-385).

        Node-partitioned runs (layer.partition set, dist.py) reorder the work of a depth so that collectives overlap
        rank-local kernels: node types whose aggregate is a partial sum (replicated destinations) go FIRST and only
        launch their all-reduce; the rank-local node types run while it is in flight; then the replicated types
        finish (activation, output Dense).  The results equal the sequential order."""
        ret = dict()
        for depth in range(len(self)):
            ret = dict()
            _prev_ids, agg_args = computing_plan[depth]
            layer = self[depth]
            part = layer.partition
            grad_pending = dict()
            if part is not None and not layer._accum_self:
                # replicated features consumed by exactly one rank-local aggregation: their gradient all-reduce is
                # launched asynchronously there and awaited here (runs last in the backward pass of this depth)
                input_dict = dict(input_dict)
                for rkey in part.replicated_keys:
                    users = [s for s in agg_args if rkey in agg_args[s][2] and part.crossing_in(s, rkey)]
                    if rkey in input_dict and len(users) == 1:
                        input_dict[rkey], pend = D.grad_wait(input_dict[rkey], exclusive=True)   # one crossing, no other reader
                        if pend is not None:
                            grad_pending[rkey] = pend
            order = list(agg_args)
            if part is not None:
                order.sort(key=lambda k: 0 if k in part.replicated_keys else 1)
            started = dict()
            for src_key in order:
                base_take, sel_take, plans = agg_args[src_key]
                neighbor_data = {dst_key: (input_dict[dst_key], plan) for dst_key, plan in plans.items()}
                started[src_key] = layer.forward_start(src_key, neighbor_data, grad_pending)
            for src_key in reversed(order):          # rank-local types finish first, replicated types last
                base_take, sel_take, plans = agg_args[src_key]
                base = SF.take_rows(input_dict[src_key], base_take) if layer._accum_self else None
                out = layer.forward_finish(src_key, base, started[src_key])
                if depth == len(self) - 1 and sel_take is not None:
                    out = SF.take_rows(out, sel_take)
                ret[src_key] = out
            ret = {k: ret[k] for k in agg_args}      # keep the plan's key order for callers that iterate
            input_dict = ret
        return ret
