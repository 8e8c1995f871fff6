"""Graph-conv aggregators with the constructor / call signatures of reference
mxgraph/layers/aggregators.py (GCNAggregator :21-55, MultiLinkGCNAggregator :58-163).

What differs underneath: the reference issues, per rating level, one FullyConnected and one
`contrib.seg_weighted_pool` (2R+1 operator launches, R intermediates of size n_src x units); here the R levels
are fused into a cached `MultiLinkPlan` and executed as one gather launch + one MFMA contraction
(`functional.multilink_aggregate`), with the association order chosen so the R-times expanded matrix sits on
the smaller node side.  Results equal the reference's A_r (X W_r^T + b_r) to fp32 rounding.
"""
import numpy as np
import torch
from torch import nn

from .._native import functional as SF
from .._native import plan as P
from .common import Activation, get_activation, xavier_in_uniform_


class BaseAggregator(nn.Module):
    @property
    def use_multi_link(self):
        raise NotImplementedError

    @property
    def use_support(self):
        raise NotImplementedError

    @property
    def use_edge_type(self):
        raise NotImplementedError


class MultiLinkGCNAggregator(BaseAggregator):
    def __init__(self, units, num_links, act=None, dropout_rate=0.0, ordinal_sharing=True, accum='stack',
                 in_units=0, order='auto'):
        super().__init__()
        if accum not in ('stack', 'sum'):
            raise NotImplementedError(accum)
        if accum == 'stack':
            assert units % num_links == 0, 'units should be divisible by the num_links '
            units = units // num_links
        self._units, self._num_links = units, num_links
        self._act = get_activation(act)
        self._ordinal_sharing, self._accum, self._order = ordinal_sharing, accum, order
        self.dropout = nn.Dropout(dropout_rate)
        for i in range(num_links):   # same parameter names as the reference: weight{i} (units, in), bias{i} zeros
            w = nn.UninitializedParameter() if in_units == 0 else nn.Parameter(torch.empty(units, in_units))
            setattr(self, 'weight{}'.format(i), w)
            setattr(self, 'bias{}'.format(i), nn.Parameter(torch.zeros(units)))
            if in_units:
                xavier_in_uniform_(w)
        self._plan_cache = {}

    @property
    def use_multi_link(self):
        return True

    @property
    def use_support(self):
        return True

    @property
    def use_edge_type(self):
        return False

    @property
    def activation(self):
        return self._act

    def _params(self, in_units, device):
        ws, bs = [], []
        for i in range(self._num_links):
            w = getattr(self, 'weight{}'.format(i))
            if isinstance(w, nn.UninitializedParameter):
                w.materialize((self._units, in_units), device=device, dtype=torch.float32)
                xavier_in_uniform_(w)
            b = getattr(self, 'bias{}'.format(i))
            if b.device != device:
                b.data = b.data.to(device)
            ws.append(w)
            bs.append(b)
        if self._ordinal_sharing and self._num_links > 1:   # cumulative sums, reference aggregators.py:134-137
            ws = list(torch.cumsum(torch.stack(ws), dim=0).unbind(0))
            bs = list(torch.cumsum(torch.stack(bs), dim=0).unbind(0))
        return ws, bs

    def _plan_for(self, end_points_l, indptr_l, support_l, n_src, device):
        if isinstance(end_points_l, P.MultiLinkPlan):
            return end_points_l
        key = tuple(id(a) for a in end_points_l) + tuple(id(a) for a in indptr_l) + tuple(id(a) for a in support_l)
        hit = self._plan_cache.get(key)
        if hit is None:
            plan = P.MultiLinkPlan(end_points_l, indptr_l, support_l, n_src, device)
            if len(self._plan_cache) >= 16:
                self._plan_cache.clear()
            self._plan_cache[key] = hit = (plan, end_points_l, indptr_l, support_l)  # keep the key objects alive
        return hit[0]

    def forward(self, neighbor_data, end_points_l, indptr_l=None, support_l=None, defer_act=False):
        """neighbor_data (n_src, feat); end_points_l / indptr_l / support_l: per-level arrays as in the reference
        (element shapes (nnz_r,), (n_dst+1,), (nnz_r,)), or a prebuilt MultiLinkPlan in place of end_points_l.
        defer_act=True returns the pre-activation sum (node-partitioned runs all-reduce it first) -- apply
        `self.activation` afterwards."""
        x = self.dropout(neighbor_data)
        plan = self._plan_for(end_points_l, indptr_l, support_l, x.shape[0], x.device)
        ws, bs = self._params(x.shape[1], x.device)
        act = self._act
        if defer_act:
            return SF.multilink_aggregate(x, ws, bs, plan, accum=self._accum, order=self._order)
        if isinstance(act, Activation) and act.fusable:
            return SF.multilink_aggregate(x, ws, bs, plan, accum=self._accum, act=act.fused, slope=act.slope,
                                          order=self._order)
        return act(SF.multilink_aggregate(x, ws, bs, plan, accum=self._accum, order=self._order))


class GCNAggregator(BaseAggregator):
    """Single-link wrapper, reference aggregators.py:21-55."""

    def __init__(self, units, act=None, dropout_rate=0.0, in_units=0):
        super().__init__()
        self._agg = MultiLinkGCNAggregator(units=units, num_links=1, act=act, dropout_rate=dropout_rate,
                                           in_units=in_units)

    @property
    def use_multi_link(self):
        return False

    @property
    def use_support(self):
        return True

    @property
    def use_edge_type(self):
        return False

    @property
    def activation(self):
        return self._agg.activation

    def forward(self, neighbor_data, end_points, indptr=None, support=None, defer_act=False):
        if isinstance(end_points, P.MultiLinkPlan):
            return self._agg(neighbor_data, end_points, defer_act=defer_act)
        return self._agg(neighbor_data, [end_points], [indptr], [support], defer_act=defer_act)
