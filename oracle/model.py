"""Layer-level oracle -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates, on torch-CPU tensors in the REFERENCE's operation order, the Python layers that sit on the hot path:
  * MultiLinkGCNAggregator.hybrid_forward   reference mxgraph/layers/aggregators.py:111-163
  * HeterGCNLayer.forward_single            reference mxgraph/layers/layers.py:147-187
  * Net.get_embed / decoder / rating head   reference experiments/STAR-GCN.py:264-300, 428-459
Gradients come from torch autograd on this restatement (float64 by default, so it doubles as the high-precision
reference for the fp32 1e-5 budget).  PARITY UNPINNED: MXNet/Gluon cannot be imported in this image and the
reference has no test for these layers, so this file is pinned only by reading the source.
The sparse op inside is `seg_weighted_pool` in the reference's own loop order; `use_c_oracle=True` routes it
through oracle/seg_oracle.c (fp32) instead of torch index_add (any dtype).
"""
import numpy as np
import torch

from . import seg as O


def leaky(x, slope=0.1):
    """reference mxgraph/layers/common.py:47 -- LeakyReLU(0.1)."""
    return torch.where(x > 0, x, slope * x)


ACTS = {None: lambda x: x, "identity": lambda x: x, "leaky": leaky, "relu": torch.relu,
        "sigmoid": torch.sigmoid, "tanh": torch.tanh}


def seg_weighted_pool(data, weights, indices, indptr):
    """data (T,C), weights (nnz), indices (nnz), indptr (S+1) -> (S,C): dst[i] = sum_j w_j * data[idx_j]
    (reference seg_op.cc:180-207), differentiable, any dtype."""
    S = indptr.shape[0] - 1
    E = int(indptr[-1])
    seg = torch.repeat_interleave(torch.arange(S), torch.as_tensor(np.diff(indptr.numpy() if isinstance(indptr, torch.Tensor) else indptr)))
    idx = torch.as_tensor(np.asarray(indices[:E]), dtype=torch.long)
    rows = data[idx] * weights[:E].unsqueeze(1)
    out = torch.zeros((S, data.shape[1]), dtype=data.dtype)
    return out.index_add(0, seg, rows)


def multilink_aggregator(x, weights, biases, end_points_l, indptr_l, support_l, accum="stack", act="leaky",
                         ordinal_sharing=False):
    """reference aggregators.py:111-163 (dropout omitted = rate 0): per level FullyConnected THEN
    seg_weighted_pool; concat ('stack') or add_n ('sum'); activation."""
    outs = []
    w, b = weights[0], biases[0]
    for i in range(len(weights)):
        if i > 0 and ordinal_sharing:
            w, b = w + weights[i], b + biases[i]
        else:
            w, b = weights[i], biases[i]
        h = x @ w.t() + b                                   # FullyConnected(flatten=False)
        sup = torch.as_tensor(np.asarray(support_l[i]), dtype=x.dtype)
        outs.append(seg_weighted_pool(h, sup, np.asarray(end_points_l[i]), np.asarray(indptr_l[i])))
    out = outs[0] if len(outs) == 1 else (torch.cat(outs, dim=1) if accum == "stack" else sum(outs))
    return ACTS[act](out)


def dense(x, w, b, act=None):
    y = x @ w.t()
    if b is not None:
        y = y + b
    return ACTS[act](y)


def masked_embed(table, ids, noise=None):
    """reference STAR-GCN.py:290-299."""
    ids = torch.as_tensor(np.asarray(ids), dtype=torch.long)
    if noise is None:
        return table[ids]
    nid = torch.as_tensor(np.asarray(noise), dtype=torch.long)[ids]
    mask = (nid != -1)
    return table[nid * mask] * mask.unsqueeze(1).to(table.dtype)


def c_seg_weighted_pool(data, weights, indices, indptr):
    """fp32 numpy in/out through the C restatement (bit-level reference order)."""
    return O.seg_weighted_pool(data[None], weights[None], indices, indptr)[0]
