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


# ------------------------------------------------------------------------------------------------------------------
# Whole-network dense oracle (independent of any plan / unique / re-indexing logic).
# With NUM_NEIGHBORS = -1 (every shipped yaml) the sampled computation of reference
# Net.forward (STAR-GCN.py:311-461) equals the full-graph computation restricted to the requested rows, so it can be
# restated with dense per-level adjacency matrices: A[(dst, src)][r] (n_dst, n_src) holding the support of the
# level-r edges (reference graph.py:414-429 / graph_sampler.cpp:393-420 for the support formula).
# ------------------------------------------------------------------------------------------------------------------
def dense_level_adjacency(row_ind, col_ind, values, levels, n_rows, n_cols, symm=True, dtype=torch.float64,
                          sparse=False):
    """sparse=True returns torch sparse COO matrices (same values, `A @ X` still works and differentiates): lets the
    oracle run at real MovieLens-1M size without dense (n_dst, n_src) float64 matrices."""
    row_ind, col_ind = np.asarray(row_ind), np.asarray(col_ind)
    dr = np.bincount(row_ind, minlength=n_rows).astype(np.float32)
    dc = np.bincount(col_ind, minlength=n_cols).astype(np.float32)
    if symm:
        sup = np.sqrt(np.float32(1.0) / dr[row_ind] / dc[col_ind]).astype(np.float32)
    else:
        sup = (np.float32(1.0) / dr[row_ind]).astype(np.float32)
    out = []
    for lv in levels:
        sel = np.asarray(values) == lv
        r = torch.as_tensor(row_ind[sel], dtype=torch.long)
        c = torch.as_tensor(col_ind[sel], dtype=torch.long)
        v = torch.as_tensor(sup[sel], dtype=dtype)
        if sparse:
            out.append(torch.sparse_coo_tensor(torch.stack([r, c]), v, (n_rows, n_cols)).coalesce())
        else:
            a = torch.zeros((n_rows, n_cols), dtype=dtype)
            a[r, c] = v
            out.append(a)
    return out


def dense_star_gcn(tables, noise, adj, blocks, maps, projs, rating_pairs, recon_ids, accum="sum", act="leaky",
                   features=None, fea_maps=None, recon_fea=False):
    """tables {key: (n, D)}; noise {key: int array or None}; adj {(dst, src): [A_r]};
    blocks[b] = list of layers, layer = {dst: dict(src=..., W=[..], b=[..], ow=, ob=)};
    maps[b] = {key: (w0, b0, w1, b1)} or None; projs[b] = {key: (w, b)};
    rating_pairs (user_key, item_key, u_idx, i_idx) or None; recon_ids {key: ids} or None.
    features {key: (n, F)} + fea_maps {key: (w0, b0, w1, b1)}: MODEL.USE_FEA_PROJ of reference STAR-GCN.py:182-192,
    405-413 -- the mapped features are concatenated to every block's input (and to the reconstruction target when
    recon_fea, :364-370; otherwise re-attached after the decoder map, :455-459)."""
    f = ACTS[act]
    x = {k: masked_embed(t, np.arange(t.shape[0]), noise.get(k) if noise else None) for k, t in tables.items()}
    gt = {k: tables[k][torch.as_tensor(np.asarray(v), dtype=torch.long)] for k, v in (recon_ids or {}).items()}
    fea = None
    if fea_maps is not None:
        fea = {k: dense(dense(features[k], fea_maps[k][0], fea_maps[k][1], act), fea_maps[k][2], fea_maps[k][3])
               for k in tables}
        x = {k: torch.cat([x[k], fea[k]], dim=1) for k in x}
        if recon_fea:
            gt = {k: torch.cat([g, fea[k][torch.as_tensor(np.asarray(recon_ids[k]), dtype=torch.long)]], dim=1)
                  for k, g in gt.items()}
    preds, recons = [], []
    for b, layers in enumerate(blocks):
        for layer in layers:
            nxt = {}
            for dst, p in layer.items():
                outs = []
                for r, a in enumerate(adj[(dst, p["src"])]):
                    outs.append(a @ (x[p["src"]] @ p["W"][r].t() + p["b"][r]))   # FullyConnected, then seg_weighted_pool
                h = f(torch.cat(outs, dim=1) if accum == "stack" else sum(outs))
                nxt[dst] = f(h @ p["ow"].t() + p["ob"])
            x = nxt
        out = x
        if rating_pairs is not None:
            uk, ik, ui, ii = rating_pairs
            pu = dense(out[uk], *projs[b][uk])[torch.as_tensor(np.asarray(ui), dtype=torch.long)]
            pi = dense(out[ik], *projs[b][ik])[torch.as_tensor(np.asarray(ii), dtype=torch.long)]
            preds.append((pu * pi).sum(dim=1, keepdim=True))
        if maps is not None and maps[b] is not None:
            emap = lambda k, t: dense(dense(t, maps[b][k][0], maps[b][k][1], act), maps[b][k][2], maps[b][k][3])
            if recon_ids is not None:
                recons.append({k: emap(k, out[k][torch.as_tensor(np.asarray(v), dtype=torch.long)])
                               for k, v in recon_ids.items()})
            if b < len(blocks) - 1:
                x = {k: emap(k, out[k]) for k in out}
                if fea is not None and not recon_fea:
                    x = {k: torch.cat([x[k], fea[k]], dim=1) for k in x}
    return preds, recons, gt
