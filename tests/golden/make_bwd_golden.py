#!/usr/bin/env python3
"""Golden vectors for the DATA GRADIENT of seg_weighted_pool (= `_backward_seg_take_k_corr_embed2`, reference
seg_op.cc:209-240, 718-752) generated from the REFERENCE's own numpy models.

Run in the build container only (needs /root/reference):   python tests/golden/make_bwd_golden.py

The reference test file has no numpy model of this gradient (it checks it against finite differences).  Both operators
are LINEAR in the tensor the gradient is taken for, so the reference's FORWARD models pin the gradient exactly:

  (a) npy_seg_weighted_pool(data, weights, indices, indptr) is linear in `data`.  Evaluated on data = identity
      (B, T, T) it returns the matrix P (B, S, T) with  out = P . data  for every data.  The data gradient of
      <ograd, out> is therefore  P^T . ograd  -- computed here in float64 from the reference model's P.
  (b) npy_seg_take_k_corr(embed1, embed2, ids, indptr) is linear in `embed2`.  Evaluated on the unit tensors e_(m,c)
      it returns, column by column, the matrix of that map; the gradient w.r.t. embed2 of <ograd, out> is
      d[k,m,c] = <ograd[k], take_k_corr(embed1, e_(m,c))[k]>.  (small shapes: M*C evaluations of the python model)

As in make_golden.py the reference functions are extracted with `ast` (the module imports mxnet) and executed from the
file where it lies; the committed artefact is data only.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import REF_TEST, indptr_with_empties, load_reference_models  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bwd_data_golden.npz")


def main():
    ref = load_reference_models()
    rng = np.random.default_rng(20240918)
    np.random.seed(20240918)
    out = {}
    shapes = {"g0": (1, 5, 10, 30, 128), "g1": (10, 50, 20, 500, 4), "g2": (4, 1000, 10000, 50000, 4),
              "h64": (1, 40, 60, 600, 64), "h250": (1, 30, 45, 400, 250), "h256": (2, 30, 45, 400, 256)}
    for tag, (B, S, T, nnz, C) in shapes.items():
        for kind in ("dense", "empties"):
            if tag == "g2" and kind == "empties":
                continue
            weights = rng.normal(0, 1, (B, nnz)).astype(np.float32)
            ograd = rng.normal(0, 1, (B, S, C)).astype(np.float32)
            indices = rng.integers(0, T, size=(nnz,)).astype(np.int32)
            indptr = ref["rand_indptr"](S, nnz) if kind == "dense" else indptr_with_empties(rng, S, nnz)
            eye = np.eye(T, dtype=np.float32)[None]
            ddata = np.empty((B, T, C), np.float64)
            for b in range(B):      # P_b (S, T) from the reference forward model, one batch entry at a time
                P = ref["npy_seg_weighted_pool"](eye, weights[b:b + 1], indices, indptr)[0].astype(np.float64)
                ddata[b] = P.T @ ograd[b].astype(np.float64)
            p = "wp_%s_%s_" % (tag, kind)
            out[p + "weights"], out[p + "ograd"], out[p + "indices"], out[p + "indptr"] = weights, ograd, indices, indptr
            out[p + "total_ind_num"] = np.int64(T)
            out[p + "ddata"] = ddata.astype(np.float32)
    # (b) gradient of seg_take_k_corr w.r.t. embed2 from unit tensors, small shapes only
    for tag, (K, N, M, nnz, C) in {"g0": (1, 5, 10, 30, 16), "g1": (3, 12, 9, 80, 4)}.items():
        embed1 = rng.normal(0, 1, (K, N, C)).astype(np.float32)
        ograd = rng.normal(0, 1, (K, nnz)).astype(np.float32)
        ids = rng.integers(0, M, size=(nnz,)).astype(np.int32)
        indptr = indptr_with_empties(rng, N, nnz)
        d2 = np.zeros((K, M, C), np.float64)
        for m in range(M):
            for c in range(C):
                unit = np.zeros((K, M, C), np.float32)
                unit[:, m, c] = 1.0
                col = ref["npy_seg_take_k_corr"](embed1, unit, ids, indptr).astype(np.float64)      # (K, nnz)
                d2[:, m, c] = (col * ograd.astype(np.float64)).sum(axis=1)
        p = "tk_%s_" % tag
        out[p + "embed1"], out[p + "ograd"], out[p + "ids"], out[p + "indptr"] = embed1, ograd, ids, indptr
        out[p + "total_ind_num"] = np.int64(M)
        out[p + "dembed2"] = d2.astype(np.float32)
    np.savez_compressed(OUT, **out)
    print("wrote %s: %d arrays, %.1f KiB" % (OUT, len(out), os.path.getsize(OUT) / 1024.0))


if __name__ == "__main__":
    if not os.path.exists(REF_TEST):
        sys.exit("needs /root/reference (build container only)")
    main()
