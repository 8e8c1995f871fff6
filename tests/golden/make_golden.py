#!/usr/bin/env python3
"""Generate tests/golden/seg_ops_golden.npz from the REFERENCE's own numpy models.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

What it does
------------
The reference's unit tests (reference/seg_ops_cuda/mxnet_op/test_seg_ops.py:11-99) define
plain-numpy models of every segment operator (npy_seg_sum, npy_seg_broadcast_*, npy_seg_softmax,
npy_seg_take_k_corr, npy_seg_weighted_pool, npy_seg_pool, grad_seg_max_pool) and compare the MXNet
ops against them.  That test module imports `mxnet` at the top, which is not installable here, so
instead of importing the module this script parses it with `ast`, keeps ONLY the top-level
`def npy_*` / `def grad_seg_max_pool` / `def rand_indptr` function definitions (pure numpy), and
executes those definitions from the file where it lies.  No reference text is copied into this
repository: the committed artefact is data (seeded inputs + the outputs those reference functions
produced), i.e. golden vectors.

Shapes are the reference tests' own (test_seg_ops.py:118,163,225,269,314,382,449) plus the
hot-path feature widths of SURVEY.md section 8 (C in {50, 64, 75, 250, 256}) and edge cases
(empty segments, a padding edge past indptr[-1] with weight 0).
"""
import ast
import os
import sys

import numpy as np

REF_TEST = "/root/reference/seg_ops_cuda/mxnet_op/test_seg_ops.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seg_ops_golden.npz")
KEEP = {"rand_indptr", "npy_seg_sum", "npy_seg_broadcast_add", "npy_seg_broadcast_mul",
        "npy_seg_broadcast_to", "npy_softmax_contig", "npy_seg_softmax", "npy_seg_take_k_corr",
        "npy_seg_weighted_pool", "npy_seg_pool", "grad_seg_max_pool"}


def load_reference_models():
    with open(REF_TEST, "r") as f:
        tree = ast.parse(f.read(), REF_TEST)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in KEEP]
    missing = KEEP - {n.name for n in body}
    if missing:
        raise RuntimeError("reference test file lacks %s" % sorted(missing))
    ns = {"np": np}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF_TEST, "exec"), ns)
    return ns


def indptr_with_empties(rng, seg_num, nnz):
    """indptr with some empty segments (the reference's rand_indptr never makes one)."""
    cuts = np.sort(rng.integers(0, nnz + 1, size=seg_num - 1))
    return np.concatenate([[0], cuts, [nnz]]).astype(np.int32)


def main():
    ref = load_reference_models()
    out = {}
    rng = np.random.default_rng(20240917)
    np.random.seed(20240917)  # rand_indptr uses the global numpy RNG

    # ---- (B, S, nnz) ops: seg_sum, broadcast_*, softmax  (test_seg_ops.py:118) --------------
    for tag, (B, S, nnz) in {"s0": (1, 5, 10), "s1": (10, 50, 100), "s2": (4, 1000, 10000)}.items():
        data = rng.normal(0, 1, (B, nnz)).astype(np.float32)
        rhs = rng.normal(0, 1, (B, S)).astype(np.float32)
        for kind in ("dense", "empties"):
            indptr = ref["rand_indptr"](S, nnz) if kind == "dense" else indptr_with_empties(rng, S, nnz)
            p = "flat_%s_%s_" % (tag, kind)
            out[p + "data"] = data
            out[p + "rhs"] = rhs
            out[p + "indptr"] = indptr
            out[p + "seg_sum"] = ref["npy_seg_sum"](data, indptr)
            out[p + "bcast_add"] = ref["npy_seg_broadcast_add"](data, rhs, indptr)
            out[p + "bcast_mul"] = ref["npy_seg_broadcast_mul"](data, rhs, indptr)
            out[p + "bcast_to"] = ref["npy_seg_broadcast_to"](rhs, indptr, nnz)
            if kind == "dense":  # the reference softmax model marks uncovered slots -1; only used dense
                out[p + "softmax"] = ref["npy_seg_softmax"](data, indptr)

    # ---- gather ops: (K/B, S, T, nnz, C)  (test_seg_ops.py:314-316, 382-384, 449-451) --------
    shapes = {
        "g0": (1, 5, 10, 30, 128),
        "g1": (10, 50, 20, 500, 4),
        "g2": (4, 1000, 10000, 50000, 4),
        # hot-path widths: C = 50 (stack 250/5), 64, 75, 250, 256
        "h50": (1, 40, 60, 600, 50),
        "h64": (1, 40, 60, 600, 64),
        "h75": (1, 40, 60, 600, 75),
        "h250": (1, 30, 45, 400, 250),
        "h256": (2, 30, 45, 400, 256),
    }
    for tag, (B, S, T, nnz, C) in shapes.items():
        for kind in ("dense", "empties"):
            if tag == "g2" and kind == "empties":
                continue
            data = rng.normal(0, 1, (B, T, C)).astype(np.float32)
            embed1 = rng.normal(0, 1, (B, S, C)).astype(np.float32)
            weights = rng.normal(0, 1, (B, nnz)).astype(np.float32)
            indices = rng.integers(0, T, size=(nnz,)).astype(np.int32)
            indptr = ref["rand_indptr"](S, nnz) if kind == "dense" else indptr_with_empties(rng, S, nnz)
            p = "gather_%s_%s_" % (tag, kind)
            out[p + "data"] = data
            out[p + "embed1"] = embed1
            out[p + "weights"] = weights
            out[p + "indices"] = indices
            out[p + "indptr"] = indptr
            out[p + "weighted_pool"] = ref["npy_seg_weighted_pool"](data, weights, indices, indptr)
            if tag != "g2":  # the pure-python triple loop of the reference model is O(K*nnz) python steps
                out[p + "take_k_corr"] = ref["npy_seg_take_k_corr"](embed1, data, indices, indptr)
            out[p + "pool_sum"] = ref["npy_seg_pool"](data, indices, indptr, "sum")
            if kind == "dense":  # reference avg/max models are undefined (nan / error) on empty segments
                out[p + "pool_avg"] = ref["npy_seg_pool"](data, indices, indptr, "avg")
                out[p + "pool_max"] = ref["npy_seg_pool"](data, indices, indptr, "max")
                if tag in ("g0", "g1", "h64"):
                    ograd = rng.normal(0, 1, (B, S, C)).astype(np.float32)
                    out[p + "pool_max_ograd"] = ograd
                    out[p + "pool_max_grad"] = ref["grad_seg_max_pool"](ograd, data, indices, indptr)

    np.savez_compressed(OUT, **out)
    print("wrote %s: %d arrays, %.1f KiB" % (OUT, len(out), os.path.getsize(OUT) / 1024.0))


if __name__ == "__main__":
    if not os.path.exists(REF_TEST):
        sys.exit("needs /root/reference (build container only)")
    main()
