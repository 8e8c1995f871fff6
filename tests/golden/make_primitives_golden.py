#!/usr/bin/env python3
"""Generate tests/golden/graph_primitives_golden.npz: inputs and outputs of the REFERENCE's GraphSampler C++ core
(GraphSampler/graph_sampler.{h,cpp}), compiled from its own sources by `make -C oracle _ref` (no stand-in headers, see
oracle/gs_ref_wrap.cpp) and called through oracle/gs_ref.GraphSamplerRef, on seeded inputs.  Build container only:

    OMP_NUM_THREADS=4 python tests/golden/make_primitives_golden.py

These vectors pin, primitive by primitive, what the product's host builders (csrc/graph_host.cpp, `sg_*_cpu`) and their
device twins (csrc/plan_build.hip / edge_mask.hip, `sg_*_hip`) must reproduce:

  get_support                graph_sampler.cpp:393-420   both modes, real degrees and override degrees containing zeros;
                                                         IEEE build and the reference's own -O3 -ffast-math build
  multi_link_split_by_value  graph_sampler.cpp:277-376   <= 10 000 nnz (serial form) and > 10 000 nnz (the _omp form)
  remove_edges               graph_sampler.cpp:154-201   repeated pairs, pairs that are not edges, a row that loses all
                                                         its edges; + degrees / support of the graph that is left
  slice_csr_mat              graph_sampler.cpp:31-152    rows only / columns only / both / neither
  gen_row_indices_by_indptr  graph_sampler.cpp:378-391
  unique_inverse, unique_cnt graph_sampler.h:441-534     <= 10 000 (first-occurrence order) and > 10 000 (_omp forms)
  random_sample_fix_neighbor graph_sampler.cpp:742-779   the copy branches bit for bit; the sampled branch with the
                                                         reference's Mersenne-Twister stream (1 thread, seed 7) as a
                                                         record of its output FORMAT (row pointer, per-row draw size)

The committed artefact is data (int32 / float32 arrays); no reference text enters the repository.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.gs_ref import GraphSamplerRef  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_primitives_golden.npz")

CASES = {
    # name: (seed, n_rows, n_cols, nnz, multi_link)
    "s": (101, 120, 70, 2600, np.arange(1, 6, dtype=np.float32)),                 # <= 10 000 nnz: the serial forms
    "l": (202, 900, 400, 24000, np.arange(1, 11, dtype=np.float32) * np.float32(0.5)),   # > 10 000: the _omp forms
}


def make_csr(seed, n_rows, n_cols, nnz, multi_link):
    """Duplicate-free rating matrix with column-sorted rows (scipy tocsr order, datasets.py:116-121), skewed degrees
    and a few EMPTY rows and columns."""
    rng = np.random.default_rng(seed)
    pr = np.exp(rng.normal(size=n_rows))
    pc = np.exp(1.5 * rng.normal(size=n_cols))
    pr[rng.permutation(n_rows)[:max(2, n_rows // 40)]] = 0.0
    pc[rng.permutation(n_cols)[:max(2, n_cols // 40)]] = 0.0
    p = np.outer(pr / pr.sum(), pc / pc.sum()).ravel()
    cells = np.sort(rng.choice(n_rows * n_cols, nnz, replace=False, p=p / p.sum()))
    rows, cols = (cells // n_cols).astype(np.int32), (cells % n_cols).astype(np.int32)
    ind_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n_rows))]).astype(np.int32)
    vals = multi_link[rng.integers(0, multi_link.size, nnz)].astype(np.float32)
    return rng, rows, cols, vals, ind_ptr


def main():
    ref, ref_fm = GraphSamplerRef(), GraphSamplerRef(fastmath=True)
    out = {}
    for tag, (seed, n_rows, n_cols, nnz, ml) in CASES.items():
        rng, rows, cols, vals, ip = make_csr(seed, n_rows, n_cols, nnz, ml)
        P = tag + "_"
        out.update({P + "ep": cols, P + "ip": ip, P + "val": vals, P + "ml": ml,
                    P + "shape": np.array([n_rows, n_cols], np.int32)})
        rd = np.diff(ip).astype(np.int32)
        cd = np.bincount(cols, minlength=n_cols).astype(np.int32)
        assert (rd == 0).any() and (cd == 0).any()
        # ---- get_support -------------------------------------------------------------------------------------------
        rdz, cdz = rd.copy(), cd.copy()                 # override degrees (a rank-local block normalised with global
        rdz[rng.permutation(n_rows)[:n_rows // 5]] = 0  # degrees, or degrees after an edge removal): zeros on rows /
        cdz[rng.permutation(n_cols)[:n_cols // 5]] = 0  # columns that DO have edges exercise the != 0 guards
        cdz[cdz > 0] += rng.integers(0, 50, int((cdz > 0).sum())).astype(np.int32)
        out.update({P + "rdz": rdz, P + "cdz": cdz})
        for name, (r_, c_) in (("", (rd, cd)), ("z", (rdz, cdz))):
            for symm in (1, 0):
                key = P + "sup%s_%s" % (name, "symm" if symm else "row")
                out[key] = ref.get_support(r_, c_, ip, cols, symm)
                out[key + "_fastmath"] = ref_fm.get_support(r_, c_, ip, cols, symm)
        out[P + "row_idx"] = ref.gen_row_indices_by_indptr(ip, nnz)
        assert np.array_equal(out[P + "row_idx"], rows)
        # ---- multi_link_split ----------------------------------------------------------------------------------------
        pos, ptr = ref.multi_link_split(vals, ip, ml)                 # dispatching entry, as the binding calls it
        pos_o, ptr_o = ref.multi_link_split(vals, ip, ml, omp=True)   # the _omp form explicitly
        for a, b in zip(pos + ptr, pos_o + ptr_o):
            assert np.array_equal(a, b), "serial and _omp forms of multi_link_split disagree"
        for r in range(ml.size):
            out.update({P + "split_pos%d" % r: pos[r], P + "split_ptr%d" % r: ptr[r]})
        # ---- remove_edges ----------------------------------------------------------------------------------------------
        n_rm = nnz // 8
        sel = rng.choice(nnz, n_rm, replace=False)
        victim = int(np.argmax(rd == np.sort(rd[rd > 0])[len(rd[rd > 0]) // 2]))     # a median-degree row loses everything
        whole = np.arange(ip[victim], ip[victim + 1])
        rr = np.concatenate([rows[sel], rows[sel[:9]], rows[whole], rng.integers(0, n_rows, 40).astype(np.int32)])
        rc = np.concatenate([cols[sel], cols[sel[:9]], cols[whole], rng.integers(0, n_cols, 40).astype(np.int32)])
        perm = rng.permutation(rr.size)
        rr, rc = np.ascontiguousarray(rr[perm], np.int32), np.ascontiguousarray(rc[perm], np.int32)
        # the form the binding calls.  (remove_edges_omp is dead code in the reference and racy -- all threads write the
        # bit-packed std::vector<bool> find_row, graph_sampler.cpp:235-240 -- so it is not a source of vectors.)
        ep2, val2, ip2 = ref.remove_edges_by_indices(cols, vals, ip, rr, rc)
        rd2 = np.diff(ip2).astype(np.int32)
        cd2 = np.bincount(ep2, minlength=n_cols).astype(np.int32)
        out.update({P + "rm_rows": rr, P + "rm_cols": rc, P + "rm_ep": ep2, P + "rm_val": val2, P + "rm_ip": ip2,
                    P + "rm_sup_symm": ref.get_support(rd2, cd2, ip2, ep2, 1),
                    P + "rm_sup_row": ref.get_support(rd2, cd2, ip2, ep2, 0)})
        assert rd2[victim] == 0
        # the transposed matrix of what is left (scipy transposes in the reference, graph.py:585-593): its support is what
        # the item-side plans carry -- sqrt(1/d_col/d_row) has the fp32 divisions in the other order
        order = np.lexsort((np.repeat(np.arange(n_rows), rd2), ep2))
        t_ep = np.repeat(np.arange(n_rows), rd2)[order].astype(np.int32)
        t_ip = np.concatenate([[0], np.cumsum(cd2)]).astype(np.int32)
        out.update({P + "rm_t_sup_symm": ref.get_support(cd2, rd2, t_ip, t_ep, 1),
                    P + "rm_t_sup_row": ref.get_support(cd2, rd2, t_ip, t_ep, 0)})
        # ---- slice_csr_mat -------------------------------------------------------------------------------------------
        row_ids = (np.arange(n_rows) * 3 + 7).astype(np.int32)          # ids != indices
        col_ids = (np.arange(n_cols) * 2 + 1).astype(np.int32)
        sr = rng.permutation(n_rows)[:n_rows // 3].astype(np.int32)     # given order, not sorted
        sc = rng.permutation(n_cols)[:n_cols // 2].astype(np.int32)
        out.update({P + "row_ids": row_ids, P + "col_ids": col_ids, P + "sub_rows": sr, P + "sub_cols": sc})
        for name, (a, b) in (("r", (sr, None)), ("c", (None, sc)), ("rc", (sr, sc)), ("all", (None, None))):
            e, v, p_, ri, ci = ref.csr_submat(cols, vals, ip, row_ids, col_ids, a, b)
            out.update({P + "sub_%s_ep" % name: e, P + "sub_%s_val" % name: v, P + "sub_%s_ip" % name: p_,
                        P + "sub_%s_rid" % name: ri, P + "sub_%s_cid" % name: ci})
        e, v, p_, ri, ci = ref.csr_submat(cols, None, ip, row_ids, col_ids, sr, sc)     # no values
        assert v is None and np.array_equal(e, out[P + "sub_rc_ep"])
        # ---- random_sample_fix_neighbor ------------------------------------------------------------------------------
        sel_rows = np.concatenate([rng.permutation(n_rows)[:n_rows // 2], [victim, victim]]).astype(np.int32)
        out[P + "fix_sel"] = sel_rows
        for k in (-1, int(rd.max()), int(rd.max()) + 5):               # copy branches: everything, in CSR order
            s, p_ = ref.random_sample_fix_neighbor(ip, sel_rows, k)
            out.update({P + "fix_pos_k%d" % k: s, P + "fix_ptr_k%d" % k: p_})
        out[P + "fix_k_list"] = np.array([-1, int(rd.max()), int(rd.max()) + 5], np.int32)
        os.environ["OMP_NUM_THREADS"] = "1"
        ref.set_seed(7)
        for k in (0, 3, 17):
            s, p_ = ref.random_sample_fix_neighbor(ip, sel_rows, k)
            out.update({P + "fix_mt_pos_k%d" % k: s, P + "fix_mt_ptr_k%d" % k: p_})
    # ---- unique_inverse / unique_cnt ---------------------------------------------------------------------------------
    rng = np.random.default_rng(303)
    for tag, n, hi in (("s", 5000, 300), ("e", 10000, 20000), ("l", 30000, 2000), ("one", 1, 5)):
        d = rng.integers(0, hi, n).astype(np.int32)
        u, inv = ref.unique_inverse(d)
        uc, cnt = ref.unique_cnt(d)
        out.update({"uq_%s_data" % tag: d, "uq_%s_uniq" % tag: u, "uq_%s_inv" % tag: inv, "uq_%s_cnt_vals" % tag: uc,
                    "uq_%s_cnt" % tag: cnt})
        assert np.array_equal(u[inv], d)
    # ---- seg_mul / seg_add / seg_sum / take_1d_omp (py_ext.cpp:230-372, 535-563; host-side helpers of graph.py:14-62) ---
    ip = out["s_ip"]
    lhs = rng.normal(size=int(ip[-1])).astype(np.float32)
    rhs = rng.normal(size=ip.size - 1).astype(np.float32)
    out.update(seg_lhs=lhs, seg_rhs=rhs, seg_mul=ref.seg_mul(lhs, ip, rhs), seg_add=ref.seg_add(lhs, ip, rhs),
               seg_sum=ref.seg_sum(lhs, ip), take_sel=out["s_fix_sel"],
               take_out=ref.take_1d_omp(rhs, out["s_fix_sel"]))
    np.savez_compressed(OUT, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote %s: %d arrays, %d bytes" % (OUT, len(out), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
