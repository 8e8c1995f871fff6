#!/usr/bin/env python3
"""Generate tests/golden/graph_glue_golden.npz: outputs of the REFERENCE's own Python graph layer
(mxgraph/graph.py CSRMat / HeterGraph / merge_nodes / merge_node_ids_dict / empty_as_zero and
mxgraph/iterators.py DataIterator, mxgraph/layers/layers.py StackedHeterGCNLayers.gen_plan) on seeded inputs.  Run in the build container only (needs /root/reference):

    python tests/golden/make_graph_golden.py

How the reference code is executed
----------------------------------
`graph.py` imports `mxnet` (absent) and the compiled extension `mxgraph._graph_sampler` at its top.  This script parses
the reference files with `ast` where they lie, drops exactly those two import statements (and iterators.py's
`from mxgraph.graph import ...`), and executes the remaining, unmodified reference definitions.  `_graph_sampler` is
the REFERENCE's OWN C++ core (GraphSampler/graph_sampler.{h,cpp}) compiled from its sources by `make -C oracle _ref`
(oracle/_ref/libgs_ref.so; -D_WIN32 selects the std::unordered_* branches the reference carries, so google/sparsehash
is not needed and no stand-in is written) behind `oracle/gs_ref.GraphSamplerRef`, a ctypes twin of the method table of
py_ext.cpp:612-627 (the CPython binding itself does not compile against NumPy 2).  Until round 3 the eight primitives
were builder-written numpy stand-ins; swapping in the compiled reference left all 380 arrays bit-identical.
So these vectors PIN, against the reference's Python AND C++: id<->index mapping, CSR transposition, both-direction
edge removal, per-level neighbour lists (`sample_neighbors`), support values, node merging / re-indexing, sub-matrix
selection, and the samplers' RNG call sequence -- i.e. the integer plan arrays that enter the hot path.
No reference text is copied into this repository: the committed artefact is data.
"""
import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.gs_ref import GraphSamplerRef  # noqa: E402

REF_GRAPH = "/root/reference/mxgraph/graph.py"
REF_ITER = "/root/reference/mxgraph/iterators.py"
REF_LAYERS = "/root/reference/mxgraph/layers/layers.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_glue_golden.npz")


def load_reference():
    def keep(node):
        if isinstance(node, ast.Import):
            return not any(a.name in ("mxnet", "mxgraph._graph_sampler") for a in node.names)
        if isinstance(node, ast.ImportFrom):
            return node.module != "mxgraph.graph"
        return True

    ns = {"_graph_sampler": GraphSamplerRef(), "mx": None, "__name__": "reference_graph"}
    with open(REF_GRAPH) as f:
        tree = ast.parse(f.read(), REF_GRAPH)
    exec(compile(ast.Module(body=[n for n in tree.body if keep(n)], type_ignores=[]), REF_GRAPH, "exec"), ns)
    it = {"HeterGraph": ns["HeterGraph"], "CSRMat": ns["CSRMat"], "__name__": "reference_iterators"}
    with open(REF_ITER) as f:
        tree = ast.parse(f.read(), REF_ITER)
    exec(compile(ast.Module(body=[n for n in tree.body if keep(n)], type_ignores=[]), REF_ITER, "exec"), it)
    return ns, it


def load_reference_gen_plan(graph_ns):
    """`StackedHeterGCNLayers.gen_plan` (layers.py:260-337) lives in a Gluon subclass (mxnet absent): take the
    method's FunctionDef out of the class body and execute it, unmodified, as a plain function of a stub `self`.
    Its debugging `print(...); ch = input()` (layers.py:319-320) is silenced by shadowing the two builtins."""
    with open(REF_LAYERS) as f:
        tree = ast.parse(f.read(), REF_LAYERS)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "StackedHeterGCNLayers"][0]
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "gen_plan"][0]
    ns = {"np": np, "unordered_unique": graph_ns["unordered_unique"], "merge_nodes": graph_ns["merge_nodes"],
          "print": lambda *a, **k: None, "input": lambda *a: ""}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF_LAYERS, "exec"), ns)
    return ns["gen_plan"]


class _StubStack(object):
    """What gen_plan touches of `self`: len(self) and self[depth].aggregators[(src, dst)].use_multi_link."""

    class _Agg(object):
        use_multi_link = True

    class _Layer(object):
        def __init__(self, keys):
            self.aggregators = {k: _StubStack._Agg() for k in keys}

    def __init__(self, depth, keys):
        self._layers = [self._Layer(keys) for _ in range(depth)]

    def __len__(self):
        return len(self._layers)

    def __getitem__(self, i):
        return self._layers[i]


def make_inputs(seed, n_user, n_item, nnz, levels):
    """Seeded duplicate-free bipartite rating graph in CSR order (node ids = 0..n-1, as the reference's samplers index
    their noise arrays with them, iterators.py:338-346)."""
    rng = np.random.default_rng(seed)
    cells = rng.choice(n_user * n_item, nnz, replace=False)
    cells.sort()                                    # CSR order: rows, then columns (scipy tocsr order)
    ru, ci = (cells // n_item).astype(np.int32), (cells % n_item).astype(np.int32)
    vals = np.asarray(levels, np.float32)[rng.integers(0, len(levels), nnz)]
    ind_ptr = np.concatenate([[0], np.cumsum(np.bincount(ru, minlength=n_user))]).astype(np.int32)
    return ru, ci, vals, ind_ptr


def main():
    ref, ref_it = load_reference()
    out = {}

    # ---- node merging (graph.py:142-219) -------------------------------------------------------------------------
    rng = np.random.default_rng(7)
    a, b, c = (rng.integers(0, 40, n).astype(np.int32) for n in (25, 1, 60))
    uniq, idx = ref["merge_nodes"]([a, b, c])
    out.update(mn_a=a, mn_b=b, mn_c=c, mn_uniq=uniq, mn_idx0=idx[0], mn_idx1=idx[1], mn_idx2=idx[2])
    uniq1, idx1 = ref["merge_nodes"](c)
    out.update(mn1_uniq=uniq1, mn1_idx=idx1)
    pair = rng.integers(0, 30, (3, 12)).astype(np.int32)        # (1 + K, n): src row, K dst rows
    d0, d1 = {"user": a, ("user", "movie"): pair}, {"movie": c[:20]}
    ud, nl = ref["merge_node_ids_dict"]([d0, d1])
    out.update(md_pair=pair, md_u_user=ud["user"], md_u_movie=ud["movie"], md_0_user=nl[0]["user"],
               md_0_pair=nl[0][("user", "movie")], md_1_movie=nl[1]["movie"])
    ez = ref["empty_as_zero"]([np.zeros(0, np.float32), np.array([2.5, 1.0])], np.float32)
    out.update(ez0=ez[0], ez1=ez[1])

    # ---- CSRMat / HeterGraph glue on a rating graph ---------------------------------------------------------------
    levels = [0.5, 1.0, 2.0, 3.5, 5.0]
    n_user, n_item, nnz = 30, 22, 260
    ru, ci, vals, ind_ptr = make_inputs(11, n_user, n_item, nnz, levels)
    user_ids = np.arange(n_user, dtype=np.int32)       # HeterGraph needs ids usable as feature row indices
    item_ids = np.arange(n_item, dtype=np.int32)
    out.update(g_ru=ru, g_ci=ci, g_vals=vals, g_ind_ptr=ind_ptr, g_levels=np.asarray(levels, np.float32))
    CSRMat, HeterGraph = ref["CSRMat"], ref["HeterGraph"]
    mat = CSRMat(ci, ind_ptr, user_ids, item_ids, values=vals, multi_link=np.asarray(levels, np.float32))  # the binding insists on float32 (py_ext.cpp:513)
    graph = HeterGraph({"user": np.zeros((n_user, 1), np.float32), "movie": np.zeros((n_item, 1), np.float32)},
                       {"user": user_ids, "movie": item_ids}, {("user", "movie"): mat})

    def dump_csr(prefix, m):
        out.update({prefix + "ep": m.end_points, prefix + "ip": m.ind_ptr, prefix + "val": m.values,
                    prefix + "rdeg": m.row_degrees.astype(np.int32), prefix + "cdeg": m.col_degrees.astype(np.int32),
                    prefix + "sup_symm": m.get_support(True), prefix + "sup_row": m.get_support(False)})

    dump_csr("um_", graph["user", "movie"])
    dump_csr("mu_", graph["movie", "user"])            # = mat.T via scipy (graph.py:586-593)

    def dump_neighbors(prefix, m, src_ids, symm):
        eps, vs, ips, sps = m.sample_neighbors(src_ids=src_ids, symm=symm, use_multi_link=True, num_neighbors=-1)
        for l in range(len(levels)):
            out.update({"%sep%d" % (prefix, l): eps[l], "%sval%d" % (prefix, l): vs[l], "%sip%d" % (prefix, l): ips[l],
                        "%ssup%d" % (prefix, l): sps[l]})
        ep, v, ip, sp = m.sample_neighbors(src_ids=src_ids, symm=symm, use_multi_link=False)
        out.update({prefix + "flat_ep": ep, prefix + "flat_val": v, prefix + "flat_ip": ip, prefix + "flat_sup": sp})

    src = np.array([4, 0, 17, 4, 29], np.int32)
    out["nb_src"] = src
    dump_neighbors("nb_all_um_", graph["user", "movie"], None, True)
    dump_neighbors("nb_all_mu_", graph["movie", "user"], None, True)
    dump_neighbors("nb_sub_um_", graph["user", "movie"], src, True)
    dump_neighbors("nb_sub_um_row_", graph["user", "movie"], src, False)

    # batch edge removal in both directions (graph.py:952-974) and value fetch (:920-934)
    sel = rng.choice(nnz, 40, replace=False)
    pairs = np.stack([user_ids[ru[sel]], item_ids[ci[sel]]]).astype(np.int32)
    out["rm_pairs"] = pairs
    out["fetch_vals"] = graph.fetch_edges_by_id("user", "movie", pairs)
    g2 = graph.remove_edges_by_id("user", "movie", pairs)
    dump_csr("rm_um_", g2["user", "movie"])
    dump_csr("rm_mu_", g2["movie", "user"])
    dump_neighbors("rm_nb_um_", g2["user", "movie"], None, True)
    out["um_pair_ids"] = graph["user", "movie"].node_pair_ids

    # ---- gen_plan: the 2-layer top-down plan with re-indexing (layers.py:260-337) ----------------------------------
    gen_plan = load_reference_gen_plan(ref)
    keys = [("user", "movie"), ("movie", "user")]
    sel = {"user": np.array([3, 9, 3, 28, 0, 9], np.int32), "movie": np.array([5, 5, 1, 20], np.int32)}
    out.update(gp_sel_user=sel["user"], gp_sel_movie=sel["movie"])
    req, plan = gen_plan(_StubStack(2, keys), g2, sel, {k: -1 for k in keys}, True)
    for key in ("user", "movie"):
        out["gp_req_" + key] = req[key]
    for depth in range(2):
        prev_ids, agg_args = plan[depth]
        for key in ("user", "movie"):
            out["gp%d_prev_%s" % (depth, key)] = prev_ids[key]
            base_inds, sel_idx, info = agg_args[key]
            out["gp%d_base_%s" % (depth, key)] = base_inds
            if sel_idx is not None:
                out["gp%d_selidx_%s" % (depth, key)] = sel_idx
            for dst_key, (eps, vals_l, ips, sps) in info.items():
                for l in range(len(levels)):
                    out["gp%d_%s_%s_ep%d" % (depth, key, dst_key, l)] = eps[l]
                    out["gp%d_%s_%s_ip%d" % (depth, key, dst_key, l)] = ips[l]
                    out["gp%d_%s_%s_sup%d" % (depth, key, dst_key, l)] = sps[l]

    # ---- DataIterator: splits + the samplers' RNG call sequence (iterators.py:120-236, 264-370) -------------------
    test_pairs, valid_pairs = pairs[:, :15], pairs[:, 15:]
    it = ref_it["DataIterator"](graph, "user", "movie", test_node_pairs=test_pairs, valid_node_pairs=valid_pairs,
                                embed_P_mask=0.3, embed_p_zero=0.5, embed_p_self=0.5, seed=123)
    out.update(it_train_pairs=it._train_node_pairs, it_train_ratings=it._train_ratings,
               it_valid_ratings=it._valid_ratings, it_test_ratings=it._test_ratings,
               it_eval_noise_user=it.evaluate_embed_noise_dict["user"],
               it_eval_noise_movie=it.evaluate_embed_noise_dict["movie"])
    rs = it.rating_sampler(batch_size=32, segment="train")
    for k in range(3):
        p, r = next(rs)
        out.update({"it_rs%d_pairs" % k: p, "it_rs%d_ratings" % k: r})
    ns = it.recon_nodes_sampler(batch_size=4)
    for k in range(3):
        noise, batch, allr = next(ns)
        for key in ("user", "movie"):
            out.update({"it_ns%d_noise_%s" % (k, key): noise[key], "it_ns%d_batch_%s" % (k, key): batch[key],
                        "it_ns%d_all_%s" % (k, key): allr[key]})
    vs = list(it.rating_sampler(batch_size=10, segment="valid"))
    out["it_valid_batches"] = np.array([p.shape[1] for p, _ in vs], np.int32)

    # ---- sub-matrices / sub-graphs and the INDUCTIVE DataIterator (graph.py:493-538, 1001-1030; iterators.py:171-176).
    # Appended after everything else and drawn from its own generator, so the vectors above do not move. -----------
    rng2 = np.random.default_rng(2024)
    um = graph["user", "movie"]
    sub_rows = rng2.permutation(n_user)[:11].astype(np.int32)
    sub_cols = rng2.permutation(n_item)[:9].astype(np.int32)
    out.update(sm_rows=sub_rows, sm_cols=sub_cols)
    for tag, kw in (("sm_r_", dict(row_ids=sub_rows)), ("sm_c_", dict(col_ids=sub_cols)),
                    ("sm_rc_", dict(row_ids=sub_rows, col_ids=sub_cols))):
        m = um.submat_by_id(**kw)
        out.update({tag + "ep": m.end_points, tag + "ip": m.ind_ptr, tag + "val": m.values, tag + "rid": m.row_ids,
                    tag + "cid": m.col_ids})
    # inductive split over the items: every item is a train, validation or test node
    perm = rng2.permutation(n_item).astype(np.int32)
    test_ids, valid_ids, train_ids = perm[:4], perm[4:8], perm[8:]
    mu = graph["movie", "user"]

    def held_out_pairs(ids, frac):
        cols = []
        for i in ids:
            p = mu.submat_by_id(row_ids=np.array([i], np.int32)).node_pair_ids            # (movie, user) pairs
            take = rng2.permutation(p.shape[1])[:max(1, int(p.shape[1] * frac))]
            cols.append(np.stack([p[1, take], p[0, take]]))                                # -> (user, movie)
        return np.hstack(cols).astype(np.int32)

    ind_test_pairs, ind_valid_pairs = held_out_pairs(test_ids, 0.8), held_out_pairs(valid_ids, 0.8)
    out.update(ind_test_ids=test_ids, ind_valid_ids=valid_ids, ind_train_ids=train_ids, ind_test_pairs=ind_test_pairs,
               ind_valid_pairs=ind_valid_pairs)
    sg = graph.sel_subgraph_by_id("movie", train_ids)
    for tag, m in (("sg_um_", sg["user", "movie"]), ("sg_mu_", sg["movie", "user"])):
        out.update({tag + "ep": m.end_points, tag + "ip": m.ind_ptr, tag + "val": m.values, tag + "rid": m.row_ids,
                    tag + "cid": m.col_ids, tag + "sup": m.get_support(True)})
    it2 = ref_it["DataIterator"](graph, "user", "movie", is_inductive=True, test_node_pairs=ind_test_pairs,
                                 valid_node_pairs=ind_valid_pairs, inductive_key="movie",
                                 inductive_valid_ids=valid_ids, inductive_train_ids=train_ids,
                                 embed_P_mask=0.3, embed_p_zero=0.5, embed_p_self=0.5, seed=321)
    for tag, g_ in (("ind_train_", it2.train_graph), ("ind_val_", it2.val_graph), ("ind_test_", it2.test_graph)):
        for d, m in (("um_", g_["user", "movie"]), ("mu_", g_["movie", "user"])):
            out.update({tag + d + "ep": m.end_points, tag + d + "ip": m.ind_ptr, tag + d + "val": m.values,
                        tag + d + "rid": m.row_ids, tag + d + "cid": m.col_ids})
    out.update(ind_train_pairs=it2._train_node_pairs, ind_train_ratings=it2._train_ratings,
               ind_valid_ratings=it2._valid_ratings, ind_test_ratings=it2._test_ratings,
               ind_eval_noise_user=it2.evaluate_embed_noise_dict["user"],
               ind_eval_noise_movie=it2.evaluate_embed_noise_dict["movie"])
    rs2 = it2.rating_sampler(batch_size=20, segment="train")
    for k in range(2):
        p, r = next(rs2)
        out.update({"ind_rs%d_pairs" % k: p, "ind_rs%d_ratings" % k: r})
    ns2 = it2.recon_nodes_sampler(batch_size=3)
    for k in range(2):
        noise, batch, allr = next(ns2)
        for key in ("user", "movie"):
            out.update({"ind_ns%d_noise_%s" % (k, key): noise[key], "ind_ns%d_batch_%s" % (k, key): batch[key],
                        "ind_ns%d_all_%s" % (k, key): allr[key]})
    # full-neighbourhood plan arrays of the inductive train graph (unsorted rows of the column-selected direction)
    eps, vs_, ips, sps = it2.train_graph["user", "movie"].sample_neighbors(None, True, True, -1)
    for l in range(len(levels)):
        out.update({"ind_nb_ep%d" % l: eps[l], "ind_nb_ip%d" % l: ips[l], "ind_nb_sup%d" % l: sps[l]})

    np.savez_compressed(OUT, **{k: np.asarray(v) for k, v in out.items()})
    print("wrote %s: %d arrays, %d bytes" % (OUT, len(out), os.path.getsize(OUT)))


if __name__ == "__main__":
    main()
