"""does a destination / source ORDERING of the ML-10M-shaped plan raise the L2 hit rate of the gathers?
Relabels users and items of the synthetic graph by descending degree (and, as a control, by a random permutation) before
anything else sees it -- a consistent relabelling is just another input graph, every check still holds -- and times the step
and the gather launches exactly as bench.py does."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import star_gcn_amd.synthetic as S  # noqa: E402
from star_gcn_amd.mxgraph.graph import CSRMat, HeterGraph  # noqa: E402

orig = S.make_graph


def relabelled(mode):
    def make(shape="ml-10m", **kw):
        graph, eu, ei, vals = orig(shape, **kw)
        csr = graph["user", "movie"]
        nu, ni = csr.shape
        du, di = np.bincount(eu, minlength=nu), np.bincount(ei, minlength=ni)
        rng = np.random.default_rng(1)
        if mode == "degree":
            pu, pi = np.argsort(-du, kind="stable"), np.argsort(-di, kind="stable")
        elif mode == "random":
            pu, pi = rng.permutation(nu), rng.permutation(ni)
        else:
            return graph, eu, ei, vals
        inv_u, inv_i = np.empty(nu, np.int64), np.empty(ni, np.int64)
        inv_u[pu], inv_i[pi] = np.arange(nu), np.arange(ni)
        u2, i2 = inv_u[eu].astype(np.int32), inv_i[ei].astype(np.int32)
        order = np.lexsort((i2, u2))
        c2 = CSRMat.from_edges(u2[order], i2[order], vals[order], nu, ni, multi_link=csr.multi_link)
        g2 = HeterGraph({"user": np.arange(nu, dtype=np.int32), "movie": np.arange(ni, dtype=np.int32)}, {("user", "movie"): c2})
        return g2, c2.edge_row_indices, c2.end_points, c2.values
    return make


dev = torch.device("cuda", 0)
for mode in ("as generated", "random", "degree"):
    S.make_graph = relabelled(mode)
    c = bench.main_case("ml-10m", 256, "auto", dev)
    elapsed, loss, timeline = bench.timed_steps(c.step, 20, 5, dev, False)
    roof = bench.gather_roofline(timeline, c.E_local, 256, 20)
    cls = sorted(roof["_classes"].items())
    print("%-13s step %.3f ms  loss %.6f  gather per aggregation %.3f ms  classes %s" % (
        mode, elapsed / 20 * 1e3, float(loss), roof["avg_aggregation_ms"],
        ["%d MB%s x%d: %.3f ms" % (sb >> 20, " phased" if ph else "", n // 20, t * 1e3) for (sb, ph), (n, t, e) in cls]), flush=True)
    del c
    torch.cuda.empty_cache()
