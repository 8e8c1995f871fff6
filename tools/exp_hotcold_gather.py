#!/usr/bin/env python3
"""Experiment: two gather launches split by source-row POPULARITY instead of source-row range: phase 0 = the edges whose
source row is among the H most referenced rows (their column slices fit the L2s), phase 1 = the rest.  ML-10M shapes."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import star_gcn_amd.synthetic as S
from star_gcn_amd import ops
from star_gcn_amd.plan import MultiLinkPlan
from tools.exp_phase_gather import timeit

def split(idx, indptr, w, hot_rows):
    idx, indptr, w = idx.cpu().numpy(), indptr.cpu().numpy().astype(np.int64), w.cpu().numpy()
    nseg = indptr.size - 1
    seg = np.repeat(np.arange(nseg), np.diff(indptr))
    ishot = hot_rows[idx[:seg.size]]
    out = []
    for m in (ishot, ~ishot):
        ip = np.concatenate([[0], np.cumsum(np.bincount(seg[m], minlength=nseg))]).astype(np.int32)
        out.append((torch.from_numpy(idx[:seg.size][m].copy()).cuda(), torch.from_numpy(ip).cuda(), torch.from_numpy(w[:seg.size][m].copy()).cuda()))
    return out, float(ishot.mean())

if __name__ == "__main__":
    graph, eu, ei, vals = S.make_graph("ml-10m")
    for name, (a, b) in {"users<-items": ("user", "movie"), "items<-users": ("movie", "user")}.items():
        m = graph[a, b]
        eps, _, ips, sps = m.sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
        plan = MultiLinkPlan(eps, ips, sps, m.shape[1], "cuda")
        R = plan.R
        cases = [("TF-type fwd (grouped src rows)", plan.c_q, plan.d_indptr, plan.c_w, plan.n_dst, plan.n_src * R, dict(src_group=R, src_ld=R * 256)),
                 ("AF-type fwd (dst grouped)", plan.c_idx, plan.c_indptr, plan.c_w, plan.n_dst * R, plan.n_src, dict(dst_group=R, dst_ld=R * 256))]
        for label, idx, indptr, w, nseg, nrows, kw in cases:
            if not (24 << 20) <= nrows * 1024 <= (256 << 20):
                continue
            src = torch.randn(nrows * 256, device="cuda").view(-1, (R * 256) if "src_group" in kw else 256)
            dst = torch.empty(nseg * 256, device="cuda").view(-1, (R * 256) if "dst_group" in kw else 256)
            print("%s | %s | source %.0f MB, %d segments" % (name, label, nrows * 1024 / 2**20, nseg))
            base = timeit(lambda: ops.gather_sum(dst, src, idx, indptr, w, nseg, 256, **kw))
            print("   1 launch            %7.3f ms" % base)
            deg = np.bincount(idx.cpu().numpy(), minlength=nrows)
            order = np.argsort(-deg, kind="stable")
            for H in (8192, 12288, 16384, 24576, 32768):
                hot = np.zeros(nrows, bool); hot[order[:H]] = True
                ph, frac = split(idx, indptr, w, hot)
                def run():
                    for p, (i_, ip_, w_) in enumerate(ph):
                        ops.gather_sum(dst, src, i_, ip_, w_, nseg, 256, req=ops.REQ_WRITE if p == 0 else ops.REQ_ADD, **kw)
                t = timeit(run)
                t0 = timeit(lambda: ops.gather_sum(dst, src, ph[0][0], ph[0][1], ph[0][2], nseg, 256, **kw))
                print("   hot %5d rows (%.2f of the edges)  %7.3f ms  (%.2f of 1 launch; hot phase alone %.3f)" % (H, frac, t, t / base, t0))
