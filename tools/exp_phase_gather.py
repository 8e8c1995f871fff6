#!/usr/bin/env python3
"""Experiment: split a gather into P launches by SOURCE-ROW RANGE (phase p only touches source rows [p T/P, (p+1) T/P),
later phases accumulate with req=add).  Together with the 4-way column slicing each XCD's L2 then faces a working set
of  T x 256 B / P  during a launch.  Measures the summed time of the P launches."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import star_gcn_amd.synthetic as S
from star_gcn_amd import ops
from star_gcn_amd.plan import MultiLinkPlan

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

def phases(idx, indptr, w, T, P):
    """split CSR (idx, indptr, w) into P CSRs by source range; edges keep their order inside a segment"""
    idx, indptr, w = idx.cpu().numpy(), indptr.cpu().numpy().astype(np.int64), w.cpu().numpy()
    nseg = indptr.size - 1
    seg = np.repeat(np.arange(nseg), np.diff(indptr))
    ph = np.minimum(idx[:seg.size].astype(np.int64) * P // T, P - 1)
    out = []
    for p in range(P):
        m = ph == p
        ip = np.concatenate([[0], np.cumsum(np.bincount(seg[m], minlength=nseg))]).astype(np.int32)
        out.append((torch.from_numpy(idx[:seg.size][m].copy()).cuda(), torch.from_numpy(ip).cuda(),
                    torch.from_numpy(w[:seg.size][m].copy()).cuda()))
    return out

graph, eu, ei, vals = S.make_graph("ml-10m")
for name, (a, b) in {"users<-items": ("user", "movie"), "items<-users": ("movie", "user")}.items():
    m = graph[a, b]
    eps, _, ips, sps = m.sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
    plan = MultiLinkPlan(eps, ips, sps, m.shape[1], "cuda")
    R = plan.R
    for label, idx, indptr, w, nseg, nrows, kw in (
            ("TF-type fwd (grouped src rows)", plan.c_q, plan.d_indptr, plan.c_w, plan.n_dst, plan.n_src * R, dict(src_group=R, src_ld=R * 256)),
            ("AF-type fwd (dst grouped)", plan.c_idx, plan.c_indptr, plan.c_w, plan.n_dst * R, plan.n_src, dict(dst_group=R, dst_ld=R * 256))):
        src = torch.randn(nrows * 256, device="cuda").view(-1, (R * 256) if "src_group" in kw else 256)
        dst = torch.empty(nseg * 256, device="cuda").view(-1, (R * 256) if "dst_group" in kw else 256)
        print("%s | %s | source %.0f MB, %d segments" % (name, label, nrows * 1024 / 2**20, nseg))
        base = timeit(lambda: ops.gather_sum(dst, src, idx, indptr, w, nseg, 256, **kw))
        print("   1 phase   %7.3f ms" % base)
        ref = dst.clone()
        for P in (2, 3, 4):
            ph = phases(idx, indptr, w, nrows, P)
            def run():
                for p, (i_, ip_, w_) in enumerate(ph):
                    ops.gather_sum(dst, src, i_, ip_, w_, nseg, 256, req=ops.REQ_WRITE if p == 0 else ops.REQ_ADD, **kw)
            t = timeit(run)
            err = float((dst - ref).abs().max() / ref.abs().max())
            print("   %d phases  %7.3f ms  (%.2f of 1 phase)  rel diff %.1e" % (P, t, t / base, err))
