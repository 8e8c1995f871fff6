#!/usr/bin/env python3
"""Experiment: would column-slicing the gather across XCDs raise the L2 hit rate enough to pay?
Each XCD would own a 32-float (128-byte = one cache line) slice of every source row, so its private 4 MB L2 faces a
working set of n_rows x 128 B instead of n_rows x 1 KiB.  Proxy with the existing kernel: ONE launch that reads only a
C-float slice of every row (row stride still 256 floats) has the same per-XCD working set and hit rate as the proposed
kernel; the proposal would cost (256 / C) x that launch."""
import sys, os, time
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

graph, eu, ei, vals = S.make_graph(sys.argv[1] if len(sys.argv) > 1 else "ml-10m")
for name, (a, b) in {"users<-items (reads (item,level) rows)": ("user", "movie"), "items<-users (reads user rows)": ("movie", "user")}.items():
    m = graph[a, b]
    eps, _, ips, sps = m.sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
    plan = MultiLinkPlan(eps, ips, sps, m.shape[1], "cuda")
    R = plan.R
    for label, idx, indptr, w, nseg, nrows in (("fwd TF-type: idx=c_q over (n_src*R) rows", plan.c_q, plan.d_indptr, plan.c_w, plan.n_dst, plan.n_src * R),
                                                ("fwd AF-type: idx=c_idx over n_src rows", plan.c_idx, plan.c_indptr, plan.c_w, plan.n_dst * R, plan.n_src)):
        src = torch.randn(nrows, 256, device="cuda")
        dst = torch.empty(nseg, 256, device="cuda")
        print("%s | %s | source %.0f MB, %d segments" % (name, label, nrows * 1024 / 2**20, nseg))
        full = timeit(lambda: ops.gather_sum(dst, src, idx, indptr, w, nseg, 256))
        print("   C=256 full rows              %7.3f ms" % full)
        for C in (128, 64, 32):
            t = timeit(lambda: ops.gather_sum(dst, src, idx, indptr, w, nseg, C, dst_ld=256, src_ld=256))
            print("   C=%3d slice (ld 256)         %7.3f ms  x%d = %7.3f ms  (%.2f of full)" % (C, t, 256 // C, t * 256 // C, t * (256 // C) / full))
