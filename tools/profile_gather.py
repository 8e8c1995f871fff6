#!/usr/bin/env python3
"""Minimal driver for rocprofv3 PMC passes on the dominant kernel: builds the ML-10M-shaped plan and launches ONLY the
fused aggregation gathers (forward both directions) a few times, so counter collection stays cheap.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- python tools/profile_gather.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out -- python tools/profile_gather.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import star_gcn_amd.synthetic as S  # noqa: E402
from star_gcn_amd import ops  # noqa: E402
from star_gcn_amd.plan import MultiLinkPlan  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "ml-10m"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
D = 256
graph, eu, ei, vals = S.make_graph(shape)
dev = torch.device("cuda", 0)
for dst, src in (("user", "movie"), ("movie", "user")):
    m = graph[dst, src]
    eps, _, ips, sps = m.sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
    plan = MultiLinkPlan([m.col_id_to_ind(e) for e in eps], ips, sps, m.shape[1], dev)
    R = plan.R
    x = torch.randn(plan.n_src, D, device=dev)
    h = torch.randn(plan.n_src, R * D, device=dev)
    out = torch.empty(plan.n_dst, D, device=dev)
    zext = torch.empty(plan.n_dst, R * D + 16, device=dev)
    for _ in range(reps):
        # transform-first forward: grouped SOURCE rows (H is n_src x R*D), un-split destination CSR
        ops.gather_sum(out, h, plan.c_q, plan.d_indptr, plan.c_w, plan.n_dst, D, src_group=R, src_ld=R * D, act="leaky")
        # aggregate-first forward: grouped DESTINATION rows (Zext is n_dst x (R*D+16))
        ops.gather_sum(zext, x, plan.c_idx, plan.c_indptr, plan.c_w, plan.n_dst * R, D, dst_group=R, dst_ld=R * D + 16)
    torch.cuda.synchronize()
    print(dst, "<-", src, "edges", plan.nnz, "algorithmic bytes/launch", (8 + 4 * D) * plan.nnz)
