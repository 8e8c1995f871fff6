"""Round-4 experiment: ways of shrinking the per-XCD working set of the two cache-resident launch classes of the ML-10M step
(68 MB plain user rows -> 106 770 (item, level) segments; 104 MB grouped (item, level) rows -> 69 878 user segments):

  1 launch, column slices 1 / 2 / 4 / 8
  the shipped form: two source-range phases = two launches (sg_seg_gather_sum_phased_hip), slices 4 / 8
  P source-range parts inside ONE launch + join pass (sg_seg_gather_sum_parts_hip over a plan.SourcePartition; parts ordered
  in TIME for P != 8: all XCDs work on part p before part p + 1), slices 4 / 8
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L  # noqa: E402
from star_gcn_amd.plan import SourcePartition  # noqa: E402
from tools.microbench import timeit  # noqa: E402

rng = np.random.default_rng(0)
nnz, C = 10_000_000, 256
lib = L.lib()
for name, S, T, sigma in (("(item,level)<-user rows (68 MB)", 106770, 69878, 1.0), ("users<-(item,level) rows (104 MB)", 69878, 106770, 1.5)):
    lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 2.0))
    indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    seg = np.repeat(np.arange(S), lens)
    w = torch.rand(nnz).cuda()
    x = torch.randn(T, C, device="cuda")
    out = torch.empty(S, C, device="cuda")
    pop = rng.lognormal(0.0, sigma, T)
    idx = rng.choice(T, size=nnz, p=pop / pop.sum()).astype(np.int64)
    order = np.lexsort((idx, seg))
    idx_d = torch.from_numpy(idx[order].astype(np.int32)).cuda()
    ref = None
    for sl in (1, 2, 4, 8):
        lib.sg_gather_tuning(-1, sl)
        t = timeit(lambda: ops.gather_sum(out, x, idx_d, indptr, w, S, C))
        print("%-36s 1 launch, slices %d              %7.3f ms" % (name, sl, t * 1e3), flush=True)
        if ref is None:
            ref = out.clone()
    # shipped: two launches
    idx_p = torch.empty(nnz, dtype=torch.int32, device="cuda")
    wpos_p = torch.empty(nnz, dtype=torch.int32, device="cuda")
    indptr_p = torch.empty(2 * (S + 1), dtype=torch.int32, device="cuda")
    nnz_p = torch.empty(2, dtype=torch.int32, device="cuda")
    ws, wsn = L.workspace(lib.sg_gather_phases_workspace_bytes(nnz), x.device)
    L.check(lib.sg_gather_phases_build_hip(L.ptr(idx_p), L.ptr(wpos_p), L.ptr(indptr_p), L.ptr(nnz_p), L.ptr(idx_d), L.ptr(indptr), S,
                                           nnz, T, L.ptr(ws), wsn, L.stream_ptr()), "phases")
    n0, n1 = (int(v) for v in nnz_p.cpu())
    ph = L.GatherPhasesStruct()
    ph.num_phases, ph.idx, ph.wpos, ph.indptr = 2, idx_p.data_ptr(), wpos_p.data_ptr(), indptr_p.data_ptr()
    ph.nnz_p[0], ph.nnz_p[1] = n0, n1
    import ctypes
    gws, gwsn = L.workspace(lib.sg_seg_weighted_pool_workspace_bytes(1, S, nnz, C), x.device)

    def phased():
        L.check(lib.sg_seg_gather_sum_phased_hip(L.ptr(out), 1, C, L.ptr(x), 1, C, L.ptr(w), ctypes.byref(ph), S, C, 1, 0, 0.0,
                                                 L.ptr(gws), gwsn, L.stream_ptr(), T * C * 4), "phased")
    for sl in (4, 8):
        lib.sg_gather_tuning(-1, sl)
        t = timeit(phased)
        err = float((out - ref).abs().max() / ref.abs().max())
        print("%-36s 2 phases = 2 launches, slices %d   %7.3f ms   rel diff %.1e" % (name, sl, t * 1e3, err), flush=True)
    for P in (2, 4, 8, 16):
        sp = SourcePartition(indptr, idx_d, T, parts=P)
        for sl in (4, 8):
            lib.sg_gather_tuning(-1, sl)
            t = timeit(lambda: ops.gather_sum_parts(out, x, sp, w, C))
            err = float((out - ref).abs().max() / ref.abs().max())
            print("%-36s %2d parts in ONE launch + join, slices %d%s %7.3f ms   rel diff %.1e" %
                  (name, P, sl, " (spatial: XCD x = part x)" if P == 8 else "", t * 1e3, err), flush=True)
        del sp
    lib.sg_gather_tuning(-1, 0)
