#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (HIP events, median of N): fp32 MFMA GEMM shapes of the ML-10M step,
the gather kernel at cache-resident and HBM-bound sizes, and seg_take_k_corr.  Development aid; not a test."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops  # noqa: E402


def timeit(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    return float(np.median(ts))


def gemm_cases():
    nu, ni, R, D = 69878, 10677, 10, 256
    ld = (R * D + R + 63) // 64 * 64
    cases = [("TF fwd   H=X Wcat^T", ni, R * D, D, False, True), ("AF fwd   Zext Wext^T", ni, D, ld, False, True),
             ("out_fc user", nu, D, D, False, True), ("dX = dH Wcat (NN)", ni, D, R * D, False, False),
             ("dW = dH^T X (TN)", R * D, D, ni, True, False), ("dZ = dpre Wext (NN)", ni, ld, D, False, False),
             ("dWext = dpre^T Zext (TN)", D, ld, ni, True, False), ("dW out_fc user (TN)", D, D, nu, True, False),
             ("square 4096", 4096, 4096, 4096, False, True)]
    for name, M, N, K, ta, tb in cases:
        a = torch.randn((K, M) if ta else (M, K), device="cuda")
        b = torch.randn((N, K) if tb else (K, N), device="cuda")
        from star_gcn_amd import _lib as L
        res = []
        for be, nm in ((0, "fp32"), (2, "x6v2"), (3, "f16x3")):
            L.lib().sg_gemm_backend(be)
            t = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb))
            res.append("%s %6.3f ms %5.1f TF/s" % (nm, t * 1e3, 2.0 * M * N * K / t / 1e12))
        L.lib().sg_gemm_backend(-1)
        tt = timeit(lambda: torch.matmul(a.t() if ta else a, b.t() if tb else b))
        print("gemm %-26s M=%6d N=%5d K=%6d  %s   (torch %6.3f ms %5.1f TF/s)" %
              (name, M, N, K, "  ".join(res), tt * 1e3, 2.0 * M * N * K / tt / 1e12), flush=True)


def gemm_big_cases():
    """GEMM shapes of the 1-GPU shard of BASELINE config 5 (1.25 M x 1 M nodes, R = 16, dim 256), tile-height sweep"""
    nu, ni, R, D = 1250000, 1000000, 16, 256
    ld = (R * D + R + 63) // 64 * 64
    cases = [("c5 TF fwd H=X Wcat^T", ni, R * D, D, False, True), ("c5 AF fwd Zext Wext^T", ni, D, ld, False, True),
             ("c5 out_fc user", nu, D, D, False, True), ("c5 dX = dH Wcat (NN)", ni, D, R * D, False, False),
             ("c5 dW = dH^T X (TN)", R * D, D, ni, True, False), ("c5 dZ = dpre Wext (NN)", ni, ld, D, False, False),
             ("c5 dWext (TN)", D, ld, ni, True, False)]
    for name, M, N, K, ta, tb in cases:
        a = torch.randn((K, M) if ta else (M, K), device="cuda")
        b = torch.randn((N, K) if tb else (K, N), device="cuda")
        res = []
        from star_gcn_amd import _lib as L
        for be, nm in ((2, "x6v2"), (3, "f16x3")):
            L.lib().sg_gemm_backend(be)
            t = timeit(lambda: ops.gemm(a, b, trans_a=ta, trans_b=tb), n=5, warm=2)
            res.append("%s %8.3f ms %6.1f TF/s" % (nm, t * 1e3, 2.0 * M * N * K / t / 1e12))
        L.lib().sg_gemm_backend(-1)
        tt = timeit(lambda: torch.matmul(a.t() if ta else a, b.t() if tb else b), n=5, warm=2)
        print("gemm %-24s M=%7d N=%5d K=%7d  %s   (torch %8.3f ms %6.1f TF/s)" %
              (name, M, N, K, "  ".join(res), tt * 1e3, 2.0 * M * N * K / tt / 1e12), flush=True)
        del a, b


def gather_cases():
    g = torch.Generator().manual_seed(0)
    for name, S, T, nnz, C in [("ml-10m users<-items", 69878, 10677, 10_000_000, 256),
                               ("ml-10m items<-users", 10677, 69878, 10_000_000, 256),
                               ("hbm-bound 2M src rows", 400_000, 2_000_000, 20_000_000, 256),
                               ("ml-1m C=128", 6040, 3706, 1_000_000, 128), ("pair width 64", 69878, 10677, 10_000_000, 64)]:
        lens = torch.distributions.Multinomial(nnz, torch.rand(S, generator=g) ** 2 + 1e-3).sample().long()
        indptr = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)]).int().cuda()
        idx = torch.randint(0, T, (nnz,), generator=g).int().cuda()
        w = torch.rand(1, nnz, generator=g).cuda()
        x = torch.randn(1, T, C, device="cuda")
        out = torch.empty(1, S, C, device="cuda")
        t = timeit(lambda: ops.seg_weighted_pool(x, w, idx, indptr, out=out))
        by = (8 + 4 * C) * nnz
        print("gather %-24s S=%7d T=%8d nnz=%9d C=%3d  %7.3f ms  %7.1f GB/s algorithmic (%.2f of 8 TB/s)" %
              (name, S, T, nnz, C, t * 1e3, by / t / 1e9, by / t / 8e12))
        if C == 64:
            e1 = torch.randn(1, S, C, device="cuda")
            t = timeit(lambda: ops.seg_take_k_corr(e1, x, idx, indptr))
            print("take_k_corr %-19s %7.3f ms  %7.1f GB/s (4C+8 B/edge)" % (name, t * 1e3, (4 * C + 8) * nnz / t / 1e9))


def small_cases():
    """The step's small kernels: rating-head inner products, long split-K weight gradients, Dense backward's
    activation + bias gradient."""
    g = torch.Generator().manual_seed(0)
    S, T, nnz, C = 69878, 10677, 10_000_000, 64
    lens = torch.distributions.Multinomial(nnz, torch.rand(S, generator=g) ** 2 + 1e-3).sample().long()
    indptr = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)]).int().cuda()
    idx = torch.randint(0, T, (nnz,), generator=g).int().cuda()
    x = torch.randn(1, T, C, device="cuda")
    e1 = torch.randn(1, S, C, device="cuda")
    t = timeit(lambda: ops.seg_take_k_corr(e1, x, idx, indptr))
    print("take_k_corr 10M pairs C=64        %7.3f ms  %7.1f GB/s (4C+8 B/edge)" % (t * 1e3, (4 * C + 8) * nnz / t / 1e9))
    from star_gcn_amd.plan import SourcePartition, TransposePlan
    tp = TransposePlan(idx, indptr, T, idx.device)
    gw = torch.randn(1, nnz, device="cuda")
    t1 = timeit(lambda: ops.seg_weighted_pool_bwd_data(gw, e1, tp, T))
    sp = SourcePartition(tp.t_indptr, tp.t_seg, S, pos=tp.t_pos, parts=8)
    out = torch.empty(T, C, device="cuda")
    t2 = timeit(lambda: ops.gather_sum_parts(out, e1[0], sp, gw, C))
    print("rating head item-side gradient (10M x 256 B from 17.9 MB): plain %7.3f ms, source-partitioned %7.3f ms" %
          (t1 * 1e3, t2 * 1e3))
    for M, N, K in [(256, 256, 69878), (64, 256, 69878), (256, 256, 10677), (2560, 256, 10677)]:
        a = torch.randn(K, M, device="cuda")
        b = torch.randn(K, N, device="cuda")
        t = timeit(lambda: ops.gemm(a, b, trans_a=True))
        print("gemm TN (weight gradient) M=%5d N=%4d K=%6d  %7.3f ms  %6.1f TF/s" % (M, N, K, t * 1e3, 2.0 * M * N * K / t / 1e12))
    for M, N in [(69878, 256), (10677, 256)]:
        dy = torch.randn(M, N, device="cuda")
        y = torch.randn(M, N, device="cuda")
        t1 = timeit(lambda: ops.colsum(ops.act_bwd(dy, y, "leaky", 0.1)))
        t2 = timeit(lambda: ops.act_bwd_colsum(dy, y, "leaky", 0.1))
        print("Dense backward %6d x %3d: act_bwd + colsum %7.3f ms, fused %7.3f ms" % (M, N, t1 * 1e3, t2 * 1e3))


def segop_cases():
    """The rest of the seg-op surface (API parity ops, not executed by STAR-GCN training) at ML-10M-like sizes, for both
    segment granularities of the step: 69 878 user segments (143 edges on average) and 698 780 (user, level) segments (14)."""
    g = torch.Generator().manual_seed(0)
    nnz, T, C = 10_000_000, 10677, 256
    for S in (69878, 698780):
        lens = torch.distributions.Multinomial(nnz, torch.rand(S, generator=g) ** 2 + 1e-3).sample().long()
        indptr = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)]).int().cuda()
        idx = torch.randint(0, T, (nnz,), generator=g).int().cuda()
        d = torch.randn(1, nnz, device="cuda")
        t = timeit(lambda: ops.seg_sum(d, indptr))
        print("S=%6d seg_sum            %7.3f ms  %7.1f GB/s" % (S, t * 1e3, 4 * nnz / t / 1e9))
        t = timeit(lambda: ops.seg_softmax(d, indptr))
        print("S=%6d seg_softmax        %7.3f ms  %7.1f GB/s (read + write)" % (S, t * 1e3, 8 * nnz / t / 1e9))
        sm = ops.seg_softmax(d, indptr)
        t = timeit(lambda: ops.seg_softmax_bwd(d, sm, indptr))
        print("S=%6d seg_softmax_bwd    %7.3f ms  %7.1f GB/s (2 reads + write)" % (S, t * 1e3, 12 * nnz / t / 1e9))
        lhs = torch.randn(1, nnz, device="cuda")
        rhs = torch.randn(1, S, device="cuda")
        t = timeit(lambda: ops.seg_broadcast(lhs, rhs, indptr, 0))
        print("S=%6d seg_broadcast_add  %7.3f ms  %7.1f GB/s (read + write)" % (S, t * 1e3, 8 * nnz / t / 1e9))
        x = torch.randn(1, T, C, device="cuda")
        from star_gcn_amd.plan import TransposePlan
        tp = TransposePlan(idx, indptr, T, idx.device)
        og = torch.randn(1, S, C, device="cuda")
        for pt in ("sum", "max"):
            t = timeit(lambda: ops.seg_pool(x, idx, indptr, pt), n=5, warm=2)
            print("S=%6d seg_pool %-3s C=256   %7.3f ms  %7.1f GB/s algorithmic (4C + 4 B per edge)" %
                  (S, pt, t * 1e3, (4 * C + 4) * nnz / t / 1e9))
            _, arg = ops.seg_pool(x, idx, indptr, pt)
            t = timeit(lambda: ops.seg_pool_bwd(og, arg, indptr, tp, T, pt), n=5, warm=2)
            print("S=%6d seg_pool %-3s bwd     %7.3f ms" % (S, pt, t * 1e3))


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "gather"]
    if "gemm" in which:
        gemm_cases()
    if "gemm5" in which:
        gemm_big_cases()
    if "gather" in which:
        gather_cases()
    if "small" in which:
        small_cases()
    if "segops" in which:
        segop_cases()
