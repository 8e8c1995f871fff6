"""Round 5: the fused aggregate -> contract kernel (csrc/agg_fused.hip) -- correctness against float64 on small graphs and
timing against the unfused pair (gather into the R-expanded matrix + the 256-wide GEMM).
  python tools/exp_r5_fused.py check                        six graphs (hub row, empty segments, both weight orientations) vs float64
  python tools/exp_r5_fused.py time [n_dst n_src nnz R]     uniform graph of the config-5 shard's size: fused / fused + saved
                                                            aggregates / gather + GEMM
  python tools/exp_r5_fused.py bench-graph                  fused forward of either direction on bench.py's config-5 shard graph
                                                            (honours SG_FUSED_ABLATE, SG_FUSED_NT, SG_LIB_OVERRIDE)
  python tools/exp_r5_fused.py ml10m                        forward of either direction at the MovieLens-10M shape, all orders
Results: profiles/r5_fused_kernel.md."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from star_gcn_amd import _lib as L      # noqa: E402
from star_gcn_amd import ops            # noqa: E402

D = 256


def make_graph(n_dst, n_src, nnz, R, dev, seed=0, hub=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    dst = torch.randint(0, n_dst, (nnz,), device=dev, generator=g)
    if hub:
        dst[:hub] = n_dst // 3
    lvl = torch.randint(0, R, (nnz,), device=dev, generator=g)
    key = dst * R + lvl
    key, _ = torch.sort(key)
    counts = torch.bincount(key, minlength=n_dst * R)
    indptr = torch.zeros(n_dst * R + 1, dtype=torch.int32, device=dev)
    indptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    idx = torch.randint(0, n_src, (nnz,), device=dev, generator=g, dtype=torch.int32)
    w = torch.rand(nnz, device=dev, generator=g) + 0.1
    return indptr, idx, w


def build_plan(indptr, idx, w, n_dst, R, order=None, with_pos=False):
    lib = L.lib()
    dev = idx.device
    tiles = lib.sg_agg_fused_tiles(n_dst)
    f_ptr = torch.empty(tiles * R * 65, dtype=torch.int32, device=dev)
    f_idx = torch.empty_like(idx)
    f_w = torch.empty_like(w)
    f_pos = torch.empty_like(idx) if with_pos else None
    L.check(lib.sg_agg_fused_plan_build_hip(L.ptr(f_ptr), L.ptr(f_idx), L.ptr(f_w), L.ptr(f_pos), L.ptr(order), L.ptr(indptr),
                                            L.ptr(idx), L.ptr(w), n_dst, R, idx.numel(), L.stream_ptr()), "plan_build")
    return f_ptr, f_idx, f_w, f_pos


def fused(x, Ws, bs, rowsum, plan, order, n_dst, R, nnz, act, trans, zsave=None, nt=0):
    lib = L.lib()
    f_ptr, f_idx, f_w, _ = plan
    out = torch.empty(n_dst, D, dtype=torch.float32, device=x.device)
    ws, wsn = L.workspace(lib.sg_agg_fused_workspace_bytes(R), x.device)
    wp = ops._ptr_array(Ws)
    bp = ops._ptr_array(bs) if bs is not None else None
    L.check(lib.sg_agg_fused_hip(L.ptr(out), D, L.ptr(zsave), zsave.shape[1] if zsave is not None else 0, L.ptr(x), x.shape[1],
                                 wp, D, trans, bp, L.ptr(rowsum), L.ptr(f_ptr), L.ptr(f_idx), L.ptr(f_w), L.ptr(order), n_dst, x.shape[0], R,
                                 nnz, D, D, ops._act_id(act), 0.1, nt, L.ptr(ws), wsn, L.stream_ptr()), "sg_agg_fused_hip")
    return out


def reference(x, Ws, bs, indptr, idx, w, n_dst, R, act, trans):
    dev = x.device
    seg = torch.repeat_interleave(torch.arange(n_dst * R, device=dev), (indptr[1:] - indptr[:-1]).long())
    Z = torch.zeros(n_dst * R, D, dtype=torch.float64, device=dev)
    Z.index_add_(0, seg, x.double()[idx.long()] * w.double()[:, None])
    Z = Z.view(n_dst, R, D)
    rs = torch.zeros(n_dst * R, dtype=torch.float64, device=dev).index_add_(0, seg, w.double()).view(n_dst, R)
    out = torch.zeros(n_dst, D, dtype=torch.float64, device=dev)
    mag = torch.zeros(n_dst, D, dtype=torch.float64, device=dev)
    for r in range(R):
        B = Ws[r].double() if trans else Ws[r].double().t()
        out += Z[:, r] @ B
        mag += Z[:, r].abs() @ B.abs()
        if bs is not None:
            out += rs[:, r:r + 1] * bs[r].double()[None]
            mag += (rs[:, r:r + 1] * bs[r].double()[None]).abs()
    if act == "leaky":
        out = torch.where(out > 0, out, 0.1 * out)
    return out, mag, Z, rs


def check():
    dev = torch.device("cuda")
    torch.manual_seed(1)
    worst = 0.0
    for (n_dst, n_src, nnz, R, hub, trans, bias, act, perm, zs) in [
            (1000, 800, 60000, 5, 0, 0, True, "leaky", False, False),
            (1000, 800, 60000, 5, 3000, 1, False, None, True, True),
            (64, 50, 10, 1, 0, 0, True, None, False, True),
            (130, 4000, 200000, 16, 20000, 0, True, "leaky", True, False),
            (5000, 3000, 400000, 10, 0, 1, True, None, True, True),
            (33000, 9000, 3000000, 16, 100000, 0, True, "leaky", True, True)]:
        indptr, idx, w = make_graph(n_dst, n_src, nnz, R, dev, seed=nnz, hub=hub)
        x = torch.randn(n_src, D, device=dev) * torch.exp(torch.randn(n_src, 1, device=dev))
        Ws = [torch.randn(D, D, device=dev) / 16 for _ in range(R)]
        bs = [torch.randn(D, device=dev) for _ in range(R)] if bias else None
        tiles = (n_dst + 63) // 64
        order = torch.randperm(tiles, device=dev).to(torch.int32) if perm else None
        plan = build_plan(indptr, idx, w, n_dst, R, order, with_pos=True)
        ref, mag, Z, rs = reference(x, Ws, bs, indptr, idx, w, n_dst, R, act, trans)
        zsave = torch.full((n_dst, R * D + 64), -7.0, device=dev) if zs else None
        for nt in (0, 1):
            out = fused(x, Ws, bs, rs.float().contiguous() if bias else None, plan, order, n_dst, R, nnz, act, trans, zsave, nt)
            torch.cuda.synchronize()
            err = ((out.double() - ref).abs() / mag.clamp_min(1e-30)).max().item()
            worst = max(worst, err)
            msg = "n_dst %6d R %2d nnz %8d trans %d bias %d perm %d nt %d: max |err| / sum|a||b| = %.3g" % (n_dst, R, nnz, trans, bias, perm, nt, err)
            if zs:
                zerr = ((zsave[:, :R * D].double().view(n_dst, R, D) - Z).abs().max() / Z.abs().max()).item()
                msg += "   zsave rel err %.3g, pad untouched %s" % (zerr, bool((zsave[:, R * D:] == -7.0).all()))
            print(msg, flush=True)
        # refresh path: new weights through f_pos
        w2 = torch.rand_like(w) + 0.5
        f_ptr, f_idx, f_w, f_pos = plan
        L.check(L.lib().sg_agg_fused_refresh_hip(L.ptr(f_w), L.ptr(f_pos), L.ptr(w2), nnz, L.stream_ptr()), "refresh")
        ref2, mag2, _, rs2 = reference(x, Ws, bs, indptr, idx, w2, n_dst, R, act, trans)
        out2 = fused(x, Ws, bs, rs2.float().contiguous() if bias else None, plan, order, n_dst, R, nnz, act, trans)
        err2 = ((out2.double() - ref2).abs() / mag2.clamp_min(1e-30)).max().item()
        print("   after refresh: %.3g" % err2, flush=True)
        worst = max(worst, err2)
    print("WORST %.3g (%s)" % (worst, "ok" if worst < 2e-6 else "FAIL"))


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def time_case(n_dst, n_src, nnz, R, skew=False):
    dev = torch.device("cuda")
    indptr, idx, w = make_graph(n_dst, n_src, nnz, R, dev, seed=3)
    x = torch.randn(n_src, D, device=dev)
    Ws = [torch.randn(D, D, device=dev) / 16 for _ in range(R)]
    bs = [torch.randn(D, device=dev) for _ in range(R)]
    lens = (indptr[1:] - indptr[:-1]).view(n_dst, R).sum(1)
    tiles = (n_dst + 63) // 64
    pad = tiles * 64 - n_dst
    work = torch.cat([lens, lens.new_zeros(pad)]).view(tiles, 64).sum(1)
    order = torch.argsort(work, descending=True).to(torch.int32)
    rs = torch.zeros(n_dst * R, device=dev).index_add_(0, torch.repeat_interleave(
        torch.arange(n_dst * R, device=dev), (indptr[1:] - indptr[:-1]).long()), w).view(n_dst, R).contiguous()
    gb = nnz * (8 + 4 * D) / 1e9
    print("shape: n_dst %d n_src %d nnz %d R %d; algorithmic %.1f GB; max tile work %d, mean %.0f" %
          (n_dst, n_src, nnz, R, gb, int(work.max()), float(work.float().mean())), flush=True)
    for name, od in (("identity order", None), ("work-sorted order", order)):
        plan = build_plan(indptr, idx, w, n_dst, R, od)
        for nt in (0, 1):
            t = timeit(lambda: fused(x, Ws, bs, rs, plan, od, n_dst, R, nnz, "leaky", 0, None, nt))
            print("fused  %-18s nt %d: %.3f ms = %.2f TB/s algorithmic" % (name, nt, t, gb / t), flush=True)
        del plan
    plan = build_plan(indptr, idx, w, n_dst, R, order)
    zsave = torch.empty(n_dst, R * D + 64, device=dev)
    for nt in (0, 1):
        t = timeit(lambda: fused(x, Ws, bs, rs, plan, order, n_dst, R, nnz, "leaky", 0, zsave, nt))
        print("fused + zsave (work-sorted) nt %d: %.3f ms" % (nt, t), flush=True)
    del plan
    # the unfused pair: gather into the R-expanded matrix, then the contraction
    zext = zsave
    ld = R * D + 64
    t_g = timeit(lambda: ops.gather_sum(zext, x, idx, indptr, w, n_dst * R, D, dst_group=R, dst_ld=ld))
    wext = torch.randn(D, ld, device=dev) / 16
    out = torch.empty(n_dst, D, device=dev)
    t_m = timeit(lambda: ops.gemm(zext, wext, trans_b=True, act="leaky", out=out))
    print("unfused: gather %.3f ms (%.2f TB/s) + contraction %.3f ms = %.3f ms" % (t_g, gb / t_g, t_m, t_g + t_m), flush=True)


def bench_graph(nu=1_250_000, ni=1_000_000, ne=125_000_000, R=16):
    """the config-5 shard of bench.py (log-normal propensities, MovieLens-like level skew): fused forward of either direction"""
    from star_gcn_amd import functional as F
    from star_gcn_amd.device_graph import synthetic_device_graph
    dev = torch.device("cuda")
    dg = synthetic_device_graph(nu, ni, ne, R, dev, seed=5)
    gb = dg.nnz * (8 + 4 * D) / 1e9
    for dst in (dg.U, dg.I):
        plan = dg.plan(dst)
        lv = (plan.c_indptr[1:] - plan.c_indptr[:-1]).view(plan.n_dst, R).sum(0).float()
        x = torch.randn(plan.n_src, D, device=dev)
        Ws = [torch.randn(D, D, device=dev) / 16 for _ in range(R)]
        bs = [torch.randn(D, device=dev) for _ in range(R)]
        t = timeit(lambda: F.multilink_aggregate(x, Ws, bs, plan, accum="sum", act="leaky", order="fused"))
        work = (plan.c_indptr[R::R] - plan.c_indptr[:-1:R])
        tw = torch.cat([work, work.new_zeros((-plan.n_dst) % 64)]).view(-1, 64).sum(1)
        print("ablate %s  into %-5s: %.3f ms = %.2f TB/s; level shares %s; tile work max %d mean %.0f" % (
            os.environ.get("SG_FUSED_ABLATE", "0"), dst, t, gb / t, " ".join("%.3f" % v for v in (lv / lv.sum()).tolist()),
            int(tw.max()), float(tw.float().mean())), flush=True)


def ml10m_forward():
    """forward of either direction at the MovieLens-10M shape: fused against the unfused orders (is the tile kernel usable where
    the gathered matrix is small?)"""
    import star_gcn_amd.synthetic as S
    from star_gcn_amd import functional as F
    from star_gcn_amd.plan import MultiLinkPlan
    graph, eu, ei, vals = S.make_graph("ml-10m")
    for dst, src in (("user", "movie"), ("movie", "user")):
        m = graph[dst, src]
        eps, _, ips, sps = m.sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
        plan = MultiLinkPlan(eps, ips, sps, m.shape[1], "cuda")
        x = torch.randn(plan.n_src, D, device="cuda")
        Ws = [torch.randn(D, D, device="cuda") / 16 for _ in range(plan.R)]
        bs = [torch.randn(D, device="cuda") for _ in range(plan.R)]
        with torch.no_grad():
            for order in ("fused", "transform_first", "aggregate_first"):
                t = timeit(lambda: F.multilink_aggregate(x, Ws, bs, plan, accum="sum", act="leaky", order=order), n=9, warm=3)
                print("into %-5s (%d x %d, %d tiles) %-16s %.3f ms" % (dst, plan.n_dst, plan.n_src, (plan.n_dst + 63) // 64, order, t), flush=True)



def imbalance(nu=1_250_000, ni=1_000_000, ne=125_000_000, R=16, G=256, GW=8):
    """How evenly does the fused kernel's work split?  Per (tile, level) the edges go to the 8 gather waves at ROW boundaries (a
    row belongs to the wave whose share holds its first edge): a hub row is one wave's.  Prints, per direction of the config-5
    shard graph: sum over items of the busiest wave's edges against edges / 8 (1.0 = even), and the same per workgroup after
    the descending-work boustrophedon dealing (the launch ends with the slowest workgroup)."""
    from star_gcn_amd.device_graph import synthetic_device_graph
    dev = torch.device("cuda")
    dg = synthetic_device_graph(nu, ni, ne, R, dev, seed=5)
    for dst in (dg.U, dg.I):
        plan = dg.plan(dst)
        cnt = (plan.c_indptr[1:] - plan.c_indptr[:-1]).view(plan.n_dst, R).long()
        pad = (-plan.n_dst) % 64
        cnt = torch.cat([cnt, cnt.new_zeros(pad, R)]).view(-1, 64, R).permute(0, 2, 1).contiguous()      # (T, R, 64)
        T = cnt.shape[0]
        total = cnt.sum(2)                                                                              # (T, R)
        pre = cnt.cumsum(2) - cnt
        wj = torch.zeros_like(cnt)
        for q in range(1, GW):
            wj += (pre >= (total * q // GW).unsqueeze(2)).long()
        wj = torch.where(total.unsqueeze(2) > 0, wj, torch.zeros_like(wj))
        loads = torch.zeros(T, R, GW, dtype=torch.long, device=dev).scatter_add_(2, wj, cnt)
        busiest = loads.max(2).values                                                                   # (T, R)
        even = (total + GW - 1) // GW
        tile_edges = total.sum(1)
        order = torch.argsort(tile_edges, descending=True)
        slot = torch.arange(T, device=dev)
        ti, pos = slot // G, slot % G
        wg = torch.where(ti % 2 == 1, G - 1 - pos, pos)
        wg_of_tile = torch.empty(T, dtype=torch.long, device=dev)
        wg_of_tile[order] = wg
        per_wg_busy = torch.zeros(G, dtype=torch.long, device=dev).scatter_add_(0, wg_of_tile, busiest.sum(1))
        per_wg_even = torch.zeros(G, dtype=torch.long, device=dev).scatter_add_(0, wg_of_tile, even.sum(1))
        per_wg_edges = torch.zeros(G, dtype=torch.long, device=dev).scatter_add_(0, wg_of_tile, tile_edges)
        # the same tiles dealt stratum by stratum (G tiles of similar size) to the workgroups in ascending order of their load so far
        te = tile_edges[order].cpu()
        load = torch.zeros(G, dtype=torch.long)
        for s0 in range(0, T, G):
            chunk = te[s0:s0 + G]
            wgs = torch.argsort(load)[:chunk.numel()]
            load[wgs] += chunk
        print("      stratum-greedy dealing: per-workgroup edges max/mean %.4f (boustrophedon: %.4f)" % (
            float(load.max()) / float(load.float().mean()), float(per_wg_edges.max()) / float(per_wg_edges.float().mean())), flush=True)
        big = (cnt.max(2).values > even * 2).float().mean()
        print("into %-5s: %d tiles; busiest-wave edges / (edges / 8) over all items %.3f; per workgroup: edges max/mean %.3f, "
              "busiest-wave sum max / even mean %.3f (mean / even mean %.3f); items whose largest row exceeds 2 x its even share: %.3f; "
              "largest (row, level) %d edges" % (
                  dst, T, float(busiest.sum()) / float(even.sum()), float(per_wg_edges.max()) / float(per_wg_edges.float().mean()),
                  float(per_wg_busy.max()) / float(per_wg_even.float().mean()), float(per_wg_busy.float().mean()) / float(per_wg_even.float().mean()),
                  float(big), int(cnt.max())), flush=True)


def dir_ab(shapes):
    """fused against the better unfused order, PER DIRECTION, forward + backward through autograd, on bench.py's synthetic graph
    (log-normal propensities) at the given (users, items, ratings, levels) shapes: the data behind the routing rule of
    sg_multilink_agg_resolve_order2"""
    from star_gcn_amd import functional as F
    from star_gcn_amd.device_graph import synthetic_device_graph
    dev = torch.device("cuda")
    for nu, ni, ne, R in shapes:
        dg = synthetic_device_graph(nu, ni, ne, R, dev, seed=5)
        for dst in (dg.U, dg.I):
            plan = dg.plan(dst)
            x = torch.randn(plan.n_src, D, device=dev, requires_grad=True)
            Ws = [(torch.randn(D, D, device=dev) / 16).requires_grad_(True) for _ in range(R)]
            bs = [torch.randn(D, device=dev).requires_grad_(True) for _ in range(R)]
            gy = torch.randn(plan.n_dst, D, device=dev)
            res = {}
            for order in ("fused", "transform_first", "aggregate_first"):
                def step():
                    out = F.multilink_aggregate(x, Ws, bs, plan, accum="sum", act="leaky", order=order)
                    out.backward(gy)
                res[order] = timeit(step, n=5, warm=2)
            work = (plan.c_indptr[R::R] - plan.c_indptr[:-1:R])
            tw = torch.cat([work, work.new_zeros((-plan.n_dst) % 64)]).view(-1, 64).sum(1)
            unf = min(res["transform_first"], res["aggregate_first"])
            print("%9d x %9d %10d R %2d into %-5s: tiles %6d (%.2f per CU) n_dst/n_src %.2f expanded dst %5.0f MB src %5.0f MB  fused %8.3f  "
                  "unfused %8.3f (tf %.3f af %.3f)  fused/unfused %.2f  tile max/mean %.1f" % (
                      nu, ni, dg.nnz, R, dst, tw.numel(), tw.numel() / 256, plan.n_dst / plan.n_src, plan.n_dst * R * D * 4 / 2**20,
                      plan.n_src * R * D * 4 / 2**20, res["fused"], unf, res["transform_first"], res["aggregate_first"], res["fused"] / unf,
                      float(tw.max()) / float(tw.float().mean())), flush=True)
        del dg, plan
        torch.cuda.empty_cache()


def fuzz(n_cases=60):
    """random shapes (1 .. 700 rows, 1 .. 32 levels, 1 .. 150 000 edges) with heavy-tailed (Zipf) row and level popularity -- rows
    that are a level's whole share, levels with a handful of edges, empty tiles -- forward with saved aggregates, either operand
    orientation, against float64; exercises the equal-share cuts of the gather waves"""
    dev = torch.device("cuda")
    rng = torch.Generator(device="cpu").manual_seed(20240)
    worst, worst_z = 0.0, 0.0
    for case in range(n_cases):
        n_dst = int(torch.randint(1, 701, (1,), generator=rng))
        n_src = int(torch.randint(1, 3001, (1,), generator=rng))
        R = int(torch.randint(1, 33, (1,), generator=rng))
        nnz = int(10 ** (float(torch.rand(1, generator=rng)) * 5.17)) + 1
        a_row = 0.5 + 2.0 * float(torch.rand(1, generator=rng))
        a_lvl = 2.5 * float(torch.rand(1, generator=rng))
        g = torch.Generator(device=dev).manual_seed(case)
        pr = (torch.arange(1, n_dst + 1, device=dev, dtype=torch.float64) ** -a_row)[torch.randperm(n_dst, device=dev, generator=g)]
        pl = (torch.arange(1, R + 1, device=dev, dtype=torch.float64) ** -a_lvl)[torch.randperm(R, device=dev, generator=g)]
        dst = torch.multinomial(pr, nnz, replacement=True, generator=g)
        lvl = torch.multinomial(pl, nnz, replacement=True, generator=g)
        key, _ = torch.sort(dst * R + lvl)
        indptr = torch.zeros(n_dst * R + 1, dtype=torch.int32, device=dev)
        indptr[1:] = torch.cumsum(torch.bincount(key, minlength=n_dst * R), 0).to(torch.int32)
        idx = torch.randint(0, n_src, (nnz,), device=dev, generator=g, dtype=torch.int32)
        w = torch.rand(nnz, device=dev, generator=g) + 0.1
        x = torch.randn(n_src, D, device=dev, generator=g) * torch.exp(torch.randn(n_src, 1, device=dev, generator=g))
        Ws = [torch.randn(D, D, device=dev, generator=g) / 16 for _ in range(R)]
        bs = [torch.randn(D, device=dev, generator=g) for _ in range(R)]
        trans, act = case & 1, ("leaky" if case & 2 else None)
        tiles = (n_dst + 63) // 64
        order = torch.randperm(tiles, device=dev, generator=g).to(torch.int32) if case & 4 else None
        plan = build_plan(indptr, idx, w, n_dst, R, order)
        ref, mag, Z, rs = reference(x, Ws, bs, indptr, idx, w, n_dst, R, act, trans)
        zsave = torch.full((n_dst, R * D), -7.0, device=dev)
        out = fused(x, Ws, bs, rs.float().contiguous(), plan, order, n_dst, R, nnz, act, trans, zsave)
        torch.cuda.synchronize()
        err = ((out.double() - ref).abs() / mag.clamp_min(1e-30)).max().item()
        zerr = ((zsave.double().view(n_dst, R, D) - Z).abs().max() / Z.abs().max().clamp_min(1e-300)).item()
        big = int((indptr[1:] - indptr[:-1]).max())
        worst, worst_z = max(worst, err), max(worst_z, zerr)
        print("case %2d: %3d rows %2d levels %6d edges (largest (row, level) %6d) trans %d: out %.2e  saved aggregates %.2e%s" % (
            case, n_dst, R, nnz, big, trans, err, zerr, "   <-- FAIL" if not (err < 2e-6 and zerr < 2e-6) else ""), flush=True)
    print("WORST out %.3g aggregates %.3g (%s)" % (worst, worst_z, "ok" if worst < 2e-6 and worst_z < 2e-6 else "FAIL"))


if __name__ == "__main__":
    if sys.argv[1] == "check":
        check()
    elif sys.argv[1] == "bench-graph":
        bench_graph()
    elif sys.argv[1] == "ml10m":
        ml10m_forward()
    elif sys.argv[1] == "imbalance":
        imbalance()
    elif sys.argv[1] == "fuzz":
        fuzz(int(sys.argv[2]) if len(sys.argv) > 2 else 60)
    elif sys.argv[1] == "dir-ab":
        dir_ab([tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]])
    else:
        a = [int(v) for v in sys.argv[2:6]] if len(sys.argv) >= 6 else [1_000_000, 1_250_000, 125_000_000, 16]
        time_case(*a)
