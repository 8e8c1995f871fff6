"""Source-range phases of the aggregation gathers (sg_gather_phases_build_hip, sg_seg_gather_sum_phased_hip, the
`phases` of sg_multilink_plan): the device builder against a numpy stable partition, the phased gather against the
single launch, and the fused aggregator (both orders, both accumulations, forward and backward) with and without phases."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _graph(n_dst, n_src, nnz, R, seed):
    rng = np.random.default_rng(seed)
    dst = np.sort(rng.integers(0, n_dst, nnz))
    src = rng.integers(0, n_src, nnz)
    lvl = rng.integers(0, R, nnz)
    indptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n_dst))]).astype(np.int32)
    sup = rng.random(nnz).astype(np.float32) + 0.1
    return indptr, src.astype(np.int32), lvl.astype(np.int32), sup


def test_phase_builder_is_a_stable_partition_by_source_half():
    from star_gcn_amd import _lib as L
    rng = np.random.default_rng(0)
    for n_seg, n_rows, nnz in [(50, 7, 300), (1, 2, 5), (1000, 999, 40000), (13, 100, 0), (3000, 4, 20000)]:
        seg = np.sort(rng.integers(0, n_seg, nnz))
        indptr = np.concatenate([[0], np.cumsum(np.bincount(seg, minlength=n_seg))]).astype(np.int32)
        idx = rng.integers(0, n_rows, nnz).astype(np.int32)
        d_idx, d_ip = torch.from_numpy(idx).cuda(), torch.from_numpy(indptr).cuda()
        idx_p = torch.full((max(nnz, 1),), -1, dtype=torch.int32, device="cuda")
        wpos_p = torch.full((max(nnz, 1),), -1, dtype=torch.int32, device="cuda")
        ip_p = torch.empty(2 * (n_seg + 1), dtype=torch.int32, device="cuda")
        nnz_p = torch.empty(2, dtype=torch.int32, device="cuda")
        lib = L.lib()
        ws, wsn = L.workspace(lib.sg_gather_phases_workspace_bytes(nnz), d_idx.device)
        L.check(lib.sg_gather_phases_build_hip(L.ptr(idx_p), L.ptr(wpos_p), L.ptr(ip_p), L.ptr(nnz_p), L.ptr(d_idx), L.ptr(d_ip),
                                               n_seg, nnz, n_rows, L.ptr(ws), wsn, L.stream_ptr()), "build")
        split = (n_rows + 1) // 2
        low = idx < split
        order = np.concatenate([np.flatnonzero(low), np.flatnonzero(~low)])          # stable partition
        assert nnz_p.cpu().tolist() == [int(low.sum()), int((~low).sum())]
        assert np.array_equal(wpos_p.cpu().numpy()[:nnz], order)
        assert np.array_equal(idx_p.cpu().numpy()[:nnz], idx[order])
        ip = ip_p.cpu().numpy().reshape(2, n_seg + 1)
        assert np.array_equal(ip[0], np.concatenate([[0], np.cumsum(np.bincount(seg[low], minlength=n_seg))]))
        assert np.array_equal(ip[1], np.concatenate([[0], np.cumsum(np.bincount(seg[~low], minlength=n_seg))]))


def _plan(n_dst, n_src, nnz, R, seed):
    from star_gcn_amd.plan import MultiLinkPlan
    indptr, src, lvl, sup = _graph(n_dst, n_src, nnz, R, seed)
    return MultiLinkPlan.from_device_csr(torch.from_numpy(indptr).cuda(), torch.from_numpy(src).cuda(),
                                         torch.from_numpy(lvl).cuda(), torch.from_numpy(sup).cuda(), n_src, R)


@pytest.mark.parametrize("order", ["transform_first", "aggregate_first"])
@pytest.mark.parametrize("accum", ["sum", "stack"])
def test_fused_aggregator_with_phases_equals_single_launches(order, accum):
    """Sources of 40-120 MB (the range in which the library phases a launch): the phased plan must take the phased path
    (its phase arrays exist) and agree with the same plan without phases to fp32 association, forward and backward."""
    from star_gcn_amd import ops
    from star_gcn_amd import _lib as L
    n_dst, n_src, nnz, R, D, U = 30000, 45000, 1_300_000, 3, 256, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n_src, D, generator=g).cuda()
    ws = [(torch.randn(U, D, generator=g) / 16).cuda() for _ in range(R)]
    bs = [torch.randn(U, generator=g).cuda() for _ in range(R)]
    res = []
    for phased in (False, True):
        plan = _plan(n_dst, n_src, nnz, R, seed=11)
        if not phased:
            plan._phases = {v: None for v in range(L.NUM_VIEWS)}        # "nothing to build": single launches
        out, saved = ops.multilink_agg_fwd(x, ws, bs, plan, accum, "leaky", 0.1, order)
        dout = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).cuda()
        dx, dws, dbs = ops.multilink_agg_bwd(dout, out, saved, x, ws, plan, accum, "leaky", 0.1, order, True, True, True)
        torch.cuda.synchronize()
        built = [v for v, p in plan._phases.items() if p is not None]
        assert bool(built) == phased and (not phased or len(built) == 2)      # the forward's and the backward's view
        if phased:
            for v in built:
                n0, n1 = plan._phases[v][3:]
                assert n0 + n1 == nnz and n0 > 0 and n1 > 0
        res.append((out, dx, dws, dbs))
    (o0, x0, w0, b0), (o1, x1, w1, b1) = res
    def close(a, b, what):
        err = float((a - b).abs().max() / b.abs().max())
        assert err <= 2e-6, (what, err)
    close(o1, o0, "out")
    close(x1, x0, "dx")
    for r in range(R):
        close(w1[r], w0[r], "dW%d" % r)
        close(b1[r], b0[r], "db%d" % r)
    # deterministic
    plan = _plan(n_dst, n_src, nnz, R, seed=11)
    out2, _ = ops.multilink_agg_fwd(x, ws, bs, plan, accum, "leaky", 0.1, order)
    assert torch.equal(out2, o1)


def test_phased_gather_entry_with_masked_weights_and_add():
    """sg_seg_gather_sum_phased_hip: the weights are read through wpos (a weight array rewritten in place -- resident edge
    masking -- needs no rebuild); req = add accumulates in both phases; empty phases / segments are fine."""
    from star_gcn_amd import _lib as L
    from star_gcn_amd import ops
    rng = np.random.default_rng(4)
    n_seg, n_rows, nnz, C = 500, 64, 20000, 64
    seg = np.sort(rng.integers(0, n_seg // 2, nnz))                   # the upper half of the segments is empty
    indptr = np.concatenate([[0], np.cumsum(np.bincount(seg, minlength=n_seg))]).astype(np.int32)
    for hi in (n_rows, n_rows // 2):                                   # hi = n_rows / 2: phase 1 has no edge at all
        idx = rng.integers(0, hi, nnz).astype(np.int32)
        w = rng.random(nnz).astype(np.float32)
        w[rng.random(nnz) < 0.3] = 0.0
        d_idx, d_ip, d_w = torch.from_numpy(idx).cuda(), torch.from_numpy(indptr).cuda(), torch.from_numpy(w).cuda()
        src = torch.randn(n_rows, C, device="cuda")
        idx_p = torch.empty(nnz, dtype=torch.int32, device="cuda"); wpos_p = torch.empty_like(idx_p)
        ip_p = torch.empty(2 * (n_seg + 1), dtype=torch.int32, device="cuda"); nnz_p = torch.empty(2, dtype=torch.int32, device="cuda")
        lib = L.lib()
        ws, wsn = L.workspace(lib.sg_gather_phases_workspace_bytes(nnz), src.device)
        L.check(lib.sg_gather_phases_build_hip(L.ptr(idx_p), L.ptr(wpos_p), L.ptr(ip_p), L.ptr(nnz_p), L.ptr(d_idx), L.ptr(d_ip),
                                               n_seg, nnz, n_rows, L.ptr(ws), wsn, L.stream_ptr()), "build")
        ph = L.GatherPhasesStruct()
        ph.num_phases, ph.idx, ph.wpos, ph.indptr = 2, idx_p.data_ptr(), wpos_p.data_ptr(), ip_p.data_ptr()
        ph.nnz_p[0], ph.nnz_p[1] = (int(v) for v in nnz_p.cpu())
        import ctypes
        ref = ctypes.cast(ctypes.pointer(ph), ctypes.c_void_p)
        start = torch.randn(n_seg, C, device="cuda")
        for req in (ops.REQ_WRITE, ops.REQ_ADD):
            dst = start.clone()
            want = ops.gather_sum(start.clone(), src, d_idx, d_ip, d_w, n_seg, C, req=req)
            ws2, wsn2 = L.workspace(lib.sg_seg_weighted_pool_workspace_bytes(1, n_seg, nnz, C), src.device)
            L.check(lib.sg_seg_gather_sum_phased_hip(L.ptr(dst), 1, C, L.ptr(src), 1, C, L.ptr(d_w), ref, n_seg, C, req, 0, 0.0,
                                                     L.ptr(ws2), wsn2, L.stream_ptr(), 0), "phased")
            torch.testing.assert_close(dst, want, rtol=2e-6, atol=2e-6)


def test_phases_are_built_only_for_views_the_library_would_phase():
    """the Python side used to gate on edge and row counts only and built 8 bytes per edge of phase arrays (plus a
    blocking read-back) for views the C side can never phase.  Now ONE rule decides (sg_multilink_agg_phased_view): width
    below a 256-byte column slice, or a gathered matrix outside 24 MB .. 256 MB -> nothing is built."""
    from star_gcn_amd import ops
    from star_gcn_amd import _lib as L
    n_dst, n_src, nnz, R = 30000, 45000, 1_300_000, 3
    g = torch.Generator().manual_seed(3)
    for D, U, expect in ((32, 32, 0), (256, 256, 2)):       # 32-wide rows: never phased; 256-wide over 46 / 31 MB: both calls
        plan = _plan(n_dst, n_src, nnz, R, seed=11)
        x = torch.randn(n_src, D, generator=g).cuda()
        ws = [(torch.randn(U, D, generator=g) / 16).cuda() for _ in range(R)]
        bs = [torch.randn(U, generator=g).cuda() for _ in range(R)]
        st = ops._byref(plan.c_struct(False))
        views = [L.lib().sg_multilink_agg_phased_view(st, D, U, 2, 0, b) for b in (0, 1)]
        assert (views == [-1, -1]) if expect == 0 else (views == [L.VIEW_C_IDX_C, L.VIEW_T_Q_S])
        out, saved = ops.multilink_agg_fwd(x, ws, bs, plan, "sum", "leaky", 0.1, "aggregate_first")
        ops.multilink_agg_bwd(torch.ones_like(out), out, saved, x, ws, plan, "sum", "leaky", 0.1, "aggregate_first", True, True, True)
        built = [v for v, p in plan.__dict__.get("_phases", {}).items() if p is not None]
        assert len(built) == expect
    # ahead of time, outside any step
    plan = _plan(n_dst, n_src, nnz, R, seed=11)
    plan.prepare_phases(256, 256, "transform_first", "sum")
    assert sorted(v for v, p in plan._phases.items() if p is not None) == [L.VIEW_C_Q_D, L.VIEW_T_IDX_T]
