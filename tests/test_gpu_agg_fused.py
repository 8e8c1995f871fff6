"""GPU parity of the fused aggregate -> contract kernel (csrc/agg_fused.hip; order='fused' of the multi-link aggregator)
against the float64 layer oracle in the reference's operation order (oracle/model.py; reference aggregators.py:111-163:
R FullyConnected + R seg_weighted_pool + add_n + activation), through autograd (forward, data gradient, weight and bias
gradients), through the raw C ABI, and -- at the MovieLens-10M size -- against the float64 definition on sampled rows and
against the unfused orders.  fp32 tolerance 1e-5 of each tensor's scale (north star)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import model as OM
from tests.test_abi_and_host import make_multilink
from tests.test_gpu_dense_multilink import _rows_vs_definition, rel_close

pytestmark = pytest.mark.gpu

D = 256

CASES = [  # n_dst, n_src, nnz, R
    (60, 45, 900, 5),
    (300, 40, 6000, 10),          # ragged last tile, few sources (every row re-read)
    (64, 500, 4000, 3),           # exactly one tile
    (65, 70, 50, 4),              # mostly empty (row, level) segments
    (1000, 700, 90000, 16),
    (130, 90, 3000, 1),           # one level
    (300, 200, 20000, 32),        # SG_MAX_LINKS levels
    (200, 150, 9000, 17),         # an odd level count
]


@pytest.mark.parametrize("n_dst,n_src,nnz,R", CASES)
@pytest.mark.parametrize("act", ["leaky", None])
def test_fused_order_matches_reference_order(n_dst, n_src, nnz, R, act):
    rng = np.random.default_rng(n_dst + nnz + R)
    eps, ips, sps = make_multilink(rng, n_dst, n_src, nnz, R)
    _check_against_reference_order(eps, ips, sps, n_dst, n_src, R, act, seed=R + nnz)


def test_fused_order_rows_cut_between_gather_waves():
    """A level's edges of a tile go to the eight gather waves in equal shares cut at ANY edge: a row that IS its level (one
    piece in every wave), a long row in the middle of short ones, a long last row, a level with a single edge (seven empty
    shares), an empty level, rows of exactly one share -- against the reference order, forward and all gradients."""
    n_dst, n_src, R = 100, 300, 6
    cnt = np.zeros((R, n_dst), np.int64)
    cnt[0, 5] = 5000                                   # the whole level is one row
    cnt[1, :64] = 1; cnt[1, 0] = 3; cnt[1, 1] = 900    # a long row after a short one
    cnt[2, :64] = 2; cnt[2, 63] = 2000                 # a long LAST row of the tile
    cnt[3, 70] = 1                                     # one edge in the second tile, none in the first
    cnt[4, :64] = 16; cnt[4, 64:] = np.arange(36) % 2 * 40   # every share exactly two rows / alternating empty and 40-edge rows
    rng = np.random.default_rng(7)                     # (level 5 stays empty)
    eps, ips, sps = [], [], []
    for r in range(R):
        ip = np.concatenate([[0], np.cumsum(cnt[r])]).astype(np.int32)
        n = int(ip[-1])
        e = rng.integers(0, n_src, n).astype(np.int32)
        sp = rng.uniform(0.05, 1.0, n).astype(np.float32)
        if n == 0:
            e, sp = np.zeros(1, np.int32), np.zeros(1, np.float32)      # reference graph.py:221-222 empty_as_zero
        eps.append(e); ips.append(ip); sps.append(sp)
    for act in ("leaky", None):
        _check_against_reference_order(eps, ips, sps, n_dst, n_src, R, act, seed=3)


def _check_against_reference_order(eps, ips, sps, n_dst, n_src, R, act, seed, d_in=D, units=D, accum="sum"):
    from star_gcn_amd import functional as F
    from star_gcn_amd.plan import MultiLinkPlan
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n_src, d_in, generator=g) * 0.1 * torch.exp(torch.randn(n_src, 1, generator=g))    # rows of very different scale
    ws = [torch.randn(units, d_in, generator=g) * (3.0 / d_in) ** 0.5 for _ in range(R)]
    bs = [torch.randn(units, generator=g) * 0.1 for _ in range(R)]
    gy = torch.randn(n_dst, units * (R if accum == "stack" else 1), generator=g)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    br = [b.double().requires_grad_(True) for b in bs]
    ref = OM.multilink_aggregator(xr, wr, br, eps, ips, sps, accum=accum, act=act)
    ref.backward(gy.double())
    plan = MultiLinkPlan(eps, ips, sps, n_src, "cuda")
    xd = x.cuda().requires_grad_(True)
    wd = [w.cuda().requires_grad_(True) for w in ws]
    bd = [b.cuda().requires_grad_(True) for b in bs]
    out = F.multilink_aggregate(xd, wd, bd, plan, accum=accum, act=act, slope=0.1, order="fused")
    out.backward(gy.cuda())
    rel_close(out, ref, 1e-5, "out")
    rel_close(xd.grad, xr.grad, 1e-5, "dx")
    for r in range(R):
        rel_close(wd[r].grad, wr[r].grad, 2e-5, "dW%d" % r)
        rel_close(bd[r].grad, br[r].grad, 2e-5, "db%d" % r)
    # the data gradient alone (frozen parameters): the kernel variant that does not save the aggregates
    xd2 = x.cuda().requires_grad_(True)
    out2 = F.multilink_aggregate(xd2, [w.cuda() for w in ws], [b.cuda() for b in bs], plan, accum=accum, act=act, order="fused")
    out2.backward(gy.cuda())
    assert torch.equal(out2, out) and torch.equal(xd2.grad, xd.grad)          # deterministic, and the same with / without zsave


# The reference's own widths (experiments/cfg: EMBED.UNITS 32 / 64 -> GCN.AGG.UNITS 250, 'sum' or 'stack' = 250 // 5 = 50 units per
# level, aggregators.py:79-81) and the halves of the bench width.  (d_in, units per level, accum)
WIDTHS = [(128, 128, "sum"), (64, 250, "sum"), (32, 250, "sum"), (256, 250, "sum"), (252, 64, "sum"),
          (64, 50, "stack"), (32, 50, "stack"), (128, 75, "stack"), (256, 256, "stack"), (256, 250, "stack"), (4, 1, "stack")]


@pytest.mark.parametrize("d_in,units,accum", WIDTHS)
@pytest.mark.parametrize("act", ["leaky", None])
def test_fused_order_other_widths_and_stack(d_in, units, accum, act):
    """the fused kernel at the reference's widths and with accum 'stack' (every level's product to its own column
    block, no cross-level sum; the data gradient reads level r's column block of the output gradient), against the float64
    layer oracle in the reference's operation order: forward, data gradient, weight and bias gradients, 1e-5.  Units that are
    not a multiple of 4 (250) exercise the padded level pitch of the backward; graphs with ragged tiles, empty levels, cut rows."""
    for (n_dst, n_src, nnz, R) in [(300, 40, 6000, 5), (65, 70, 50, 4), (200, 150, 9000, 3), (130, 90, 3000, 1)]:
        rng = np.random.default_rng(n_dst + nnz + R + d_in + units)
        eps, ips, sps = make_multilink(rng, n_dst, n_src, nnz, R)
        _check_against_reference_order(eps, ips, sps, n_dst, n_src, R, act, seed=R + nnz + units, d_in=d_in, units=units, accum=accum)


def test_fused_order_stack_with_cut_rows_and_more_sources_than_destinations():
    """'stack' through both backward forms: n_dst < n_src (the forward saves the aggregates, R per-level weight-gradient products)
    and n_dst > n_src (the data-gradient launch writes the R-expanded gradient), with a hub row cut between the gather waves."""
    for (n_dst, n_src) in [(100, 300), (300, 100)]:
        R = 4
        cnt = np.zeros((R, n_dst), np.int64)
        cnt[0, 5] = 3000
        cnt[1, :64] = 2; cnt[1, 63] = 1500
        cnt[2, 70] = 1
        rng = np.random.default_rng(n_dst)
        eps, ips, sps = [], [], []
        for r in range(R):
            ip = np.concatenate([[0], np.cumsum(cnt[r])]).astype(np.int32)
            n = int(ip[-1])
            e = rng.integers(0, n_src, n).astype(np.int32)
            sp = rng.uniform(0.05, 1.0, n).astype(np.float32)
            if n == 0:
                e, sp = np.zeros(1, np.int32), np.zeros(1, np.float32)
            eps.append(e); ips.append(ip); sps.append(sp)
        for (d_in, units) in [(64, 50), (256, 256)]:
            _check_against_reference_order(eps, ips, sps, n_dst, n_src, R, "leaky", seed=9, d_in=d_in, units=units, accum="stack")


def test_fused_order_is_refused_where_it_does_not_apply():
    from star_gcn_amd import _lib as L
    from star_gcn_amd import functional as F
    from star_gcn_amd.plan import MultiLinkPlan
    rng = np.random.default_rng(5)
    eps, ips, sps = make_multilink(rng, 40, 30, 500, 3)
    plan = MultiLinkPlan(eps, ips, sps, 30, "cuda")
    for (d_in, units) in [(30, 64), (260, 64), (64, 300)]:       # rows not a multiple of 4 floats / wider than 256 / > 256 units
        x = torch.randn(30, d_in, device="cuda")
        ws = [torch.randn(units, d_in, device="cuda") for _ in range(3)]
        bs = [torch.zeros(units, device="cuda") for _ in range(3)]
        with pytest.raises(L.StarGCNError):
            F.multilink_aggregate(x, ws, bs, plan, accum="sum", order="fused")
        assert not L.lib().sg_agg_fused_supported(d_in, units, 3)
    assert L.lib().sg_agg_fused_supported(64, 250, 5) and L.lib().sg_agg_fused_supported(32, 50, 5)
    assert L.lib().sg_agg_fused_supported(128, 128, 10) and L.lib().sg_agg_fused_supported(256, 256, 32)
    assert not L.lib().sg_agg_fused_supported(256, 256, 33)
    # 'auto' stays unfused on a small graph (the R-expanded matrix is cache-resident), and says so -- and for every width other
    # than 256 -> 256 'sum' whatever the size (the kernel is tuned for that shape only)
    from star_gcn_amd import ops
    assert ops.multilink_resolve_order(plan, "auto", 256, 256, "sum") in ("transform_first", "aggregate_first")
    assert ops.multilink_resolve_order(plan, "auto", 64, 250, "sum") in ("transform_first", "aggregate_first")
    assert ops.multilink_resolve_order(plan, "auto", 256, 256, "stack") in ("transform_first", "aggregate_first")


def test_fused_kernel_raw_c_abi_rows_of_other_pitch_and_saved_aggregates():
    """sg_agg_fused_plan_build_hip + sg_agg_fused_hip called directly: strided x / out / zsave rows, a permuted launch order,
    trans_w = 1 (the data-gradient form), no bias; compared with float64; then the weights are rewritten and the plan
    refreshed through f_pos (sg_agg_fused_refresh_hip)."""
    from star_gcn_amd import _lib as L
    from star_gcn_amd import ops
    lib = L.lib()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(11)
    n_dst, n_src, nnz, R = 777, 300, 50000, 7
    key, _ = torch.sort(torch.randint(0, n_dst * R, (nnz,), device=dev, generator=g))
    indptr = torch.zeros(n_dst * R + 1, dtype=torch.int32, device=dev)
    indptr[1:] = torch.cumsum(torch.bincount(key, minlength=n_dst * R), 0).to(torch.int32)
    idx = torch.randint(0, n_src, (nnz,), device=dev, generator=g, dtype=torch.int32)
    w = torch.rand(nnz, device=dev, generator=g) + 0.1
    xbig = torch.randn(n_src, D + 32, device=dev, generator=g)
    x = xbig[:, :D]
    Ws = [torch.randn(D, D, device=dev, generator=g) / 16 for _ in range(R)]
    tiles = int(lib.sg_agg_fused_tiles(n_dst))
    assert tiles == (n_dst + 63) // 64 and lib.sg_agg_fused_supported(256, 256, R) == 1 and lib.sg_agg_fused_supported(130, 256, R) == 0
    order = torch.randperm(tiles, device=dev, generator=g).to(torch.int32)
    f_ptr = torch.empty(tiles * R * 65, dtype=torch.int32, device=dev)
    f_idx, f_w, f_pos = torch.empty_like(idx), torch.empty_like(w), torch.empty_like(idx)
    L.check(lib.sg_agg_fused_plan_build_hip(L.ptr(f_ptr), L.ptr(f_idx), L.ptr(f_w), L.ptr(f_pos), L.ptr(order), L.ptr(indptr),
                                            L.ptr(idx), L.ptr(w), n_dst, R, nnz, L.stream_ptr()), "plan")
    assert torch.equal(torch.sort(f_pos)[0], torch.arange(nnz, device=dev, dtype=torch.int32))      # a permutation of the edges
    assert torch.equal(f_idx, idx[f_pos.long()]) and torch.equal(f_w, w[f_pos.long()])

    def run(wts):
        out = torch.full((n_dst, D + 64), 3.0, device=dev)
        zs = torch.full((n_dst, R * D + 128), -7.0, device=dev)
        ws, wsn = L.workspace(lib.sg_agg_fused_workspace_bytes(R), dev)
        L.check(lib.sg_agg_fused_hip(L.ptr(out), out.shape[1], L.ptr(zs), zs.shape[1], L.ptr(x), xbig.shape[1], ops._ptr_array(Ws), D,
                                     1, None, None, L.ptr(f_ptr), L.ptr(f_idx), L.ptr(f_w), L.ptr(order), n_dst, n_src if wts is w else 0, R, nnz,
                                     D, D, 0, 0.0, 0, L.ptr(ws), wsn, L.stream_ptr()), "fused")      # second run: row extent unknown
        seg = torch.repeat_interleave(torch.arange(n_dst * R, device=dev), (indptr[1:] - indptr[:-1]).long())
        Z = torch.zeros(n_dst * R, D, dtype=torch.float64, device=dev)
        Z.index_add_(0, seg, x.double()[idx.long()] * wts.double()[:, None])
        Z = Z.view(n_dst, R, D)
        ref = sum(Z[:, r] @ Ws[r].double() for r in range(R))
        rel_close(out[:, :D], ref, 1e-5, "out")
        rel_close(zs[:, :R * D], Z.view(n_dst, R * D), 1e-6, "saved aggregates")
        assert bool((out[:, D:] == 3.0).all()) and bool((zs[:, R * D:] == -7.0).all())     # nothing outside the rows' 256 floats

    run(w)
    w2 = torch.rand(nnz, device=dev, generator=g) + 0.5
    L.check(lib.sg_agg_fused_refresh_hip(L.ptr(f_w), L.ptr(f_pos), L.ptr(w2), nnz, L.stream_ptr()), "refresh")
    run(w2)
    # argument checks
    ws, wsn = L.workspace(lib.sg_agg_fused_workspace_bytes(R), dev)
    out = torch.empty(n_dst, D, device=dev)
    rc = lib.sg_agg_fused_hip(L.ptr(out), D, None, 0, L.ptr(x), xbig.shape[1], ops._ptr_array(Ws), D, 0, None, None, L.ptr(f_ptr),
                              L.ptr(f_idx), L.ptr(f_w), None, n_dst, n_src, R, nnz, 130, D, 0, 0.0, 0, L.ptr(ws), wsn, L.stream_ptr())
    assert rc == -2 and b"256" in lib.sg_last_error()          # SG_ERR_UNSUPPORTED: rows must be a multiple of 4 floats, <= 256
    rc = lib.sg_agg_fused_hip(L.ptr(out), D, None, 0, L.ptr(x), xbig.shape[1], ops._ptr_array(Ws), D, 0, None, None, L.ptr(f_ptr),
                              L.ptr(f_idx), L.ptr(f_w), None, n_dst, n_src, R, nnz, D, D, 0, 0.0, 0, L.ptr(ws), ctypes.c_size_t(16),
                              L.stream_ptr())
    assert rc < 0 and b"workspace" in lib.sg_last_error()


def test_fused_kernel_on_heavy_tailed_random_graphs():
    """80 random shapes (1 .. 700 rows, 1 .. 32 levels, 1 .. 150 000 edges) with Zipf row AND level popularity: rows that are
    a level's whole share of a tile, levels with fewer edges than gather waves (empty shares inside a cut row), empty tiles,
    one-row graphs -- forward with saved aggregates, both weight orientations, bias on, against float64.  (Found a bug the
    hand-made cases missed: a two-edge row in a three-edge level.)"""
    from star_gcn_amd import _lib as L
    from star_gcn_amd import ops
    lib = L.lib()
    dev = torch.device("cuda")
    rng = torch.Generator(device="cpu").manual_seed(20240)
    for case in range(80):
        n_dst = int(torch.randint(1, 701, (1,), generator=rng))
        n_src = int(torch.randint(1, 3001, (1,), generator=rng))
        R = int(torch.randint(1, 33, (1,), generator=rng))
        nnz = int(10 ** (float(torch.rand(1, generator=rng)) * 5.17)) + 1
        a_row = 0.5 + 2.0 * float(torch.rand(1, generator=rng))
        a_lvl = 2.5 * float(torch.rand(1, generator=rng))
        g = torch.Generator(device=dev).manual_seed(case)
        pr = (torch.arange(1, n_dst + 1, device=dev, dtype=torch.float64) ** -a_row)[torch.randperm(n_dst, device=dev, generator=g)]
        pl = (torch.arange(1, R + 1, device=dev, dtype=torch.float64) ** -a_lvl)[torch.randperm(R, device=dev, generator=g)]
        key, _ = torch.sort(torch.multinomial(pr, nnz, replacement=True, generator=g) * R +
                            torch.multinomial(pl, nnz, replacement=True, generator=g))
        indptr = torch.zeros(n_dst * R + 1, dtype=torch.int32, device=dev)
        indptr[1:] = torch.cumsum(torch.bincount(key, minlength=n_dst * R), 0).to(torch.int32)
        idx = torch.randint(0, n_src, (nnz,), device=dev, generator=g, dtype=torch.int32)
        w = torch.rand(nnz, device=dev, generator=g) + 0.1
        x = torch.randn(n_src, D, device=dev, generator=g) * torch.exp(torch.randn(n_src, 1, device=dev, generator=g))
        Ws = [torch.randn(D, D, device=dev, generator=g) / 16 for _ in range(R)]
        bs = [torch.randn(D, device=dev, generator=g) for _ in range(R)]
        trans = case & 1
        tiles = (n_dst + 63) // 64
        order = torch.randperm(tiles, device=dev, generator=g).to(torch.int32) if case & 4 else None
        f_ptr = torch.empty(tiles * R * 65, dtype=torch.int32, device=dev)
        f_idx, f_w = torch.empty_like(idx), torch.empty_like(w)
        L.check(lib.sg_agg_fused_plan_build_hip(L.ptr(f_ptr), L.ptr(f_idx), L.ptr(f_w), None, L.ptr(order), L.ptr(indptr), L.ptr(idx),
                                                L.ptr(w), n_dst, R, nnz, L.stream_ptr()), "plan")
        seg = torch.repeat_interleave(torch.arange(n_dst * R, device=dev), (indptr[1:] - indptr[:-1]).long())
        Z = torch.zeros(n_dst * R, D, dtype=torch.float64, device=dev).index_add_(0, seg, x.double()[idx.long()] * w.double()[:, None])
        Z = Z.view(n_dst, R, D)
        rs = torch.zeros(n_dst * R, dtype=torch.float64, device=dev).index_add_(0, seg, w.double()).view(n_dst, R)
        ref = torch.zeros(n_dst, D, dtype=torch.float64, device=dev)
        mag = torch.zeros_like(ref)
        for r in range(R):
            B = Ws[r].double() if trans else Ws[r].double().t()
            ref += Z[:, r] @ B + rs[:, r:r + 1] * bs[r].double()[None]
            mag += Z[:, r].abs() @ B.abs() + (rs[:, r:r + 1] * bs[r].double()[None]).abs()
        out = torch.empty(n_dst, D, device=dev)
        zs = torch.full((n_dst, R * D), -7.0, device=dev)
        ws, wsn = L.workspace(lib.sg_agg_fused_workspace_bytes(R), dev)
        L.check(lib.sg_agg_fused_hip(L.ptr(out), D, L.ptr(zs), R * D, L.ptr(x), D, ops._ptr_array(Ws), D, trans, ops._ptr_array(bs),
                                     L.ptr(rs.float().contiguous()), L.ptr(f_ptr), L.ptr(f_idx), L.ptr(f_w), L.ptr(order), n_dst,
                                     n_src if case % 3 else 0, R, nnz, D, D, 0, 0.0, 0, L.ptr(ws), wsn, L.stream_ptr()), "fused")
        err = float(((out.double() - ref).abs() / mag.clamp_min(1e-30)).max())
        zerr = float((zs.double().view(n_dst, R, D) - Z).abs().max() / Z.abs().max().clamp_min(1e-300))
        what = (case, n_dst, R, nnz, int((indptr[1:] - indptr[:-1]).max()))
        assert err < 2e-6 and zerr < 2e-6, (what, err, zerr)          # |err| / sum |a||b|: fp32 products at f16x3 accuracy


def test_fused_order_through_autograd_on_heavy_tailed_random_graphs():
    """40 random graphs with Zipf destination, source and level popularity through the whole aggregator (forward, data gradient
    over the transposed plan, weight gradient from the saved aggregates -- either side smaller --, bias gradient): the fused order
    against the transform-first order of the same library, and the outputs against float64."""
    from star_gcn_amd import functional as F
    from star_gcn_amd.plan import MultiLinkPlan
    rng = np.random.default_rng(77)
    for case in range(40):
        n_dst, n_src = int(rng.integers(1, 500)), int(rng.integers(1, 500))
        R = int(rng.integers(1, 33))
        nnz = int(10 ** rng.uniform(0, 4.6)) + 1
        zipf = lambda n, a: (np.arange(1, n + 1) ** -a)[rng.permutation(n)]
        pd, ps, pl = zipf(n_dst, rng.uniform(0.3, 2.5)), zipf(n_src, rng.uniform(0.0, 1.5)), zipf(R, rng.uniform(0.0, 2.5))
        dst = rng.choice(n_dst, nnz, p=pd / pd.sum())
        src = rng.choice(n_src, nnz, p=ps / ps.sum()).astype(np.int32)
        lev = rng.choice(R, nnz, p=pl / pl.sum())
        sup = rng.uniform(0.05, 1.0, nnz).astype(np.float32)
        eps, ips, sps = [], [], []
        for r in range(R):
            sel = np.flatnonzero(lev == r)
            sel = sel[np.argsort(dst[sel], kind="stable")]
            ips.append(np.concatenate([[0], np.cumsum(np.bincount(dst[sel], minlength=n_dst))]).astype(np.int32))
            e, sp = src[sel], sup[sel]
            if e.size == 0:
                e, sp = np.zeros(1, np.int32), np.zeros(1, np.float32)      # reference graph.py:221-222 empty_as_zero
            eps.append(e); sps.append(sp)
        plan = MultiLinkPlan(eps, ips, sps, n_src, "cuda")
        g = torch.Generator().manual_seed(case)
        x = torch.randn(n_src, D, generator=g) * torch.exp(torch.randn(n_src, 1, generator=g))
        ws = [torch.randn(D, D, generator=g) * (3.0 / D) ** 0.5 for _ in range(R)]
        bs = [torch.randn(D, generator=g) * 0.1 for _ in range(R)]
        gy = torch.randn(n_dst, D, generator=g).cuda()
        res = {}
        for order in ("fused", "transform_first"):
            xd = x.cuda().requires_grad_(True)
            wd = [w.cuda().requires_grad_(True) for w in ws]
            bd = [b.cuda().requires_grad_(True) for b in bs]
            # (no activation under the gradients: two summation orders may put a pre-activation on either side of LeakyReLU's kink,
            #  and on a 10 000-edge hub row one flipped derivative is 1e-5 of the gradient -- tools/dbg_case2.py)
            out = F.multilink_aggregate(xd, wd, bd, plan, accum="sum", act=None, order=order)
            out.backward(gy)
            res[order] = (out.detach(), xd.grad, torch.stack([w.grad for w in wd]), torch.stack([b.grad for b in bd]))
        for name, a, b in zip(("out", "dx", "dW", "db"), res["fused"], res["transform_first"]):
            rel_close(a, b.double(), 2e-5, "case %d (%d x %d, %d levels, %d edges) %s" % (case, n_dst, n_src, R, nnz, name))
        ref = OM.multilink_aggregator(x.double(), [w.double() for w in ws], [b.double() for b in bs], eps, ips, sps, accum="sum", act="leaky")
        with torch.no_grad():
            out = F.multilink_aggregate(x.cuda(), [w.cuda() for w in ws], [b.cuda() for b in bs], plan, accum="sum", act="leaky",
                                        slope=0.1, order="fused")
        rel_close(out, ref, 1e-5, "case %d out (leaky) vs float64" % case)


def test_fused_order_at_ml10m_size_against_the_definition_and_the_unfused_orders():
    """BASELINE config 4 size (69878 x 10677, 10 M ratings, 10 levels, dim 256), both directions of the bipartite graph
    (heavy item rows up to 35 k ratings): >= 64 sampled output rows and gradient rows against the float64 definition,
    agreement with the transform-first order, the adjoint identity through autograd, and the weight / bias gradients
    against the unfused order."""
    import star_gcn_amd.synthetic as S
    from star_gcn_amd import functional as F
    from star_gcn_amd.plan import MultiLinkPlan
    graph, eu, ei, vals = S.make_graph("ml-10m")
    for dst, src in (("user", "movie"), ("movie", "user")):
        m = graph[dst, src]
        eps, _, ips, sps = m.sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
        plan = MultiLinkPlan(eps, ips, sps, m.shape[1], "cuda")
        R = plan.R
        g = torch.Generator(device="cuda").manual_seed(3)
        x1 = torch.randn(plan.n_src, D, device="cuda", generator=g) * 0.1
        ws = [torch.randn(D, D, device="cuda", generator=g) * (3.0 / D) ** 0.5 for _ in range(R)]
        bs = [torch.randn(D, device="cuda", generator=g) * 0.1 for _ in range(R)]
        y = torch.randn(plan.n_dst, D, device="cuda", generator=g)
        res = {}
        for order in ("fused", "transform_first"):
            xg = x1.clone().requires_grad_(True)
            wg = [w.clone().requires_grad_(True) for w in ws]
            bg = [b.clone().requires_grad_(True) for b in bs]
            out = F.multilink_aggregate(xg, wg, bg, plan, accum="sum", act=None, order=order)
            out.backward(y)
            res[order] = (out.detach(), xg.grad, [w.grad for w in wg], [b.grad for b in bg])
        out, dx, dws, dbs = res["fused"]
        nd, ns, _w = _rows_vs_definition(plan, x1, ws, bs, out, y, dx, 64, 4, 1e-5)
        assert nd >= 64 and ns >= 64
        o2, dx2, dws2, dbs2 = res["transform_first"]
        assert float((out - o2).abs().max()) <= 1e-5 * float(o2.abs().max())
        assert float((dx - dx2).abs().max()) <= 1e-5 * float(dx2.abs().max())
        for r in range(R):
            assert float((dws[r] - dws2[r]).abs().max()) <= 2e-5 * max(float(dws2[r].abs().max()), 1e-3), r
            assert float((dbs[r] - dbs2[r]).abs().max()) <= 2e-5 * max(float(dbs2[r].abs().max()), 1e-3), r
        const = F.multilink_aggregate(torch.zeros_like(x1), ws, bs, plan, accum="sum", act=None, order="fused")
        terms = (out - const).double() * y.double()
        lhs, rhs = float(terms.sum()), float((dx.double() * x1.double()).sum())
        assert abs(lhs - rhs) <= 1e-9 * float(terms.abs().sum()), (dst, lhs, rhs, float(terms.abs().sum()))


def test_fused_order_replays_inside_a_hip_graph():
    """forward + backward of the fused order captured into one hipGraph (after an eager step built the plan's fused edge orders)
    and replayed on new inputs: the same bits as the eager call -- no allocation, host read-back or stream switch hides in the
    fused entry points."""
    from star_gcn_amd import functional as F
    from star_gcn_amd.plan import MultiLinkPlan
    rng = np.random.default_rng(77)
    n_dst, n_src, nnz, R = 500, 400, 40000, 6
    eps, ips, sps = make_multilink(rng, n_dst, n_src, nnz, R)
    plan = MultiLinkPlan(eps, ips, sps, n_src, "cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    ws = [(torch.randn(D, D, device="cuda", generator=g) / 16).requires_grad_(True) for _ in range(R)]
    bs = [(torch.randn(D, device="cuda", generator=g) * 0.1).requires_grad_(True) for _ in range(R)]
    x = torch.randn(n_src, D, device="cuda", generator=g).requires_grad_(True)
    gy = torch.randn(n_dst, D, device="cuda", generator=g)

    def step():
        for t in [x] + ws + bs:
            t.grad = None
        out = F.multilink_aggregate(x, ws, bs, plan, accum="sum", act="leaky", order="fused")
        out.backward(gy)
        return out.detach(), x.grad, ws[0].grad, bs[-1].grad

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = step()
    with torch.no_grad():
        x.copy_(torch.randn(n_src, D, device="cuda", generator=g))
        gy.copy_(torch.randn(n_dst, D, device="cuda", generator=g))
    graph.replay()
    torch.cuda.synchronize()
    replayed = [t.clone() for t in captured]
    eager = step()
    for a, b in zip(replayed, eager):
        assert torch.equal(a, b)
    # the resident training iteration: the plan's weights are rewritten (edge masking) and refreshed INSIDE the captured region --
    # the fused kernel reads its own level-major copy of them, which the refresh must rebuild during capture, not skip
    cw0, tw0 = plan.c_w.clone(), plan.t_w.clone()
    fac = torch.ones(1, device="cuda")

    def masked_step():
        plan.c_w.copy_(cw0 * fac)
        plan.t_w.copy_(tw0 * fac)
        plan.refresh_rowsum()
        return step()

    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        masked_step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2):
        captured2 = masked_step()
    fac.fill_(0.25)
    graph2.replay()
    torch.cuda.synchronize()
    replayed2 = [t.clone() for t in captured2]
    assert not torch.equal(replayed2[0], replayed[0])
    sps4 = [sp * np.float32(0.25) for sp in sps]
    plan4 = MultiLinkPlan(eps, ips, sps4, n_src, "cuda")
    for t in [x] + ws + bs:
        t.grad = None
    out4 = F.multilink_aggregate(x, ws, bs, plan4, accum="sum", act="leaky", order="fused")
    out4.backward(gy)
    for a, b in zip(replayed2, (out4.detach(), x.grad, ws[0].grad, bs[-1].grad)):
        assert torch.equal(a, b)
