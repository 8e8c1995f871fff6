"""Edge cases of the segment operators on the GPU (C ABI path) against the CPU oracle: empty and ragged inputs, odd
feature widths (scalar / float2 / float4 kernels), batches, colliding indices, single giant segment."""
import numpy as np
import pytest
import torch

from oracle import seg as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(a, b, tol=1e-5):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    scale = max(1.0, float(np.abs(b).max())) if b.size else 1.0
    assert a.shape == b.shape
    if b.size:
        assert float(np.abs(a.astype(np.float64) - b).max()) <= tol * scale


@pytest.mark.parametrize("C", [1, 2, 3, 6, 10, 50, 66, 130, 250, 257, 512, 1000])
@pytest.mark.parametrize("B", [1, 3])
def test_widths_and_batches(C, B):
    from star_gcn_amd import contrib
    rng = np.random.default_rng(C * 10 + B)
    S, T, nnz = 37, 29, 900
    cuts = np.sort(rng.integers(0, nnz + 1, S - 1))
    indptr = np.concatenate([[0], cuts, [nnz]]).astype(np.int32)
    idx = rng.integers(0, T, nnz).astype(np.int32)            # many collisions: 900 edges onto 29 rows
    data = rng.normal(size=(B, T, C)).astype(np.float32)
    w = rng.normal(size=(B, nnz)).astype(np.float32)
    og = rng.normal(size=(B, S, C)).astype(np.float32)
    x = dev(data).requires_grad_(True)
    ww = dev(w).requires_grad_(True)
    out = contrib.seg_weighted_pool(x, ww, dev(idx), dev(indptr))
    close(out, O.seg_weighted_pool(data, w, idx, indptr))
    out.backward(dev(og))
    close(x.grad, O.seg_weighted_pool_bwd_data(w, og, idx, indptr, T), 2e-5)
    close(ww.grad, O.seg_take_k_corr(og, data, idx, indptr), 2e-5)
    for pt in ("sum", "avg", "max"):
        ref = O.seg_pool(data, idx, indptr, pt)
        close(contrib.seg_pool(dev(data), dev(idx), dev(indptr), pool_type=pt), ref[0] if pt == "max" else ref)


def test_no_edges_and_all_empty_segments():
    from star_gcn_amd import contrib, ops
    S, T, C = 11, 7, 20
    data = np.random.default_rng(0).normal(size=(2, T, C)).astype(np.float32)
    indptr = np.zeros(S + 1, np.int32)
    # nnz = 0 is represented, as the reference does (graph.py:221-222), by one padding element with weight 0
    out = contrib.seg_weighted_pool(dev(data), dev(np.zeros((2, 1), np.float32)), dev(np.zeros(1, np.int32)), dev(indptr))
    assert out.shape == (2, S, C) and float(out.abs().max()) == 0.0
    val, arg = ops.seg_pool(dev(data), dev(np.zeros(1, np.int32)), dev(indptr), "max")
    assert float(val.abs().max()) == 0.0 and int(arg.max()) == -1      # empty segment -> value 0, index -1
    for pt in ("sum", "avg"):
        v, _ = ops.seg_pool(dev(data), dev(np.zeros(1, np.int32)), dev(indptr), pt)
        assert float(v.abs().max()) == 0.0
    sm = contrib.seg_softmax(dev(np.ones((2, 1), np.float32)), dev(indptr))
    assert float(sm.abs().max()) == 0.0                                  # positions outside every segment stay 0
    ss = contrib.seg_sum(dev(np.ones((2, 1), np.float32)), dev(indptr))
    assert ss.shape == (2, S) and float(ss.abs().max()) == 0.0
    # gradient w.r.t. data of an op with no covered edge is exactly zero
    x = dev(data).requires_grad_(True)
    contrib.seg_weighted_pool(x, dev(np.zeros((2, 1), np.float32)), dev(np.zeros(1, np.int32)), dev(indptr)).sum().backward()
    assert float(x.grad.abs().max()) == 0.0


def test_one_giant_segment_and_trailing_empties():
    """A single 300k-edge segment (spans ~1200 chunks: partial rows + fix-up) followed by empty segments."""
    from star_gcn_amd import contrib
    rng = np.random.default_rng(5)
    S, T, nnz, C = 6, 5000, 300_000, 64
    indptr = np.array([0, 0, nnz, nnz, nnz, nnz, nnz], np.int32)
    idx = rng.integers(0, T, nnz).astype(np.int32)
    data = rng.normal(size=(1, T, C)).astype(np.float32)
    w = (rng.normal(size=(1, nnz)) / np.sqrt(nnz)).astype(np.float32)
    out = contrib.seg_weighted_pool(dev(data), dev(w), dev(idx), dev(indptr)).cpu().numpy()
    ref64 = (data[0, idx].astype(np.float64) * w[0].astype(np.float64)[:, None]).sum(axis=0)
    assert float(np.abs(out[0, 1] - ref64).max()) <= 1e-5 * max(1.0, float(np.abs(ref64).max()))
    assert float(np.abs(out[0, [0, 2, 3, 4, 5]]).max()) == 0.0
    pooled = contrib.seg_pool(dev(data), dev(idx), dev(indptr), pool_type="avg").cpu().numpy()
    avg64 = data[0, idx].astype(np.float64).mean(axis=0)
    assert float(np.abs(pooled[0, 1] - avg64).max()) <= 1e-5


def test_errors_are_python_exceptions():
    """Shape / req violations surface as StarGCNError with the native message (the reference LOG(FATAL)s or exits)."""
    from star_gcn_amd import _lib as L
    from star_gcn_amd import ops
    data = torch.zeros(1, 4, 8, device="cuda")
    w = torch.zeros(1, 3, device="cuda")
    idx = torch.zeros(3, dtype=torch.int32, device="cuda")
    indptr = torch.tensor([0, 3], dtype=torch.int32, device="cuda")
    with pytest.raises(L.StarGCNError, match="req"):
        ops.seg_weighted_pool(data, w, idx, indptr, req=2)
    with pytest.raises(L.StarGCNError, match="AddTo"):
        L.check(L.lib().sg_seg_softmax_hip(L.ptr(w), L.ptr(w), L.ptr(indptr), 1, 1, 3, 3, None), "sg_seg_softmax_hip")
    with pytest.raises(L.StarGCNError, match="inner dimensions"):
        ops.gemm(torch.zeros(3, 4, device="cuda"), torch.zeros(5, 6, device="cuda"))
    with pytest.raises(L.StarGCNError, match="CUDA/HIP tensor"):
        ops.seg_sum(torch.zeros(1, 3), indptr)


@pytest.mark.parametrize("slices", ["2", "4", "8"])
def test_column_sliced_gather_matches_unsliced(slices):
    """XCD column slicing of the gather (sg_gather_tuning(-1, slices) == SG_GATHER_SLICES_FORCE): every addressing mode, write / add, fused activation,
    empty segments, a hub segment spanning many chunks -- against the unsliced launch (same values up to the different
    but fixed summation order) and the float64 definition."""
    from star_gcn_amd import _lib as L
    from star_gcn_amd import ops
    rng = np.random.default_rng(int(slices))
    g = torch.Generator().manual_seed(int(slices))
    n_seg, n_src, nnz, R = 300, 90, 20000, 3
    lens = rng.multinomial(nnz - 6000, rng.dirichlet(np.ones(n_seg) * 0.3))
    lens[17] += 6000                                            # hub segment: > 23 chunks
    lens[5] = 0
    nnz = int(lens.sum())
    indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    w = torch.randn(nnz, generator=g).cuda()
    for C in (256, 320):
        for grouped in (False, True):
            idx = torch.from_numpy(rng.integers(0, n_src * (R if grouped else 1), nnz).astype(np.int32)).cuda()
            src = torch.randn(n_src, R * C if grouped else C, generator=g).cuda()
            kw = dict(src_group=R, src_ld=R * C) if grouped else {}
            rows = src.view(-1, C) if grouped else src
            ref = torch.zeros(n_seg, C, dtype=torch.float64)
            seg = np.repeat(np.arange(n_seg), lens)
            ref.index_add_(0, torch.from_numpy(seg), (w.double().cpu()[:, None] * rows.double().cpu()[idx.cpu().long()]))
            for req, act in ((ops.REQ_WRITE, None), (ops.REQ_ADD, None), (ops.REQ_WRITE, "leaky")):
                base = torch.randn(n_seg, C, generator=g).cuda()
                L.lib().sg_gather_tuning(-1, 0)
                plain = ops.gather_sum(base.clone(), src, idx, indptr, w, n_seg, C, req=req, act=act, **kw)
                L.lib().sg_gather_tuning(-1, int(slices))
                try:
                    sliced = ops.gather_sum(base.clone(), src, idx, indptr, w, n_seg, C, req=req, act=act, **kw)
                finally:
                    L.lib().sg_gather_tuning(-1, 0)
                want = ref + (base.double().cpu() if req == ops.REQ_ADD else 0)
                if act:
                    want = torch.where(want > 0, want, 0.1 * want)
                scale = float(want.abs().max())
                assert float((sliced.double().cpu() - want).abs().max()) <= 2e-6 * scale
                assert float((sliced - plain).abs().max()) <= 2e-6 * scale


@pytest.mark.parametrize("C", [4, 12, 64, 100, 256, 260, 1024])
def test_take_k_corr_chunk_staging(C):
    """The chunked kernel's segment staging: runs of empty segments (also at chunk boundaries), segments longer than a
    256-edge chunk, padding edges past indptr[-1] (zero-filled on write, untouched on add), K > 1; lane groups of
    every width (C/4 = 1 ... 64) and the multi-pass path (C > 256)."""
    from star_gcn_amd import ops
    rng = np.random.default_rng(C)
    K, S, T = 2, 400, 57
    lens = rng.integers(0, 12, S)
    lens[rng.choice(S, 150, replace=False)] = 0              # runs of empty segments
    lens[[3, 200]] = [700, 300]                              # segments spanning several chunks
    lens[255:262] = 0
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    E = int(indptr[-1])
    nnz = E + 300                                            # padding edges, a whole chunk of them included
    ids = rng.integers(0, T, nnz).astype(np.int32)
    e1 = rng.normal(size=(K, S, C)).astype(np.float32)
    e2 = rng.normal(size=(K, T, C)).astype(np.float32)
    ref = O.seg_take_k_corr(e1, e2, ids, indptr)
    got = ops.seg_take_k_corr(dev(e1), dev(e2), dev(ids), dev(indptr))
    close(got[:, :E], ref[:, :E], 2e-5)
    assert float(got[:, E:].abs().max()) == 0.0
    prev = rng.normal(size=(K, nnz)).astype(np.float32)
    acc = dev(prev)
    ops.seg_take_k_corr(dev(e1), dev(e2), dev(ids), dev(indptr), out=acc, req=ops.REQ_ADD)
    close(acc[:, :E], prev[:, :E] + ref[:, :E], 2e-5)
    assert np.array_equal(acc[:, E:].cpu().numpy(), prev[:, E:])


@pytest.mark.parametrize("S", [3, 40, 150, 700, 3000])          # average segment length 1000 ... 1: every lane-group width
@pytest.mark.parametrize("B", [1, 2])
def test_lane_group_kernels_for_every_segment_length(S, B):
    """seg_sum / seg_softmax (fwd, bwd) / seg_broadcast pick 4 ... 64 lanes per segment from the average segment length,
    and seg_broadcast stages segment ids per 256-position chunk: all widths, empty segments, padding past indptr[-1]."""
    from star_gcn_amd import contrib, ops
    rng = np.random.default_rng(S * 7 + B)
    nnz, pad = 3000, 77
    lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 0.5))
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    d = rng.normal(size=(B, nnz + pad)).astype(np.float32)
    dd, di = dev(d), dev(indptr)
    close(ops.seg_sum(dd, di), O.seg_sum(d, indptr), 2e-5)
    sm = ops.seg_softmax(dd, di)
    ref_sm = O.seg_softmax(d, indptr)
    close(sm, ref_sm, 2e-6)
    assert float(sm[:, nnz:].abs().max()) == 0.0
    og = rng.normal(size=(B, nnz + pad)).astype(np.float32)
    close(ops.seg_softmax_bwd(dev(og), sm, di), O.seg_softmax_bwd(og, ref_sm, indptr), 2e-5)
    rhs = rng.normal(size=(B, S)).astype(np.float32)
    for op, name in ((0, "add"), (1, "mul"), (2, "to")):
        got = ops.seg_broadcast(dd if op < 2 else None, dev(rhs), di, op, nnz=nnz + pad)
        want = getattr(O, "seg_broadcast_" + name)(*((d, rhs, indptr) if op < 2 else (rhs, indptr, nnz + pad)))
        close(got[:, :nnz], want[:, :nnz], 1e-6)
        assert float(got[:, nnz:].abs().max()) == 0.0
