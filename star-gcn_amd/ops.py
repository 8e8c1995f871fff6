"""Thin, autograd-free wrappers over the C ABI (include/stargcn.h) on torch CUDA tensors.

Every function enqueues on the current torch stream and never synchronises.  Arguments are validated by
the native side; errors surface as `StarGCNError` carrying `sg_last_error()`.
"""
import torch

from . import _lib as L
from ._lib import ACT, REQ_ADD, REQ_NULL, REQ_WRITE  # noqa: F401  (re-exported)


def gather_profile(enable):
    """bench.py: switch the library's HIP-event bracketing of gather launches on/off (sg_gather_profile_enable)."""
    return L.lib().sg_gather_profile_enable(int(bool(enable)))


def gather_profile_read(capacity=1 << 16, with_src_bytes=False):
    """-> list of (seconds, edges visited, feature width[, bytes of the gathered matrix]) per gather launch since the
    profile was enabled."""
    import ctypes
    ms = (ctypes.c_float * capacity)()
    nnz = (ctypes.c_int64 * capacity)()
    fd = (ctypes.c_int64 * capacity)()
    sb = (ctypes.c_int64 * capacity)()
    n = L.lib().sg_gather_profile_read2(ms, nnz, fd, sb, capacity)
    if with_src_bytes:
        return [(ms[i] * 1e-3, int(nnz[i]), int(fd[i]), int(sb[i])) for i in range(n)]
    return [(ms[i] * 1e-3, int(nnz[i]), int(fd[i])) for i in range(n)]


def fused_profile(enable):
    """bench.py: HIP-event bracketing of the fused aggregate -> contract launches on/off (sg_agg_fused_profile_enable)."""
    return L.lib().sg_agg_fused_profile_enable(int(bool(enable)))


def fused_profile_read(capacity=1 << 14):
    """-> list of (seconds, edges, 1 if the launch also wrote the aggregates) per fused launch since the profile was enabled."""
    import ctypes
    ms = (ctypes.c_float * capacity)()
    nnz = (ctypes.c_int64 * capacity)()
    zs = (ctypes.c_int32 * capacity)()
    n = L.lib().sg_agg_fused_profile_read(ms, nnz, zs, capacity)
    return [(ms[i] * 1e-3, int(nnz[i]), int(zs[i])) for i in range(n)]


def gemm_profile(enable):
    """bench.py: HIP-event bracketing of every GEMM call on/off (sg_gemm_profile_enable)."""
    return L.lib().sg_gemm_profile_enable(int(bool(enable)))


def gemm_profile_read(capacity=1 << 16):
    """-> list of (seconds, M, N, K, backend) per sg_gemm_f32_hip call since the profile was enabled."""
    import ctypes
    ms = (ctypes.c_float * capacity)()
    mnk = (ctypes.c_int64 * (3 * capacity))()
    be = (ctypes.c_int * capacity)()
    n = L.lib().sg_gemm_profile_read(ms, mnk, be, capacity)
    return [(ms[i] * 1e-3, int(mnk[3 * i]), int(mnk[3 * i + 1]), int(mnk[3 * i + 2]), int(be[i])) for i in range(n)]


def _act_id(act):
    if isinstance(act, int):
        return act
    if act not in ACT:
        raise L.StarGCNError("unsupported activation %r for the fused path" % (act,))
    return ACT[act]


def gather_sum(dst, src, indices, indptr, weights, seg_num, feat_dim, dst_group=1, dst_ld=None, src_group=1,
               src_ld=None, req=REQ_WRITE, act=None, slope=0.1):
    """dst[row(s)] (+)= act( sum_j w[j] * src[row(idx[j])] )  -- sg_seg_gather_sum_hip (strided/grouped rows)."""
    L.require_gpu(dst, src, indices, indptr, weights)
    nnz = indices.numel()
    dst_ld = feat_dim * dst_group if dst_ld is None else dst_ld
    src_ld = feat_dim * src_group if src_ld is None else src_ld
    lib = L.lib()
    wsb = lib.sg_seg_weighted_pool_workspace_bytes(1, seg_num, nnz, feat_dim)
    ws, wsn = L.workspace(wsb, dst.device)
    L.check(lib.sg_seg_gather_sum_hinted_hip(L.ptr(dst), dst_group, dst_ld, L.ptr(src), src_group, src_ld,
                                             L.ptr(weights), L.ptr(indices), L.ptr(indptr), seg_num, nnz, feat_dim, req,
                                             _act_id(act), float(slope), L.ptr(ws), wsn, L.stream_ptr(),
                                             src.numel() * 4), "sg_seg_gather_sum_hinted_hip")
    return dst


def gather_sum_parts(dst, src, sp, weights, feat_dim, dst_group=1, dst_ld=None, src_group=1, src_ld=None, req=REQ_WRITE,
                     act=None, slope=0.1):
    """gather_sum over a plan.SourcePartition (sg_seg_gather_sum_parts_hip): dst[s] (+)= act(sum_j w[pos[j]] src[row(j)])."""
    L.require_gpu(dst, src, weights)
    dst_ld = feat_dim * dst_group if dst_ld is None else dst_ld
    src_ld = feat_dim * src_group if src_ld is None else src_ld
    lib = L.lib()
    ws, wsn = L.workspace(lib.sg_seg_gather_sum_parts_workspace_bytes(sp.n_seg, sp.parts, sp.nnz, feat_dim), dst.device)
    L.check(lib.sg_seg_gather_sum_parts_hip(L.ptr(dst), dst_group, dst_ld, L.ptr(src), src_group, src_ld, L.ptr(weights),
                                            L.ptr(sp.pos), L.ptr(sp.src), L.ptr(sp.indptr), sp.n_seg, sp.parts, sp.nnz,
                                            feat_dim, req, _act_id(act), float(slope), L.ptr(ws), wsn, L.stream_ptr(),
                                            src.numel() * 4), "sg_seg_gather_sum_parts_hip")
    return dst


def pair_l2(src, other, y, indices, indptr, seg_num, scale, parts=1, scale_dev=None, loss_scale=None, out=None,
            req=REQ_WRITE):
    """sg_pair_l2_hip: rows[seg] (+)= sum_j g_j src[indices[j]] with g_j = scale * (*scale_dev) * (<src[indices[j]], other[seg]>
    - y[j]); returns (rows, loss) with loss = loss_scale * sum_j (..)^2 as a 1-element tensor (None when loss_scale is)."""
    L.require_gpu(src, other, y, indices, indptr)
    C = src.shape[1]
    nnz = indices.numel()
    if out is None:
        out = torch.empty((seg_num, C), dtype=torch.float32, device=src.device)
    loss = torch.empty(1, dtype=torch.float32, device=src.device) if loss_scale is not None else None
    lib = L.lib()
    ws, wsn = L.workspace(lib.sg_pair_l2_workspace_bytes(seg_num, parts, nnz, C), src.device)
    L.check(lib.sg_pair_l2_hip(L.ptr(out), L.ptr(loss) if loss is not None else None, L.ptr(src), L.ptr(other), L.ptr(y),
                               L.ptr(indices), L.ptr(indptr), seg_num, parts, nnz, C, float(scale),
                               L.ptr(scale_dev) if scale_dev is not None else None,
                               float(loss_scale) if loss_scale is not None else 0.0, req, L.ptr(ws), wsn, L.stream_ptr(),
                               src.numel() * 4), "sg_pair_l2_hip")
    return out, loss


def seg_weighted_pool(data, weights, indices, indptr, out=None, req=REQ_WRITE):
    """reference `_contrib_seg_weighted_pool` forward (seg_op.cc:665-716)."""
    L.require_gpu(data, weights, indices, indptr)
    B, T, C = data.shape
    nnz = indices.numel()
    S = indptr.numel() - 1
    if out is None:
        out = torch.empty((B, S, C), dtype=torch.float32, device=data.device)
    lib = L.lib()
    ws, wsn = L.workspace(lib.sg_seg_weighted_pool_workspace_bytes(B, S, nnz, C), data.device)
    L.check(lib.sg_seg_weighted_pool_hip(L.ptr(out), L.ptr(data), L.ptr(weights), L.ptr(indices), L.ptr(indptr), B, S,
                                         T, nnz, C, req, L.ptr(ws), wsn, L.stream_ptr()), "sg_seg_weighted_pool_hip")
    return out


def seg_weighted_pool_bwd_data(weights, ograd, tplan, total_ind_num, out=None, req=REQ_WRITE):
    """gradient of seg_weighted_pool w.r.t. data through a cached TransposePlan (reference seg_op.cc:700-703)."""
    L.require_gpu(weights, ograd)
    B, S, C = ograd.shape
    nnz = weights.shape[1]
    T = int(total_ind_num)
    if out is None:
        out = torch.empty((B, T, C), dtype=torch.float32, device=ograd.device)
    lib = L.lib()
    ws, wsn = L.workspace(lib.sg_seg_weighted_pool_bwd_data_workspace_bytes(B, T, nnz, C), ograd.device)
    L.check(lib.sg_seg_weighted_pool_bwd_data_hip(L.ptr(out), L.ptr(weights), L.ptr(ograd), L.ptr(tplan.t_indptr),
                                                  L.ptr(tplan.t_pos), L.ptr(tplan.t_seg), B, S, T, nnz, C, req,
                                                  L.ptr(ws), wsn, L.stream_ptr()), "sg_seg_weighted_pool_bwd_data_hip")
    return out


def seg_take_k_corr(embed1, embed2, neighbor_ids, neighbor_indptr, out=None, req=REQ_WRITE):
    L.require_gpu(embed1, embed2, neighbor_ids, neighbor_indptr)
    K, N, C = embed1.shape
    M = embed2.shape[1]
    nnz = neighbor_ids.numel()
    if out is None:
        out = torch.empty((K, nnz), dtype=torch.float32, device=embed1.device)
    L.check(L.lib().sg_seg_take_k_corr_hip(L.ptr(out), L.ptr(embed1), L.ptr(embed2), L.ptr(neighbor_ids),
                                           L.ptr(neighbor_indptr), K, N, M, nnz, C, req, L.stream_ptr()),
            "sg_seg_take_k_corr_hip")
    return out


def seg_sum(data, indptr, out=None, req=REQ_WRITE):
    L.require_gpu(data, indptr)
    B, nnz = data.shape
    S = indptr.numel() - 1
    if out is None:
        out = torch.empty((B, S), dtype=torch.float32, device=data.device)
    L.check(L.lib().sg_seg_sum_hip(L.ptr(out), L.ptr(data), L.ptr(indptr), B, S, nnz, req, L.stream_ptr()),
            "sg_seg_sum_hip")
    return out


def seg_broadcast(lhs, rhs, indptr, op, nnz=None, out=None, req=REQ_WRITE):
    L.require_gpu(rhs, indptr, lhs)
    B, S = rhs.shape
    nnz = lhs.shape[1] if lhs is not None else int(nnz)
    if out is None:
        out = torch.empty((B, nnz), dtype=torch.float32, device=rhs.device)
    L.check(L.lib().sg_seg_broadcast_hip(L.ptr(out), L.ptr(lhs), L.ptr(rhs), L.ptr(indptr), B, S, nnz, op, req,
                                         L.stream_ptr()), "sg_seg_broadcast_hip")
    return out


def seg_softmax(data, indptr):
    L.require_gpu(data, indptr)
    B, nnz = data.shape
    out = torch.empty_like(data)
    L.check(L.lib().sg_seg_softmax_hip(L.ptr(out), L.ptr(data), L.ptr(indptr), B, indptr.numel() - 1, nnz, REQ_WRITE,
                                       L.stream_ptr()), "sg_seg_softmax_hip")
    return out


def seg_softmax_bwd(ograd, val, indptr, out=None, req=REQ_WRITE):
    L.require_gpu(ograd, val, indptr)
    B, nnz = ograd.shape
    if out is None:
        out = torch.empty_like(ograd)
    L.check(L.lib().sg_seg_softmax_bwd_hip(L.ptr(out), L.ptr(ograd), L.ptr(val), L.ptr(indptr), B, indptr.numel() - 1,
                                           nnz, req, L.stream_ptr()), "sg_seg_softmax_bwd_hip")
    return out


def seg_pool(data, indices, indptr, pool_type):
    L.require_gpu(data, indices, indptr)
    B, T, C = data.shape
    nnz = indices.numel()
    S = indptr.numel() - 1
    out = torch.empty((B, S, C), dtype=torch.float32, device=data.device)
    arg = torch.empty((B, S, C), dtype=torch.int32, device=data.device) if pool_type == "max" else None
    lib = L.lib()
    ws, wsn = L.workspace(lib.sg_seg_pool_workspace_bytes(B, S, nnz, C), data.device)
    L.check(lib.sg_seg_pool_hip(L.ptr(out), L.ptr(arg), L.ptr(data), L.ptr(indices), L.ptr(indptr), B, S, T, nnz, C,
                                L.POOL[pool_type], REQ_WRITE, L.ptr(ws), wsn, L.stream_ptr()), "sg_seg_pool_hip")
    return out, arg


def seg_pool_bwd(ograd, pool_indices, indptr, tplan, total_ind_num, pool_type, out=None, req=REQ_WRITE):
    L.require_gpu(ograd, indptr)
    B, S, C = ograd.shape
    T = int(total_ind_num)
    nnz = tplan.nnz
    if out is None:
        out = torch.empty((B, T, C), dtype=torch.float32, device=ograd.device)
    lib = L.lib()
    ws, wsn = L.workspace(lib.sg_seg_pool_bwd_workspace_bytes(B, T, nnz, C), ograd.device)
    L.check(lib.sg_seg_pool_bwd_hip(L.ptr(out), L.ptr(ograd), L.ptr(pool_indices), L.ptr(indptr), L.ptr(tplan.t_indptr),
                                    L.ptr(tplan.t_pos), L.ptr(tplan.t_seg), B, S, T, nnz, C, L.POOL[pool_type], req,
                                    L.ptr(ws), wsn, L.stream_ptr()), "sg_seg_pool_bwd_hip")
    return out


def gemm(a, b, trans_a=False, trans_b=False, bias=None, act=None, slope=0.1, out=None, accumulate=False):
    """out[M,N] = act(op(a) @ op(b) + bias (+ out)) on the fp32 MFMA kernel.  2-D tensors whose last stride is 1
    (row strides = leading dims are honoured, so column slices of a wider matrix work)."""
    L.require_gpu(a, b, bias, out)
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = (a.shape[1], a.shape[0]) if trans_a else (a.shape[0], a.shape[1])
    Kb, N = (b.shape[1], b.shape[0]) if trans_b else (b.shape[0], b.shape[1])
    if K != Kb:
        raise L.StarGCNError("gemm: inner dimensions differ (%d vs %d)" % (K, Kb))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.stride(1) == 1 or N == 1
    lda = a.stride(0) if a.shape[0] > 1 else max(a.shape[1], 1)
    ldb = b.stride(0) if b.shape[0] > 1 else max(b.shape[1], 1)
    ldc = out.stride(0) if out.shape[0] > 1 else max(N, 1)
    lib = L.lib()
    ws, wsn = L.workspace(lib.sg_gemm_f32_workspace_bytes(M, N, K, int(trans_a)), a.device)
    L.check(lib.sg_gemm_f32_hip(L.ptr(out), ldc, L.ptr(a), lda, int(trans_a), L.ptr(b), ldb, int(trans_b), M, N, K,
                                L.ptr(bias), _act_id(act), float(slope), int(accumulate), L.ptr(ws), wsn,
                                L.stream_ptr()), "sg_gemm_f32_hip")
    return out


def act_fwd(x, act, slope=0.1):
    """act(x) elementwise on the native kernel (sg_act_hip); identity returns x itself."""
    if _act_id(act) == 0:
        return x
    L.require_gpu(x)
    x = L.f32c(x)
    out = torch.empty_like(x)
    L.check(L.lib().sg_act_hip(L.ptr(out), L.ptr(x), x.numel(), _act_id(act), float(slope), L.stream_ptr()), "sg_act_hip")
    return out


def act_bwd(dout, out_act, act, slope=0.1):
    """dpre = dout * act'(.) computed from the activation output."""
    if _act_id(act) == 0:
        return dout
    L.require_gpu(dout, out_act)
    dout = L.f32c(dout)
    dpre = torch.empty_like(dout)
    L.check(L.lib().sg_act_bwd_hip(L.ptr(dpre), L.ptr(dout), L.ptr(out_act), dout.numel(), _act_id(act), float(slope),
                                   L.stream_ptr()), "sg_act_bwd_hip")
    return dpre


def act_bwd_colsum(dout, out_act, act, slope=0.1):
    """(dpre, column sums of dpre) in one pass -- the Dense backward's activation gradient + bias gradient."""
    if _act_id(act) == 0:
        return dout, colsum(dout)
    L.require_gpu(dout, out_act)
    dout = L.f32c(dout)
    assert dout.dim() == 2 and out_act.is_contiguous()
    M, N = dout.shape
    dpre = torch.empty_like(dout)
    db = torch.empty((N,), dtype=torch.float32, device=dout.device)
    lib = L.lib()
    ws, wsn = L.workspace(lib.sg_colsum_workspace_bytes(M, N), dout.device)
    L.check(lib.sg_act_bwd_colsum_hip(L.ptr(dpre), L.ptr(db), L.ptr(dout), L.ptr(out_act), M, N, _act_id(act),
                                      float(slope), REQ_WRITE, L.ptr(ws), wsn, L.stream_ptr()), "sg_act_bwd_colsum_hip")
    return dpre, db


def colsum(x):
    L.require_gpu(x)
    assert x.dim() == 2 and x.stride(1) == 1
    M, N = x.shape
    out = torch.empty((N,), dtype=torch.float32, device=x.device)
    lib = L.lib()
    ws, wsn = L.workspace(lib.sg_colsum_workspace_bytes(M, N), x.device)
    L.check(lib.sg_colsum_hip(L.ptr(out), L.ptr(x), x.stride(0) if M > 1 else N, M, N, REQ_WRITE, L.ptr(ws), wsn,
                              L.stream_ptr()), "sg_colsum_hip")
    return out


def masked_embed(table, ids, noise=None):
    """reference Net.get_embed (STAR-GCN.py:290-299): rows of `table` at noise[ids] (or ids), zero where -1."""
    L.require_gpu(table, ids, noise)
    n = ids.numel()
    out = torch.empty((n, table.shape[1]), dtype=torch.float32, device=table.device)
    L.check(L.lib().sg_masked_embed_hip(L.ptr(out), L.ptr(table), L.ptr(ids), L.ptr(noise), n, table.shape[0],
                                        table.shape[1], L.stream_ptr()), "sg_masked_embed_hip")
    return out


_ORDER = {"auto": 0, "transform_first": 1, "aggregate_first": 2, "fused": 3}
_ORDER_NAMES = ("auto", "transform_first", "aggregate_first", "fused")
_ACCUM = {"sum": 0, "stack": 1}


def _ptr_array(tensors):
    import ctypes
    arr = (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])
    return arr


def multilink_resolve_order(plan, order, in_dim=None, units_per_level=None, accum="sum"):
    """The order the native entries run: with the widths given, the library's full rule (sg_multilink_agg_resolve_order2:
    'auto' becomes 'fused' for 256-wide 'sum' aggregations over graphs whose R-expanded matrix would cost HBM time, and the
    plan's fused edge orders are built here if so); without them the size rule of the two unfused orders."""
    if in_dim is None:
        rc = L.lib().sg_multilink_agg_resolve_order(_byref(plan.c_struct(False)), _ORDER[order])
        L.check(min(rc, 0), "sg_multilink_agg_resolve_order")
        return _ORDER_NAMES[rc]
    rc = L.lib().sg_multilink_agg_resolve_order2(_byref(plan.c_struct(False)), _ORDER[order], int(in_dim), int(units_per_level),
                                                 _ACCUM[accum])
    L.check(min(rc, 0), "sg_multilink_agg_resolve_order2")
    name = _ORDER_NAMES[rc]
    if name == "fused" and not plan.ensure_fused():
        if order == "fused":
            raise L.StarGCNError("the plan's fused edge orders are not built and cannot be built during stream capture "
                                 "(run one eager step first)")
        import warnings
        warnings.warn("MultiLinkPlan: fused edge orders cannot be built during stream capture; this graph replays the unfused "
                      "order (run one eager step first)")
        return multilink_resolve_order(plan, "auto")
    return name


def _gather_view(plan, order, accum, backward):
    """SG_VIEW_* of the gather that sg_multilink_agg_{fwd,bwd}_hip issues for this order / accumulation."""
    if order == "auto":
        order = multilink_resolve_order(plan, order)
    if order == "transform_first":
        if backward:
            return L.VIEW_T_Q_T if accum == "stack" else L.VIEW_T_IDX_T
        return L.VIEW_C_Q_C if accum == "stack" else L.VIEW_C_Q_D
    return L.VIEW_T_Q_S if backward else L.VIEW_C_IDX_C


def _byref(struct):
    import ctypes
    return ctypes.cast(ctypes.pointer(struct), ctypes.c_void_p)


def _ensure_phases(plan, D, upl, o, a, backward):
    """Build the source-range phases of the view THIS call gathers through, if the library would use them for these sizes
    (sg_multilink_agg_phased_view: the C side's own rule -- width, footprint, SG_GATHER_PHASES) and the plan is large
    enough for two launches to pay.  First use only; `MultiLinkPlan.prepare_phases` does the same ahead of time."""
    if plan.nnz < plan.PHASE_MIN_EDGES or (o, a, backward, D, upl) in plan.__dict__.setdefault("_phase_checked", set()):
        return
    view = L.lib().sg_multilink_agg_phased_view(_byref(plan.c_struct(False)), D, upl, o, a, backward)
    if view < -1:
        L.check(view, "sg_multilink_agg_phased_view")
    if view >= 0 and not plan.ensure_phases(view):
        return              # not built now (stream capture in progress): ask again at the next eager call
    plan._phase_checked.add((o, a, backward, D, upl))


def multilink_agg_fwd(x, weights, biases, plan, accum, act, slope, order):
    """Fused aggregator forward (sg_multilink_agg_fwd_hip).  Returns (out, saved) -- `saved` is the opaque buffer
    the backward needs (None for transform-first)."""
    L.require_gpu(x, *weights)
    lib = L.lib()
    D, upl = x.shape[1], weights[0].shape[0]
    o, a = _ORDER[order], _ACCUM[accum]
    _ensure_phases(plan, D, upl, o, a, 0)
    if order == "fused" and not plan.ensure_fused():
        raise L.StarGCNError("fused edge orders missing during stream capture")
    st = plan.c_struct(order != "transform_first")
    outw = upl * (plan.R if accum == "stack" else 1)
    out = torch.empty((plan.n_dst, outw), dtype=torch.float32, device=x.device)
    nsaved = lib.sg_multilink_agg_saved_bytes(_byref(st), D, upl, o, a)
    saved = torch.empty(nsaved // 4, dtype=torch.float32, device=x.device) if nsaved else None
    ws, wsn = L.workspace(lib.sg_multilink_agg_workspace_bytes(_byref(st), D, upl, o, a, 0), x.device)
    wp, bp = _ptr_array(weights), _ptr_array(biases)
    L.check(lib.sg_multilink_agg_fwd_hip(L.ptr(out), L.ptr(saved), L.ptr(x), wp, bp, _byref(st), D, upl, o, a,
                                         _act_id(act), float(slope), L.ptr(ws), wsn, L.stream_ptr()),
            "sg_multilink_agg_fwd_hip")
    return out, saved


def multilink_agg_bwd(dout, out, saved, x, weights, plan, accum, act, slope, order, need_dx, need_dw, need_db):
    """Fused aggregator backward (sg_multilink_agg_bwd_hip) -> (dx | None, [dW_r] | None, [db_r] | None)."""
    L.require_gpu(dout, x, *weights)
    lib = L.lib()
    D, upl = x.shape[1], weights[0].shape[0]
    o, a = _ORDER[order], _ACCUM[accum]
    _ensure_phases(plan, D, upl, o, a, 1)
    st = plan.c_struct(order != "transform_first")
    dx = torch.empty((plan.n_src, D), dtype=torch.float32, device=x.device) if need_dx else None
    dws = [torch.empty_like(w) for w in weights] if need_dw else None
    dbs = [torch.empty(upl, dtype=torch.float32, device=x.device) for _ in weights] if need_db else None
    ws, wsn = L.workspace(lib.sg_multilink_agg_workspace_bytes(_byref(st), D, upl, o, a, 1), x.device)
    L.check(lib.sg_multilink_agg_bwd_hip(L.ptr(dx), _ptr_array(dws) if dws else None, _ptr_array(dbs) if dbs else None,
                                         L.ptr(dout), L.ptr(out), L.ptr(saved), L.ptr(x), _ptr_array(weights),
                                         _byref(st), D, upl, o, a, _act_id(act), float(slope), L.ptr(ws), wsn,
                                         L.stream_ptr()), "sg_multilink_agg_bwd_hip")
    return dx, dws, dbs


def l2_loss_fwd(pred, target, scale):
    """(loss scalar tensor, grad) with loss = scale * sum 0.5 (pred - target)^2, grad = scale * (pred - target)."""
    L.require_gpu(pred, target)
    pred, target = L.f32c(pred).reshape(-1), L.f32c(target).reshape(-1)
    n = pred.numel()
    if target.numel() != n:
        raise L.StarGCNError("l2_loss: %d predictions vs %d targets" % (n, target.numel()))
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred)
    lib = L.lib()
    ws, wsn = L.workspace(lib.sg_l2_loss_workspace_bytes(n), pred.device)
    L.check(lib.sg_l2_loss_hip(L.ptr(loss), L.ptr(grad), L.ptr(pred), L.ptr(target), n, float(scale), L.ptr(ws), wsn,
                               L.stream_ptr()), "sg_l2_loss_hip")
    return loss, grad
