"""`contrib` namespace: the segment operators of the reference with the reference's keyword names
(`mx.nd.contrib.seg_*` / `F.contrib.seg_*`, registered at reference seg_op.cc:339-861), differentiable through
torch autograd with the gradient wiring of the reference's FGradient graphs:

  seg_weighted_pool : d data = _backward_seg_take_k_corr_embed2(weights, ograd, ...), d weights =
                      seg_take_k_corr(ograd, data, ...)                                  (seg_op.cc:700-712)
  seg_take_k_corr   : d embed1 = seg_weighted_pool(embed2, ograd, ...), d embed2 =
                      _backward_seg_take_k_corr_embed2(ograd, embed1, ...)               (seg_op.cc:647-659)
  seg_sum           : broadcast_to(ograd)                                                (seg_op.cc:370-395)
  seg_broadcast_add : identity / seg_sum(ograd);  _mul: broadcast_mul(ograd, rhs) / seg_sum(ograd*lhs);
  seg_broadcast_to  : seg_sum(ograd)                                                     (seg_op.cc:427-538)
  seg_softmax       : val * (ograd - sum_seg(ograd*val))                                 (seg_op.cc:575-600)
  seg_pool          : sum / avg / max(argmax = edge position)                            (seg_op.cc:754-861)

All tensors must live on the GPU (fp32 data, int32 indices); there is no CPU path.
"""
import torch

from . import _lib as L
from . import ops
from .plan import transpose_plan_for

__all__ = ["seg_weighted_pool", "seg_take_k_corr", "seg_sum", "seg_broadcast_add", "seg_broadcast_mul",
           "seg_broadcast_to", "seg_softmax", "seg_pool"]


def _prep(*ts):
    out = []
    for t in ts:
        if t is None:
            out.append(None)
        elif t.dtype in (torch.int32, torch.int64, torch.int16, torch.uint8):
            out.append(L.i32c(t))
        else:
            out.append(L.f32c(t))
    return out


class _SegWeightedPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, weights, indices, indptr):
        data, weights, indices, indptr = _prep(data, weights, indices, indptr)
        ctx.save_for_backward(data, weights, indices, indptr)
        return ops.seg_weighted_pool(data, weights, indices, indptr)

    @staticmethod
    def backward(ctx, ograd):
        data, weights, indices, indptr = ctx.saved_tensors
        ograd = L.f32c(ograd)
        d_data = d_w = None
        if ctx.needs_input_grad[0]:
            tplan = transpose_plan_for(indices, indptr, data.shape[1])
            d_data = ops.seg_weighted_pool_bwd_data(weights, ograd, tplan, data.shape[1])
        if ctx.needs_input_grad[1]:
            d_w = ops.seg_take_k_corr(ograd, data, indices, indptr)
        return d_data, d_w, None, None


class _SegTakeKCorr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, embed1, embed2, neighbor_ids, neighbor_indptr):
        embed1, embed2, ids, indptr = _prep(embed1, embed2, neighbor_ids, neighbor_indptr)
        ctx.save_for_backward(embed1, embed2, ids, indptr)
        return ops.seg_take_k_corr(embed1, embed2, ids, indptr)

    @staticmethod
    def backward(ctx, ograd):
        embed1, embed2, ids, indptr = ctx.saved_tensors
        ograd = L.f32c(ograd)
        d1 = d2 = None
        if ctx.needs_input_grad[0]:
            d1 = ops.seg_weighted_pool(embed2, ograd, ids, indptr)
        if ctx.needs_input_grad[1]:
            tplan = transpose_plan_for(ids, indptr, embed2.shape[1])
            d2 = ops.seg_weighted_pool_bwd_data(ograd, embed1, tplan, embed2.shape[1])
        return d1, d2, None, None


class _SegSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, indptr):
        data, indptr = _prep(data, indptr)
        ctx.save_for_backward(indptr)
        ctx.nnz = data.shape[1]
        return ops.seg_sum(data, indptr)

    @staticmethod
    def backward(ctx, ograd):
        (indptr,) = ctx.saved_tensors
        return ops.seg_broadcast(None, L.f32c(ograd), indptr, 2, nnz=ctx.nnz), None


class _SegBroadcast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lhs, rhs, indptr, op, nnz):
        lhs, rhs, indptr = _prep(lhs, rhs, indptr)
        ctx.op = op
        ctx.save_for_backward(lhs, rhs, indptr)
        return ops.seg_broadcast(lhs, rhs, indptr, op, nnz=nnz)

    @staticmethod
    def backward(ctx, ograd):
        lhs, rhs, indptr = ctx.saved_tensors
        ograd = L.f32c(ograd)
        d_lhs = d_rhs = None
        if ctx.op == 0:  # add
            if ctx.needs_input_grad[0]:
                d_lhs = ograd
            if ctx.needs_input_grad[1]:
                d_rhs = ops.seg_sum(ograd, indptr)
        elif ctx.op == 1:  # mul
            if ctx.needs_input_grad[0]:
                d_lhs = ops.seg_broadcast(ograd, rhs, indptr, 1)
            if ctx.needs_input_grad[1]:
                d_rhs = ops.seg_sum(ograd * lhs, indptr)
        else:  # to
            if ctx.needs_input_grad[1]:
                d_rhs = ops.seg_sum(ograd, indptr)
        return d_lhs, d_rhs, None, None, None


class _SegSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, indptr):
        data, indptr = _prep(data, indptr)
        val = ops.seg_softmax(data, indptr)
        ctx.save_for_backward(val, indptr)
        return val

    @staticmethod
    def backward(ctx, ograd):
        val, indptr = ctx.saved_tensors
        return ops.seg_softmax_bwd(L.f32c(ograd), val, indptr), None


class _SegPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, indices, indptr, pool_type):
        data, indices, indptr = _prep(data, indices, indptr)
        out, arg = ops.seg_pool(data, indices, indptr, pool_type)
        ctx.pool_type, ctx.total = pool_type, data.shape[1]
        ctx.save_for_backward(indices, indptr, arg if arg is not None else indptr)
        return out

    @staticmethod
    def backward(ctx, ograd):
        indices, indptr, arg = ctx.saved_tensors
        tplan = transpose_plan_for(indices, indptr, ctx.total)
        pi = arg if ctx.pool_type == "max" else None
        return ops.seg_pool_bwd(L.f32c(ograd), pi, indptr, tplan, ctx.total, ctx.pool_type), None, None, None


def seg_weighted_pool(data, weights, indices, indptr):
    """data (B,T,C), weights (B,nnz), indices (nnz), indptr (S+1) -> (B,S,C)."""
    return _SegWeightedPool.apply(data, weights, indices, indptr)


def seg_take_k_corr(embed1, embed2, neighbor_ids, neighbor_indptr):
    """embed1 (K,N,C), embed2 (K,M,C), neighbor_ids (nnz), neighbor_indptr (N+1) -> (K,nnz)."""
    return _SegTakeKCorr.apply(embed1, embed2, neighbor_ids, neighbor_indptr)


def seg_sum(data, indptr):
    return _SegSum.apply(data, indptr)


def seg_broadcast_add(lhs, rhs, indptr):
    return _SegBroadcast.apply(lhs, rhs, indptr, 0, lhs.shape[1])


def seg_broadcast_mul(lhs, rhs, indptr):
    return _SegBroadcast.apply(lhs, rhs, indptr, 1, lhs.shape[1])


def seg_broadcast_to(rhs, indptr, nnz):
    return _SegBroadcast.apply(None, rhs, indptr, 2, int(nnz))


def seg_softmax(data, indptr):
    return _SegSoftmax.apply(data, indptr)


def seg_pool(data, indices, indptr, pool_type="sum"):
    if pool_type not in ("sum", "avg", "max"):
        raise L.StarGCNError("pool_type must be 'avg', 'sum' or 'max'")
    return _SegPool.apply(data, indices, indptr, pool_type)
