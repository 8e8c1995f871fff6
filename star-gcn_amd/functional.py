"""Differentiable building blocks of the hot path, each backed by the C ABI (no torch math on the path):

  linear             Dense / FullyConnected with fused bias + activation on the fp32 MFMA GEMM
  multilink_aggregate the multi-link graph-conv aggregation of reference aggregators.py:111-163 as ONE native call
                     (sg_multilink_agg_fwd_hip: one gather launch + one MFMA contraction; the reference runs
                     R FullyConnected + R seg_weighted_pool + add_n/concat + activation)
  take_rows / masked_embed   row gathers with atomic-free, plan-based gradients

Association order of the aggregation.  The reference computes  A_r (X W_r^T + 1 b_r^T)  ("transform first").
Algebraically that equals  (A_r X) W_r^T + (A_r 1) b_r^T  ("aggregate first").  Either way ONE side of the
bipartite graph carries an R-times expanded matrix; its GEMM costs 2*n*R*D*U flops with n the number of nodes
on THAT side.  `order='auto'` puts the expansion on the smaller side (transform first when n_src <= n_dst,
aggregate first otherwise) -- at MovieLens-10M shape that is 6.5x fewer flops for the user-side aggregation.
"""
import torch

from . import _lib as L
from . import ops


class _Linear(torch.autograd.Function):
    """y = act(x @ W^T + b)   x (N,K)  W (U,K)  b (U) | None.  Saves y to evaluate act' in backward."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, slope):
        x, weight = L.f32c(x), L.f32c(weight)
        bias = L.f32c(bias) if bias is not None else None
        y = ops.gemm(x, weight, trans_b=True, bias=bias, act=act, slope=slope)
        ctx.act, ctx.slope, ctx.has_bias = act, slope, bias is not None
        ctx.save_for_backward(x, weight, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dx = dw = db = None
        if ctx.has_bias and ctx.needs_input_grad[2] and dy.dim() == 2:
            dpre, db = ops.act_bwd_colsum(L.f32c(dy), y, ctx.act, ctx.slope)      # one pass for both
        else:
            dpre = ops.act_bwd(L.f32c(dy), y, ctx.act, ctx.slope)
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dpre, weight)                     # (N,U) @ (U,K)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm(dpre, x, trans_a=True)            # (U,N) @ (N,K), split-K over N
        if ctx.has_bias and ctx.needs_input_grad[2] and db is None:
            db = ops.colsum(dpre)
        return dx, dw, db, None, None


def linear(x, weight, bias=None, act=None, slope=0.1):
    return _Linear.apply(x, weight, bias, act, slope)


class _MultiLinkAgg(torch.autograd.Function):
    """out = act( accum_r A_r (x W_r^T + b_r) ) through the fused native entry points
    sg_multilink_agg_{fwd,bwd}_hip (csrc/multilink.hip): one C call forward, one backward; the per-level
    parameters are passed in the reference layout (R weights (U', D), R biases (U'))."""

    @staticmethod
    def forward(ctx, x, plan, accum, act, slope, order, *params):
        R = plan.R
        x = L.f32c(x)
        weights = [L.f32c(w) for w in params[:R]]
        biases = [L.f32c(b) for b in params[R:]]
        out, saved = ops.multilink_agg_fwd(x, weights, biases, plan, accum, act, slope, order)
        ctx.plan, ctx.accum, ctx.act, ctx.slope, ctx.order = plan, accum, act, slope, order
        ctx.saved_z = saved            # opaque native buffer (Zext of the aggregate-first order), not a graph tensor
        # the output is only needed to evaluate act'; without an activation (the pre-activation partial of a
        # node-partitioned run) it is NOT saved, so dist.reduce_start may all-reduce it in place
        ctx.has_out = ops._act_id(act) != 0
        if ctx.has_out:
            ctx.save_for_backward(x, out, *weights)
        else:
            ctx.save_for_backward(x, *weights)
        return out

    @staticmethod
    def backward(ctx, dout):
        x = ctx.saved_tensors[0]
        out = ctx.saved_tensors[1] if ctx.has_out else None
        weights = list(ctx.saved_tensors[2 if ctx.has_out else 1:])
        R = ctx.plan.R
        need_dw = any(ctx.needs_input_grad[6:6 + R])
        need_db = any(ctx.needs_input_grad[6 + R:6 + 2 * R])
        dx, dws, dbs = ops.multilink_agg_bwd(L.f32c(dout), out, ctx.saved_z, x, weights, ctx.plan, ctx.accum, ctx.act,
                                             ctx.slope, ctx.order, ctx.needs_input_grad[0], need_dw, need_db)
        ctx.saved_z = None
        return (dx, None, None, None, None, None) + tuple(dws or [None] * R) + tuple(dbs or [None] * R)


def multilink_aggregate(x, weights, biases, plan, accum="stack", act=None, slope=0.1, order="auto"):
    """act( accum_r  A_r (x W_r^T + b_r) )  for the R levels of `plan`; weights[r] (U', D), biases[r] (U')."""
    if accum not in ("sum", "stack"):
        raise NotImplementedError(accum)
    if order not in ("auto", "transform_first", "aggregate_first", "fused"):
        raise L.StarGCNError("order must be 'auto', 'transform_first', 'aggregate_first' or 'fused'")
    if len(weights) != plan.R or len(biases) != plan.R:
        raise L.StarGCNError("need one weight/bias per link level (%d)" % plan.R)
    order = ops.multilink_resolve_order(plan, order, x.shape[1], weights[0].shape[0], accum)
    return _MultiLinkAgg.apply(x, plan, accum, act, slope, order, *weights, *biases)


class _Activation(torch.autograd.Function):
    """y = act(x) on the native elementwise kernel; the derivative is evaluated from the OUTPUT, as the fused epilogues do."""

    @staticmethod
    def forward(ctx, x, act, slope):
        y = ops.act_fwd(x, act, slope)
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.act_bwd(L.f32c(dy), y, ctx.act, ctx.slope), None, None


def activation(x, act, slope=0.1):
    """leaky / relu / sigmoid / tanh as ONE native pass each way (used where the activation cannot be fused into the
    producing kernel: after the all-reduce of a node-partitioned aggregate)."""
    if act is None or ops._act_id(act) == 0:
        return x
    return _Activation.apply(x, act, slope)


class _TakeRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, tplan):
        table = L.f32c(table)
        ctx.tplan, ctx.shape = tplan, table.shape
        return ops.masked_embed(table, tplan.ids, None)

    @staticmethod
    def backward(ctx, dout):
        tp = ctx.tplan
        dout = L.f32c(dout)
        if tp.inv_ids is not None:
            return ops.masked_embed(dout, tp.inv_ids, None), None
        dt = torch.empty(ctx.shape, dtype=torch.float32, device=dout.device)
        ops.gather_sum(dt, dout, tp.t_pos, tp.t_indptr, None, tp.n_rows, ctx.shape[1])
        return dt, None


def take_rows(table, take_plan):
    """out[i] = table[ids[i]] (zero row where ids[i] == -1); gradient = segment sum over the TakePlan."""
    if take_plan.identity:
        return table
    return _TakeRows.apply(table, take_plan)


class _L2Loss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, scale):
        loss, grad = ops.l2_loss_fwd(pred, target, scale)
        ctx.shape = pred.shape
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).view(ctx.shape), None, None


def l2_loss(pred, target, scale):
    """scale * sum 0.5 (pred - target)^2 -- gluon L2Loss of the reference (STAR-GCN.py:550,612) with the reduction;
    value and gradient come from one native pass (sg_l2_loss_hip).  `target` gets no gradient."""
    return _L2Loss.apply(pred, target, scale)
