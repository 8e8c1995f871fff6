"""1-D node partition of the bipartite graph across the GPUs of one node (one process per GPU, torch.distributed
backend 'nccl' == RCCL over xGMI).  The reference has no multi-device code at all; this is
new design (section 8e):

  * users are split into contiguous blocks balanced by edge count; each rank owns its users' rows of the
    user->item CSR and the matching item->user CSR restricted to its users; item features are REPLICATED.
  * user-side aggregation is local.  Item-side aggregation produces a partial (n_item, width) matrix per rank
    -> one all-reduce(sum) of n_item*width*4 bytes BEFORE the activation ("boundary messages").
  * autograd crossings follow the f/g pattern: a replicated tensor entering rank-local work passes through
    `copy_to_local` (forward identity, backward all-reduce); a rank-local partial leaving for the replicated side
    passes through `reduce_from_local` (forward all-reduce, backward identity).  Parameters used inside the local
    region get partial gradients (summed by `allreduce_grads`), parameters of the replicated region already see
    the total gradient on every rank.
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU): messages are kept few and large (one per layer and
    direction, one flat buffer for all local-region parameter gradients).
"""
import numpy as np
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


_FORCE_ONE_RANK = __import__("os").environ.get("SG_BENCH_FORCE_DIST") == "1"   # read once: this sits on the hot path


def _active():
    """Collectives are issued when there is more than one rank -- or, for a development check of the RCCL code path
    on a single GPU, when SG_BENCH_FORCE_DIST=1 initialised a one-rank group."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_ONE_RANK)


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def all_reduce_sum(t):
    """Sum `t` over ranks, returning a NEW tensor (never mutates autograd-owned buffers).  RCCL ('nccl') reduces
    device tensors in place over xGMI; with the 'gloo' backend (CPU tests, or two test ranks sharing one GPU) device
    tensors are staged through the host explicitly."""
    if dist.get_backend() == "gloo" and t.is_cuda:
        h = t.detach().cpu().contiguous()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        return h.to(t.device)
    y = t.detach().contiguous().clone()
    dist.all_reduce(y, op=dist.ReduceOp.SUM)
    return y


class _CopyToLocal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return all_reduce_sum(g)


class _ReduceFromLocal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return all_reduce_sum(x)

    @staticmethod
    def backward(ctx, g):
        return g


def copy_to_local(x):
    return _CopyToLocal.apply(x) if _active() else x


def reduce_from_local(x):
    return _ReduceFromLocal.apply(x) if _active() else x


def allreduce_grads(params):
    """Sum the gradients of local-region parameters over ranks through ONE flat buffer."""
    if not _active():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = all_reduce_sum(torch.cat([g.reshape(-1) for g in grads]))
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


class NodePartition(object):
    """Which node types are rank-local (sharded by rows) and which are replicated."""

    def __init__(self, local_keys, replicated_keys):
        self.local_keys, self.replicated_keys = set(local_keys), set(replicated_keys)

    def crossing_in(self, dst_key, src_key):
        """replicated source features consumed by a rank-local aggregation"""
        return src_key in self.replicated_keys and dst_key in self.local_keys

    def crossing_out(self, dst_key, src_key):
        """partial aggregate over rank-local sources destined to a replicated node type"""
        return dst_key in self.replicated_keys and src_key in self.local_keys


def balanced_row_blocks(ind_ptr, n_parts):
    """Contiguous row blocks with (almost) equal edge counts: boundaries by searching the CSR row pointer."""
    ind_ptr = np.asarray(ind_ptr, dtype=np.int64)
    n_rows, nnz = ind_ptr.size - 1, int(ind_ptr[-1])
    cuts = [0]
    for p in range(1, n_parts):
        cuts.append(int(np.searchsorted(ind_ptr, nnz * p // n_parts, side="left")))
    cuts.append(n_rows)
    cuts = np.maximum.accumulate(np.minimum(cuts, n_rows))
    return [(int(cuts[i]), int(cuts[i + 1])) for i in range(n_parts)]
