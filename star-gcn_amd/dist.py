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
  * OVERLAP: every collective of the data path runs on a dedicated communication stream.  The split forms
    `reduce_start` / `reduce_wait` (forward) and `grad_wait` / `copy_to_local_async` (backward) let the layer put
    independent rank-local work between the launch of a collective and the first use of its result: the item-side
    all-reduce of a layer travels over xGMI while the user-side aggregation of the same layer runs, and in the
    backward pass the all-reduce of d(item features) overlaps the item-side aggregator gradient
    (mxgraph/layers/layers.py: StackedHeterGCNLayers.heter_sage).
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU): messages are kept few and large (one per layer and
    direction, one flat buffer for all local-region parameter gradients).
"""
import os

import numpy as np
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


_FORCE_ONE_RANK = os.environ.get("SG_BENCH_FORCE_DIST") == "1"   # read once: this sits on the hot path


def _active():
    """Collectives are issued when there is more than one rank -- or, for a development check of the RCCL code path
    on a single GPU, when SG_BENCH_FORCE_DIST=1 initialised a one-rank group."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE_ONE_RANK)


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


# ---- communication stream + statistics --------------------------------------------------------------------------
class _CommStats(object):
    """Bytes / calls / device time of the data-path collectives (bench.py reports them per step)."""

    def __init__(self):
        self.enabled = False
        self.reset()

    def reset(self):
        self.bytes, self.calls, self.events, self.host_s, self.waits = 0, 0, [], 0.0, []

    def read(self):
        """device_ms: time of the collectives on the communication stream; exposed_ms: time the COMPUTE stream spent
        blocked on a collective's completion event (what overlap did not hide).  gloo runs are synchronous: all of
        their time is exposed."""
        ms = exposed = 0.0
        for a, b in self.events:
            b.synchronize()
            ms += a.elapsed_time(b)
        for a, b in self.waits:
            b.synchronize()
            exposed += a.elapsed_time(b)
        return {"bytes": int(self.bytes), "calls": int(self.calls), "device_ms": ms + self.host_s * 1e3,
                "exposed_ms": exposed + self.host_s * 1e3}


STATS = _CommStats()
_comm_streams = {}


def comm_stream(device):
    """The side stream every data-path collective of this process is enqueued on (one per device)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    s = _comm_streams.get(key)
    if s is None:
        s = _comm_streams[key] = torch.cuda.Stream(device=device)
    return s


def capture_error_mode():
    """The `capture_error_mode` a step with RCCL collectives should be captured under (torch.cuda.graph(...,
    capture_error_mode=...)): 'thread_local' confines the capture's "unsafe call" checks to the capturing thread, so the
    hipEventQuery polls of torch's RCCL watchdog THREAD on the warm-up collectives' completion events are not answered with
    hipErrorCapturedEvent (which the watchdog turns into a process abort).  'global' (torch's default) without RCCL."""
    return "thread_local" if (_active() and dist.get_backend() == "nccl") else "global"


def quiesce_for_capture(device=None, settle_s=0.5):
    """Call right before a step with RCCL collectives is captured into a hipGraph.  torch's RCCL watchdog thread polls the
    completion events of the collectives enqueued so far (eagerly: warm-up steps) about every 100 ms and retires them; HIP
    answers hipEventQuery with hipErrorCapturedEvent when the event's stream has MEANWHILE entered a GLOBAL-mode capture,
    which the watchdog turns into a process abort.  The guarantee is `capture_error_mode()` (capture in 'thread_local' mode:
    the watchdog's queries are not the capturing thread's); this function is the belt to those braces and BEST EFFORT only
    (ADVICE r5): it drains the device, waits until every warm-up collective this module launched reports completion
    (Work.is_completed on the handles kept by _launch_sum), then gives the watchdog one polling period to retire them
    (SG_CAPTURE_SETTLE_S overrides the wait).  No-op without RCCL."""
    if device is not None:
        torch.cuda.synchronize(device)
    elif torch.cuda.is_available():
        torch.cuda.synchronize()
    if _active() and dist.get_backend() == "nccl":
        import time
        deadline = time.perf_counter() + 10.0
        while _works and time.perf_counter() < deadline:
            w = _works[0]
            try:
                done = w.is_completed()
            except Exception:
                done = True
            if done:
                _works.pop(0)
            else:
                time.sleep(0.001)
        del _works[:]
        time.sleep(float(os.environ.get("SG_CAPTURE_SETTLE_S", settle_s)))


_works = []          # Work handles of the eager RCCL collectives since the last quiesce (bounded: see _launch_sum)


def _wait_on(cur, done):
    """make stream `cur` wait for event `done`; with statistics on, bracket the wait with two events on `cur` whose
    distance is the time `cur` sat blocked (0 when the collective had already finished)"""
    if STATS.enabled and not torch.cuda.is_current_stream_capturing():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(cur)
        cur.wait_event(done)
        b.record(cur)
        STATS.waits.append((a, b))
    else:
        cur.wait_event(done)


class _Pending(object):
    """Completion handle of collectives launched on the communication stream (no event: already complete).
    `users` counts the crossings that were given this handle in the forward pass: a replicated tensor that feeds MORE
    than one rank-local consumer must not hand out in-flight gradient buffers (autograd would add them on the compute
    stream before the wait), so `_CopyToLocalAsync` falls back to a blocking all-reduce in that case."""
    __slots__ = ("events", "users", "exclusive")

    def __init__(self, exclusive=False):
        self.events = []
        self.users = 0
        self.exclusive = exclusive      # the tensor guarded by this handle is consumed through crossings ONLY

    def wait(self):
        if self.events:
            cur = torch.cuda.current_stream()
            for e in self.events:
                _wait_on(cur, e)
            self.events = []


class _SumCheck(object):
    """Self-check of the all-reduces (bench.py `partition_check`, outside any timed region): while enabled, every _launch_sum
    also forms the float64 sum and sum of magnitudes of the local buffer BEFORE the collective, all-reduces those two scalars
    on their own, and compares the sum of the reduced buffer with the reduced scalar: |sum(result) - sum_r sum(local_r)| /
    sum_r sum |local_r|.  A checksum of checksums: it proves the collective added every rank's buffer, whatever the buffers
    hold, to fp32 summation accuracy.  Records (elements, relative error) per collective; forces completion of each
    collective on the spot (so it serialises the step it watches)."""

    def __init__(self):
        self.enabled = False
        self.records = []

    def reset(self):
        self.records = []


CHECK = _SumCheck()


def _check_before(y):
    if not CHECK.enabled or (y.is_cuda and torch.cuda.is_current_stream_capturing()):
        return None
    yd = y.detach().double()
    return torch.stack([yd.sum(), yd.abs().sum()])


def _check_after(y, before):
    after = y.detach().double().sum()
    c = before.cpu() if dist.get_backend() != "nccl" else before
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    c = c.cpu()
    CHECK.records.append((int(y.numel()), abs(float(after) - float(c[0])) / max(float(c[1]), 1e-300)))


def _launch_sum(y, pending=None):
    """Sum `y` over ranks IN PLACE.  RCCL ('nccl'): enqueued on the communication stream behind the work already
    queued on the current stream; the caller's stream only waits when `pending.wait()` is called (immediately when
    `pending` is None).  'gloo' (CPU tests, or two test ranks sharing one GPU): synchronous, device tensors are
    staged through the host explicitly."""
    if STATS.enabled:
        STATS.bytes += y.numel() * y.element_size()
        STATS.calls += 1
    chk = _check_before(y)
    if dist.get_backend() != "nccl" or not y.is_cuda:
        import time
        t0 = time.perf_counter() if STATS.enabled else 0.0
        if y.is_cuda:
            h = y.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            y.copy_(h)
        else:
            dist.all_reduce(y, op=dist.ReduceOp.SUM)
        if STATS.enabled:
            STATS.host_s += time.perf_counter() - t0
        if chk is not None:
            _check_after(y, chk)
        return
    cur = torch.cuda.current_stream(y.device)
    cs = comm_stream(y.device)
    cs.wait_stream(cur)
    timed = STATS.enabled and not torch.cuda.is_current_stream_capturing()     # timing events cannot be captured
    with torch.cuda.stream(cs):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cs)
        if torch.cuda.is_current_stream_capturing():
            dist.all_reduce(y, op=dist.ReduceOp.SUM)
        else:       # same stream semantics (the communication stream waits for the collective), but the handle is kept for
            work = dist.all_reduce(y, op=dist.ReduceOp.SUM, async_op=True)      # quiesce_for_capture
            work.wait()
            _works.append(work)
            if len(_works) > 256:
                del _works[:128]
        if timed:
            e1.record(cs)
            STATS.events.append((e0, e1))
        done = torch.cuda.Event()
        done.record(cs)
    y.record_stream(cs)
    if pending is None or chk is not None:
        _wait_on(cur, done)
    if pending is not None:
        pending.events.append(done)
    if chk is not None:
        _check_after(y, chk)


def all_reduce_sum(t):
    """Sum `t` over ranks, returning a NEW tensor (never mutates autograd-owned buffers); blocking with respect to
    the current stream."""
    y = t.detach().contiguous().clone()
    _launch_sum(y)
    return y


# ---- autograd crossings -------------------------------------------------------------------------------------------
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


class _ReduceStart(torch.autograd.Function):
    """forward: launch the all-reduce of a rank-local partial on the communication stream and return the buffer it
    lands in (NOT valid on the compute stream before `_ReduceWait`); backward: identity."""

    @staticmethod
    def forward(ctx, x, pending, owned):
        # owned=True (the layer's promise): `x` is the aggregator's pre-activation partial, a fresh buffer that nothing
        # else reads (the aggregator does not save an un-activated output, functional._MultiLinkAgg) -- reduced IN PLACE,
        # no 1 GB clone per collective at config 5.  Without that promise, or for a view / non-contiguous input, the
        # sum lands in a copy and the caller's tensor is left alone.
        if not owned or x._base is not None or not x.is_contiguous():
            y = x.detach().contiguous().clone()
            _launch_sum(y, pending)
            return y
        ctx.mark_dirty(x)
        _launch_sum(x, pending)
        return x

    @staticmethod
    def backward(ctx, g):
        return g, None, None


class _ReduceWait(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, pending):
        pending.wait()
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _GradWait(torch.autograd.Function):
    """forward: identity.  backward: make the compute stream wait for the gradient all-reduces that
    `_CopyToLocalAsync` launched for this tensor.  Applied where the replicated tensor is PRODUCED, so that in the
    backward pass it runs after every node created later -- i.e. after the rank-local gradient work that the
    all-reduce is meant to overlap."""

    @staticmethod
    def forward(ctx, x, pending):
        ctx.pending = pending
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.pending.wait()
        return g, None


class _CopyToLocalAsync(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pending, owned):
        ctx.pending, ctx.owned = pending, owned
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        # owned=True (the layer's promise): `g` is the data gradient the rank-local aggregator (or its dropout) has just
        # produced, a fresh buffer that only this edge of the graph holds, so it is summed over the ranks IN PLACE.
        # Autograd may hand one gradient tensor OBJECT to several edges (identity / add consumers, retain_grad, hooks):
        # without the promise -- and for views / strided gradients -- the sum lands in a copy.
        y = g if (ctx.owned and g._base is None and g.is_contiguous()) else g.detach().contiguous().clone()
        p = ctx.pending
        if p.users > 1 or not p.exclusive:
            # several gradients meet at the replicated tensor (more than one crossing shares this handle, or the tensor
            # has consumers that are not crossings): autograd ADDS them on the compute stream before `_GradWait` runs, so
            # every buffer must be complete when it is returned -- blocking all-reduce
            _launch_sum(y)
        else:
            _launch_sum(y, p)
        return y, None, None


def copy_to_local(x):
    return _CopyToLocal.apply(x) if _active() else x


def reduce_from_local(x):
    return _ReduceFromLocal.apply(x) if _active() else x


def reduce_start(x, owned=False):
    """-> (buffer, pending).  The all-reduce of `x` is in flight; call `reduce_wait(buffer, pending)` before use.
    owned=True: the caller guarantees that nothing else reads `x` (a fresh, unsaved buffer) -- it is then reduced in
    place and returned; by default the sum lands in a copy."""
    if not _active():
        return x, None
    p = _Pending()
    return _ReduceStart.apply(x, p, bool(owned)), p


def reduce_wait(y, pending):
    return y if pending is None else _ReduceWait.apply(y, pending)


def grad_wait(x, exclusive=False):
    """-> (x', pending) for a replicated tensor about to enter rank-local work through `copy_to_local_async`.
    exclusive=True: the caller guarantees that x' is consumed by ONE `copy_to_local_async` crossing and by nothing else;
    only then may the crossing return a gradient buffer whose all-reduce is still in flight (awaited by x' 's backward
    node).  Any other consumer of x' would have its gradient ADDED to that in-flight buffer before the wait, so without
    the guarantee the crossing falls back to a blocking all-reduce."""
    if not _active() or not x.requires_grad:
        return x, None
    p = _Pending(exclusive=bool(exclusive))
    return _GradWait.apply(x, p), p


def copy_to_local_async(x, pending, owned=False):
    """owned=True: the gradient that will arrive at this crossing is a fresh buffer no other graph edge holds (the data
    gradient of the consuming aggregator): it is all-reduced in place; by default it is copied first."""
    if not _active():
        return x
    if pending is None:
        return copy_to_local(x)
    pending.users += 1
    return _CopyToLocalAsync.apply(x, pending, bool(owned))


def allreduce_grads(params):
    """Sum the gradients of local-region parameters over ranks through ONE flat buffer."""
    if not _active():
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    flat = all_reduce_sum(torch.cat([g.reshape(-1) for g in grads]))
    # one multi-tensor copy back instead of one launch per parameter (46 parameters = 92 tiny copies and 0.5 ms of an
    # 8-rank ML-10M step, measured with SG_BENCH_EMULATE_WORLD=8)
    views = [v.view_as(g) for v, g in zip(torch.split(flat, [g.numel() for g in grads]), grads)]
    torch._foreach_copy_(grads, views)


def broadcast_parameters(params, src=0):
    """Make replicated parameters bit-identical on every rank (rank `src` wins) through ONE flat buffer."""
    if not _active():
        return
    params = [p for p in params if not isinstance(p, torch.nn.UninitializedParameter)]
    if not params:
        return
    flat = torch.cat([p.detach().reshape(-1) for p in params])
    if dist.get_backend() != "nccl" and flat.is_cuda:
        h = flat.cpu()
        dist.broadcast(h, src=src)
        flat = h.to(flat.device)
    else:
        dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            p.copy_(flat[off:off + n].view_as(p))
            off += n


_rep_generators = {}
_rep_seed = [None]


def set_replicated_seed(seed=None, src=0):
    """Seed of the mask stream of `replicated_dropout`; every rank must end up with the SAME value.  seed=None: rank
    `src` draws one from torch's default generator (so `torch.manual_seed` governs it) and broadcasts it.  Calling
    it again resets the stream (e.g. between models, or to replay a run).  Returns the seed in use."""
    if seed is None:
        t = torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
        if _active() and dist.get_world_size() > 1:
            if dist.get_backend() == "nccl":
                d = t.cuda()
                dist.broadcast(d, src=src)
                t = d.cpu()
            else:
                dist.broadcast(t, src=src)
        seed = int(t.item())
    _rep_seed[0] = int(seed)
    _rep_generators.clear()
    return _rep_seed[0]


def replicated_dropout(x, p, training):
    """Dropout for a REPLICATED tensor of a node-partitioned run: the mask comes from a generator that is seeded
    identically on every rank (`set_replicated_seed`) and advanced by the same sequence of calls, so replicas stay
    identical (a per-rank mask would let the replicated activations -- and the gradients of replicated parameters --
    drift apart).  The generator is not registered with a hipGraph: a step that uses dropout on replicated tensors
    cannot be captured (the captured benchmark step runs with dropout 0, which returns before touching it)."""
    if not training or p <= 0.0:
        return x
    if torch.cuda.is_available() and x.is_cuda and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("replicated_dropout cannot run inside a hipGraph capture (its generator is not graph-registered)")
    if _rep_seed[0] is None:
        set_replicated_seed()
    key = str(x.device)
    g = _rep_generators.get(key)
    if g is None:
        g = _rep_generators[key] = torch.Generator(device=x.device)
        g.manual_seed(_rep_seed[0])
    keep = (torch.rand(x.shape, generator=g, device=x.device) >= p).to(x.dtype)
    return x * keep / (1.0 - p)


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
    """Contiguous row blocks with (almost) equal edge counts: boundaries by searching the CSR row pointer.  Every block
    holds at least one row whenever there are at least `n_parts` rows (a hub row that alone carries several blocks' worth of
    edges would otherwise leave its neighbours with EMPTY blocks -- a rank without users); with fewer rows than parts the
    trailing blocks are empty."""
    ind_ptr = np.asarray(ind_ptr, dtype=np.int64)
    n_rows, nnz = ind_ptr.size - 1, int(ind_ptr[-1])
    cuts = [0]
    for p in range(1, n_parts):
        cuts.append(int(np.searchsorted(ind_ptr, nnz * p // n_parts, side="left")))
    cuts.append(n_rows)
    cuts = np.maximum.accumulate(np.minimum(cuts, n_rows))
    if n_rows >= n_parts:
        for p in range(1, n_parts):                      # strictly increasing from the left ...
            cuts[p] = max(cuts[p], cuts[p - 1] + 1)
        for p in range(n_parts - 1, 0, -1):              # ... leaving at least one row for every block to the right
            cuts[p] = min(cuts[p], cuts[p + 1] - 1)
    return [(int(cuts[i]), int(cuts[i + 1])) for i in range(n_parts)]
