"""CHECKER (test infrastructure, not product): float64 evaluation of the DEFINITION of the benchmark network -- the
2-layer multi-link GCN of SURVEY.md section 8(d) -- with plain torch ops on whatever device the inputs live on.

Nothing of the product is used: no HIP kernel of libstargcn_hip.so, no plan, no transposed CSR.  The only inputs are
the raw user->item CSR (row pointer, item of every rating, rating level of every rating), the degrees the support is
normalised with, and the parameter VALUES of the network under test.  Every formula cites the reference line it
restates (paths relative to /root/reference):

  support        sqrt(1 / d_row / d_col) evaluated in fp32 in that order       GraphSampler/graph_sampler.cpp:393-420
  aggregator     per level r: FullyConnected(x, W_r, b_r) then seg_weighted_pool over the level's edges, add_n over the
                 levels, activation                                             mxgraph/layers/aggregators.py:141-160
                 seg_weighted_pool: out[s] = sum_{j in seg s} w_j data[idx_j]   seg_ops_cuda/mxnet_op/seg_op.cc:180-207
  layer          out Dense + activation on the aggregate                        mxgraph/layers/layers.py:147-187
  rating head    score = <Dense_u(out_u)[u], Dense_i(out_i)[i]>                 experiments/STAR-GCN.py:249-261, 428-438
  loss           gluon L2Loss = 0.5 (score - y)^2, mean over the ratings        experiments/STAR-GCN.py:550, 610-613

The forward AND the backward pass (chain rule written out, no autograd) run over the WHOLE graph in float64; the
product's fp32 results -- loss, every layer's output rows for both node types, the rating projections, the gradient of
every embedding row and every weight / bias gradient -- are compared element by element.  A 2-hop network has a
receptive field of (nearly) the whole graph, so anything short of a full evaluation could not check a single output row
end to end.  Cost at the 125 M-rating shard of BASELINE config 5 on an MI355X: a few seconds (fp64 GEMMs + chunked
index_add_), about 25 GB of temporaries.

Used by tests/test_gpu_bench_verify.py and by bench.py's `verify` block (outside every timed region).
"""
import math

import torch

SLOPE = 0.1          # LeakyReLU(0.1), mxgraph/layers/common.py:47


def _leaky(x):
    return torch.where(x > 0, x, SLOPE * x)


AMBIGUOUS = 1e-6     # |pre-activation| <= AMBIGUOUS * max |pre-activation|: within the rounding of an fp32 evaluation


def _dleaky(pre, prod_out=None, stats=None):
    """LeakyReLU'(pre).  The derivative is DISCONTINUOUS at 0: an element whose float64 pre-activation lies within fp32
    rounding distance of 0 (a few hundred of the 3e8 elements of a config-5 layer) may legitimately come out on either
    side in an fp32 evaluation, and the two sides differ by a factor of 10 in that element of the gradient.  For exactly
    those elements -- |pre| <= AMBIGUOUS * max|pre|, 4x the forward error this checker measures -- the side the network
    under test took (sign of its activation OUTPUT `prod_out`) is adopted; they are counted in `stats`.  Everywhere
    else the float64 sign decides."""
    pos = pre > 0
    if prod_out is not None:
        amb = pre.abs() <= AMBIGUOUS * float(pre.abs().max())
        ppos = prod_out > 0
        if stats is not None:
            stats["ambiguous_act_elements"] += int(amb.sum())
            stats["adopted_sign_flips"] += int((amb & (ppos != pos)).sum())
            stats["act_elements"] += pre.numel()
        pos = torch.where(amb, ppos, pos)
    return torch.where(pos, torch.ones_like(pre), torch.full_like(pre, SLOPE))


class RawGraph(object):
    """user->item ratings in CSR order: ind_ptr (n_user+1), item (E), level (E) in [0, R); degrees for the support."""

    def __init__(self, ind_ptr, item, level, n_item, R, item_degrees=None, chunk=1 << 22):
        dev = ind_ptr.device
        self.n_user, self.n_item, self.R = int(ind_ptr.numel() - 1), int(n_item), int(R)
        self.E = int(item.numel())
        deg_u = (ind_ptr[1:] - ind_ptr[:-1]).long()
        self.user = torch.repeat_interleave(torch.arange(self.n_user, device=dev, dtype=torch.int32), deg_u)
        self.item = item.to(torch.int32)
        self.level = level.to(torch.int32)
        assert self.user.numel() == self.E and int(self.level.max()) < R and int(self.level.min()) >= 0
        deg_i = torch.bincount(item.long(), minlength=self.n_item) if item_degrees is None else item_degrees.long()
        du, di = deg_u.double(), deg_i.double()
        # std::sqrt(1.0f / float(r) / float(c)): three correctly rounded fp32 operations in the reference's order (the
        # user->item matrix and its transpose divide in different orders).  Each is evaluated in float64 and rounded to
        # fp32 -- exact emulation (53 >= 2*24+2 bits: no double-rounding error for / and sqrt), independent of how the
        # backend's own fp32 sqrt / division round.
        f32 = lambda t: t.float().double()
        self.w_ui = torch.empty(self.E, dtype=torch.float32, device=dev)
        self.w_iu = torch.empty(self.E, dtype=torch.float32, device=dev)
        for a in range(0, self.E, chunk):
            u, i = self.user[a:a + chunk].long(), self.item[a:a + chunk].long()
            self.w_ui[a:a + chunk] = torch.sqrt(f32(f32(1.0 / du[u]) / di[i])).float()
            self.w_iu[a:a + chunk] = torch.sqrt(f32(f32(1.0 / di[i]) / du[u])).float()
        self.chunk = chunk

    def edges(self, dst):
        """(dst row, src row, support) of every rating for the aggregation INTO node type `dst` ('user' | 'item')"""
        return (self.user, self.item, self.w_ui) if dst == "user" else (self.item, self.user, self.w_iu)

    def n(self, key):
        return self.n_user if key == "user" else self.n_item


class _Agg(object):
    """act( sum_r A_r (x W_r^T + b_r) ) and its gradients, float64, definition order: level by level, FullyConnected of
    the level then seg_weighted_pool over the level's edges (memory stays at a few (n, U) float64 matrices; evaluating all
    levels at once would need a 41 GB (n_src, R U) matrix at the config-5 shard)."""

    def __init__(self, g, dst, Ws, bs):
        self.g, self.dst = g, dst
        self.W = [w.double() for w in Ws]                          # R x (U, D)
        self.b = [b.double() for b in bs]
        self.U = Ws[0].shape[0]

    def _level_edges(self, r):
        idx = (self.g.level == r).nonzero().view(-1)
        for a in range(0, idx.numel(), self.g.chunk):
            yield idx[a:a + self.g.chunk]

    def forward(self, x):
        g = self.g
        d_e, s_e, w_e = g.edges(self.dst)
        out = torch.zeros(g.n(self.dst), self.U, dtype=torch.float64, device=x.device)
        for r in range(g.R):
            H = x @ self.W[r].t() + self.b[r]                       # FullyConnected_r(x): (n_src, U)
            for sel in self._level_edges(r):
                rows = H[s_e[sel].long()]
                rows *= w_e[sel].double()[:, None]
                out.index_add_(0, d_e[sel].long(), rows)
            del H
        self.x, self.pre = x, out
        return _leaky(out)

    def backward(self, dh, prod_out=None, stats=None):
        g = self.g
        d_e, s_e, w_e = g.edges(self.dst)
        dpre = dh * _dleaky(self.pre, prod_out, stats)
        dx = torch.zeros_like(self.x)
        dW, db = [], []
        for r in range(g.R):
            G = torch.zeros(self.x.shape[0], self.U, dtype=torch.float64, device=dh.device)     # d H_r
            for sel in self._level_edges(r):
                rows = dpre[d_e[sel].long()]
                rows *= w_e[sel].double()[:, None]
                G.index_add_(0, s_e[sel].long(), rows)
            dW.append(G.t() @ self.x)
            db.append(G.sum(0))
            dx += G @ self.W[r]
            del G
        self.pre = self.x = None
        return dx, dW, db


class _Dense(object):
    def __init__(self, W, b, act):
        self.W, self.b, self.act = W.double(), b.double(), act

    def forward(self, x):
        self.x = x
        self.pre = x @ self.W.t() + self.b
        return _leaky(self.pre) if self.act else self.pre

    def backward(self, dy, prod_out=None, stats=None):
        dpre = dy * _dleaky(self.pre, prod_out, stats) if self.act else dy
        dW, db, dx = dpre.t() @ self.x, dpre.sum(0), dpre @ self.W
        self.pre = self.x = None
        return dx, dW, db


def evaluate(g, params, y, scale, product=None):
    """-> dict of float64 results.  params: {"embed": {key: table}, "layers": [{key: {"W": [R], "b": [R], "Wo", "bo"}}],
    "proj": {key: (W, b)}} with key in ('user', 'item'); y: standardised rating of every edge (CSR order).
    product: a `Capture` of the network under test -- only the SIGN of its activation outputs is read, and only where
    the float64 pre-activation is within fp32 rounding of zero (see _dleaky)."""
    other = {"user": "item", "item": "user"}
    x = {k: params["embed"][k].double() for k in ("user", "item")}
    aggs, outs, res = [], [], {"layer_out": []}
    for lp in params["layers"]:
        a = {k: _Agg(g, k, lp[k]["W"], lp[k]["b"]) for k in ("user", "item")}
        o = {k: _Dense(lp[k]["Wo"], lp[k]["bo"], True) for k in ("user", "item")}
        h = {k: a[k].forward(x[other[k]]) for k in ("user", "item")}
        x = {k: o[k].forward(h[k]) for k in ("user", "item")}
        aggs.append(a)
        outs.append(o)
        res["layer_out"].append(dict(x))
    proj = {k: _Dense(params["proj"][k][0], params["proj"][k][1], False) for k in ("user", "item")}
    p = {k: proj[k].forward(x[k]) for k in ("user", "item")}
    res["proj"] = dict(p)
    # rating head + loss (chunked: a (E, width) float64 gather would be 64 GB at config 5)
    yd = y.double().view(-1)
    loss = torch.zeros((), dtype=torch.float64, device=yd.device)
    ssq = torch.zeros((), dtype=torch.float64, device=yd.device)
    dp = {k: torch.zeros_like(p[k]) for k in ("user", "item")}
    for a in range(0, g.E, g.chunk):
        sl = slice(a, a + g.chunk)
        u, i = g.user[sl].long(), g.item[sl].long()
        ru, ri = p["user"][u], p["item"][i]
        score = (ru * ri).sum(1)
        ssq += (score * score).sum()
        diff = score - yd[sl]
        loss += 0.5 * (diff * diff).sum()
        gs = (scale * diff)[:, None]
        dp["user"].index_add_(0, u, gs * ri)
        dp["item"].index_add_(0, i, gs * ru)
    res["loss"] = loss * scale
    res["score_rms"] = math.sqrt(float(ssq) / max(g.E, 1))
    grads = {"layers": [None] * len(aggs), "proj": {}, "embed": {}}
    stats = {"ambiguous_act_elements": 0, "adopted_sign_flips": 0, "act_elements": 0}
    res["act_stats"] = stats
    dx = {}
    for k in ("user", "item"):
        dx[k], dW, db = proj[k].backward(dp[k])
        grads["proj"][k] = (dW, db)
    for l in range(len(aggs) - 1, -1, -1):
        gl, nxt = {}, {}
        for k in ("user", "item"):
            dh, dWo, dbo = outs[l][k].backward(dx[k], None if product is None else product.layer_out[l][k], stats)
            dsrc, dW, db = aggs[l][k].backward(dh, None if product is None else product.agg_out[l][k], stats)
            gl[k] = {"W": dW, "b": db, "Wo": dWo, "bo": dbo}
            nxt[other[k]] = dsrc
        grads["layers"][l] = gl
        dx = nxt
    grads["embed"] = dx
    res["grads"] = grads
    return res


# ---------------------------------------------------------------------------------------------------------------------
# reading the network under test (parameter VALUES and the tensors it produced) -- no product code is executed here
def net_params(net, name_user="user", name_item="movie"):
    key = {"user": name_user, "item": name_item}
    other = {"user": name_item, "item": name_user}
    assert len(net.encoders) == 1, "the benchmark network is ONE encoder of stacked layers"
    layers = []
    for layer in net.encoders[0]._blocks:
        lp = {}
        for k in ("user", "item"):
            agg = layer.aggregators[(key[k], other[k])]
            assert not agg._ordinal_sharing and agg._accum == "sum"
            R = agg._num_links
            fc = layer._out_fcs[key[k]]
            lp[k] = {"W": [getattr(agg, "weight%d" % r).detach() for r in range(R)],
                     "b": [getattr(agg, "bias%d" % r).detach() for r in range(R)],
                     "Wo": fc.weight.detach(), "bo": fc.bias.detach()}
        layers.append(lp)
    proj = {"user": (net.rating_user_projs[0].weight.detach(), net.rating_user_projs[0].bias.detach()),
            "item": (net.rating_item_projs[0].weight.detach(), net.rating_item_projs[0].bias.detach())}
    embed = {k: net.embed_layers[key[k]].weight.detach() for k in ("user", "item")}
    return {"embed": embed, "layers": layers, "proj": proj}


class Capture(object):
    """forward hooks that keep what the product computed: every layer's output per node type and the rating projections"""

    def __init__(self, net, name_user="user", name_item="movie"):
        self.key = {"user": name_user, "item": name_item}
        self.layer_out = [dict() for _ in net.encoders[0]._blocks]
        self.agg_out = [dict() for _ in net.encoders[0]._blocks]
        self.proj = {}
        self._h = []
        other = {"user": name_item, "item": name_user}
        for l, layer in enumerate(net.encoders[0]._blocks):
            for k in ("user", "item"):
                self._h.append(layer._out_fcs[self.key[k]].register_forward_hook(
                    lambda m, i, o, l=l, k=k: self.layer_out[l].__setitem__(k, o.detach())))
                agg = layer.aggregators[(self.key[k], other[k])]
                self._h.append(agg.register_forward_hook(
                    lambda m, i, o, l=l, k=k: self.agg_out[l].__setitem__(k, o.detach())))
                # node-partitioned runs: the aggregator returns the pre-activation PARTIAL sum of a replicated destination and
                # the layer applies `agg.activation` after the all-reduce -- that call's output is the aggregate proper
                act = getattr(agg, "activation", None)
                if isinstance(act, torch.nn.Module):
                    self._h.append(act.register_forward_hook(
                        lambda m, i, o, l=l, k=k: self.agg_out[l].__setitem__(k, o.detach())))
        self._h.append(net.rating_user_projs[0].register_forward_hook(
            lambda m, i, o: self.proj.__setitem__("user", o.detach())))
        self._h.append(net.rating_item_projs[0].register_forward_hook(
            lambda m, i, o: self.proj.__setitem__("item", o.detach())))

    def close(self):
        for h in self._h:
            h.remove()
        self._h = []


def _rel(got, ref):
    """max |got - ref| relative to the tensor's scale (max |ref|), the measure of every fp32 parity test of this repo"""
    s = float(ref.abs().max())
    return float((got.double() - ref).abs().max()) / max(s, 1e-30), s


def compare(net, cap, loss, ref, name_user="user", name_item="movie"):
    """-> {"max_rel_err", "worst", "rows", "tensors", "per_tensor": {name: rel err}} of the product's fp32 results
    against `ref` = evaluate(...)."""
    key = {"user": name_user, "item": name_item}
    other = {"user": name_item, "item": name_user}
    per, rows = {}, 0
    per["loss"] = abs(float(loss) - float(ref["loss"])) / max(abs(float(ref["loss"])), 1e-30)
    for l, lo in enumerate(ref["layer_out"]):
        for k in ("user", "item"):
            per["layer%d.out.%s" % (l, k)], _ = _rel(cap.layer_out[l][k], lo[k])
            rows += lo[k].shape[0]
    for k in ("user", "item"):
        per["proj.%s" % k], _ = _rel(cap.proj[k], ref["proj"][k])
        per["grad.embed.%s" % k], _ = _rel(net.embed_layers[key[k]].weight.grad, ref["grads"]["embed"][k])
        rows += ref["grads"]["embed"][k].shape[0]
    for l, layer in enumerate(net.encoders[0]._blocks):
        for k in ("user", "item"):
            agg, fc, gl = layer.aggregators[(key[k], other[k])], layer._out_fcs[key[k]], ref["grads"]["layers"][l][k]
            # the R per-level gradients are one contraction: scale of the whole (R, U, D) block
            gW = torch.stack([getattr(agg, "weight%d" % r).grad for r in range(agg._num_links)])
            gb = torch.stack([getattr(agg, "bias%d" % r).grad for r in range(agg._num_links)])
            per["grad.layer%d.%s.W" % (l, k)], _ = _rel(gW, torch.stack(gl["W"]))
            per["grad.layer%d.%s.b" % (l, k)], _ = _rel(gb, torch.stack(gl["b"]))
            per["grad.layer%d.%s.Wo" % (l, k)], _ = _rel(fc.weight.grad, gl["Wo"])
            per["grad.layer%d.%s.bo" % (l, k)], _ = _rel(fc.bias.grad, gl["bo"])
    pj = {"user": net.rating_user_projs[0], "item": net.rating_item_projs[0]}
    for k in ("user", "item"):
        per["grad.proj.%s.W" % k], _ = _rel(pj[k].weight.grad, ref["grads"]["proj"][k][0])
        per["grad.proj.%s.b" % k], _ = _rel(pj[k].bias.grad, ref["grads"]["proj"][k][1])
    worst = max(per, key=lambda n: per[n])
    act_worst = max((n for n in per if not n.startswith("grad.") and n != "loss"), key=lambda n: per[n])
    return {"max_rel_err": per[worst], "worst": worst, "max_rel_err_outputs": per[act_worst], "rows": rows,
            "tensors": len(per), "per_tensor": {n: float("%.3g" % v) for n, v in per.items()},
            "loss_f64": float(ref["loss"]), "activation_derivative": dict(ref["act_stats"], rule=(
                "LeakyReLU' is discontinuous at 0: where |float64 pre-activation| <= %g * max|pre| (inside fp32 rounding) "
                "the sign the network under test took is adopted; float64 decides everywhere else" % AMBIGUOUS))}


def verify_step(net, run_step, graph_arrays, y, scale, name_user="user", name_item="movie"):
    """Run ONE step of the network under test (`run_step()` -> loss, gradients left in .grad) with capture hooks on,
    evaluate the definition in float64 and compare.  graph_arrays = (ind_ptr, item, level, n_item, R, item_degrees)."""
    cap = Capture(net, name_user, name_item)
    try:
        loss = run_step()
    finally:
        cap.close()
    if loss.is_cuda:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()          # hand the step's cached temporaries back before the float64 pass allocates
    g = RawGraph(*graph_arrays)
    ref = evaluate(g, net_params(net, name_user, name_item), y, scale, product=cap)
    out = compare(net, cap, loss.detach(), ref, name_user, name_item)
    out["score_rms"] = float("%.4g" % ref["score_rms"])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# node-partitioned runs (star-gcn_amd/dist.py: 1-D user-block partition, items replicated)
def _rel_rows(got_block, ref_full, lo, hi):
    """error of rows [lo, hi) against the float64 tensor of ALL rows, relative to the WHOLE tensor's scale (the N = 1 measure)"""
    s = float(ref_full.abs().max())
    return float((got_block.double() - ref_full[lo:hi]).abs().max()) / max(s, 1e-30)


def verify_step_partitioned(net, run_step, graph_arrays, y, scale, lo, hi, n_user, assemble_rows, reduce_scalar,
                            name_user="user", name_item="movie", one_at_a_time=None):
    """One rank's view of a node-partitioned step against the float64 evaluation of the definition over the WHOLE graph.

    net: this rank's network (user table = rows [lo, hi) of the global one, everything else replicated); run_step() runs
    one partitioned step INCLUDING the gradient all-reduce of the replicated parameters and returns the local loss.
    graph_arrays / y: the whole graph, as for verify_step.  assemble_rows(block) -> (n_user, width) matrix made of every
    rank's row block; reduce_scalar(t) -> sum over ranks (both are collectives: every rank calls them in the same order).
    one_at_a_time(fn) -> fn(): optional turnstile that lets ONE rank at a time through the float64 evaluation (ranks that share a
    GPU over gloo: eight processes time-slicing one device made the evaluation 40x slower than eight turns).
    Compared on every rank: the loss, item-side layer outputs / projection (replicated: all rows), this rank's user rows of
    every layer output / projection / embedding gradient, every replicated parameter's all-reduced gradient and the item
    embedding gradient.  The activation-derivative rule of the single-GPU check applies unchanged: the side the product
    took is adopted only where the float64 pre-activation lies inside fp32 rounding of zero -- the product's outputs of
    ALL user rows are assembled for that."""
    key = {"user": name_user, "item": name_item}
    other = {"user": name_item, "item": name_user}
    cap = Capture(net, name_user, name_item)
    try:
        loss_local = run_step()
    finally:
        cap.close()
    loss = reduce_scalar(loss_local.detach().view(1))[0]
    if loss_local.is_cuda:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    params = net_params(net, name_user, name_item)
    params["embed"]["user"] = assemble_rows(params["embed"]["user"])
    assert params["embed"]["user"].shape[0] == n_user
    block = {"layer": [c["user"] for c in cap.layer_out], "agg": [c["user"] for c in cap.agg_out]}
    for l in range(len(cap.layer_out)):          # same order on every rank
        cap.layer_out[l]["user"] = assemble_rows(block["layer"][l])
        cap.agg_out[l]["user"] = assemble_rows(block["agg"][l])
    def _evaluate_and_compare():
        g = RawGraph(*graph_arrays)
        assert g.n_user == n_user
        ref = evaluate(g, params, y, scale, product=cap)
        return _compare_partitioned(net, cap, block, loss, ref, lo, hi, key, other)

    return one_at_a_time(_evaluate_and_compare) if one_at_a_time is not None else _evaluate_and_compare()


def _compare_partitioned(net, cap, block, loss, ref, lo, hi, key, other):
    per = {"loss": abs(float(loss) - float(ref["loss"])) / max(abs(float(ref["loss"])), 1e-30)}
    for l, lo_ in enumerate(ref["layer_out"]):
        per["layer%d.out.user[block]" % l] = _rel_rows(block["layer"][l], lo_["user"], lo, hi)
        per["layer%d.out.item" % l], _ = _rel(cap.layer_out[l]["item"], lo_["item"])
    per["proj.user[block]"] = _rel_rows(cap.proj["user"], ref["proj"]["user"], lo, hi)
    per["proj.item"], _ = _rel(cap.proj["item"], ref["proj"]["item"])
    per["grad.embed.user[block]"] = _rel_rows(net.embed_layers[key["user"]].weight.grad, ref["grads"]["embed"]["user"], lo, hi)
    per["grad.embed.item"], _ = _rel(net.embed_layers[key["item"]].weight.grad, ref["grads"]["embed"]["item"])
    for l, layer in enumerate(net.encoders[0]._blocks):
        for k in ("user", "item"):
            agg, fc, gl = layer.aggregators[(key[k], other[k])], layer._out_fcs[key[k]], ref["grads"]["layers"][l][k]
            gW = torch.stack([getattr(agg, "weight%d" % r).grad for r in range(agg._num_links)])
            gb = torch.stack([getattr(agg, "bias%d" % r).grad for r in range(agg._num_links)])
            per["grad.layer%d.%s.W" % (l, k)], _ = _rel(gW, torch.stack(gl["W"]))
            per["grad.layer%d.%s.b" % (l, k)], _ = _rel(gb, torch.stack(gl["b"]))
            per["grad.layer%d.%s.Wo" % (l, k)], _ = _rel(fc.weight.grad, gl["Wo"])
            per["grad.layer%d.%s.bo" % (l, k)], _ = _rel(fc.bias.grad, gl["bo"])
    pj = {"user": net.rating_user_projs[0], "item": net.rating_item_projs[0]}
    for k in ("user", "item"):
        per["grad.proj.%s.W" % k], _ = _rel(pj[k].weight.grad, ref["grads"]["proj"][k][0])
        per["grad.proj.%s.b" % k], _ = _rel(pj[k].bias.grad, ref["grads"]["proj"][k][1])
    worst = max(per, key=lambda n: per[n])
    gworst = max((n for n in per if n.startswith("grad.")), key=lambda n: per[n])
    return {"max_rel_err": per[worst], "worst": worst, "gradient_max_rel_err": per[gworst], "gradient_worst": gworst,
            "tensors": len(per), "per_tensor": {n: float("%.3g" % v) for n, v in per.items()},
            "loss_f64": float(ref["loss"]), "user_rows": [int(lo), int(hi)],
            "activation_derivative": dict(ref["act_stats"], rule=(
                "LeakyReLU' is discontinuous at 0: where |float64 pre-activation| <= %g * max|pre| (inside fp32 rounding) "
                "the sign the partitioned run took is adopted; float64 decides everywhere else" % AMBIGUOUS))}
