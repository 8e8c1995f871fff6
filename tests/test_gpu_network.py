"""GPU parity of the whole hot path as STAR-GCN uses it (reference experiments/STAR-GCN.py:311-461 + losses
:610-628): 2 blocks, decoder, masked embedding input, rating mini-batch, units 250 / 75 of the shipped yamls --
against the dense float64 whole-network oracle (oracle/model.py:dense_star_gcn), which shares no planning /
unique / re-indexing code with the product.  Tolerance 1e-5 x the tensor's scale (north star) for every output AND
every parameter gradient (measured worst case: 8e-7, tools/measure_network_tol.py)."""
import numpy as np
import pytest
import torch

from oracle import model as OM

pytestmark = pytest.mark.gpu
U, I = "user", "movie"


def rel_close(got, ref, tol, what):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    scale = max(float(ref.abs().max()), 1e-3)
    err = float((got - ref).abs().max())
    assert err <= tol * scale, "%s: err %.3e > %.1e * %.3e" % (what, err, tol, scale)


def extract(net, dtype=torch.float64):
    """Parameters of star_gcn_amd.model.Net in the structure dense_star_gcn expects (requires_grad leaves)."""
    leaf = {}

    def cv(p):
        if id(p) not in leaf:
            leaf[id(p)] = p.detach().to("cpu", dtype).requires_grad_(True)
        return leaf[id(p)]

    tables = {k: cv(net.embed_layers[k].weight) for k in (U, I)}
    blocks, maps, projs = [], [], []
    for b, enc in enumerate(net.encoders):
        layers = []
        for layer in enc._blocks:
            d = {}
            for dst, src in ((U, I), (I, U)):
                agg = layer.aggregators[(dst, src)]
                R = agg._num_links
                fc = layer._out_fcs[dst]
                d[dst] = dict(src=src, W=[cv(getattr(agg, "weight%d" % r)) for r in range(R)],
                              b=[cv(getattr(agg, "bias%d" % r)) for r in range(R)], ow=cv(fc.weight), ob=cv(fc.bias))
            layers.append(d)
        blocks.append(layers)
        maps.append({k: (cv(net.embed_maps[b][k][0].weight), cv(net.embed_maps[b][k][0].bias),
                         cv(net.embed_maps[b][k][1].weight), cv(net.embed_maps[b][k][1].bias)) for k in (U, I)}
                    if net._use_dae else None)
        projs.append({U: (cv(net.rating_user_projs[b].weight), cv(net.rating_user_projs[b].bias)),
                      I: (cv(net.rating_item_projs[b].weight), cv(net.rating_item_projs[b].bias))})
    return tables, blocks, maps, projs, leaf


@pytest.mark.parametrize("accum,agg_units,order", [("sum", 250, "auto"), ("stack", 250, "auto"),
                                                  ("sum", 60, "aggregate_first"), ("stack", 60, "transform_first"),
                                                  # round 6: the fused aggregate -> contract kernel at the shipped widths
                                                  # (embed 32 -> AGG 250; 'stack' = 5 x 50 units), whole network, all gradients
                                                  ("sum", 250, "fused"), ("stack", 250, "fused")])
def test_two_block_star_gcn_matches_dense_oracle(accum, agg_units, order):
    import star_gcn_amd.model as M
    import star_gcn_amd.synthetic as S
    dev = torch.device("cuda", 0)
    graph, eu, ei, vals = S.make_graph("custom", seed=21, n_user=70, n_item=45, n_edges=800, n_levels=5)
    rng = np.random.default_rng(4)
    torch.manual_seed(2)
    net = M.Net(graph, U, I, embed_units=32, agg_units=(agg_units,), out_units=(75,), nblocks=2, use_dae=True,
                agg_accum=accum, agg_order=order).to(dev)
    # masked-embedding sampler state as reference iterators.py:309-370 produces it: -1 = zero-mask, i = keep
    noise, recon = {}, {}
    for key, n in ((U, 70), (I, 45)):
        perm = rng.permutation(n).astype(np.int32)
        k = int(np.ceil(0.2 * n))
        recon[key] = perm[:k]
        nz = np.arange(n, dtype=np.int32)
        nz[perm[:k // 2]] = -1                                   # half of the recon nodes zero-masked, half keep self
        noise[key] = nz
    sel = rng.choice(eu.size, 300, replace=False)                # unsorted rating mini-batch
    pairs = np.stack([eu[sel], ei[sel]])
    y = torch.from_numpy(((vals[sel] - vals.mean()) / vals.std()).astype(np.float32))

    preds, recons, gt = net(graph, rating_node_pairs=pairs, embed_noise_dict=noise, recon_node_ids_dict=recon,
                            device=dev)
    loss = M.star_gcn_loss(preds, recons, gt, y.to(dev), recon_lambda=0.1)
    loss.backward()

    tables, blocks, maps, projs, leaf = extract(net)
    levels = graph[U, I].multi_link
    adj = {(U, I): OM.dense_level_adjacency(eu, ei, vals, levels, 70, 45),
           (I, U): OM.dense_level_adjacency(ei, eu, vals, levels, 45, 70)}
    opreds, orecons, ogt = OM.dense_star_gcn(tables, noise, adj, blocks, maps, projs, (U, I, pairs[0], pairs[1]), recon,
                                             accum=accum)
    oloss = 0.0
    for pr in opreds:
        oloss = oloss + (0.5 * (pr.view(-1) - y.double()) ** 2).mean()
    for blk in orecons:
        for key, pred in blk.items():
            oloss = oloss + 0.1 * ((ogt[key] - pred) ** 2).sum(dim=1).mean()
    oloss.backward()

    assert len(preds) == 2 and len(recons) == 2
    for b in range(2):
        rel_close(preds[b], opreds[b], 1e-5, "pred_ratings[%d]" % b)
        for key in (U, I):
            rel_close(recons[b][key], orecons[b][key], 1e-5, "pred_embeddings[%d][%s]" % (b, key))
    for key in (U, I):
        rel_close(gt[key], ogt[key], 0.0 + 1e-12, "gt[%s]" % key)
    rel_close(loss, oloss, 1e-5, "loss")
    for name, p in net.named_parameters():
        ref = leaf[id(p)].grad
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        rel_close(p.grad, ref, 1e-5, "grad " + name)


@pytest.mark.parametrize("recon_fea", [False, True])
def test_feature_projection_matches_dense_oracle(recon_fea):
    """MODEL.USE_FEA_PROJ / RECON_FEA (reference STAR-GCN.py:182-192, 364-370, 405-413, 455-459): mapped node features
    concatenated to every block's input, and to the reconstruction target when RECON_FEA."""
    import star_gcn_amd.model as M
    import star_gcn_amd.synthetic as S
    dev = torch.device("cuda", 0)
    graph, eu, ei, vals = S.make_graph("custom", seed=5, n_user=60, n_item=40, n_edges=700, n_levels=5)
    rng = np.random.default_rng(9)
    feats = {U: rng.normal(size=(60, 23)).astype(np.float32), I: rng.normal(size=(40, 19)).astype(np.float32)}
    torch.manual_seed(3)
    net = M.Net(graph, U, I, embed_units=32, agg_units=(60,), out_units=(75,), nblocks=2, use_dae=True, agg_accum="stack",
                use_fea_proj=True, recon_fea=recon_fea, fea_mid_map=16, fea_units=12, features=feats).to(dev)
    noise, recon = {}, {}
    for key, n in ((U, 60), (I, 40)):
        perm = rng.permutation(n).astype(np.int32)
        recon[key] = perm[:int(np.ceil(0.25 * n))]
        nz = np.arange(n, dtype=np.int32)
        nz[perm[:5]] = -1
        noise[key] = nz
    sel = rng.choice(eu.size, 250, replace=False)
    pairs = np.stack([eu[sel], ei[sel]])
    y = torch.from_numpy(((vals[sel] - vals.mean()) / vals.std()).astype(np.float32))
    preds, recons, gt = net(graph, rating_node_pairs=pairs, embed_noise_dict=noise, recon_node_ids_dict=recon, device=dev)
    loss = M.star_gcn_loss(preds, recons, gt, y.to(dev), recon_lambda=0.1)
    loss.backward()

    tables, blocks, maps, projs, leaf = extract(net)

    def cv(p):
        if id(p) not in leaf:
            leaf[id(p)] = p.detach().to("cpu", torch.float64).requires_grad_(True)
        return leaf[id(p)]
    fmaps = {k: (cv(net.fea_mappings[k][0].weight), cv(net.fea_mappings[k][0].bias), cv(net.fea_mappings[k][1].weight),
                 cv(net.fea_mappings[k][1].bias)) for k in (U, I)}
    levels = graph[U, I].multi_link
    adj = {(U, I): OM.dense_level_adjacency(eu, ei, vals, levels, 60, 40),
           (I, U): OM.dense_level_adjacency(ei, eu, vals, levels, 40, 60)}
    opreds, orecons, ogt = OM.dense_star_gcn(tables, noise, adj, blocks, maps, projs, (U, I, pairs[0], pairs[1]), recon,
                                             accum="stack", features={k: torch.from_numpy(v).double() for k, v in feats.items()},
                                             fea_maps=fmaps, recon_fea=recon_fea)
    oloss = 0.0
    for pr in opreds:
        oloss = oloss + (0.5 * (pr.view(-1) - y.double()) ** 2).mean()
    for blk in orecons:
        for key, pred in blk.items():
            oloss = oloss + 0.1 * ((ogt[key] - pred) ** 2).sum(dim=1).mean()
    oloss.backward()
    width = 32 + (12 if recon_fea else 0)
    for b in range(2):
        rel_close(preds[b], opreds[b], 1e-5, "pred_ratings[%d]" % b)
        for key in (U, I):
            assert recons[b][key].shape[1] == width
            rel_close(recons[b][key], orecons[b][key], 1e-5, "pred_embeddings[%d][%s]" % (b, key))
    for key in (U, I):
        rel_close(gt[key], ogt[key], 1e-6, "gt[%s]" % key)
    rel_close(loss, oloss, 1e-5, "loss")
    for name, p in net.named_parameters():
        ref = leaf[id(p)].grad
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        rel_close(p.grad, ref, 1e-5, "grad " + name)


def test_layer_api_with_reference_style_lists():
    """The aggregator accepts the reference's per-level lists (incl. empty_as_zero padding) and GCNAggregator works."""
    from star_gcn_amd.mxgraph.layers import GCNAggregator, MultiLinkGCNAggregator
    from tests.test_abi_and_host import make_multilink
    rng = np.random.default_rng(8)
    n_dst, n_src, nnz, R, D, Uo = 40, 33, 500, 4, 24, 20
    eps, ips, sps = make_multilink(rng, n_dst, n_src, nnz, R)
    x = torch.randn(n_src, D)
    for sharing in (False, True):
        agg = MultiLinkGCNAggregator(units=Uo, num_links=R, act="leaky", ordinal_sharing=sharing, accum="stack").cuda()
        d_eps = [torch.from_numpy(e).cuda() for e in eps]
        d_ips = [torch.from_numpy(e).cuda() for e in ips]
        d_sps = [torch.from_numpy(e).cuda() for e in sps]
        out = agg(x.cuda(), d_eps, d_ips, d_sps)
        ws = [getattr(agg, "weight%d" % r).detach().double().cpu() for r in range(R)]
        bs = [getattr(agg, "bias%d" % r).detach().double().cpu() for r in range(R)]
        ref = OM.multilink_aggregator(x.double(), ws, bs, eps, ips, sps, accum="stack", act="leaky",
                                      ordinal_sharing=sharing)
        rel_close(out, ref, 1e-5, "stack sharing=%s" % sharing)
    g = GCNAggregator(units=Uo, act="tanh").cuda()
    ep = np.concatenate([e[:ip[-1]] for e, ip in zip(eps, ips)])   # single-link view of level 0 only is enough
    out = g(x.cuda(), torch.from_numpy(eps[0]).cuda(), torch.from_numpy(ips[0]).cuda(), torch.from_numpy(sps[0]).cuda())
    ref = OM.multilink_aggregator(x.double(), [g._agg.weight0.detach().double().cpu()],
                                  [g._agg.bias0.detach().double().cpu()], eps[:1], ips[:1], sps[:1], act="tanh")
    rel_close(out, ref, 1e-5, "GCNAggregator")


@pytest.mark.parametrize("shape,embed,batch,order", [("ml-100k", 64, 10000, "auto"), ("ml-1m", 128, 100000, "auto"),
                                                     ("ml-100k", 64, 10000, "fused")])
def test_two_block_star_gcn_at_baseline_config_sizes(shape, embed, batch, order):
    """BASELINE configs 2 and 3 (MovieLens-100k dim 64 / MovieLens-1M dim 128, 5 rating levels): the real 2-block
    network with the shipped yaml widths (AGG 250, OUT 75, mask 0.1, recon lambda 0.1, symmetric support, rating
    mini-batch) on the full-size synthetic graph, vs the float64 oracle with sparse float64 adjacency."""
    import star_gcn_amd.model as M
    import star_gcn_amd.synthetic as S
    dev = torch.device("cuda", 0)
    graph, eu, ei, vals = S.make_graph(shape)
    nu, ni = graph[U, I].shape
    rng = np.random.default_rng(1)
    torch.manual_seed(3)
    net = M.Net(graph, U, I, embed_units=embed, agg_units=(250,), out_units=(75,), nblocks=2, use_dae=True,
                agg_accum="sum", agg_order=order).to(dev)
    noise, recon = {}, {}
    for key, n in ((U, nu), (I, ni)):
        perm = rng.permutation(n).astype(np.int32)
        k = int(np.ceil(0.1 * n))
        recon[key] = perm[:k]
        noise[key] = np.arange(n, dtype=np.int32)          # P_ZERO = 0: masked nodes keep their own embedding
    sel = rng.choice(eu.size, batch, replace=False)
    pairs = np.stack([eu[sel], ei[sel]])
    y = torch.from_numpy(((vals[sel] - vals.mean()) / vals.std()).astype(np.float32))
    g = graph.remove_edges_by_id(U, I, pairs)               # the batch's own ratings are not aggregated over
    m = g[U, I]
    preds, recons, gt = net(g, rating_node_pairs=pairs, embed_noise_dict=noise, recon_node_ids_dict=recon, device=dev)
    loss = M.star_gcn_loss(preds, recons, gt, y.to(dev), recon_lambda=0.1)
    loss.backward()

    tables, blocks, maps, projs, leaf = extract(net)
    ru, ci, rv = m.edge_row_indices, m.end_points, m.values
    adj = {(U, I): OM.dense_level_adjacency(ru, ci, rv, m.multi_link, nu, ni, sparse=True),
           (I, U): OM.dense_level_adjacency(ci, ru, rv, m.multi_link, ni, nu, sparse=True)}
    opreds, orecons, ogt = OM.dense_star_gcn(tables, noise, adj, blocks, maps, projs, (U, I, pairs[0], pairs[1]), recon)
    oloss = 0.0
    for pr in opreds:
        oloss = oloss + (0.5 * (pr.view(-1) - y.double()) ** 2).mean()
    for blk in orecons:
        for key, pred in blk.items():
            oloss = oloss + 0.1 * ((ogt[key] - pred) ** 2).sum(dim=1).mean()
    oloss.backward()
    for b in range(2):
        rel_close(preds[b], opreds[b], 1e-5, "pred_ratings[%d]" % b)
        for key in (U, I):
            rel_close(recons[b][key], orecons[b][key], 1e-5, "pred_embeddings[%d][%s]" % (b, key))
    rel_close(loss, oloss, 1e-5, "loss")
    for name, p in net.named_parameters():
        ref = leaf[id(p)].grad
        if ref is not None:
            rel_close(p.grad, ref, 1e-5, "grad " + name)

def test_one_block_star_gcn_ml100k_against_cpu_seg_ops_reference():
    """BASELINE config 1 (MovieLens-100k transductive, 1-block STAR-GCN, CPU seg_ops reference): the HIP network vs the
    SAME network evaluated the way the reference runs on mx.cpu() -- fp32, per rating level one FullyConnected then one
    `seg_weighted_pool` through the C restatement of the reference CPU kernel (seg_op.cc:180-207, oracle/seg_oracle.c),
    concat / add_n, LeakyReLU, Dense (aggregators.py:141-160, layers.py:169-184) -- with the shipped yaml widths
    (embed 32, AGG 250, OUT 75, MID_MAP 64, mask 0.1; transductive_ml_100k.yml).  Forward outputs: rating predictions
    and reconstructed embeddings."""
    import star_gcn_amd.model as M
    import star_gcn_amd.synthetic as S
    from oracle import seg as O
    dev = torch.device("cuda", 0)
    graph, eu, ei, vals = S.make_graph("ml-100k")
    nu, ni = graph[U, I].shape
    rng = np.random.default_rng(11)
    torch.manual_seed(5)
    net = M.Net(graph, U, I, embed_units=32, agg_units=(250,), out_units=(75,), nblocks=1, use_dae=True,
                agg_accum="sum").to(dev)
    noise, recon = {}, {}
    for key, n in ((U, nu), (I, ni)):
        perm = rng.permutation(n).astype(np.int32)
        recon[key] = perm[:int(np.ceil(0.1 * n))]
        noise[key] = np.arange(n, dtype=np.int32)
    sel = rng.choice(eu.size, 10000, replace=False)
    pairs = np.stack([eu[sel], ei[sel]])
    g = graph.remove_edges_by_id(U, I, pairs)
    with torch.no_grad():
        preds, recons, gt = net(g, rating_node_pairs=pairs, embed_noise_dict=noise, recon_node_ids_dict=recon, device=dev)

    tables, blocks, maps, projs, _leaf = extract(net, dtype=torch.float32)
    x = {k: t.detach().numpy() for k, t in tables.items()}
    lk = lambda a: np.where(a > 0, a, np.float32(0.1) * a).astype(np.float32)
    nxt = {}
    for dst, src in ((U, I), (I, U)):
        p = blocks[0][0][dst]
        eps, _v, ips, sps = g[dst, src].sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
        acc = None
        for r in range(len(eps)):
            h = (x[src] @ p["W"][r].detach().numpy().T + p["b"][r].detach().numpy()).astype(np.float32)
            ep, sp = G_empty(eps[r], np.int32), G_empty(sps[r], np.float32)
            o = O.seg_weighted_pool(h[None], sp[None], ep, np.ascontiguousarray(ips[r], np.int32))[0]
            acc = o if acc is None else acc + o
        nxt[dst] = lk(lk(acc) @ p["ow"].detach().numpy().T + p["ob"].detach().numpy())
    pu = nxt[U] @ projs[0][U][0].detach().numpy().T + projs[0][U][1].detach().numpy()
    pi = nxt[I] @ projs[0][I][0].detach().numpy().T + projs[0][I][1].detach().numpy()
    ref_pred = (pu[pairs[0]] * pi[pairs[1]]).sum(axis=1)
    rel_close(preds[0].view(-1), torch.from_numpy(ref_pred), 1e-5, "pred_ratings (config 1)")
    for key in (U, I):
        w0, b0, w1, b1 = (t.detach().numpy() for t in maps[0][key])
        ref_rec = lk(nxt[key][recon[key]] @ w0.T + b0) @ w1.T + b1
        rel_close(recons[0][key], torch.from_numpy(ref_rec), 1e-5, "pred_embeddings[%s] (config 1)" % key)
        rel_close(gt[key], torch.from_numpy(x[key][recon[key]]), 0.0, "gt_embeddings[%s]" % key)


def G_empty(a, dtype):
    """reference graph.py:221-222 empty_as_zero: an empty per-level array is passed as one zero element"""
    a = np.ascontiguousarray(a, dtype=dtype)
    return a if a.size else np.zeros(1, dtype=dtype)
