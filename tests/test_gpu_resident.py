"""Device-side per-batch edge removal on a resident full-graph plan (star_gcn_amd/resident.py,
sg_mask_edges_hip) vs the reference-style host path: remove_edges_by_id in both directions -> fresh degrees / support
-> gen_plan -> upload (reference graph.py:952-974, layers.py:260-337)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
U, I = "user", "movie"


def dense_levels(mp):
    """(n_dst*R, n_src) float64 matrix of a MultiLinkPlan's forward CSR and of its transpose."""
    R = mp.R
    c_indptr, c_idx, c_w = mp.c_indptr.cpu().numpy(), mp.c_idx.cpu().numpy(), mp.c_w.cpu().numpy()
    t_indptr, t_idx, t_w = mp.t_indptr.cpu().numpy(), mp.t_idx.cpu().numpy(), mp.t_w.cpu().numpy()
    A = np.zeros((mp.n_dst * R, mp.n_src), np.float32)
    rows = np.repeat(np.arange(mp.n_dst * R), np.diff(c_indptr))
    A[rows, c_idx[:rows.size]] = c_w[:rows.size]
    B = np.zeros_like(A)
    seg = np.repeat(np.arange(mp.n_src * R), np.diff(t_indptr))
    B[t_idx[:seg.size] * R + seg % R, seg // R] = t_w[:seg.size]
    return A, B


def make(symm=True, accum="sum", nblocks=2):
    import star_gcn_amd.model as M
    import star_gcn_amd.synthetic as S
    graph, eu, ei, vals = S.make_graph("custom", seed=31, n_user=80, n_item=50, n_edges=1100, n_levels=5)
    torch.manual_seed(5)
    net = M.Net(graph, U, I, embed_units=24, agg_units=(40,), out_units=(30,), nblocks=nblocks, use_dae=True,
                agg_accum=accum, norm_symm=symm).cuda()
    return net, graph, eu, ei, vals


@pytest.mark.parametrize("symm", [True, False])
def test_masked_weights_equal_host_edge_removal_bit_exactly(symm):
    from star_gcn_amd.resident import ResidentPlan
    net, graph, eu, ei, vals = make(symm=symm)
    res = ResidentPlan(net, graph, symm=symm)
    before = [(w.clone()) for w in res._w_arrays]
    rng = np.random.default_rng(0)
    sel = rng.choice(eu.size, 200, replace=False)
    # one user loses ALL its ratings (degree 0 -> support 0), ids repeat, and -1 / out-of-range ids are ignored
    all_of_user = np.nonzero(eu == eu[sel[0]])[0]
    pairs = np.stack([np.concatenate([eu[sel], eu[all_of_user]]), np.concatenate([ei[sel], ei[all_of_user]])])
    ids = graph[U, I].edge_positions(pairs)
    assert np.all(ids >= 0)
    res.mask_edges(np.concatenate([ids, ids[:7], [-1, 10 ** 6]]))
    # host path: new graph, new plan
    g2 = graph.remove_edges_by_id(U, I, pairs)
    full = {k: g2.node_ids_dict[k] for k in g2.meta_graph}
    host = net.make_plan(g2, symm=symm, device="cuda", full_node_ids=full)
    for b in range(2):
        for depth in range(len(host["enc"][b])):
            for key in (U, I):
                for src, hp in host["enc"][b][depth][1][key][2].items():
                    rp = res.plan["enc"][b][depth][1][key][2][src]
                    Ah, Bh = dense_levels(hp)
                    Ar, Br = dense_levels(rp)
                    assert np.array_equal(Ah, Ar) and np.array_equal(Bh, Br), (b, depth, key)
                    assert np.array_equal(Ah, Bh)
                    assert torch.equal(rp.rowsum, hp.rowsum) or torch.allclose(rp.rowsum, hp.rowsum, rtol=1e-6, atol=0)
    # restoring the full graph reproduces the original weights bit for bit
    res.mask_edges(np.zeros(0, np.int32))
    for w0, w in zip(before, res._w_arrays):
        assert torch.equal(w0, w)


@pytest.mark.parametrize("accum,order", [("sum", "auto"), ("stack", "aggregate_first")])
def test_training_step_on_resident_plan_matches_replanned_step(accum, order):
    import star_gcn_amd.model as M
    from star_gcn_amd.resident import ResidentPlan
    net, graph, eu, ei, vals = make(accum=accum)
    for enc in net.encoders:
        for layer in enc._blocks:
            for agg in layer.aggregators.values():
                agg._order = order
    rng = np.random.default_rng(2)
    noise, recon = {}, {}
    for key, n in ((U, 80), (I, 50)):
        perm = rng.permutation(n).astype(np.int32)
        k = int(np.ceil(0.2 * n))
        recon[key] = perm[:k]
        nz = np.arange(n, dtype=np.int32)
        nz[perm[:k // 2]] = -1
        noise[key] = nz
    sel = rng.choice(eu.size, 256, replace=False)
    pairs = np.stack([eu[sel], ei[sel]])
    y = torch.from_numpy(((vals[sel] - vals.mean()) / vals.std()).astype(np.float32)).cuda()

    def grads():
        return {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}

    # reference-style iteration: remove the batch's edges on the host, re-plan, upload
    net.zero_grad(set_to_none=True)
    g2 = graph.remove_edges_by_id(U, I, pairs)
    preds, recons, gt = net(g2, rating_node_pairs=pairs, embed_noise_dict=noise, recon_node_ids_dict=recon, device="cuda")
    loss = M.star_gcn_loss(preds, recons, gt, y, recon_lambda=0.1)
    loss.backward()
    want = grads()
    # resident iteration: weights masked on the device, heads re-indexed
    net.zero_grad(set_to_none=True)
    res = ResidentPlan(net, graph)
    plan = res.set_batch(rating_node_pairs=pairs, embed_noise_dict=noise, recon_node_ids_dict=recon)
    preds2, recons2, gt2 = net.run(plan)
    loss2 = M.star_gcn_loss(preds2, recons2, gt2, y, recon_lambda=0.1)
    loss2.backward()
    got = grads()

    def close(a, b, what, tol=2e-6):
        a, b = a.detach(), b.detach()
        scale = max(float(b.abs().max()), 1e-3)
        assert float((a - b).abs().max()) <= tol * scale, (what, float((a - b).abs().max()), scale)

    for b in range(2):
        close(preds2[b], preds[b], "pred %d" % b)
        for key in (U, I):
            close(recons2[b][key], recons[b][key], "recon %d %s" % (b, key))
            close(gt2[key], gt[key], "gt " + key, tol=0.0 + 1e-12)
    close(loss2, loss, "loss")
    assert set(got) == set(want)
    for n in want:
        close(got[n], want[n], "grad " + n, tol=1e-5)
    # a second batch on the same resident plan (different removal set) still matches a fresh host plan
    sel = rng.choice(eu.size, 100, replace=False)
    pairs = np.stack([eu[sel], ei[sel]])
    with torch.no_grad():
        p_host, _, _ = net(graph.remove_edges_by_id(U, I, pairs), rating_node_pairs=pairs, device="cuda")
        p_res, _, _ = net.run(res.set_batch(rating_node_pairs=pairs, edge_ids=graph[U, I].edge_positions(pairs)))
    close(p_res[1], p_host[1], "second batch")


def test_device_sampler_draws_are_distinct_uniform_and_reproducible():
    from star_gcn_amd import _lib as L
    lib = L.lib()
    n, k = 1_000_003, 200_000
    out = torch.empty(k, dtype=torch.int32, device="cuda")
    L.check(lib.sg_sample_distinct_hip(L.ptr(out), n, k, 7, 0, None), "sample")
    a = out.cpu().numpy()
    assert a.min() >= 0 and a.max() < n and np.unique(a).size == k              # distinct, in range
    L.check(lib.sg_sample_distinct_hip(L.ptr(out), n, k, 7, 0, None), "sample")
    assert np.array_equal(a, out.cpu().numpy())                                 # (seed, counter) -> same draw
    L.check(lib.sg_sample_distinct_hip(L.ptr(out), n, k, 7, 1, None), "sample")
    b = out.cpu().numpy()
    assert np.intersect1d(a, b).size < 0.25 * k                                 # next counter: an independent draw (E = 0.2 k)
    # uniformity: 50 equal bins of [0, n), chi-square with 49 dof (99.9 % quantile = 85.4)
    hist = np.bincount(a // (n // 50 + 1), minlength=50)[:50].astype(np.float64)
    exp = k * np.diff(np.minimum(np.arange(51) * (n // 50 + 1), n)) / n
    assert ((hist - exp) ** 2 / exp).sum() < 86.0
    # a full permutation: k = n
    m = 12345
    perm = torch.empty(m, dtype=torch.int32, device="cuda")
    L.check(lib.sg_sample_distinct_hip(L.ptr(perm), m, m, 1, 5, None), "sample")
    assert np.array_equal(np.sort(perm.cpu().numpy()), np.arange(m))
    # recon mask: k distinct nodes; noise = identity except (with probability p_zero) -1 on the picked nodes
    nz, rc = torch.empty(m, dtype=torch.int32, device="cuda"), torch.empty(1235, dtype=torch.int32, device="cuda")
    L.check(lib.sg_recon_mask_hip(L.ptr(nz), L.ptr(rc), m, 1235, 0.3, 3, 9, None), "recon")
    nzh, rch = nz.cpu().numpy(), rc.cpu().numpy()
    assert np.unique(rch).size == 1235
    rest = np.setdiff1d(np.arange(m), rch)
    assert np.array_equal(nzh[rest], rest) and set(np.unique(nzh[rch] - rch * (nzh[rch] >= 0))) <= {0, -1}
    frac = float((nzh[rch] == -1).mean())
    assert 0.22 < frac < 0.38


def test_device_recon_mask_inductive_rule_unseen_nodes_are_minus_one():
    """Reference iterators.py:332-346: the noise array starts as -1 for EVERY node of the graph ("nodes unseen in the training
    graph are masked as -1.0"), the nodes of the training graph (`_recon_train_candidates`) get their own id, and
    ceil(P_mask |candidates|) of THEM are drawn for reconstruction, -1 with probability p_zero.  The device form
    (sg_recon_mask_cand_dev_hip) is held to exactly those set-level facts -- the reference's Mersenne-Twister stream is
    deliberately not reproduced (include/stargcn.h section 12) -- and to the degenerate probabilities, where the noise
    array is a deterministic function of the drawn set and must equal the reference expression evaluated on it."""
    from star_gcn_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(5)
    n = 50_000
    cand = np.sort(rng.choice(n, 31_007, replace=False)).astype(np.int32)       # nodes seen in the training graph
    unseen = np.setdiff1d(np.arange(n), cand)
    k = int(np.ceil(0.1 * cand.size))
    cd = torch.from_numpy(cand).cuda()
    for p_zero in (0.0, 1.0, 0.35):
        nz = torch.full((n,), 12345, dtype=torch.int32, device="cuda")
        rc = torch.empty(k, dtype=torch.int32, device="cuda")
        L.check(lib.sg_recon_mask_cand_dev_hip(L.ptr(nz), L.ptr(rc), n, L.ptr(cd), cand.size, k, p_zero, 11, 4, None, None), "recon")
        nzh, rch = nz.cpu().numpy(), rc.cpu().numpy()
        assert np.unique(rch).size == k and np.isin(rch, cand).all()            # k distinct CANDIDATES
        assert (nzh[unseen] == -1).all()                                        # the inductive rule
        remain = np.setdiff1d(cand, rch)
        assert np.array_equal(nzh[remain], remain)                              # embed_noise[remain] = remain
        if p_zero in (0.0, 1.0):
            # the reference expression on the drawn set: mask_type = multinomial(1, [p_zero, p_self]) is one-hot and certain
            mask_type = np.tile(np.array([[1, 0]] if p_zero == 1.0 else [[0, 1]]), (k, 1))
            expect = -np.ones(n, dtype=np.int32)
            expect[remain] = remain
            expect[rch] = (mask_type * np.stack([-np.ones(rch.shape), rch], axis=1)).sum(axis=1).astype(np.int32)
            assert np.array_equal(nzh, expect)
        else:
            on = nzh[rch]
            assert set(np.unique(on - rch * (on >= 0))) <= {0, -1}
            assert 0.30 < float((on == -1).mean()) < 0.40
    # the sampler class: `recon_candidates` switches the inductive form on
    import types
    from star_gcn_amd.device_sampler import DeviceBatchSampler
    csr = types.SimpleNamespace(values=np.ones(10, dtype=np.float32))
    res = types.SimpleNamespace(nnz=10, device=torch.device("cuda"), csr=csr, U="user", I="movie", n_user=n, n_item=1000,
                                _edge_row=torch.zeros(10, dtype=torch.int32, device="cuda"),
                                _edge_col=torch.zeros(10, dtype=torch.int32, device="cuda"))
    smp = DeviceBatchSampler(res, 4, embed_P_mask=0.1, embed_p_zero=0.0, seed=2, recon_candidates={"user": cand})
    b = smp.next_batch()
    nu = b["noise"]["user"].cpu().numpy()
    assert (nu[unseen] == -1).all() and np.isin(b["recon"]["user"].cpu().numpy(), cand).all()
    assert b["recon"]["user"].numel() == k
    assert np.array_equal(b["noise"]["movie"].cpu().numpy(), np.arange(1000))   # no candidate list: transductive, as before
    # the list is validated at construction: the device kernel assumes distinct ids inside the node range
    from star_gcn_amd._lib import StarGCNError
    for bad in (np.concatenate([cand, cand[:1]]), np.array([0, n], dtype=np.int64), np.array([-1, 3], dtype=np.int64)):
        with pytest.raises(StarGCNError):
            DeviceBatchSampler(res, 4, recon_candidates={"user": bad})
    with pytest.raises(StarGCNError):
        DeviceBatchSampler(res, 4, recon_candidates={"actor": cand})


def test_device_batch_equals_host_batch():
    """DeviceBatchSampler + ResidentPlan.set_batch_device (samplers, pair plan, take plans, edge masking all on the
    device) give bit-identical network outputs to the host-planned batch (set_batch) of the SAME edges / noise / nodes."""
    import star_gcn_amd.model as M
    from star_gcn_amd.device_sampler import DeviceBatchSampler
    from star_gcn_amd.resident import ResidentPlan
    net, graph, eu, ei, vals = make()
    res = ResidentPlan(net, graph)
    smp = DeviceBatchSampler(res, 300, embed_P_mask=0.2, embed_p_zero=0.5, seed=11)
    for _ in range(2):
        batch = smp.next_batch()
        ids = batch["edge_ids"].cpu().numpy()
        assert np.all(np.diff(ids) > 0) and ids.size == 300
        np.testing.assert_array_equal(batch["users"].cpu().numpy(), eu[ids])
        np.testing.assert_array_equal(batch["items"].cpu().numpy(), ei[ids])
        np.testing.assert_array_equal(batch["ratings"].cpu().numpy(), vals[ids])
        y = batch["ratings"]
        outs = []
        for mode in ("device", "host"):
            if mode == "device":
                plan = res.set_batch_device(batch)
            else:
                plan = res.set_batch(rating_node_pairs=np.stack([eu[ids], ei[ids]]), edge_ids=ids,
                                     embed_noise_dict={k: v.cpu().numpy() for k, v in batch["noise"].items()},
                                     recon_node_ids_dict={k: v.cpu().numpy() for k, v in batch["recon"].items()})
            net.zero_grad(set_to_none=True)
            preds, recons, gt = net.run(plan)
            loss = M.star_gcn_loss(preds, recons, gt, (y - y.mean()) / y.std(), recon_lambda=0.1)
            loss.backward()
            outs.append((preds, recons, gt, loss.detach().clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
        d, h = outs
        for b in range(2):
            assert torch.equal(d[0][b], h[0][b])
            for key in (U, I):
                assert torch.equal(d[1][b][key], h[1][b][key])
        for key in (U, I):
            assert torch.equal(d[2][key], h[2][key])
        assert torch.equal(d[3], h[3])
        for k in d[4]:
            assert torch.equal(d[4][k], h[4][k]), k


def test_whole_iteration_replays_as_one_hipgraph():
    """examples/train_star_gcn.py --resident --device-sampler --graph: sampling, edge masking, batch plans, forward,
    backward, clipping and Adam captured once and replayed; every replay draws a new batch (device-resident counter) and
    the model learns (validation RMSE falls)."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "examples", "train_star_gcn.py"), "--shape", "ml-100k",
                          "--iters", "120", "--eval-every", "60", "--resident", "--device-sampler", "--graph"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    rmse = [float(x) for x in re.findall(r"valid RMSE ([0-9.]+)", out.stdout)]
    assert len(rmse) >= 3 and rmse[-1] < rmse[0] - 0.03, out.stdout
    assert "one hipGraph replay per iteration" in out.stdout


def test_device_counter_advances_draws():
    from star_gcn_amd.device_sampler import DeviceBatchSampler
    from star_gcn_amd.resident import ResidentPlan
    net, graph, eu, ei, vals = make()
    res = ResidentPlan(net, graph)
    smp = DeviceBatchSampler(res, 100, seed=3)
    a = smp.next_batch(advance_on_device=True)["edge_ids"].cpu().numpy()
    b = smp.next_batch(advance_on_device=True)["edge_ids"].cpu().numpy()
    assert int(smp.dev_counter.item()) == 6 and not np.array_equal(a, b)
    # the device-counter draws are the host-counter draws of the same index
    smp2 = DeviceBatchSampler(res, 100, seed=3)
    np.testing.assert_array_equal(smp2.next_batch()["edge_ids"].cpu().numpy(), a)
    np.testing.assert_array_equal(smp2.next_batch()["edge_ids"].cpu().numpy(), b)
