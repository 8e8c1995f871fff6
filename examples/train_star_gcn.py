#!/usr/bin/env python3
"""Minimal STAR-GCN training driver on the MI355X-native hot path -- the loop of reference
experiments/STAR-GCN.py:583-632 reduced to what exercises the path: rating + masked-reconstruction mini-batches,
per-batch removal of the batch's rating edges from the aggregation graph (both directions, reference
graph.py:952-974), 2-block network with decoder, the two losses, Adam + global-norm clipping, RMSE on held-out
ratings.  Data: a MovieLens-shaped synthetic graph (no dataset files in this environment), or an extracted MovieLens
directory through star_gcn_amd.datasets.LoadData (reference mxgraph/datasets.py) with the reference's splits.

  python examples/train_star_gcn.py --shape ml-100k --iters 200 [--resident]
  python examples/train_star_gcn.py --data-root /data --dataset ml-1m --iters 2000 --resident --device-sampler

--resident keeps the plan of the whole training graph in HBM and removes each batch's edges ON THE DEVICE
(star_gcn_amd/resident.py, sg_mask_edges_hip) instead of rebuilding CSRs + plan on the host every iteration.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import star_gcn_amd.model as M  # noqa: E402
import star_gcn_amd.synthetic as S  # noqa: E402
from star_gcn_amd.mxgraph.iterators import DataIterator  # noqa: E402

U, I = "user", "movie"


def evaluate(net, data_iter, graph, segment, mean, std, dev, lo, hi):
    se, n = 0.0, 0
    with torch.no_grad():
        for pairs, ratings in data_iter.rating_sampler(100000, segment=segment):
            preds, _, _ = net(graph, rating_node_pairs=pairs, embed_noise_dict=data_iter.evaluate_embed_noise_dict,
                              recon_node_ids_dict=None, device=dev)
            p = torch.clamp(preds[-1].view(-1) * std + mean, lo, hi).cpu().numpy()    # last block, de-standardised
            se += float(((p - ratings) ** 2).sum())
            n += ratings.size
    return (se / max(n, 1)) ** 0.5


def run_graph(args, net, it, resident, dsampler, mean, std, lo, hi, dev):
    """The iteration has no host work left, so it is captured once and replayed: one hipGraphLaunch per training step.
    The batch changes between replays because the sampler's draw counter lives in device memory."""
    state = dict()

    def iteration(advance):
        dbatch = dsampler.next_batch(advance_on_device=advance)
        y = (dbatch["ratings"] - mean) / std
        preds, recons, gt = net.run(resident.set_batch_device(dbatch), rating_targets=y, rating_scale=1.0 / y.numel())
        loss = M.star_gcn_loss(preds, recons, gt, y, recon_lambda=0.1)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0, foreach=True)
        state["opt"].step()
        state["opt"].zero_grad(set_to_none=True)
        return loss.detach()

    # eager warm-up on a side stream (lazy parameter shapes, workspaces, allocator pools), as graph capture requires
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dbatch = dsampler.next_batch(advance_on_device=True)
        net.run(resident.set_batch_device(dbatch))                       # materialises the parameters
        state["opt"] = torch.optim.Adam(net.parameters(), lr=args.lr, capturable=True)
        for _ in range(3):
            iteration(True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        state["loss"] = iteration(True)
    torch.cuda.synchronize()
    t0 = time.time()
    t_train, n_timed = 0.0, 0
    for step in range(1, args.iters + 1):
        t_it = time.perf_counter()
        graph.replay()
        if step % args.eval_every == 0 or step == 1:
            torch.cuda.synchronize()
            net.eval()
            rmse = evaluate(net, it, it.val_graph, "valid", mean, std, dev, lo, hi)
            net.train()
            print("iter %4d  loss %.4f  valid RMSE %.4f  (%.1f s)" % (step, float(state["loss"]), rmse, time.time() - t0))
        else:
            torch.cuda.synchronize()
            t_train += time.perf_counter() - t_it
            n_timed += 1
    print("training iterations: %.2f ms/iter (one hipGraph replay per iteration: resident plan + device samplers, batch %d)" %
          (1e3 * t_train / max(n_timed, 1), args.batch))
    net.eval()
    print("test RMSE %.4f" % evaluate(net, it, it.test_graph, "test", mean, std, dev, lo, hi))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ml-100k")
    ap.add_argument("--data-root", default=None, help="directory holding the extracted ml-100k / ml-1m / ml-10M100K folder")
    ap.add_argument("--dataset", default="ml-100k", choices=["ml-100k", "ml-1m", "ml-10m"])
    ap.add_argument("--inductive", default=None, choices=["item", "user"],
                    help="with --data-root: the reference's inductive setting (datasets.py:153-171, iterators.py:171-176): "
                         "10 %% of the items / users are held out as test nodes, 10 %% of the rest as validation nodes; the "
                         "network trains on the graph of the training nodes and sees held-out nodes with a zero embedding")
    ap.add_argument("--features", action="store_true",
                    help="with --data-root: feed the data set's node features through the feature projection "
                         "(reference MODEL.USE_FEA_PROJ)")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--batch", type=int, default=10000)
    ap.add_argument("--embed", type=int, default=32)
    ap.add_argument("--lr", type=float, default=0.002)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--resident", action="store_true")
    ap.add_argument("--device-sampler", action="store_true",
                    help="with --resident: draw the rating batch / reconstruction nodes and build the batch plans on the "
                         "device as well (star_gcn_amd/device_sampler.py): no per-iteration host work besides launches")
    ap.add_argument("--graph", action="store_true",
                    help="with --resident --device-sampler: capture the whole host-free iteration (sampling, edge masking, "
                         "batch plans, forward, backward, clipping, Adam) in ONE hipGraph and replay it")
    ap.add_argument("--eval-every", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    rng = np.random.default_rng(args.seed)
    ind_kwargs = dict()
    if args.data_root is not None:
        from star_gcn_amd.datasets import LoadData
        data = LoadData(args.dataset, args.data_root, seed=args.seed, use_inductive=args.inductive is not None,
                        inductive_key=args.inductive or "item")
        print(data)
        graph, test_pairs, valid_pairs = data.graph, data.test_data[0], data.valid_data[0]
        if args.inductive is not None:
            assert not args.resident, "the resident plan needs contiguous node ids; the inductive graphs are node subsets"
            ind_kwargs = dict(is_inductive=True, inductive_key=I if args.inductive == "item" else U,
                              inductive_valid_ids=data.inductive_valid_ids, inductive_train_ids=data.inductive_train_ids)
    else:
        graph, eu, ei, vals = S.make_graph(args.shape, signal=True)
        n = eu.size
        perm = rng.permutation(n)
        n_test, n_valid = int(0.2 * n), int(0.08 * n)
        test_pairs = np.stack([eu[perm[:n_test]], ei[perm[:n_test]]])
        valid_pairs = np.stack([eu[perm[n_test:n_test + n_valid]], ei[perm[n_test:n_test + n_valid]]])
    # inductive: masked nodes are zeroed (P_ZERO = 1, the reference's inductive yamls), so that a held-out node -- whose
    # embedding was never trained -- looks to the network like a masked training node
    p_zero = 1.0 if ind_kwargs else 0.0
    it = DataIterator(graph, U, I, test_pairs, valid_pairs, embed_P_mask=0.1, embed_p_zero=p_zero, embed_p_self=1.0 - p_zero,
                      seed=args.seed, **ind_kwargs)
    train_vals = it.train_graph[U, I].values
    mean, std = float(train_vals.mean()), float(train_vals.std())
    lo, hi = float(it.possible_rating_values.min()), float(it.possible_rating_values.max())
    fea_kwargs = dict()
    if args.features:
        assert args.data_root is not None, "--features needs a data set (--data-root)"
        fea_kwargs = dict(use_fea_proj=True, features=graph.features)
    net = M.Net(graph, U, I, embed_units=args.embed, agg_units=(250,), out_units=(75,), nblocks=2, use_dae=True,
                dropout=0.5, agg_accum="sum", **fea_kwargs).to(dev)
    rating_it = it.rating_sampler(args.batch, "train", return_index=args.resident)
    recon_it = it.recon_nodes_sampler(1000000)
    opt = None
    resident = None
    dsampler = None
    if args.resident:
        from star_gcn_amd.resident import ResidentPlan
        resident = ResidentPlan(net, it.train_graph, device=dev)
        if args.device_sampler:
            from star_gcn_amd.device_sampler import DeviceBatchSampler
            dsampler = DeviceBatchSampler(resident, args.batch, embed_P_mask=0.1, embed_p_zero=0.0, seed=args.seed)
    t0 = time.time()
    t_train = 0.0
    if args.graph:
        assert dsampler is not None, "--graph needs --resident --device-sampler"
        run_graph(args, net, it, resident, dsampler, mean, std, lo, hi, dev)
        return
    for step in range(1, args.iters + 1):
        torch.cuda.synchronize()
        t_it = time.perf_counter()
        if dsampler is not None:    # samplers, batch plans and edge masking all on the device
            dbatch = dsampler.next_batch()
            y = (dbatch["ratings"] - mean) / std
            # training only needs the rating LOSS: the fused head (sg_pair_l2_hip) never materialises the scores
            preds, recons, gt = net.run(resident.set_batch_device(dbatch), rating_targets=y, rating_scale=1.0 / y.numel())
            batch = None
        else:
            batch = next(rating_it)
            pairs, ratings = batch[0], batch[1]
            noise, recon_ids, _ = next(recon_it)
        if batch is None:
            pass
        elif resident is not None:    # never aggregate over the edges being predicted: masked on the device
            preds, recons, gt = net.run(resident.set_batch(rating_node_pairs=pairs, edge_ids=batch[2],
                                                           embed_noise_dict=noise, recon_node_ids_dict=recon_ids))
        else:                       # reference-style: new CSRs, new plan, new uploads every iteration
            g = it.train_graph.remove_edges_by_id(U, I, pairs)
            preds, recons, gt = net(g, rating_node_pairs=pairs, embed_noise_dict=noise, recon_node_ids_dict=recon_ids,
                                    device=dev)
        if batch is not None:
            y = torch.from_numpy(((ratings - mean) / std).astype(np.float32)).to(dev)
        loss = M.star_gcn_loss(preds, recons, gt, y, recon_lambda=0.1)
        if opt is None:   # parameters are created lazily on the first forward
            opt = torch.optim.Adam(net.parameters(), lr=args.lr)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        torch.cuda.synchronize()
        if step > 3:
            t_train += time.perf_counter() - t_it
        if step % args.eval_every == 0 or step == 1:
            net.eval()
            rmse = evaluate(net, it, it.val_graph, "valid", mean, std, dev, lo, hi)
            net.train()
            print("iter %4d  loss %.4f  valid RMSE %.4f  (%.1f s)" % (step, float(loss.detach()), rmse, time.time() - t0))
    print("training iterations: %.2f ms/iter (%s planning, batch %d, after 3 warm-up iterations)" %
          (1e3 * t_train / max(args.iters - 3, 1), ("resident plan + device samplers" if dsampler is not None else "resident/device") if resident is not None else "host re-", args.batch))
    net.eval()
    print("test RMSE %.4f" % evaluate(net, it, it.test_graph, "test", mean, std, dev, lo, hi))


if __name__ == "__main__":
    main()
