#!/usr/bin/env python3
"""Benchmark of the STAR-GCN hot path on MI355X: edges/sec for forward+backward of a 2-layer multi-link GCN on a
MovieLens-10M-SHAPED synthetic bipartite graph (BASELINE.json metric; workload definition SURVEY.md section 8d).

  python bench.py [--gpus N --steps K --warmup W]          (N > 1: launched by torch.distributed.run, one rank/GPU)

One step = embedding gather -> 2 stacked HeterGCNLayers (both node types, all rating levels, full neighbourhood)
-> rating head over ALL ratings -> loss -> full backward to every parameter and the embedding tables.  Plans are
built once outside the timed region (inputs resident in HBM).  Prints ONE JSON line with `roofline` (dominant
kernel = seg_gather_kernel, timed with HIP events on the launch stream) and `cpu_baseline` (oracle port timed on
this host, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK = 8.0e12  # bytes/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--shape", default="ml-10m")
    p.add_argument("--dim", type=int, default=256)
    p.add_argument("--order", default="auto", choices=["auto", "transform_first", "aggregate_first"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-user-frac", type=float, default=0.125, help="share of users in the CPU-baseline sample")
    p.add_argument("--cpu-steps", type=int, default=2)
    return p.parse_args()


def main():
    args = parse()
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # SG_BENCH_FORCE_DIST=1: take the partitioned (N > 1) code path -- RCCL init, barriers, all-reduces, partition
    # crossings -- even with a single rank; development check of that path on a 1-GPU box.
    dist_on = world > 1 or os.environ.get("SG_BENCH_FORCE_DIST") == "1"
    # SG_BENCH_BACKEND=gloo lets several ranks share ONE GPU (development check of the N > 1 code path on a 1-GPU
    # box); the real multi-GPU run uses nccl == RCCL with one rank per GPU.
    backend = os.environ.get("SG_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    import star_gcn_amd.dist as SD
    import star_gcn_amd.model as M
    import star_gcn_amd.functional as SF
    import star_gcn_amd.ops as ops
    import star_gcn_amd.synthetic as S
    from star_gcn_amd.mxgraph.graph import HeterGraph

    U, I = "user", "movie"
    torch.manual_seed(1234)  # identical replicated parameters on every rank
    graph, eu, ei, vals = S.make_graph(args.shape)
    csr = graph[U, I]
    n_user, n_item, E_total, R = csr.shape[0], csr.shape[1], csr.nnz, int(csr.multi_link.size)
    mean, std = float(vals.mean()), float(vals.std())
    if dist_on:
        lo, hi = SD.balanced_row_blocks(csr.ind_ptr, world)[rank]
        sub = S.user_block(graph, U, I, lo, hi)
        lgraph = HeterGraph({U: np.arange(hi - lo, dtype=np.int32), I: np.arange(n_item, dtype=np.int32)}, {(U, I): sub})
    else:
        lgraph, sub = graph, csr
    pairs = np.stack([sub.edge_row_indices, sub.end_points])
    E_local = sub.nnz
    y = torch.from_numpy(((sub.values - mean) / std).astype(np.float32)).to(dev)

    D = args.dim
    net = M.Net(lgraph, U, I, embed_units=D, agg_units=(D, D), out_units=(D, D), nblocks=1, use_dae=False,
                activation="leaky", dropout=0.0, agg_accum="sum", agg_order=args.order).to(dev)
    if dist_on:
        part = SD.NodePartition([U], [I])
        for enc in net.encoders:
            for layer in enc._blocks:
                layer.partition = part
        net.pair_partition = part
    t_plan = time.perf_counter()
    # every node of the (local) graph is computed, in natural order: index takes between levels are identities
    plan = net.make_plan(lgraph, rating_node_pairs=pairs, device=dev,
                         full_node_ids={k: lgraph.node_ids_dict[k] for k in (U, I)})
    t_plan = time.perf_counter() - t_plan

    def step():
        net.zero_grad(set_to_none=True)
        preds, _, _ = net.run(plan)
        loss = SF.l2_loss(preds[0].view(-1), y, 1.0 / E_total)
        loss.backward()
        if dist_on:
            SD.allreduce_grads(net.local_region_parameters())
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    ops.gather_profile(True)      # HIP events around every gather launch, on the launch stream, inside the library
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.gather_profile(False)
    timeline = ops.gather_profile_read()
    if dist_on:
        tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the dominant kernel (aggregation launches: width D over all local edges) ----------
    agg = [(t, nnz, C) for t, nnz, C in timeline if C == D and nnz == max(E_local, 1) and t > 0]
    roof = None
    if agg:
        avg = sum(t for t, _, _ in agg) / len(agg)
        bytes_per_launch = (8 + 4 * D) * E_local          # idx + support + one fp32 row per edge visit
        ach = bytes_per_launch / avg
        traffic = None   # PMC bytes per launch come from separate rocprofv3 --pmc passes (committed summary)
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                rec = json.load(f).get("%s:%d" % (args.shape, D))
            if rec and world == 1:
                traffic = rec["traffic_bytes_per_launch_mean"] * (E_local / rec["edges_per_launch"])
        except (OSError, ValueError, KeyError):
            pass
        roof = {"bound": "hbm", "kernel": "seg_gather_kernel", "achieved": ach / 1e9, "peak": HBM_PEAK / 1e9,
                "unit": "GB/s", "frac": ach / HBM_PEAK, "traffic": traffic, "launches_per_step": len(agg) / args.steps,
                "avg_launch_ms": avg * 1e3, "algorithmic_bytes_per_launch": bytes_per_launch,
                "note": ("gathered matrices (%d-%d MB) fit the 256 MB Infinity Cache at this shape: the rate is a die-level "
                         "fabric rate (PMC traffic in profiles/), the HBM-bound case is --shape hbm-stress"
                         if max(max(n_user, n_item) * D * 4, min(n_user, n_item) * R * D * 4) < 256 * 2 ** 20 else
                         "gathered matrices (%d-%d MB) exceed the 256 MB Infinity Cache: HBM-bound") %
                        (max(n_user, n_item) * D * 4 // 2 ** 20, min(n_user, n_item) * R * D * 4 // 2 ** 20)}

    loss_total = loss.detach().clone()
    if dist_on:      # every rank holds its users' share of the loss; report the whole (outside the timed region)
        loss_total = SD.all_reduce_sum(loss_total.view(1))[0]
    ms = elapsed / args.steps * 1e3
    value = E_total / (elapsed / args.steps)
    out = {
        "metric": "edges/sec (fwd+bwd) 2-layer multi-link GCN, ML-10M shape, 1/2/4/8 GPU + %HBM roofline",
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s-shaped synthetic bipartite graph: %d users x %d items, %d ratings, %d rating levels, "
                               "dim %d; 2 stacked HeterGCNLayers (sum accum, symm support, leaky 0.1), both node types, "
                               "full neighbourhood, rating head over all ratings, fwd+bwd" % (args.shape, n_user, n_item,
                                                                                          E_total, R, D),
                   "partition": "single GPU" if world == 1 else "1-D user-block node partition, items replicated, "
                                "RCCL all-reduce of item-side partials", "order": args.order,
                   "plan_build_s": round(t_plan, 2), "loss": float(loss_total),
                   "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)},
        "roofline": roof,
        "step_roofline_frac": value * 8 * (8 + 4 * D) / (world * HBM_PEAK),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(graph, U, I, D, args)
    if rank == 0:
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


def cpu_baseline(graph, U, I, D, args):
    """Oracle port (oracle/cpu_step.py) timed on this host on a bounded sample: the first `cpu-user-frac` of the users
    against all items (same generator, same widths, same network)."""
    import star_gcn_amd.synthetic as S
    from oracle import cpu_step as C
    from star_gcn_amd.mxgraph.graph import HeterGraph
    csr = graph[U, I]
    n_u = max(1, int(csr.shape[0] * args.cpu_user_frac))
    sub = S.user_block(graph, U, I, 0, n_u)
    g = HeterGraph({U: np.arange(n_u, dtype=np.int32), I: np.arange(csr.shape[1], dtype=np.int32)}, {(U, I): sub})
    lv = dict()
    for dst, a, b in (("user", U, I), ("item", I, U)):
        eps, _, ips, sps = g[a, b].sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
        lv[dst] = ([np.ascontiguousarray(e, np.int32) for e in eps], ips, [np.ascontiguousarray(s, np.float32) for s in sps])
    lv["pairs"] = (sub.end_points, sub.ind_ptr, None)
    C.run_cpu_step(lv, n_u, csr.shape[1], D, steps=1)     # warm-up (page-in, BLAS threads)
    sec = C.run_cpu_step(lv, n_u, csr.shape[1], D, steps=args.cpu_steps)
    sec_fair = C.run_cpu_step(lv, n_u, csr.shape[1], D, steps=1, fair=True)   # backward parallel over rows (transposed CSR)
    info = C.host_info()
    return {"value": sub.nnz / sec, "unit": "edges/s", "cores": info["logical_cores"], "kind": "port",
            "fair_value": sub.nnz / sec_fair,
            "fair_note": "same port with the data-gradient kernel parallelised over destination rows through the "
                         "transposed CSR (the reference runs it serially for K = 1, seg_op.cc:232-233)",
            "sample": "users [0,%d) x all %d items = %d ratings of the same graph, %d step(s), %.2f s/step; seg ops = C "
                      "restatement of reference seg_op.cc CPU kernels (reference OpenMP placement: forward over rows, "
                      "backward serial), dense = torch-CPU BLAS standing in for MXNet FullyConnected" %
                      (n_u, csr.shape[1], sub.nnz, args.cpu_steps, sec),
            "host": info}


if __name__ == "__main__":
    main()
