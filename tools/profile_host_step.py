#!/usr/bin/env python3
"""cProfile of the HOST side of one benchmark step (Python + ctypes launch overhead) on a small shape where the step
is launch-bound -- the regime each rank of an 8-GPU strong-scaling run of ML-10M is in."""
import cProfile, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import star_gcn_amd.model as M, star_gcn_amd.synthetic as S
U, I, D = "user", "movie", 256
graph, eu, ei, vals = S.make_graph(sys.argv[1] if len(sys.argv) > 1 else "ml-1m")
dev = torch.device("cuda", 0)
net = M.Net(graph, U, I, embed_units=D, agg_units=(D, D), out_units=(D, D), nblocks=1, use_dae=False, agg_accum="sum").to(dev)
plan = net.make_plan(graph, rating_node_pairs=np.stack([eu, ei]), device=dev)
y = torch.from_numpy(((vals - vals.mean()) / vals.std()).astype(np.float32)).to(dev)
def step():
    net.zero_grad(set_to_none=True)
    preds, _, _ = net.run(plan)
    loss = (0.5 * (preds[0].view(-1) - y) ** 2).sum() / eu.size
    loss.backward()
for _ in range(5): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): step()
t_enq = (time.perf_counter() - t) / 50; torch.cuda.synchronize(); t_all = (time.perf_counter() - t) / 50
print("host enqueue %.2f ms/step, wall %.2f ms/step" % (t_enq * 1e3, t_all * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
