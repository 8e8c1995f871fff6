import os, sys, numpy as np, torch, torch.distributed as dist, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_partition import _build, _global_problem, U, I, D
def run(net, graph, sub, ms, E, dev):
    pairs = np.stack([sub.edge_row_indices, sub.end_points])
    plan = net.make_plan(graph, rating_node_pairs=pairs, device=dev)
    y = torch.from_numpy(((sub.values - ms[0]) / ms[1]).astype(np.float32)).to(dev)
    net.zero_grad(set_to_none=True)
    preds = net.run(plan)[0][0]
    loss = (0.5 * (preds.view(-1) - y) ** 2).sum() / E
    loss.backward()
    return preds.detach().view(-1).cpu().clone()
def worker(rank, world, port, out):
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]=str(port)
    torch.cuda.set_device(0); dev=torch.device("cuda",0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import star_gcn_amd.dist as SD, star_gcn_amd.synthetic as S
    from star_gcn_amd.mxgraph.graph import HeterGraph
    graph, vals = _global_problem(); csr = graph[U, I]; ms = (vals.mean(), vals.std())
    ref = _build(graph, dev)
    pr1 = run(ref, graph, csr, ms, csr.nnz, dev)
    state = {k: v.detach().clone() for k, v in ref.state_dict().items()}
    pr2 = run(ref, graph, csr, ms, csr.nnz, dev)
    rg = {k: p.grad.detach().cpu().clone() for k, p in ref.named_parameters()}
    print(rank, "ref preds stable", float((pr1-pr2).abs().max()), flush=True)
    lo, hi = SD.balanced_row_blocks(csr.ind_ptr, world)[rank]
    a, b = int(csr.ind_ptr[lo]), int(csr.ind_ptr[hi])
    sub = S.user_block(graph, U, I, lo, hi)
    lg = HeterGraph({U: np.arange(hi-lo, dtype=np.int32), I: np.arange(csr.shape[1], dtype=np.int32)}, {(U, I): sub})
    net = _build(lg, dev); part = SD.NodePartition([U],[I])
    for enc in net.encoders:
        for layer in enc._blocks: layer.partition = part
    net.pair_partition = part
    pp1 = run(net, lg, sub, ms, csr.nnz, dev)
    ukey = [k for k in state if k.startswith("embed_layers") and state[k].shape[0] == csr.shape[0]][0]
    st = dict(state); st[ukey] = state[ukey][lo:hi].clone()
    missing = net.load_state_dict(st)
    print(rank, "load:", missing, "ukey", ukey, flush=True)
    pp2 = run(net, lg, sub, ms, csr.nnz, dev)
    print(rank, "partition preds vs ref: err %.3e scale %.3e" % (float((pp2 - pr2[a:b]).abs().max()), float(pr2.abs().max())), flush=True)
    SD.allreduce_grads(net.local_region_parameters())
    for k, p in net.named_parameters():
        r = rg[k][lo:hi] if k == ukey else rg[k]
        e = float((p.grad.cpu() - r).abs().max()); sc = float(r.abs().max())
        if e > 1e-4 * sc: print(rank, "GRAD MISMATCH", k, "err %.2e scale %.2e" % (e, sc), flush=True)
    print(rank, "done", flush=True)
    dist.destroy_process_group()
if __name__ == "__main__":
    import socket; s=socket.socket(); s.bind(("127.0.0.1",0)); port=s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, "/tmp"), nprocs=2, join=True)
