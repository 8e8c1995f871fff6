import sys, os, socket, torch
sys.path.insert(0, os.getcwd())
import torch.multiprocessing as mp
from tests.test_gpu_bench_multirank import _config5_union_worker
if __name__ == "__main__":
    shape = tuple(int(v) for v in sys.argv[1].split(","))
    dim = int(sys.argv[2])
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    res = "/tmp/r.pt"
    mp.spawn(_config5_union_worker, args=(2, port, shape, dim, res, len(sys.argv) > 3), nprocs=2, join=True)
    r = torch.load(res)
    print("loss", r["loss"], r["ref_loss"], "fused launches", r["fused_launches"])
    for k, (e, sc, fe, fn, ro) in sorted(r["errs"].items(), key=lambda kv: -kv[1][0] / max(kv[1][1], 1e-30)):
        print("%-60s err %.3e scale %.3e ratio %.2e   fro %.2e rows off %.2e" % (k, e, sc, e / max(sc, 1e-30), fe / max(fn, 1e-300), ro))
