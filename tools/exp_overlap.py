"""Can the matrix-core GEMM (x6v2: one 8-wave workgroup per CU) hide under the L1-miss-bound gather (28 single-wave
workgroups per CU)?  Same work serial on one stream vs on two streams (GEMM stream with high priority / launched first)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops, _lib as L
rng = np.random.default_rng(0)
nnz, C, S, T = 10_000_000, 256, 69878, 106770
lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 2.0))
indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
seg = np.repeat(np.arange(S), lens)
pop = rng.lognormal(0.0, 1.5, T); rng.shuffle(pop)
idx = rng.choice(T, size=nnz, p=pop / pop.sum()).astype(np.int64)
idx_d = torch.from_numpy(idx[np.lexsort((idx, seg))].astype(np.int32)).cuda()
w = torch.rand(nnz).cuda(); x = torch.randn(T, C, device="cuda"); out = torch.empty(S, C, device="cuda")
L.lib().sg_gather_tuning(-1, 4)
a = torch.randn(10677, 2624, device="cuda"); b = torch.randn(256, 2624, device="cuda")
a2 = torch.randn(69878, 256, device="cuda"); b2 = torch.randn(256, 256, device="cuda")
cbuf = torch.empty(10677, 256, device="cuda"); c2 = torch.empty(69878, 256, device="cuda")

def gather():
    ops.gather_sum(out, x, idx_d, indptr, w, S, C)

def gemms(n):
    for _ in range(n):
        ops.gemm(a, b, trans_b=True, out=cbuf)
        ops.gemm(a2, b2, trans_b=True, out=c2)

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))

for n in (1, 2, 3):
    tg = timed(gather); tm = timed(lambda: gemms(n))
    ser = timed(lambda: (gather(), gemms(n)))
    for prio, first in ((0, "gather"), (-1, "gemm"), (-1, "gather")):
        sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=prio)
        def both():
            cur = torch.cuda.current_stream()
            sa.wait_stream(cur); sb.wait_stream(cur)
            order = ((sb, lambda: gemms(n)), (sa, gather)) if first == "gemm" else ((sa, gather), (sb, lambda: gemms(n)))
            for st, f in order:
                with torch.cuda.stream(st):
                    f()
            cur.wait_stream(sa); cur.wait_stream(sb)
        t2 = timed(both)
        print("gemms x%d: gather %.3f ms, gemm %.3f ms, serial %.3f ms | two streams (gemm prio %d, %s first) %.3f ms  -> hidden %.0f %% of the GEMM time"
              % (n, tg, tm, ser, prio, first, t2, 100 * (ser - t2) / tm), flush=True)
