"""Does a dense mix overlap with a gather when both are in flight on two HIP streams?  (ML-10M shapes.)
python tools/exp_overlap.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops

g = torch.Generator().manual_seed(0)
def gather_case(S, T, nnz, C=256):
    lens = torch.distributions.Multinomial(nnz, torch.rand(S, generator=g) ** 2 + 1e-3).sample().long()
    indptr = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)]).int().cuda()
    idx = torch.randint(0, T, (nnz,), generator=g).int().cuda()
    w = torch.rand(1, nnz, generator=g).cuda()
    x = torch.randn(1, T, C, device="cuda"); out = torch.empty(1, S, C, device="cuda")
    return lambda: ops.seg_weighted_pool(x, w, idx, indptr, out=out)

def gemm_case(M, N, K, reps):
    a = torch.randn(M, K, device="cuda"); b = torch.randn(N, K, device="cuda"); c = torch.empty(M, N, device="cuda")
    def run():
        for _ in range(reps):
            ops.gemm(a, b, trans_b=True, out=c)
    return run

def wall(fn, n=15, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both(f1, f2):
    def run():
        cur = torch.cuda.current_stream()
        e = torch.cuda.Event(); e.record(cur)
        s1.wait_event(e); s2.wait_event(e)
        with torch.cuda.stream(s1): f1()
        with torch.cuda.stream(s2): f2()
        cur.wait_stream(s1); cur.wait_stream(s2)
    return run

for gname, gat in [("users<-items (68 MB class)", gather_case(69878, 10677, 10_000_000)), ("items<-users", gather_case(10677, 69878, 10_000_000))]:
    for mname, mm in [("10677x2560x256 x4", gemm_case(10677, 2560, 256, 4)), ("69878x256x2624 x1", gemm_case(69878, 256, 2624, 1)),
                      ("4096^3 x1", gemm_case(4096, 4096, 4096, 1))]:
        with torch.cuda.stream(s1): gat()
        with torch.cuda.stream(s2): mm()
        torch.cuda.synchronize()
        tg, tm = wall(gat), wall(mm)
        seq = wall(lambda: (gat(), mm()))
        con = wall(both(gat, mm))
        print("%-28s + %-20s gather %.3f  gemm %.3f  sequential %.3f  two streams %.3f ms  (max %.3f)" % (gname, mname, tg, tm, seq, con, max(tg, tm)), flush=True)
g2 = gather_case(10677, 69878, 10_000_000)
g1 = gather_case(69878, 10677, 10_000_000)
with torch.cuda.stream(s2): g2()
torch.cuda.synchronize()
print("two gathers: %.3f + %.3f sequential %.3f, two streams %.3f ms" % (wall(g1), wall(g2), wall(lambda: (g1(), g2())), wall(both(g1, g2))))
