"""Round-4 experiment: overlap the HBM-bound gather of one node type with the matrix-core dense mix of the
other on CU-PARTITIONED streams (hipExtStreamCreateWithCUMask): gather on k CUs, GEMM on the other 256 - k.  Plain two-stream
overlap does nothing (tools/exp_overlap.py: each kernel alone fills every CU's wave slots).

Shapes: the config-5 shard (1.25 M users x 1 M items, 125 M ratings, R = 16, dim 256) -- item-side aggregation over user rows
(129 GB per launch) next to the user-side transform 1 M x 4096 x 256 (planes) / the item-side contraction 1 M x 256 x 4160
(hybrid); and the ML-10M pair for reference.   python tools/exp_cumask_overlap.py [config5|ml10m]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops  # noqa: E402

hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(bits):
    """torch stream over a HIP stream restricted to the CUs whose mask bits are set (list of bit indices)."""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


def wall(fn, n=7, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def on(stream, fn):
    def run():
        cur = torch.cuda.current_stream()
        stream.wait_stream(cur)
        with torch.cuda.stream(stream):
            fn()
        cur.wait_stream(stream)
    return run


def both(s1, f1, s2, f2):
    def run():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            f1()
        with torch.cuda.stream(s2):
            f2()
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    return run


def gather_case(S, T, nnz, C, seed, dst_group=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    lens = torch.poisson(torch.full((S,), nnz / S, device="cuda"), generator=g).long()
    indptr = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), lens.cumsum(0)]).int()
    n = int(indptr[-1])
    idx = torch.randint(0, T, (n,), generator=g, device="cuda").int()
    w = torch.rand(n, generator=g, device="cuda")
    x = torch.randn(T, C, device="cuda")
    out = torch.empty(S, C, device="cuda")
    return (lambda: ops.gather_sum(out, x, idx, indptr, w, S, C)), n


def gemm_case(M, N, K):
    a = torch.randn(M, K, device="cuda")
    b = torch.randn(N, K, device="cuda")
    c = torch.empty(M, N, device="cuda")
    return lambda: ops.gemm(a, b, trans_b=True, out=c)


which = sys.argv[1] if len(sys.argv) > 1 else "config5"
if which == "config5":
    gat, n = gather_case(16_000_000, 1_250_000, 125_000_000, 256, 1)
    cases = [("1M x 4096 x 256 (planes)", gemm_case(1_000_000, 4096, 256)), ("1M x 256 x 4160 (hybrid)", gemm_case(1_000_000, 256, 4160))]
    gbytes = n * 1032 / 1e9
else:
    gat, n = gather_case(106_770, 69_878, 10_000_000, 256, 1)
    cases = [("10677 x 2560 x 256 x4", (lambda f: (lambda: [f() for _ in range(4)]))(gemm_case(10677, 2560, 256))),
             ("69878 x 256 x 2624", gemm_case(69878, 256, 2624))]
    gbytes = n * 1032 / 1e9
for f in [gat] + [c[1] for c in cases]:
    f()
torch.cuda.synchronize()
tg = wall(gat)
print("gather alone: %.2f ms (%.2f TB/s algorithmic)" % (tg, gbytes / tg), flush=True)
plain1, plain2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, mm in cases:
    tm = wall(mm)
    seq = wall(lambda: (gat(), mm()))
    two = wall(both(plain1, gat, plain2, mm))
    print("%-28s gemm alone %.2f ms   sequential %.2f   two plain streams %.2f" % (name, tm, seq, two), flush=True)
    for layout in ("balanced",):
        for k in (64, 96, 128, 144, 160, 176, 192, 208, 224):
            # tools/cumask_probe: mask bit b is CU b // 8 of XCD b % 8, so the FIRST k bits are k / 8 CUs on every XCD --
            # both partitions keep all eight L2s and the whole fabric
            gbits = list(range(k))
            mbits = [b for b in range(256) if b not in set(gbits)]
            from star_gcn_amd import _lib as L
            L.release_workspaces()            # one scratch buffer per stream: 19 GB each at this shape
            torch.cuda.empty_cache()
            sg_, sm_ = masked_stream(gbits), masked_stream(mbits)
            tg_k = wall(on(sg_, gat), n=5)
            tm_k = wall(on(sm_, mm), n=5)
            con = wall(both(sg_, gat, sm_, mm), n=5)
            print("   %-11s gather on %3d CUs: %.2f ms alone   gemm on %3d CUs: %.2f ms alone   together %.2f ms  (sequential %.2f, x%.2f)"
                  % (layout, k, tg_k, 256 - k, tm_k, con, seq, seq / con), flush=True)
