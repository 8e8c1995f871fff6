#!/usr/bin/env python3
"""Reproducer for the round-3 in-process freeze ("Known"): after the 140 GB config-5 verification had
run INSIDE the pytest process, a later small GEMM test froze.  Modes (argv[1]):

  alloc    allocate ~150 GB of HBM through torch in 10 GB blocks, write every block, free, empty_cache(); then loop the GEMM tests
  verify   bench.hbm_case + bench.verify_leg in this process (what tests/test_gpu_bench_verify.py did until commit f5935b8),
           del + empty_cache(), then loop the GEMM tests
  both     verify, then alloc, then the loop

The loop runs tests/test_gpu_dense_multilink.py -k gemm through pytest.main in THIS process, `loops` (argv[2], default 6)
times; faulthandler dumps every thread's Python stack if one pass takes longer than 300 s.  Driven by tools/repro_freeze.sh,
which attaches rocgdb for the native stacks when the process stops making progress."""
import faulthandler
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import signal
_FH = open(os.environ.get("SG_REPRO_FH", "/tmp/repro_fh.txt"), "w")
faulthandler.enable(file=_FH)
faulthandler.register(signal.SIGUSR1, file=_FH, all_threads=True)      # the watchdog samples the Python stacks with SIGUSR1


def log(msg):
    print("[repro %7.1fs] %s" % (time.perf_counter() - T0, msg), flush=True)


def do_alloc(gb=150):
    import torch
    blocks = []
    for k in range(gb // 10):
        b = torch.empty(10 * 2 ** 30, dtype=torch.uint8, device="cuda")
        b.fill_(k)
        blocks.append(b)
    torch.cuda.synchronize()
    log("holding %d GB (max_memory_allocated %.1f GB)" % (10 * len(blocks), torch.cuda.max_memory_allocated() / 2 ** 30))
    s = sum(int(b[12345].item()) for b in blocks)
    del blocks
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    log("freed (checksum %d), reserved now %.1f GB" % (s, torch.cuda.memory_reserved() / 2 ** 30))


def do_verify():
    import torch
    import bench
    dev = torch.device("cuda", 0)
    c = bench.hbm_case("1250000,1000000,125000000,16", 256, "auto", dev)
    log("config-5 shard built: %d ratings" % c.E)
    v = bench.verify_leg(c.net, c.step, (c.dg.ind_ptr, c.dg.end_points, c.dg.level, c.ni, c.R, None), c.y, 1.0 / c.E)
    log("verify: max_rel_err %.2e, peak %.1f GB, %.1f s" % (v["max_rel_err"], v["peak_hbm_gb_incl_checker"], v["seconds"]))
    del c
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    log("released, reserved now %.1f GB" % (torch.cuda.memory_reserved() / 2 ** 30))


def micro():
    """a few primitive host <-> runtime operations, timed: which of them degrades as the process ages?"""
    import torch
    from star_gcn_amd import ops
    out = {}
    a, b = torch.randn(130, 75, device="cuda"), torch.randn(250, 75, device="cuda")
    torch.cuda.synchronize()

    def t(fn, n):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    out["gemm_small_us"] = t(lambda: ops.gemm(a, b, trans_b=True), 200)
    # the same product through the raw ABI with its OWN small workspace (not the cached one, which after a config-5 step is
    # tens of GB) and, separately, the Python-side pieces of ops.gemm
    from star_gcn_amd import _lib as L
    lib = L.lib()
    c = torch.empty(130, 250, device="cuda")
    small = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    st = L.stream_ptr()
    out["gemm_raw_small_ws_us"] = t(lambda: lib.sg_gemm_f32_hip(L.ptr(c), 250, L.ptr(a), 75, 0, L.ptr(b), 75, 1, 130, 250, 75, None, 0,
                                                                0.1, 0, L.ptr(small), small.numel(), st), 200)
    big, bign = L.workspace(1 << 20, a.device)
    out["ws_cached_mb"] = bign / 2 ** 20
    out["gemm_raw_cached_ws_us"] = t(lambda: lib.sg_gemm_f32_hip(L.ptr(c), 250, L.ptr(a), 75, 0, L.ptr(b), 75, 1, 130, 250, 75, None, 0,
                                                                 0.1, 0, L.ptr(big), bign, st), 200)
    out["ws_query_us"] = t(lambda: lib.sg_gemm_f32_workspace_bytes(130, 250, 75, 0), 200)
    out["py_workspace_us"] = t(lambda: L.workspace(1 << 20, a.device), 200)
    out["torch_empty_out_us"] = t(lambda: torch.empty((130, 250), dtype=torch.float32, device="cuda"), 200)
    out["torch_add_us"] = t(lambda: a.add_(1.0), 200)
    out["sync_us"] = t(torch.cuda.synchronize, 200)
    out["alloc_1mb_us"] = t(lambda: torch.empty(1 << 20, dtype=torch.uint8, device="cuda"), 200)
    out["alloc_fresh_64mb_us"] = t(lambda: (torch.empty(64 << 20, dtype=torch.uint8, device="cuda"), torch.cuda.empty_cache()), 5)
    out["event_us"] = t(lambda: torch.cuda.Event(enable_timing=True).record(), 200)
    out["h2d_us"] = t(lambda: torch.zeros(1000).cuda(), 100)
    out["d2h_us"] = t(lambda: a.cpu(), 100)
    log("micro: " + "  ".join("%s %.0f" % kv for kv in out.items()))


if __name__ == "__main__":
    T0 = time.perf_counter()
    mode = sys.argv[1] if len(sys.argv) > 1 else "alloc"
    loops = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    import pytest
    if mode == "none":
        pass
    if mode in ("verify", "both"):
        do_verify()
    if mode in ("alloc", "both"):
        do_alloc()
    micro()
    for k in range(loops):
        faulthandler.dump_traceback_later(200, repeat=True, file=_FH)
        t = time.perf_counter()
        rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", "--durations=8", "tests/test_gpu_dense_multilink.py", "-k",
                          os.environ.get("SG_REPRO_K", "gemm or linear or fused")])
        faulthandler.cancel_dump_traceback_later()
        micro()
        log("pass %d: pytest rc %s in %.1f s" % (k, rc, time.perf_counter() - t))
        if rc != 0:
            sys.exit(3)
    log("no freeze in %d passes (mode %s)" % (loops, mode))
