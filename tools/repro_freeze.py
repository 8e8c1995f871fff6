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
faulthandler.enable()


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


if __name__ == "__main__":
    T0 = time.perf_counter()
    mode = sys.argv[1] if len(sys.argv) > 1 else "alloc"
    loops = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    import pytest
    if mode in ("verify", "both"):
        do_verify()
    if mode in ("alloc", "both"):
        do_alloc()
    for k in range(loops):
        faulthandler.dump_traceback_later(300, exit=False)
        t = time.perf_counter()
        rc = pytest.main(["-x", "-q", "-p", "no:cacheprovider", "tests/test_gpu_dense_multilink.py", "-k", "gemm or linear or fused"])
        faulthandler.cancel_dump_traceback_later()
        log("pass %d: pytest rc %s in %.1f s" % (k, rc, time.perf_counter() - t))
        if rc != 0:
            sys.exit(3)
    log("no freeze in %d passes (mode %s)" % (loops, mode))
