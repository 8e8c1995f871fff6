"""Round-4 parameter check of the gather under the SHIPPED launch structure (two source-range phases, 4 column slices): chunk size
and rows in flight per edge group, one development build each (tools/build_gather_variants.sh).  Runs itself once per build in a
subprocess (SG_LIB_OVERRIDE is read at import).   python tools/exp_r4_gather_variants.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    import torch
    from star_gcn_amd import ops, _lib as L
    from tools.microbench import timeit
    rng = np.random.default_rng(0)
    nnz, C = 10_000_000, 256
    lib = L.lib()
    res = []
    for name, S, T, sigma in (("68MB", 106770, 69878, 1.0), ("104MB", 69878, 106770, 1.5)):
        lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 2.0))
        indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
        seg = np.repeat(np.arange(S), lens)
        w = torch.rand(nnz).cuda()
        x = torch.randn(T, C, device="cuda")
        out = torch.empty(S, C, device="cuda")
        pop = rng.lognormal(0.0, sigma, T)
        idx = rng.choice(T, size=nnz, p=pop / pop.sum()).astype(np.int64)
        idx_d = torch.from_numpy(idx[np.lexsort((idx, seg))].astype(np.int32)).cuda()
        t1 = timeit(lambda: ops.gather_sum(out, x, idx_d, indptr, w, S, C, src_bytes=T * C * 4) if False else ops.gather_sum(out, x, idx_d, indptr, w, S, C))
        idx_p = torch.empty(nnz, dtype=torch.int32, device="cuda")
        wpos_p = torch.empty(nnz, dtype=torch.int32, device="cuda")
        indptr_p = torch.empty(2 * (S + 1), dtype=torch.int32, device="cuda")
        nnz_p = torch.empty(2, dtype=torch.int32, device="cuda")
        ws, wsn = L.workspace(lib.sg_gather_phases_workspace_bytes(nnz), x.device)
        L.check(lib.sg_gather_phases_build_hip(L.ptr(idx_p), L.ptr(wpos_p), L.ptr(indptr_p), L.ptr(nnz_p), L.ptr(idx_d), L.ptr(indptr),
                                               S, nnz, T, L.ptr(ws), wsn, L.stream_ptr()), "phases")
        n0, n1 = (int(v) for v in nnz_p.cpu())
        ph = L.GatherPhasesStruct()
        ph.num_phases, ph.idx, ph.wpos, ph.indptr = 2, idx_p.data_ptr(), wpos_p.data_ptr(), indptr_p.data_ptr()
        ph.nnz_p[0], ph.nnz_p[1] = n0, n1
        gws, gwsn = L.workspace(lib.sg_seg_weighted_pool_workspace_bytes(1, S, nnz, C), x.device)
        t2 = timeit(lambda: L.check(lib.sg_seg_gather_sum_phased_hip(L.ptr(out), 1, C, L.ptr(x), 1, C, L.ptr(w), ctypes.byref(ph), S, C, 1, 0,
                                                                     0.0, L.ptr(gws), gwsn, L.stream_ptr(), T * C * 4), "phased"))
        res.append("%s: 1 launch %.3f  2 phases %.3f ms" % (name, t1 * 1e3, t2 * 1e3))
    print("   ".join(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        builds = [("shipped (chunk 256, U 4)", None)] + [(n, os.path.join(ROOT, "tools", "ablate", "g_" + n, "libstargcn_hip.so"))
                                                         for n in ("chunk128", "chunk512", "u2", "u8", "chunk512u8")]
        for name, so in builds:
            env = dict(os.environ)
            if so:
                if not os.path.exists(so):
                    print("%-26s (not built)" % name)
                    continue
                env["SG_LIB_OVERRIDE"] = so
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
            print("%-26s %s" % (name, out.stdout.strip().splitlines()[-1] if out.returncode == 0 and out.stdout.strip() else out.stderr[-300:]), flush=True)
