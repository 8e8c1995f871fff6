"""VERDICT r4 #2 (aggregate -> contract in one kernel), the feasibility number: does the contraction's matrix work hide under the
HBM-bound gather on the same CUs?  Times an HBM-bound aggregate-first gather (grouped destination, 16 levels, dim 256,
source 2 GB, destination R-expanded) with the shipped library and with builds whose gather issues 6 / 12 dummy
v_mfma_f32_32x32x16_f16 per four gathered rows (tools/build_fused_probe.sh; the fused layer needs 6.25).  One subprocess per
build (SG_LIB_OVERRIDE is read at import)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one():
    import numpy as np
    import torch
    from star_gcn_amd import ops
    from tools.microbench import timeit
    rng = np.random.default_rng(0)
    S, T, nnz, C = 400_000 * 16, 2_000_000, 40_000_000, 256          # 400 k destination rows x 16 levels, 2 M source rows (2 GB)
    lens = rng.multinomial(nnz, rng.dirichlet(np.ones(S) * 2.0))
    indptr = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).cuda()
    w = torch.rand(nnz).cuda()
    x = torch.randn(T, C, device="cuda")
    out = torch.empty(S, C, device="cuda")
    idx = torch.from_numpy(rng.integers(0, T, nnz).astype(np.int32)).cuda()
    t = timeit(lambda: ops.gather_sum(out, x, idx, indptr, w, S, C), n=7, warm=2)
    gb = nnz * (8 + 4 * C) / 1e9
    print("%.3f ms per launch of %.1f GB algorithmic (+ %.1f GB written) = %.2f TB/s" % (t * 1e3, gb, S * C * 4 / 1e9, gb / t / 1e3), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for name, so in [("shipped gather", None), ("+ 6 matrix instructions per 4 rows", "fp_6"), ("+ 12 per 4 rows", "fp_12")]:
            env = dict(os.environ)
            if so:
                env["SG_LIB_OVERRIDE"] = os.path.join(ROOT, "tools", "ablate", so, "libstargcn_hip.so")
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
            print("%-38s %s" % (name, r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else r.stderr[-400:]), flush=True)
