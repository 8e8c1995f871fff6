#!/usr/bin/env python
"""Fuzz of sg_gemm_f32_hip (default backend: f16x3): random M, N, K (1 .. 2600; also the aggregator's K = a few hundred with wide
M), the four operand layouts, row strides wider than the rows, rows whose scales span decades; error = max |C - C64| / sum |a||b|
per element against float64.  fp32-accurate means <= ~3e-7."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import ops            # noqa: E402


def main(n_cases, tol=1e-6):
    rng = np.random.default_rng(99)
    dev = torch.device("cuda")
    bad = 0
    for case in range(n_cases):
        M, N, K = (int(10 ** rng.uniform(0, 3.42)) for _ in range(3))
        ta, tb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        pad_a, pad_b = int(rng.choice([0, 0, 4, 64])), int(rng.choice([0, 0, 4, 64]))
        g = torch.Generator(device=dev).manual_seed(case)
        ash = (K, M) if ta else (M, K)
        bsh = (N, K) if tb else (K, N)
        A = torch.randn(ash[0], ash[1] + pad_a, device=dev, generator=g)[:, :ash[1]]
        B = torch.randn(bsh[0], bsh[1] + pad_b, device=dev, generator=g)[:, :bsh[1]]
        if case % 3 == 0:       # rows of very different scale (hub rows)
            A = A * torch.exp(2.5 * torch.randn(ash[0], 1, device=dev, generator=g))
            A = torch.cat([A, A.new_zeros(ash[0], pad_a)], 1)[:, :ash[1]] if pad_a else A
        use_bias, acc_into, act = bool(rng.integers(0, 2)), bool(rng.integers(0, 3) == 0), [None, "leaky", "relu", "tanh", "sigmoid"][int(rng.integers(0, 5))]
        pad_c = int(rng.choice([0, 0, 4, 64]))
        bias = torch.randn(N, device=dev, generator=g) if use_bias else None
        Cbuf = torch.randn(M, N + pad_c, device=dev, generator=g)
        C0 = Cbuf[:, :N].clone()
        C = ops.gemm(A, B, trans_a=ta, trans_b=tb, bias=bias, act=None if acc_into else act, out=Cbuf[:, :N], accumulate=acc_into)
        assert pad_c == 0 or bool((Cbuf[:, N:] == Cbuf[:, N:]).all())
        a64 = (A.double().t() if ta else A.double())
        b64 = (B.double().t() if tb else B.double())
        ref = a64 @ b64
        mag = a64.abs() @ b64.abs() + 1.0
        if use_bias:
            ref = ref + bias.double()[None]
        if acc_into:
            ref = ref + C0.double()
        elif act is not None:
            # activations are 1-Lipschitz (leaky / relu) or flatter: the pre-activation error bounds the output error
            ref = {"leaky": lambda t: torch.where(t > 0, t, 0.1 * t), "relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid}[act](ref)
        err = float(((C.double() - ref).abs() / mag.clamp_min(1e-300)).max())
        # rows of very different scale inside one 32-row scale block: the error model is block-relative (gemm_f16x3.hip), so a row
        # 100 x below its block's largest loses that factor against its OWN sum |a||b|
        flag = "" if err <= (2e-5 if case % 3 == 0 else tol) else "   <-- FAIL"
        bad += bool(flag)
        if flag or case % 20 == 0:
            print("case %3d: M %4d N %4d K %4d ta %d tb %d pads %2d %2d: %.2e%s" % (case, M, N, K, ta, tb, pad_a, pad_b, err, flag), flush=True)
    print("FAILURES %d of %d" % (bad, n_cases))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 300))
