"""The two workloads bench.py TIMES, checked at the size it times them: the 2-layer network with output
Dense, rating head through sg_pair_l2_hip (incl. the source-partitioned `parts == 8` path of the full-batch head) and
the full backward -- built by bench.py's own case builders (`bench.py --verify-only`) -- against the float64 evaluation
of the network's DEFINITION over the whole graph (tools/f64_check.py; pinned against autograd in
tests/test_f64_checker.py).

Compared: loss, every output row of both layers and node types, both rating projections, every embedding-gradient row,
every weight and bias gradient.  Tolerance: 1e-5 of each tensor's scale (north star).

Each case runs in its own process, like the multi-rank bench tests: the config-5 case holds 140 GB of HBM at its peak
and its process is gone, memory and all, when the next test starts."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(leg, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--verify-only", leg], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    v = json.loads(p.stdout.strip().splitlines()[-1])["verify"]
    print({k: v[k] for k in ("max_rel_err", "worst", "rows", "tensors", "score_rms", "loss_f64", "seconds",
                             "peak_hbm_gb_incl_checker", "activation_derivative")})
    return v


def _check(v):
    bad = {k: e for k, e in v["per_tensor"].items() if e > v["tolerance"]}
    assert not bad, bad
    assert v["ok"] and v["max_rel_err"] <= 1e-5
    assert v["rows"] >= 3 * (v["n_user"] + v["n_item"]) and v["tensors"] >= 29
    # the loss must DEPEND on the scores (round 2 printed 0.5 var(y) with scores ~ 1e-6)
    assert 0.3 < v["score_rms"] < 3.0 and abs(v["loss_f64"] - 0.5) > 0.05
    # LeakyReLU' is adopted from the network only inside fp32 rounding of zero: a vanishing share of the elements
    ad = v["activation_derivative"]
    assert ad["ambiguous_act_elements"] <= 1e-3 * ad["act_elements"]


def test_ml10m_bench_network_against_float64_definition():
    v = _run("main", 600)
    assert (v["n_user"], v["n_item"], v["levels"]) == (69878, 10677, 10) and v["edges"] >= 10000000
    assert v["rating_head_item_side_parts"] == 8          # the source-partitioned path the bench times
    assert v["deterministic"]                              # no atomics on the path: the loss repeats bit for bit
    _check(v)


def test_config5_shard_bench_network_against_float64_definition():
    """1.25 M users x 1 M items, 125 M ratings, 16 levels, dim 256: the exact `hbm_bound` leg of bench.py."""
    v = _run("hbm", 900)
    assert v["edges"] >= 125000000 and v["levels"] == 16
    assert v["peak_hbm_gb_incl_checker"] < 200             # far from the 288 GB of the device
    _check(v)
