"""The two workloads bench.py TIMES, checked at the size it times them: the 2-layer network with output
Dense, rating head through sg_pair_l2_hip (incl. the source-partitioned `parts == 8` path of the full-batch head) and
the full backward -- built by bench.py's own case builders -- against the float64 evaluation of the network's
DEFINITION over the whole graph (tools/f64_check.py; pinned against autograd in tests/test_f64_checker.py).

Compared: loss, every output row of both layers and node types, both rating projections, every embedding-gradient row,
every weight and bias gradient.  Tolerance: 1e-5 of each tensor's scale (north star)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _check(v, n_rows_min):
    bad = {k: e for k, e in v["per_tensor"].items() if e > v["tolerance"]}
    assert not bad, bad
    assert v["ok"] and v["max_rel_err"] <= 1e-5
    assert v["rows"] >= n_rows_min and v["tensors"] >= 29
    # the loss must DEPEND on the scores (round 2 printed 0.5 var(y) with scores ~ 1e-6)
    assert 0.3 < v["score_rms"] < 3.0 and abs(v["loss_f64"] - 0.5) > 0.05


def test_ml10m_bench_network_against_float64_definition():
    import bench
    dev = torch.device("cuda", 0)
    c = bench.main_case("ml-10m", 256, "auto", dev)
    pp = c.plan["idx"][0]["pair"]
    assert pp.item_side_partition(64) is not None and pp.item_side_partition(64).parts == 8     # the path the bench times
    v = bench.verify_leg(c.net, c.step, (c.dgraph.ind_ptr, c.dgraph.end_points, c.dgraph.level, c.n_item, c.R, None),
                         c.y, 1.0 / c.E_total)
    print(v)
    _check(v, 3 * (c.n_user + c.n_item))
    again = float(c.step().detach())
    assert again == float(c.step().detach())                      # deterministic: no atomics on the path


def test_config5_shard_bench_network_against_float64_definition():
    """1.25 M users x 1 M items, 125 M ratings, 16 levels, dim 256: the exact `hbm_bound` leg of bench.py."""
    import bench
    dev = torch.device("cuda", 0)
    c = bench.hbm_case("1250000,1000000,125000000,16", 256, "auto", dev)
    assert c.E >= 125000000 and c.R == 16
    v = bench.verify_leg(c.net, c.step, (c.dg.ind_ptr, c.dg.end_points, c.dg.level, c.ni, c.R, None), c.y, 1.0 / c.E)
    print(v)
    _check(v, 3 * (c.nu + c.ni))
    del c
    torch.cuda.empty_cache()
