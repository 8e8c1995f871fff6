"""bench.py as the driver runs it: `python bench.py --gpus N` must start N ranks ITSELF (no torchrun) and print one
JSON line with n_gpus = N whose loss equals the N = 1 run's.  On the 1-GPU test box the two ranks share cuda:0 and talk
over gloo (SG_BENCH_BACKEND=gloo); the RCCL flavour of the same code path runs with one rank (SG_BENCH_FORCE_DIST=1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--shape", "ml-100k", "--dim", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-hbm-leg",
          "--no-ceiling"]


def _bench(extra, env_extra):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + extra, env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_and_matches_single_rank():
    one = _bench([], {})
    two = _bench(["--gpus", "2"], {"SG_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["collectives"]["rccl_ranks"] == 2 and two["collectives"]["backend"] == "gloo"
    assert two["collectives"]["calls_per_step"] >= 5 and two["collectives"]["allreduce_bytes_per_step"] > 0
    assert len(two["config"]["edges_per_rank"]) == 2 and sum(two["config"]["edges_per_rank"]) == one["config"]["edges_per_rank"][0]
    l1, l2 = one["config"]["loss"], two["config"]["loss"]
    assert abs(l1 - l2) <= 1e-5 * max(1.0, abs(l1)), (l1, l2)
    assert one["metric"] == two["metric"] and one["roofline"] is not None


def test_bench_rccl_code_path_with_one_rank():
    """nccl (= RCCL) backend: process-group init, communication stream, async launch / wait, barriers."""
    one = _bench([], {})
    forced = _bench([], {"SG_BENCH_FORCE_DIST": "1"})
    assert forced["collectives"]["backend"] == "nccl" and forced["collectives"]["rccl_ranks"] == 1
    assert forced["collectives"]["calls_per_step"] >= 5
    assert abs(one["config"]["loss"] - forced["config"]["loss"]) <= 1e-6 * max(1.0, abs(one["config"]["loss"]))


def test_config5_mode_runs_as_user_blocks_over_two_ranks():
    """`--shape config5`: every rank generates and plans ONE user block on the device, item degrees are summed over the
    ranks, item-side partials are all-reduced; weak scaling (per-rank work fixed).  Small blocks here; the default
    block is the 1.25 M x 1 M x 125 M shard."""
    flags = ["--shape", "config5", "--hbm-shape", "3000,2000,60000,4", "--dim", "64", "--steps", "2", "--warmup", "1"]

    def run(extra, env_extra):
        env = dict(os.environ, **env_extra)
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + flags + extra, env=env, cwd=ROOT,
                             capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        return json.loads(lines[0])

    one = run([], {})
    two = run(["--gpus", "2"], {"SG_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and one["scaling"] == two["scaling"] == "weak"
    assert len(two["config"]["edges_per_rank"]) == 2 and all(e >= 60000 for e in two["config"]["edges_per_rank"])
    assert two["config"]["edges_per_rank"][0] == one["config"]["edges_per_rank"][0]      # rank 0's block is the N = 1 block
    assert two["collectives"]["rccl_ranks"] == 2 and two["collectives"]["allreduce_bytes_per_step"] > 0
    for r in (one, two):
        assert r["roofline"]["bound"] == "hbm" and 0 < r["config"]["loss"] < 10 and r["value"] > 0


def test_bench_under_torch_distributed_run():
    """The driver's N > 1 launch line: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` (ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env)."""
    env = dict(os.environ, SG_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29713", os.path.join(ROOT, "bench.py"), "--gpus", "2"] + COMMON
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 only
    two = json.loads(lines[0])
    one = _bench([], {})
    assert two["n_gpus"] == 2 and two["collectives"]["rccl_ranks"] == 2
    assert abs(one["config"]["loss"] - two["config"]["loss"]) <= 1e-5 * max(1.0, abs(one["config"]["loss"]))
