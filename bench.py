#!/usr/bin/env python3
"""Benchmark of the STAR-GCN hot path on MI355X: edges/sec for forward+backward of a 2-layer multi-link GCN on a
MovieLens-10M-SHAPED synthetic bipartite graph (BASELINE.json metric; workload definition SURVEY.md section 8d).

  python bench.py [--gpus N --steps K --warmup W]

N > 1: `python bench.py --gpus N` starts the N ranks itself (one process per GPU, RCCL); it is equally happy to be
started as a rank by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` (RANK / WORLD_SIZE in
the environment).  Rank 0 prints ONE JSON line.

One step = embedding gather -> 2 stacked HeterGCNLayers (both node types, all rating levels, full neighbourhood)
-> rating head over ALL ratings -> loss -> full backward to every parameter and the embedding tables.  The graph is
uploaded once and every plan is built ON the device (csrc/plan_build.hip) outside the timed region.

Legs of the default (N = 1) run, all in the same process and reported in the same JSON line:
  value / ms_per_step   the ML-10M-shaped step (BASELINE config 4 on one GPU)
  roofline              dominant kernel (seg_gather_kernel) of that step.  At this shape every gathered matrix sits in
                        the 256 MB Infinity Cache, so the bound is NOT HBM: per launch class, bytes x (hit / 34.5 TB/s of
                        L2 [guide] + (1 - hit) / 7.4 TB/s of Infinity Cache [profiles/r4_mall_sweep.txt]) with the class's
                        L2 hit rate from the committed PMC passes; `frac_vs_l2_peak` beside it is unconditional
  hbm_bound             the one-GPU shard of BASELINE config 5 (1.25 M users x 1 M items, 125 M ratings, 16 levels,
                        dim 256; built on the device): every gathered matrix is 1-16 GB, the gather is HBM-bound and is
                        priced against the 8 TB/s HBM peak
  cpu_baseline          the oracle port of the reference CPU path on this host's physical cores
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _physical_cores():
    cores, phys, core = set(), None, None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        cores.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    return len(cores) or (os.cpu_count() or 1)


def _cpu_quota():
    """CPUs this container may use at once (cgroup v2 cpu.max / v1 cfs quota), None when unlimited"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
        return None if q == "max" else max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = int(f.read())
        return None if q <= 0 else max(1, q // per)
    except (OSError, ValueError):
        return None


def _host_threads():
    """one thread per physical core, capped by the container's CPU quota and the affinity mask (more threads than
    schedulable CPUs only adds throttling: the round-1 baseline ran 256 threads under a 16-CPU quota)"""
    n = _physical_cores()
    q = _cpu_quota()
    if q:
        n = min(n, q)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return max(1, n)


# NOTE: this module must not touch os.environ's OpenMP settings at import.  Rounds 1-3 set OMP_NUM_THREADS / OMP_PROC_BIND=close /
# OMP_PLACES=cores here for the CPU baseline; `import bench` inside a long-lived process (round 3's in-process verification
# test) then changed the environment UNDER the already running interpreter, the library's own OpenMP runtime (LLVM libomp,
# initialised lazily by the first host builder call) picked the binding up, pinned the calling -- main -- thread to core 0,
# and every thread created afterwards inherited that one-core mask: the "unexplained freeze" of DESIGN section 5 (round 3).
# The CPU baseline now runs in its own process, which alone gets the pinned OpenMP environment (cpu_baseline()).

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK = 8.0e12  # bytes/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
BW_L2 = 34.5e12    # bytes/s, aggregate L2 (same guide, "L2 (per XCD)")
BW_MALL = 7.4e12   # bytes/s, Infinity Cache through the vector L1: profiles/r4_mall_sweep.txt (the guide gives no figure)
METRIC = "edges/sec (fwd+bwd) 2-layer multi-link GCN, ML-10M shape, 1/2/4/8 GPU + %HBM roofline"
U, I = "user", "movie"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--shape", default="ml-10m")
    p.add_argument("--dim", type=int, default=256)
    p.add_argument("--order", default="auto", choices=["auto", "transform_first", "aggregate_first", "fused"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-hbm-leg", action="store_true")
    p.add_argument("--no-ceiling", action="store_true")
    p.add_argument("--no-minibatch-leg", action="store_true", help="skip the ML-1M mini-batch training-iteration leg")
    p.add_argument("--no-verify", action="store_true", help="skip the float64 verification of both legs (profiling runs)")
    p.add_argument("--no-partition-check", action="store_true",
                   help="N > 1: skip `partition_check` (rank 0 recomputes the step unpartitioned and compares loss and the "
                        "all-reduced gradients)")
    p.add_argument("--graph-replay", action="store_true",
                   help="also time the step as ONE hipGraph replay (secondary figure `graph_replay`; opt-in: stream capture of "
                        "an autograd step depends on the torch build)")
    p.add_argument("--hbm-only", action="store_true", help="run only the HBM-bound leg (profiling)")
    p.add_argument("--verify-only", choices=["main", "hbm"], default=None,
                   help="build one leg's workload exactly as the benchmark does, run ONE step and print its float64 "
                        "verification as JSON (tests/test_gpu_bench_verify.py runs this in a subprocess)")
    p.add_argument("--hbm-steps", type=int, default=3)
    p.add_argument("--hbm-shape", default="1250000,1000000,125000000,16",
                   help="n_user,n_item,n_edges,n_levels of the HBM-bound leg (default: 1-GPU shard of BASELINE config 5)")
    p.add_argument("--cpu-sample-users", type=float, default=1.0, help="share of users in the CPU-baseline sample")
    p.add_argument("--cpu-baseline-only", action="store_true",
                   help="(internal) run only the CPU baseline leg and print its JSON: the process the default run starts with "
                        "the pinned OpenMP environment")
    return p.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves
# ---------------------------------------------------------------------------------------------------------------------
def launch_ranks(args, script=None, argv=None):
    """Start one process per rank and wait for all of them; `script` / `argv` default to this file and its own command line
    (tests/test_abi_and_host.py starts a stand-in script to check the fail-fast behaviour without a GPU)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SG_BENCH_SELF_LAUNCHED="1")
        env.pop("OMP_PROC_BIND", None)
        env.pop("OMP_PLACES", None)
        env["OMP_NUM_THREADS"] = str(max(1, _host_threads() // max(args.gpus, 1)))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv)), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        # poll ALL ranks: a rank that dies while the others sit in a collective must end the run at once, whichever
        # rank it is (waiting on rank 0 first would block until the RCCL watchdog fires)
        live = list(procs)
        while live:
            for q in list(live):
                code = q.poll()
                if code is None:
                    continue
                live.remove(q)
                if code != 0 and rc == 0:
                    rc = code
                    sys.stderr.write("bench: rank %d exited with code %d; stopping the other ranks\n" % (procs.index(q), code))
                    for o in live:
                        o.terminate()
            if live:
                time.sleep(0.05)
        if rc != 0:
            deadline = time.time() + 10.0
            for q in procs:
                try:
                    q.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    pass
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


# ---------------------------------------------------------------------------------------------------------------------
def build_net(graph_like, D, order, dev, part=None):
    import star_gcn_amd.model as M
    net = M.Net(graph_like, U, I, embed_units=D, agg_units=(D, D), out_units=(D, D), nblocks=1, use_dae=False,
                activation="leaky", dropout=0.0, agg_accum="sum", agg_order=order).to(dev)
    if part is not None:
        for enc in net.encoders:
            for layer in enc._blocks:
                layer.partition = part
        net.pair_partition = part
    return net


VERIFY_TOL = 1e-5      # north star: "within 1e-5 fp32 on embeddings"; max |err| relative to each tensor's scale


def prepare_net(net, step, forward, seed, rows, reduce=None, scores=None):
    """Materialise the lazily-shaped parameters (one step), re-draw every parameter by NAME (identical replicated
    parameters on every rank, equal to the N = 1 model's), then calibrate the layer scales so that activations, scores
    and therefore the loss are O(1) and depend on what the network computes (model.calibrate_output_scale)."""
    import star_gcn_amd.model as M
    step()
    M.deterministic_init(net, seed, rows)
    return M.calibrate_output_scale(net, forward, reduce=reduce, run_scores=scores)


def verify_leg(net, step, arrays, y, scale):
    """One extra step with capture hooks, compared tensor by tensor with the float64 evaluation of the network's
    DEFINITION over the whole graph (tools/f64_check.py: plain torch float64 on the device, none of the product's
    kernels or plans).  Outside every timed region."""
    from tools import f64_check as FC
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = FC.verify_step(net, step, arrays, y, scale, U, I)
    torch.cuda.synchronize()
    out["seconds"] = round(time.perf_counter() - t0, 2)
    out["peak_hbm_gb_incl_checker"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)
    out["tolerance"] = VERIFY_TOL
    out["ok"] = bool(out["max_rel_err"] <= VERIFY_TOL)
    out["method"] = ("float64 evaluation of the definition (aggregators.py:141-160, layers.py:147-187, STAR-GCN.py:428-438) "
                     "over the whole graph, forward and hand-written backward, by tools/f64_check.py; every layer output "
                     "row of both node types, both rating projections, every embedding-gradient row and every weight / bias "
                     "gradient compared; error = max |fp32 - fp64| / max |fp64| per tensor")
    return out


def verify_leg_exact_fp32(net, step, arrays, y, scale):
    """`verify_leg` with every dense product forced onto the EXACT fp32 MFMA kernel (sg_gemm_backend(0)): per tensor, what
    the three-f16-MFMA emulation of the default backend costs in accuracy is the difference between the two blocks."""
    from star_gcn_amd import _lib as L
    lib = L.lib()
    lib.sg_gemm_backend(0)
    try:
        out = verify_leg(net, step, arrays, y, scale)
    finally:
        lib.sg_gemm_backend(-1)
    out["dense_backend"] = "exact fp32 MFMA (sg_gemm_backend(0), v_mfma_f32_32x32x2_f32); aggregation kernels unchanged"
    return out


def exact_fp32_step_ms(step, steps):
    """ms per step with the exact-fp32 MFMA GEMM backend forced (one warm-up step, `steps` timed, HIP events)."""
    from star_gcn_amd import _lib as L
    lib = L.lib()
    lib.sg_gemm_backend(0)
    try:
        step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / steps
    finally:
        lib.sg_gemm_backend(-1)


def timed_steps(step, steps, warmup, dev, dist_on):
    import torch.distributed as dist
    import star_gcn_amd.dist as SD
    import star_gcn_amd.ops as ops
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    ops.gather_profile(True)      # HIP events around every gather launch, on the launch stream, inside the library
    ops.fused_profile(True)       # ... and around every fused aggregate -> contract launch (csrc/agg_fused.hip)
    SD.STATS.reset()
    SD.STATS.enabled = dist_on
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]     # step boundaries on the compute stream
    t0 = time.perf_counter()
    loss = None
    marks[0].record()
    for k in range(steps):
        loss = step()
        marks[k + 1].record()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.gather_profile(False)
    ops.fused_profile(False)
    timed_steps.last_fused = ops.fused_profile_read()
    SD.STATS.enabled = False
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(steps))
    timed_steps.last_median_ms = per_step[len(per_step) // 2] if steps % 2 else 0.5 * (per_step[steps // 2 - 1] + per_step[steps // 2])
    timed_steps.last_min_ms, timed_steps.last_max_ms = per_step[0], per_step[-1]
    return elapsed, loss, ops.gather_profile_read(with_src_bytes=True)


def graph_replay(step, dev, steps):
    """Capture one fwd+bwd step into a hipGraph and time `steps` replays: the step without launch gaps / host work."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    import star_gcn_amd.dist as sgdist
    sgdist.quiesce_for_capture(dev)     # the RCCL watchdog must have retired the warm-up collectives before capture begins
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode=sgdist.capture_error_mode()):      # 'thread_local' with RCCL: see dist.py
        loss = step()
    torch.cuda.synchronize()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_step": ms, "loss": float(loss.detach()),
            "note": "same step, one hipGraphLaunch per step; no HIP events inside, so not the headline"}


MFMA_F16_PEAK = 2.5e15      # dense f16 / bf16 MFMA peak of MI355X (/opt/skills/guides/MI355X_MICROARCH.md)


def dense_roofline(step, steps=3):
    """Secondary roofline of the dense mix (per-rating-level contraction + Dense layers), measured in `steps` EXTRA steps
    outside the timed region (one HIP-event pair per GEMM call would cost the headline 0.3 ms per step): every
    sg_gemm_f32_hip call bracketed on its stream inside the library, conversion passes and split-K reduce included.
    peak = the f16 MFMA peak / 3: the default backend forms an fp32-accurate product from three f16 matrix instructions."""
    import star_gcn_amd.ops as ops
    step()
    torch.cuda.synchronize()
    ops.gemm_profile(True)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ops.gemm_profile(False)
    recs = [r for r in ops.gemm_profile_read() if r[0] > 0]
    if not recs:
        return None
    t = sum(r[0] for r in recs)
    flops = sum(2.0 * r[1] * r[2] * r[3] for r in recs)
    big = sorted(recs, key=lambda r: -r[0])[:len(recs) // steps or 1]
    by_backend = dict()
    for r in recs:
        b = by_backend.setdefault({0: "fp32", 1: "bf16x6", 2: "x6v2", 3: "f16x3"}.get(r[4], str(r[4])), [0.0, 0.0])
        b[0] += r[0]
        b[1] += 2.0 * r[1] * r[2] * r[3]
    return {"kernel": "sg_gemm_f32_hip (default backend f16x3: three f16 MFMAs per fp32-accurate product)", "bound": "mfma",
            "achieved": flops / t / 1e12, "peak": MFMA_F16_PEAK / 3 / 1e12, "unit": "TFLOP/s", "frac": flops / t / (MFMA_F16_PEAK / 3),
            "gemm_ms_per_step": t / steps * 1e3, "gemm_calls_per_step": len(recs) / steps, "gflop_per_step": flops / steps / 1e9,
            "by_backend": {k: {"ms_per_step": v[0] / steps * 1e3, "tflops": (v[1] / v[0] / 1e12) if v[0] else None}
                           for k, v in by_backend.items()},
            "slowest_calls": [{"M": r[1], "N": r[2], "K": r[3], "ms": r[0] * 1e3,
                               "tflops": 2.0 * r[1] * r[2] * r[3] / r[0] / 1e12} for r in big[:6]],
            "note": "extra steps after the timed region; flops = 2 M N K of every GEMM call, time incl. operand conversion "
                    "and split-K reduction"}


def gather_roofline(timeline, E_local, D, steps):
    """HIP-event times of the aggregation launches (width D) -> algorithmic rate.  An aggregation over all local edges is
    ONE launch, or TWO source-range phases of one launch each whose edge counts add up to E_local; a launch's
    algorithmic bytes are (8 + 4 D) x ITS edges (idx + support + one fp32 row per edge visit)."""
    E = max(E_local, 1)
    agg = [(t, nnz, sb) for t, nnz, C, sb in timeline if C == D and E // 64 <= nnz <= E and t > 0]
    edges = sum(n for _, n, _ in agg)
    if not agg or edges % E != 0:                     # something else of width D ran: do not guess
        return None
    total_t = sum(t for t, _, _ in agg)
    classes = dict()                                  # launches by (footprint of the gathered matrix, phased or not)
    for t, n, sb in agg:
        c = classes.setdefault((sb, n < E), [0, 0.0, 0])
        c[0] += 1
        c[1] += t
        c[2] += n
    return {"kernel": "seg_gather_kernel", "achieved": (8 + 4 * D) * edges / total_t / 1e9, "unit": "GB/s",
            "launches_per_step": len(agg) / steps, "aggregations_per_step": edges / E / steps,
            "avg_launch_ms": total_t / len(agg) * 1e3, "avg_aggregation_ms": total_t / (edges / E) * 1e3,
            "algorithmic_bytes_per_launch": (8 + 4 * D) * edges / len(agg),
            "algorithmic_bytes_per_aggregation": (8 + 4 * D) * E,
            "_classes": {k: (n, tt / n, ne / n) for k, (n, tt, ne) in classes.items()}}


def fused_saved_rows(nu, ni):
    """Rows of the R-expanded matrix a fused launch saves for the weight gradient (multilink.hip, fused_saves_z): by default
    always the smaller node side's (the forward saves Z where it aggregates INTO the smaller side, the data gradient saves dH
    where its transposed launch does); with SG_FUSED_SAVEZ=0/1 forced, one launch of each size -> the mean."""
    return min(nu, ni) if os.environ.get("SG_FUSED_SAVEZ") not in ("0", "1") else (nu + ni) / 2


def fused_roofline(records, E_local, D, steps, rows_saved_bytes):
    """HIP-event times of the fused aggregate -> contract launches (csrc/agg_fused.hip) -> algorithmic rate.  One launch is
    one whole aggregation (every edge visited once: idx + support + one fp32 row = 8 + 4 D bytes) AND its
    per-level contraction; the launches of the backward also write the fp32 aggregates the weight gradient contracts with
    (`rows_saved_bytes` per launch, counted separately -- `achieved` uses the per-edge figure only, as in earlier rounds)."""
    recs = [(t, n, z) for t, n, z in records if t > 0 and n == E_local]
    if not recs:
        return None
    total_t = sum(t for t, _, _ in recs)
    b_edge = (8 + 4 * D) * E_local
    out = {"kernel": "agg_contract_kernel (aggregation + per-level contraction in one launch)", "unit": "GB/s",
           "achieved": b_edge * len(recs) / total_t / 1e9, "launches_per_step": len(recs) / steps,
           "aggregations_per_step": len(recs) / steps, "avg_launch_ms": total_t / len(recs) * 1e3,
           "avg_aggregation_ms": total_t / len(recs) * 1e3, "algorithmic_bytes_per_launch": b_edge,
           "algorithmic_bytes_per_aggregation": b_edge, "per_class": [],
           "note": "a launch is a whole aggregation INCLUDING its per-level contraction (rounds 1-4 timed the gather alone and ran "
                   "the contraction as a separate GEMM that re-read a 16 GB intermediate): the fraction is of the same 8 TB/s peak "
                   "on the same edges x (8 + 4 D) bytes, so it is lower than the gather-only 0.78 while the step is shorter; "
                   "SG_FUSED=0 runs the unfused pair"}
    for z, name in ((0, "launches that write only their 256-wide output"),
                    (1, "launches that also write the fp32 aggregates (the forward where the destination side is the smaller one, "
                        "else the data gradient)")):
        c = [t for t, _, zz in recs if zz == z]
        if c:
            extra = rows_saved_bytes if z else 0
            out["per_class"].append({"launch": name, "launches_per_step": len(c) / steps, "avg_launch_ms": sum(c) / len(c) * 1e3,
                                     "achieved_gbs": b_edge * len(c) / sum(c) / 1e9,
                                     "achieved_gbs_incl_saved_aggregates": (b_edge + extra) * len(c) / sum(c) / 1e9,
                                     "saved_aggregate_bytes_per_launch": extra})
    return out


def measure_stream_ceiling(dev, n_bytes, workgroups, bursts=256, strided=False):
    """streaming read of a resident buffer of n_bytes with the gather's launch geometry (one wave per workgroup, 256 row
    reads of 1 KiB per wave, 4 in flight) -> GB/s, median of 5 launches after a warming one, HIP events on the current stream.
    strided=True (sg_stream_read_strided_hip, stride = grid): no two resident waves ask for the same burst -- the clean rate of
    the level that holds the buffer (tools/mall_sweep.py).  strided=False is round 3's wrapped form, whose result depends on
    whether (buffer / 256 KiB) is a multiple of 8 (sibling waves on one XCD hit L2): kept for tools/ceiling_sweep.py only."""
    from star_gcn_amd import _lib as L
    lib = L.lib()
    buf = torch.empty(n_bytes // 4, dtype=torch.float32, device=dev).normal_()
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    st = L.stream_ptr()

    def launch():
        if strided:
            L.check(lib.sg_stream_read_strided_hip(L.ptr(buf), n_bytes, bursts, workgroups, workgroups, L.ptr(sink), st),
                    "sg_stream_read_strided_hip")
        else:
            L.check(lib.sg_stream_read_hip(L.ptr(buf), n_bytes, bursts, workgroups, L.ptr(sink), st), "sg_stream_read_hip")
    launch()
    rates = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        e1.synchronize()
        rates.append(workgroups * bursts * 1024 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    rates.sort()
    return rates[len(rates) // 2]


def _sha16(path):
    import hashlib
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        return None


def profile_record(name, source_file="seg_gather.hip"):
    """PMC traffic of the gather (profiles/pmc_traffic.json, reduced from committed rocprofv3 --pmc passes).  The record
    carries the sha of the kernel source it was measured on; when csrc/seg_gather.hip has changed since, the counters
    describe another kernel and `traffic` is reported as null instead of going stale silently."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get(name)
    except (OSError, ValueError):
        return None
    if not rec:
        return None
    now = _sha16(os.path.join(ROOT, "star-gcn_amd", "csrc", source_file))
    if rec.get("kernel_source_sha16") != now:
        return {"stale": True, "source": "%s measured on %s %s, current source is %s: traffic withheld" %
                (rec.get("source", "profiles/pmc_traffic.json"), source_file, rec.get("kernel_source_sha16"), now)}
    return rec


# ---------------------------------------------------------------------------------------------------------------------
def hbm_case(hbm_shape, D, order, dev):
    """BASELINE config 5 on ONE GPU of the 8: 1.25 M users x 1 M items, >= 125 M ratings, 16 levels, dim 256.  Graph,
    degrees, support, transposed CSR and both multi-link plans are generated / built on the device; the network is
    initialised by parameter name and scale-calibrated.  Shared by the benchmark leg and tests/test_gpu_bench_verify.py."""
    import types
    from star_gcn_amd.device_graph import synthetic_device_graph
    nu, ni, ne, R = (int(x) for x in hbm_shape.split(","))
    t0 = time.perf_counter()
    dg = synthetic_device_graph(nu, ni, ne, R, dev, seed=5)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    torch.manual_seed(4321)
    net = build_net(dg, D, order, dev)
    plan = net.make_plan_device(dg)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0
    vals = dg.values()
    y = ((vals - vals.mean()) / vals.std()).contiguous()
    del vals
    E = dg.nnz

    def step():
        net.zero_grad(set_to_none=True)
        losses, _, _ = net.run(plan, rating_targets=y, rating_scale=1.0 / E)
        loss = losses[0]
        loss.backward()
        return loss

    calib = prepare_net(net, step, lambda: net.run(plan, rating_targets=y, rating_scale=1.0 / E), 4321,
                        {U: (0, nu, nu), I: (0, ni, ni)}, scores=lambda: net.run(plan)[0][0])
    return types.SimpleNamespace(nu=nu, ni=ni, R=R, D=D, dg=dg, net=net, plan=plan, y=y, E=E, step=step, calib=calib,
                                 t_gen=t_gen, t_plan=t_plan)


def hbm_leg(args, dev):
    c = hbm_case(args.hbm_shape, args.dim, args.order, dev)
    nu, ni, R, D, dg, net, plan, y, E, step, calib, t_gen, t_plan = (c.nu, c.ni, c.R, c.D, c.dg, c.net, c.plan, c.y, c.E,
                                                                       c.step, c.calib, c.t_gen, c.t_plan)
    del c
    elapsed, loss, timeline = timed_steps(step, args.hbm_steps, 1, dev, False)
    roof = gather_roofline(timeline, E, D, args.hbm_steps)
    src_small = min(nu, ni) * D * 4
    out = {"workload": "1-GPU shard of BASELINE config 5 (10 M x 1 M nodes / 1 B edges over 8 GPUs): %d users x %d items, "
                       "%d ratings, %d rating levels, dim %d; same 2-layer network, fwd+bwd; graph generated and planned "
                       "on the device" % (nu, ni, E, R, D),
           "steps": args.hbm_steps, "warmup": 1, "ms_per_step": elapsed / args.hbm_steps * 1e3,
           "ms_per_step_median_events": timed_steps.last_median_ms,
           "edges_per_s": E / (elapsed / args.hbm_steps), "loss": float(loss.detach()),
           "graph_gen_s": round(t_gen, 2), "plan_build_s": round(t_plan, 2),
           "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
           "smallest_gathered_matrix_mb": src_small // 2 ** 20}
    froof = fused_roofline(getattr(timed_steps, "last_fused", []), E, D, args.hbm_steps, fused_saved_rows(nu, ni) * R * D * 4)
    if froof:       # the step ran the fused order (the default at this size): that kernel is the dominant one
        rec = profile_record("hbm-config5-shard-fused:%d" % D, "agg_fused.hip")
        live = rec and not rec.get("stale")
        froof.update(bound="hbm", peak=HBM_PEAK / 1e9, frac=froof["achieved"] * 1e9 / HBM_PEAK,
                     traffic=(rec["traffic_bytes_per_launch_mean"] * (E / rec["edges_per_launch"]) if live else None),
                     traffic_source=(rec.get("source") if rec else None))
        if roof:      # aggregations that still ran as gather + GEMM (none at the default routing)
            roof.pop("_classes", None)
            froof["unfused_gather_launches"] = roof
        out["roofline"] = froof
        out["step_roofline_frac"] = out["edges_per_s"] * 8 * (8 + 4 * D) / HBM_PEAK
    elif roof:
        rec = profile_record("hbm-config5-shard:%d" % D)
        live = rec and not rec.get("stale")
        roof.update(bound="hbm", peak=HBM_PEAK / 1e9, frac=roof["achieved"] * 1e9 / HBM_PEAK,
                    traffic=(rec["traffic_bytes_per_launch_mean"] * (E / rec["edges_per_launch"]) if live else None),
                    traffic_source=(rec.get("source") if rec else None))
        roof.pop("_classes", None)
        out["roofline"] = roof
        out["step_roofline_frac"] = out["edges_per_s"] * 8 * (8 + 4 * D) / HBM_PEAK
    loss = None

    def sub(fn, *a):      # a secondary measurement must not cost the primary timings of this leg
        try:
            return fn(*a)
        except Exception as e:
            return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    out["dense_roofline"] = sub(dense_roofline, step, 2)
    out["ms_per_step_exact_fp32_gemm"] = sub(exact_fp32_step_ms, step, 2)     # every dense product on the exact fp32 MFMA kernel
    if not args.no_verify:
        out["verify"] = sub(verify_leg, net, step, (dg.ind_ptr, dg.end_points, dg.level, ni, R, None), y, 1.0 / E)
        out["verify_exact_fp32"] = sub(verify_leg_exact_fp32, net, step, (dg.ind_ptr, dg.end_points, dg.level, ni, R, None), y,
                                       1.0 / E)
    out["init"] = "embeddings U(-0.1, 0.1), Xavier-in weights, zero biases, then layer-sequential scale calibration " \
                  "(model.calibrate_output_scale); pre-calibration rms per stage: %s" % json.dumps(
                      [{k: float("%.3g" % v) for k, v in st.items()} for st in calib])
    del net, plan, dg, y, step
    torch.cuda.empty_cache()
    return out


# ---------------------------------------------------------------------------------------------------------------------
def run_config5(args, dev, dist_on, world, rank, backend):
    """`--shape config5`: BASELINE config 5 as it is meant to run -- N ranks, each holding ONE user block of --hbm-shape
    (default: 1.25 M users x all 1 M items, 125 M ratings, 16 levels; 8 ranks = the 10 M x 1 M / 1 B-edge graph), the
    item side replicated, item-side partials all-reduced over RCCL.  Weak scaling: per-GPU work is fixed.  Every rank
    generates and plans its block on its own device; the item degrees of the support are summed over the ranks."""
    import torch.distributed as dist
    import star_gcn_amd.dist as SD
    import star_gcn_amd.functional as SF
    import star_gcn_amd.model as M
    from star_gcn_amd.device_graph import synthetic_device_graph
    nu, ni, ne, R = (int(x) for x in args.hbm_shape.split(","))
    D = args.dim
    t0 = time.perf_counter()
    dg = synthetic_device_graph(nu, ni, ne, R, dev, seed=5 + rank, item_seed=5)
    E_local = dg.nnz
    vals = dg.values()
    stats = torch.stack([vals.double().sum(), (vals.double() ** 2).sum(),
                         torch.tensor(float(E_local), dtype=torch.float64, device=dev)])
    if dist_on:
        deg = dg.item_degrees.clone()
        if backend == "gloo":
            deg_h, stats_h = deg.cpu(), stats.cpu()
            dist.all_reduce(deg_h)
            dist.all_reduce(stats_h)
            deg, stats = deg_h.to(dev), stats_h.to(dev)
        else:
            dist.all_reduce(deg)
            dist.all_reduce(stats)
        dg = dg.with_item_degrees(deg)
    s1, s2, n = (float(x) for x in stats.tolist())
    E_total = int(round(n))
    mean = s1 / n
    std = max((s2 / n - mean * mean) * n / max(n - 1, 1), 1e-12) ** 0.5
    y = ((vals - mean) / std).contiguous()
    del vals
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    part = SD.NodePartition([U], [I]) if dist_on else None
    torch.manual_seed(1234)
    net = build_net(dg, D, args.order, dev, part)
    plan = net.make_plan_device(dg)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t0

    def step():
        net.zero_grad(set_to_none=True)
        losses, _, _ = net.run(plan, rating_targets=y, rating_scale=1.0 / E_total)   # = l2_loss(scores, y, 1 / E)
        loss = losses[0]
        loss.backward()
        if dist_on:
            SD.allreduce_grads(net.local_region_parameters())
        return loss

    # replicated parameters (and, for simplicity, the user tables) drawn by parameter NAME: identical on every rank
    prepare_net(net, step, lambda: net.run(plan, rating_targets=y, rating_scale=1.0 / E_total), 1234,
                {U: (0, nu, nu), I: (0, ni, ni)}, reduce=SD.all_reduce_sum if dist_on else None,
                scores=lambda: net.run(plan)[0][0])
    elapsed, loss, timeline = timed_steps(step, args.steps, args.warmup, dev, dist_on)
    comm = SD.STATS.read() if dist_on else None
    rank_ms = [elapsed / args.steps * 1e3]
    loss_total = loss.detach().clone()
    edges_per_rank = [E_local]
    if dist_on:
        tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        loss_total = SD.all_reduce_sum(loss_total.view(1))[0]
        gathered = [None] * world
        dist.all_gather_object(gathered, (E_local, rank_ms[0], comm["exposed_ms"] / args.steps))
        edges_per_rank = [int(e[0]) for e in gathered]
        rank_ms = [float(e[1]) for e in gathered]
        exposed_ms = [float(e[2]) for e in gathered]
    roof = gather_roofline(timeline, E_local, D, args.steps)
    if roof:
        roof.update(bound="hbm", peak=HBM_PEAK / 1e9, frac=roof["achieved"] * 1e9 / HBM_PEAK, traffic=None)
        roof.pop("_classes", None)
    froof = fused_roofline(getattr(timed_steps, "last_fused", []), E_local, D, args.steps, fused_saved_rows(nu, ni) * R * D * 4)
    if froof:      # rank 0's fused aggregate -> contract launches (the default order at this size)
        froof.update(bound="hbm", peak=HBM_PEAK / 1e9, frac=froof["achieved"] * 1e9 / HBM_PEAK, traffic=None)
        if roof:
            froof["unfused_gather_launches"] = roof
        roof = froof
    value = E_total / (elapsed / args.steps)
    out = {"metric": METRIC, "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "BASELINE config 5, %d user block(s) of %d users x %d items, %d ratings per block, %d "
                                  "rating levels, dim %d; same 2-layer network, fwd+bwd; every block generated and planned "
                                  "on its rank's device" % (world, nu, ni, ne, R, D),
                      "partition": "single GPU (one block)" if world == 1 else
                                   "1-D user-block node partition, items replicated, RCCL all-reduce of item-side "
                                   "partials on a side stream",
                      "order": args.order, "graph_gen_s": round(t_gen, 2), "plan_build_s": round(t_plan, 2),
                      "loss": float(loss_total), "edges_per_rank": edges_per_rank,
                      "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)},
           "roofline": roof, "step_roofline_frac": value * 8 * (8 + 4 * D) / (world * HBM_PEAK),
           "cpu_baseline": None}
    if dist_on:
        out["collectives"] = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(),
                              "calls_per_step": comm["calls"] / args.steps,
                              "allreduce_bytes_per_step": comm["bytes"] / args.steps,
                              "collective_ms_per_step": comm["device_ms"] / args.steps,
                              "exposed_ms_per_step_per_rank": exposed_ms}
        out["ms_per_step_per_rank"] = rank_ms
    out["ms_per_step_median_events"] = timed_steps.last_median_ms
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


def main_case(shape, D_, order, dev, dist_on=False, world=1, rank=0):
    """The headline workload, ready to step: graph of `shape` (SURVEY 8(d) recipe), this rank's user block uploaded once,
    plans built on the device, network initialised (by parameter name) and scale-calibrated.  Used by run_rank and by
    tests/test_gpu_bench_verify.py, so the test exercises exactly what the benchmark times."""
    import types
    import star_gcn_amd.dist as SD
    import star_gcn_amd.synthetic as S
    from star_gcn_amd.device_graph import DeviceBipartite
    from star_gcn_amd.mxgraph.graph import HeterGraph
    graph, eu, ei, vals = S.make_graph(shape)
    csr = graph[U, I]
    n_user, n_item, E_total, R = csr.shape[0], csr.shape[1], csr.nnz, int(csr.multi_link.size)
    mean, std = float(vals.mean()), float(vals.std())
    lo, hi = 0, n_user
    # SG_BENCH_EMULATE_WORLD=N (development, one process): run rank 0's share of an N-rank partition through the
    # partitioned code path -- what ONE rank of the N-GPU strong-scaling run computes per step, without the other ranks
    emulate = int(os.environ.get("SG_BENCH_EMULATE_WORLD", "0")) if world == 1 else 0
    if dist_on:
        lo, hi = SD.balanced_row_blocks(csr.ind_ptr, emulate if emulate > 1 else world)[rank]
        sub = S.user_block(graph, U, I, lo, hi)       # this rank's users x ALL items, GLOBAL item degrees for the support
        lgraph = HeterGraph({U: np.arange(hi - lo, dtype=np.int32), I: np.arange(n_item, dtype=np.int32)}, {(U, I): sub})
    else:
        lgraph, sub = graph, csr
    E_local = sub.nnz
    y = torch.from_numpy(((sub.values - mean) / std).astype(np.float32)).to(dev)

    D = D_
    part = SD.NodePartition([U], [I]) if dist_on else None
    torch.manual_seed(1234)
    net = build_net(lgraph, D, order, dev, part)
    t_plan = time.perf_counter()
    dgraph = DeviceBipartite.from_host(lgraph, U, I, dev)     # one upload of the CSR; every plan is built on the device
    plan = net.make_plan_device(dgraph)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t_plan

    def step():
        net.zero_grad(set_to_none=True)
        losses, _, _ = net.run(plan, rating_targets=y, rating_scale=1.0 / E_total)   # = l2_loss(scores, y, 1 / E)
        loss = losses[0]
        loss.backward()
        if dist_on:
            SD.allreduce_grads(net.local_region_parameters())
        return loss

    # one step materialises the lazily-shaped parameters, which are then re-drawn by parameter NAME: identical
    # replicated parameters on every rank, equal to the N = 1 model's, whatever the rank-local row counts and the order
    # of first use (user table = rows [lo, hi) of the global one); then the layer scales are calibrated (global rms)
    calib = prepare_net(net, step, lambda: net.run(plan, rating_targets=y, rating_scale=1.0 / E_total), 1234,
                        {U: (lo, hi, n_user), I: (0, n_item, n_item)}, reduce=SD.all_reduce_sum if dist_on else None,
                        scores=lambda: net.run(plan)[0][0])
    return types.SimpleNamespace(**{k: v for k, v in locals().items() if k not in ("types", "SD", "S", "DeviceBipartite",
                                                                                  "HeterGraph")})


def partition_f64_check(net, step, csr, vals, mean, std, n_user, n_item, R, E_total, lo, hi, dev):
    """One rank's share of `partition_check.f64` (tools/f64_check.verify_step_partitioned): the global graph's arrays are
    taken from the HOST CSR (no product code), the user rows of the other ranks come through zero-padded all-reduces."""
    import star_gcn_amd.dist as SD
    from tools import f64_check as FC
    t0 = time.perf_counter()
    ml = np.asarray(csr.multi_link, dtype=np.float32)
    level = np.searchsorted(ml, np.asarray(csr.values, dtype=np.float32)).astype(np.int32)      # multi_link is sorted, values are its members
    assert np.array_equal(ml[level], np.asarray(csr.values, dtype=np.float32))
    arrays = (torch.from_numpy(np.ascontiguousarray(csr.ind_ptr)).to(dev), torch.from_numpy(np.ascontiguousarray(csr.end_points)).to(dev),
              torch.from_numpy(level).to(dev), n_item, R, None)
    y_all = torch.from_numpy(((vals - mean) / std).astype(np.float32)).to(dev)

    def assemble_rows(block):
        full = torch.zeros((n_user,) + tuple(block.shape[1:]), dtype=block.dtype, device=block.device)
        full[lo:hi] = block
        return SD.all_reduce_sum(full)          # the other ranks' rows arrive as sums with zeros: exact

    turn = None
    import torch.distributed as dist
    if dist.get_backend() != "nccl" and dist.get_world_size() > 1:      # the ranks share ONE GPU (development / test runs over gloo)
        def turn(fn):
            res = None
            for r_ in range(dist.get_world_size()):
                if r_ == dist.get_rank():
                    try:
                        res = fn()
                    except Exception as e:      # keep the turnstile moving; the error is this rank's result
                        res = e
                    torch.cuda.synchronize()
                    torch.cuda.empty_cache()
                dist.barrier()
            if isinstance(res, Exception):
                raise res
            return res
    out = FC.verify_step_partitioned(net, step, arrays, y_all, 1.0 / E_total, lo, hi, n_user, assemble_rows, SD.all_reduce_sum, U, I,
                                     one_at_a_time=turn)
    torch.cuda.synchronize()
    out["seconds"] = round(time.perf_counter() - t0, 2)
    return out


def guarded_all(fn, *a):
    """a collective check must not strand the other ranks: errors become an entry (the failing rank still leaves the group
    through the barrier that follows)"""
    try:
        return fn(*a)
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def run_rank(args):
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # SG_BENCH_FORCE_DIST=1: take the partitioned (N > 1) code path -- RCCL init, barriers, all-reduces, partition
    # crossings -- even with a single rank; development check of that path on a 1-GPU box.
    dist_on = world > 1 or os.environ.get("SG_BENCH_FORCE_DIST") == "1"
    # SG_BENCH_BACKEND=gloo lets several ranks share ONE GPU (development check of the N > 1 code path on a 1-GPU
    # box); the real multi-GPU run uses nccl == RCCL with one rank per GPU.
    backend = os.environ.get("SG_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # a short collective timeout: a benchmark run whose ranks fall out of step (one died, one is stuck) should fail
        # in minutes, not after torch's 10-minute default; the slowest legitimate gap between two collectives is the
        # generation + planning of a config-5 user block (tens of seconds)
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("SG_BENCH_COLLECTIVE_TIMEOUT_S", "300")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    if args.gpus != world and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)

    import star_gcn_amd.dist as SD
    import star_gcn_amd.functional as SF
    import star_gcn_amd.model as M
    import star_gcn_amd.synthetic as S
    from star_gcn_amd.device_graph import DeviceBipartite
    from star_gcn_amd.mxgraph.graph import HeterGraph

    if args.hbm_only:
        out = {"metric": METRIC, "hbm_bound": hbm_leg(args, dev)}
        print(json.dumps(out))
        return
    if args.verify_only:
        if args.verify_only == "main":
            c = main_case(args.shape, args.dim, args.order, dev)
            pp = c.plan["idx"][0]["pair"].item_side_partition(64)
            v = verify_leg(c.net, c.step, (c.dgraph.ind_ptr, c.dgraph.end_points, c.dgraph.level, c.n_item, c.R, None),
                           c.y, 1.0 / c.E_total)
            v.update(n_user=c.n_user, n_item=c.n_item, edges=c.E_total, levels=c.R,
                     rating_head_item_side_parts=(pp.parts if pp is not None else 1))
            a, b = float(c.step().detach()), float(c.step().detach())
            v["deterministic"] = (a == b)
        else:
            c = hbm_case(args.hbm_shape, args.dim, args.order, dev)
            v = verify_leg(c.net, c.step, (c.dg.ind_ptr, c.dg.end_points, c.dg.level, c.ni, c.R, None), c.y, 1.0 / c.E)
            v.update(n_user=c.nu, n_item=c.ni, edges=c.E, levels=c.R)
        print(json.dumps({"verify": v}))
        return
    if args.shape == "config5":
        run_config5(args, dev, dist_on, world, rank, backend)
        return

    c = main_case(args.shape, args.dim, args.order, dev, dist_on, world, rank)
    graph, csr, sub, lgraph, dgraph, net, plan, y, step, calib = (c.graph, c.csr, c.sub, c.lgraph, c.dgraph, c.net,
                                                                  c.plan, c.y, c.step, c.calib)
    n_user, n_item, E_total, E_local, R, D, lo, hi, t_plan = (c.n_user, c.n_item, c.E_total, c.E_local, c.R, c.D, c.lo,
                                                               c.hi, c.t_plan)
    vals, mean, std = c.vals, c.mean, c.std
    del c
    if dist_on and world > 1:      # replicas must agree bit for bit: check once
        skip = "embed_layers._layers.%d." % net.embed_layers._key2idx[U]        # the row-sharded user table
        rep = [p.detach().double().sum() for n_, p in net.named_parameters() if skip not in n_]
        chk = torch.stack(rep)
        mx, mn = chk.clone(), chk.clone()
        if backend == "gloo":
            mx, mn = mx.cpu(), mn.cpu()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        assert torch.equal(mx, mn), "replicated parameters differ between ranks"

    elapsed, loss, timeline = timed_steps(step, args.steps, args.warmup, dev, dist_on)
    comm = SD.STATS.read() if dist_on else None
    rank_ms = [elapsed / args.steps * 1e3]
    if dist_on:
        tmax = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- roofline of the dominant kernel ---------------------------------------------------------------------------
    roof = gather_roofline(timeline, E_local, D, args.steps)
    src_mb = sorted((max(hi - lo, 1) * D * 4 // 2 ** 20, n_item * D * 4 // 2 ** 20,
                     min(hi - lo, n_item) * R * D * 4 // 2 ** 20))
    cache_resident = src_mb[-1] * 2 ** 20 < 200 * 2 ** 20
    if roof:
        roof["hbm_equiv_frac"] = roof["achieved"] * 1e9 / HBM_PEAK     # secondary: algorithmic bytes over the HBM peak
        roof["gathered_matrices_mb"] = src_mb
        rec = profile_record("%s:%d" % (args.shape, D)) if world == 1 else None
        live = rec and not rec.get("stale")
        # the record describes launches of a given structure (single launches or source-range phases): it only applies
        # when this run issued the same number of launches per aggregation
        same = live and abs(rec.get("launches_per_aggregation", 1) - roof["launches_per_step"] / roof["aggregations_per_step"]) < 1e-9
        roof["traffic"] = (rec["traffic_bytes_per_launch_mean"] * (roof["algorithmic_bytes_per_launch"] / (8 + 4 * D)) /
                           rec["edges_per_launch"]) if same else None
        if live and not same:
            roof["traffic_note"] = "PMC record is for %s launch(es) per aggregation; this run issued %.3g" % (
                rec.get("launches_per_aggregation", 1), roof["launches_per_step"] / roof["aggregations_per_step"])
        roof["traffic_source"] = rec.get("source") if rec else None
        if cache_resident and world == 1:
            # Bound of a cache-resident launch class (VERDICT r3 #3 iii): its algorithmic bytes served partly by the XCD-private
            # L2s and otherwise by the Infinity Cache,   t_bound = bytes * (hit / BW_L2 + (1 - hit) / BW_MALL)
            #   BW_L2   = 34.5 TB/s, the guide's aggregate L2 figure (/opt/skills/guides/MI355X_MICROARCH.md, "L2 (per XCD)")
            #   BW_MALL = 7.4 TB/s: the guide has no Infinity-Cache bandwidth, so it comes from ONE committed size sweep
            #             (tools/mall_sweep.py -> profiles/r4_mall_sweep.txt: clean streaming reads through the vector L1
            #             plateau at 7.3-7.5 TB/s from 48 MB to 512 MB -- the L1's outstanding-miss capacity over the
            #             Infinity-Cache latency -- and fall to 6.2 TB/s, the HBM rate, past 2 GB)
            #   hit     = that class's L2 hit rate from the committed PMC passes (TCC_HIT / (TCC_HIT + TCC_MISS); stamped with
            #             the kernel source it was measured on: another source gives hit = None and no `frac`)
            # `frac` is therefore conditional on the hit rate the kernel's own column slicing / source-range phases achieve;
            # `frac_vs_l2_peak` (every byte at BW_L2) is the unconditional figure.  The HBM fraction the metric asks for is
            # `hbm_bound.roofline.frac` (config-5 shard), where HBM does bind.
            hits = (rec or {}).get("l2_hit_rate_by_source_mb") if (live and same) else None

            def hit_for(mb):      # the PMC class with the nearest source footprint (a few % apart: same column slicing)
                if not hits:
                    return None
                key = min(hits, key=lambda k: abs(float(k) - mb))
                return hits[key] if abs(float(key) - mb) <= 0.25 * max(mb, 1) else None
            classes = sorted(roof["_classes"].items())
            # the Infinity-Cache rate of THIS run and box: clean strided read over each class's footprint (falls back to the
            # committed sweep's 7.4 TB/s when the ceiling measurement is switched off)
            mall_run = {}
            if not args.no_ceiling:
                for (sb, phased), (n, t_avg, e_avg) in classes:
                    mall_run[(sb, phased)] = measure_stream_ceiling(dev, max(1 << 20, (sb >> 20) << 20),
                                                                    (int(e_avg) + 255) // 256, strided=True)
            mall_mean = (sum(mall_run.values()) / len(mall_run)) if mall_run else None
            bw_mall = mall_mean * 1e9 if mall_mean else BW_MALL
            t_bound = t_l2 = t_actual = 0.0
            per_class, have_all = [], True
            for (sb, phased), (n, t_avg, e_avg) in classes:
                b_launch = (8 + 4 * D) * e_avg
                hit = hit_for(sb >> 20)
                t_b = b_launch * (hit / BW_L2 + (1.0 - hit) / bw_mall) if hit is not None else None
                have_all = have_all and t_b is not None
                t_bound += n * (t_b or 0.0)
                t_l2 += n * b_launch / BW_L2
                t_actual += n * t_avg
                cls = {"gathered_matrix_mb": sb >> 20, "source_range_phase": bool(phased), "launches_per_step": n / args.steps,
                       "edges_per_launch": e_avg, "avg_launch_ms": t_avg * 1e3, "achieved_gbs": b_launch / t_avg / 1e9,
                       "frac": b_launch / BW_L2 / t_avg, "l2_hit_rate": hit,
                       "cache_model_bound_ms": None if t_b is None else t_b * 1e3,
                       "frac_vs_cache_model": None if t_b is None else t_b / t_avg}
                if (sb, phased) in mall_run:
                    cls["mall_rate_in_run_gbs"] = mall_run[(sb, phased)]
                per_class.append(cls)
            # Vocabulary (fixed from round 5 on): for this cache-resident leg `frac` IS the unconditional
            # fraction of the L2 peak (every algorithmic byte at 34.5 TB/s) -- it moves only when the kernel moves.  The
            # model-conditional number (bound built from the kernel's own L2 hit rate and the Infinity-Cache rate) is
            # `frac_vs_cache_model`: a diagnostic of how close the launches run to what their hit rate allows, not a target.
            # The "% HBM" the metric asks for is `hbm_bound.roofline.frac`, where HBM binds.
            roof.update(bound="l2", peak=BW_L2 / 1e9, frac=t_l2 / t_actual, frac_vs_l2_peak=t_l2 / t_actual,
                        frac_vs_cache_model=(t_bound / t_actual) if have_all else None,
                        mall_rate_in_run_gbs=mall_mean, per_class=per_class,
                        cache_model={"formula": "t_bound = bytes * (hit / BW_L2 + (1 - hit) / BW_MALL), summed over the step's "
                                                "aggregation launches; frac_vs_cache_model = t_bound / measured",
                                     "BW_L2_gbs": BW_L2 / 1e9, "BW_L2_source": "MI355X_MICROARCH.md: L2 ~34.5 TB/s aggregate",
                                     "BW_MALL_gbs": bw_mall / 1e9,
                                     "BW_MALL_source": ("mean of the clean strided reads over each class's footprint measured in this "
                                                        "run (per_class.mall_rate_in_run_gbs)") if mall_mean else
                                                       "profiles/r4_mall_sweep.txt (tools/mall_sweep.py): clean streaming-read "
                                                       "plateau 48 MB .. 512 MB through the vector L1",
                                     "hit_source": (rec or {}).get("source") if hits else "no PMC record for this kernel source: "
                                                                                          "frac_vs_cache_model withheld",
                                     "hit_note": "the hit rate is the kernel's OWN: a kernel with a worse hit rate gets a looser "
                                                 "bound -- which is why this is not `frac`"},
                        note="gathered matrices (%s MB) sit in the 256 MB Infinity Cache / partly in the 8 x 4 MB L2s at "
                             "this shape, so HBM does not bind (hbm_equiv_frac > 1 is the cache hierarchy at work); the "
                             "HBM-bound measurement is the `hbm_bound` leg" % "/".join(str(m) for m in src_mb))
        else:
            roof.update(bound="hbm", peak=HBM_PEAK / 1e9, frac=roof["achieved"] * 1e9 / HBM_PEAK)

    if roof:
        roof.pop("_classes", None)
    loss_total = loss.detach().clone()
    edges_per_rank, users_per_rank = [E_local], [hi - lo]
    if dist_on:      # every rank holds its users' share of the loss; report the whole (outside the timed region)
        loss_total = SD.all_reduce_sum(loss_total.view(1))[0]
        gathered = [None] * world
        dist.all_gather_object(gathered, (E_local, rank_ms[0], comm["exposed_ms"] / args.steps, hi - lo))
        edges_per_rank = [int(e[0]) for e in gathered]
        rank_ms = [float(e[1]) for e in gathered]
        exposed_ms = [float(e[2]) for e in gathered]
        users_per_rank = [int(e[3]) for e in gathered]
    ms = elapsed / args.steps * 1e3
    value = E_total / (elapsed / args.steps)
    out = {
        "metric": METRIC,
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "ms_per_step_median_events": timed_steps.last_median_ms,
        "ms_per_step_min_max_events": [timed_steps.last_min_ms, timed_steps.last_max_ms],
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%s-shaped synthetic bipartite graph: %d users x %d items, %d ratings, %d rating levels, "
                               "dim %d; 2 stacked HeterGCNLayers (sum accum, symm support, leaky 0.1), both node types, "
                               "full neighbourhood, rating head over all ratings, fwd+bwd" % (args.shape, n_user, n_item,
                                                                                          E_total, R, D),
                   "partition": "single GPU" if world == 1 else "1-D user-block node partition, items replicated, "
                                "RCCL all-reduce of item-side partials on a side stream, overlapped with the user-side "
                                "aggregation of the same layer",
                   "order": args.order, "plan_build_s": round(t_plan, 2), "plan_builder": "device (csrc/plan_build.hip)",
                   "rating_head": "loss and both projection gradients in two gather passes that form the pair scores in "
                                  "registers (sg_pair_l2_hip); same value / gradients as scores + L2 loss, no per-pair "
                                  "array is written",
                   "loss": float(loss_total), "edges_per_rank": edges_per_rank, "users_per_rank": users_per_rank,
                   "init": "embeddings U(-0.1, 0.1), Xavier-in weights, zero biases (by parameter name), then "
                           "layer-sequential scale calibration (model.calibrate_output_scale: outputs rms 1, scores "
                           "O(1)) so that the loss depends on the scores; pre-calibration rms per stage: %s" % json.dumps(
                               [{k: float("%.3g" % v) for k, v in st.items()} for st in calib]),
                   "peak_hbm_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2)},
        "roofline": roof,
        "step_roofline_frac": value * 8 * (8 + 4 * D) / (world * HBM_PEAK),
    }
    if dist_on:
        out["collectives"] = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(),
                              "calls_per_step": comm["calls"] / args.steps,
                              "allreduce_bytes_per_step": comm["bytes"] / args.steps,
                              "collective_ms_per_step": comm["device_ms"] / args.steps,
                              "exposed_ms_per_step_per_rank": exposed_ms,
                              "note": "calls / bytes / collective_ms: rank-0 view, device time of the collectives on the "
                                      "communication stream (they overlap compute); exposed_ms: per rank, time the "
                                      "compute stream sat blocked on a collective's completion event"}
        out["ms_per_step_per_rank"] = rank_ms
    if dist_on and world > 1 and not args.no_partition_check:
        # ---- the N > 1 line checks itself: did the collectives sum what they should? -------------------------
        # Outside the timed region.  Rank 0 builds the SAME seeded model unpartitioned (whole graph, no process group in its
        # path), runs one forward + backward, and compares (i) the loss with the all-reduced loss of the partitioned run and
        # (ii) every replicated parameter's gradient -- the item embedding table (rows summed over the ranks' user blocks), the
        # item- and user-side aggregator weights, output layers and rating projections -- with the all-reduced gradients this
        # rank holds after the last timed step.  Tolerance 1e-5 of each tensor's largest magnitude (summation order across
        # ranks differs).  The other ranks wait at the barrier below.
        t_chk = time.perf_counter()
        # (a) every collective of one more step, checked by a checksum of checksums (dist._SumCheck): all ranks take part
        SD.CHECK.reset()
        SD.CHECK.enabled = True
        try:
            step()
            torch.cuda.synchronize()
        finally:
            SD.CHECK.enabled = False
        mine = (len(SD.CHECK.records), max([e for _, e in SD.CHECK.records] or [0.0]))
        every = [None] * world
        dist.all_gather_object(every, mine)
        sums = {"collectives_checked_per_rank": [int(c) for c, _ in every], "max_rel_err": max(float(e) for _, e in every),
                "tolerance": 1e-6,
                "method": "float64 sum and sum of magnitudes of every rank's buffer before each all-reduce, all-reduced on their "
                          "own; |sum(result) - sum of local sums| / sum of local magnitude sums (star-gcn_amd/dist.py _SumCheck)"}
        sums["ok"] = bool(min(sums["collectives_checked_per_rank"]) > 0 and sums["max_rel_err"] <= sums["tolerance"])
        chk = None
        if rank == 0:
            def _check():
                one = main_case(args.shape, D, args.order, dev)
                loss_1 = one.step()
                torch.cuda.synchronize()
                l1, ln = float(loss_1.detach()), float(loss_total)
                rel = abs(l1 - ln) / max(1.0, abs(l1))
                return {"loss_partitioned": ln, "loss_unpartitioned": l1, "loss_rel_diff": rel, "loss_tolerance": 1e-5,
                        "ok": bool(rel <= 1e-5),
                        "method": "rank 0 rebuilds the same seeded, name-initialised, scale-calibrated model on the whole graph "
                                  "(bench.main_case, no process group), one forward; its loss against the all-reduced loss of "
                                  "the partitioned run.  Gradients: see `f64` (every rank against the float64 definition)"}
            try:
                chk = _check()
            except Exception as e:      # the check must never cost the line
                chk = {"error": "%s: %s" % (type(e).__name__, str(e)[:300]), "ok": False}
            chk["collective_checksums"] = sums
            chk["ok"] = bool(chk.get("ok") and sums["ok"])
        # (c) every rank: its view of one more partitioned step -- its user rows, the replicated item rows, the loss, and every
        # all-reduced gradient -- against the float64 evaluation of the DEFINITION over the WHOLE graph, to the single-GPU
        # tolerance (1e-5 of each tensor's scale), with the same activation-derivative accounting as `verify` (VERDICT r5 #4;
        # replaces the gross 2e-2 bound on the replicated gradients)
        f64 = guarded_all(partition_f64_check, net, step, csr, vals, mean, std, n_user, n_item, R, E_total, lo, hi, dev)
        every = [None] * world
        dist.all_gather_object(every, f64)
        if rank == 0:
            errs = [e for e in every if "error" in e]
            if errs:
                chk["f64"] = {"error": errs[0]["error"], "ok": False}
            else:
                wr = max(range(world), key=lambda r_: every[r_]["max_rel_err"])
                gr = max(range(world), key=lambda r_: every[r_]["gradient_max_rel_err"])
                chk["f64"] = {
                    "max_rel_err": every[wr]["max_rel_err"], "worst": every[wr]["worst"], "worst_rank": wr,
                    "gradient_max_rel_err": every[gr]["gradient_max_rel_err"], "gradient_worst": every[gr]["gradient_worst"],
                    "gradient_tolerance": VERIFY_TOL, "tolerance": VERIFY_TOL,
                    "per_rank_max_rel_err": [float("%.3g" % e["max_rel_err"]) for e in every],
                    "per_tensor_rank0": every[0]["per_tensor"], "tensors_per_rank": every[0]["tensors"],
                    "activation_derivative": every[0]["activation_derivative"],
                    "seconds_per_rank": [e["seconds"] for e in every],
                    "ok": bool(all(e["max_rel_err"] <= VERIFY_TOL for e in every)),
                    "method": "every rank: one more partitioned step with capture hooks; the product's user rows of all ranks "
                              "assembled (for the activation-derivative rule only); float64 evaluation of the definition over the "
                              "WHOLE graph by tools/f64_check.py (plain torch, no product kernel); compared: loss, the rank's "
                              "user rows and all item rows of every layer output and projection, the rank's rows of the user "
                              "embedding gradient, the all-reduced item embedding gradient and every all-reduced weight / bias "
                              "gradient; error = max |fp32 - fp64| / max |fp64| per tensor"}
            chk["ok"] = bool(chk.get("ok") and chk["f64"]["ok"])
            chk["seconds"] = round(time.perf_counter() - t_chk, 2)
        dist.barrier()
        if rank == 0:
            out["partition_check"] = chk
    if args.graph_replay and (not dist_on or backend == "nccl"):
        # secondary figure: the same step as ONE hipGraph replay (the library's launches captured through
        # torch.cuda.CUDAGraph, as examples/train_star_gcn.py --graph does for the whole training iteration).  Not the
        # headline: the HIP events that time the gathers for `roofline` cannot live inside a captured graph.
        # Partitioned runs capture their RCCL all-reduces with the step (communication stream forked from and joined
        # back into the capturing stream, dist._launch_sum): every rank captures and replays the same sequence.
        try:
            del loss                  # the last step's autograd graph (and its AccumulateGrad nodes) must be gone
            out["graph_replay"] = graph_replay(step, dev, args.steps)
            if dist_on:
                tmax = torch.tensor([out["graph_replay"]["ms_per_step"]], device=dev)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                out["graph_replay"]["ms_per_step"] = float(tmax.item())
                out["graph_replay"]["note"] += "; RCCL all-reduces captured inside the graph, max over ranks"
        except Exception as e:      # capture support is a property of the torch build, not of the path
            out["graph_replay"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
        loss = None
    def guarded(fn, *a):      # a secondary leg must never cost the headline line: its failure becomes an "error" entry
        try:
            return fn(*a)
        except Exception as e:
            return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    if world == 1 and not dist_on:
        loss = None
        out["dense_roofline"] = guarded(dense_roofline, step, 3)
        # the same step with every dense product on the EXACT fp32 MFMA kernel (sg_gemm_backend(0); v_mfma_f32_32x32x2_f32,
        # no plane splitting): what `dtype: "f32"` costs without the three-f16-MFMA emulation of the default backend
        out["ms_per_step_exact_fp32_gemm"] = guarded(exact_fp32_step_ms, step, 5)
        if isinstance(out["ms_per_step_exact_fp32_gemm"], float):      # the headline in the reference's own arithmetic
            out["value_exact_fp32_gemm"] = E_total / (out["ms_per_step_exact_fp32_gemm"] * 1e-3)
        out["dense_mix_arithmetic"] = ("default: fp32 operands as two block-scaled f16 planes, three MFMAs per product, fp32 "
                                       "accumulation, per-term error <= 7.2e-7 (include/stargcn.h, sg_gemm_backend); "
                                       "ms_per_step_exact_fp32_gemm: the same step on the exact fp32 MFMA kernel")
    if world == 1 and not dist_on and not args.no_verify:
        loss = None
        out["verify"] = guarded(verify_leg, net, step, (dgraph.ind_ptr, dgraph.end_points, dgraph.level, n_item, R, None), y,
                                1.0 / E_total)
        # the same comparison with the dense mix on the exact fp32 kernel: the per-tensor accuracy cost of the emulation
        out["verify_exact_fp32"] = guarded(verify_leg_exact_fp32, net, step,
                                           (dgraph.ind_ptr, dgraph.end_points, dgraph.level, n_item, R, None), y, 1.0 / E_total)
    # free the main leg before the big one
    del net, plan, dgraph, y, step
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not dist_on and not args.no_minibatch_leg:
        try:
            out["minibatch_iteration"] = minibatch_leg(dev)
        except Exception as e:      # a secondary figure must never cost the headline
            out["minibatch_iteration"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        torch.cuda.empty_cache()
    # the 140 GB leg and its float64 verification are the LAST device work of the process
    if rank == 0 and world == 1 and not args.no_hbm_leg:
        torch.cuda.reset_peak_memory_stats(dev)
        out["hbm_bound"] = guarded(hbm_leg, args, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


def minibatch_leg(dev, shape="ml-1m", batch=100000, iters=20):
    """Secondary figure: one TRAINING ITERATION of the reference's real loop
    (experiments/STAR-GCN.py:583-632) -- rating mini-batch + masked-reconstruction nodes sampled, the batch's rating edges
    removed from the aggregation graph in both directions (graph.py:952-974), 2-block STAR-GCN with decoder (AGG 250 /
    OUT 75, dropout 0.5), both losses, backward, global-norm clipping, Adam -- on an ML-1M-shaped synthetic graph with
    everything resident: plan of the whole training graph in HBM (resident.ResidentPlan), edge removal, samplers and
    batch plans on the device (device_sampler.DeviceBatchSampler).  Timed eagerly and as ONE hipGraph replay."""
    import star_gcn_amd.model as M
    import star_gcn_amd.synthetic as S
    from star_gcn_amd.device_sampler import DeviceBatchSampler
    from star_gcn_amd.mxgraph.iterators import DataIterator
    from star_gcn_amd.resident import ResidentPlan
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    graph, eu, ei, vals = S.make_graph(shape, signal=True)
    n = eu.size
    perm = rng.permutation(n)
    n_test, n_valid = int(0.2 * n), int(0.08 * n)
    it = DataIterator(graph, U, I, np.stack([eu[perm[:n_test]], ei[perm[:n_test]]]),
                      np.stack([eu[perm[n_test:n_test + n_valid]], ei[perm[n_test:n_test + n_valid]]]),
                      embed_P_mask=0.1, embed_p_zero=0.0, embed_p_self=1.0, seed=0)
    tv = it.train_graph[U, I].values
    mean, std = float(tv.mean()), float(tv.std())
    net = M.Net(graph, U, I, embed_units=64, agg_units=(250,), out_units=(75,), nblocks=2, use_dae=True, dropout=0.5,
                agg_accum="sum").to(dev)
    resident = ResidentPlan(net, it.train_graph, device=dev)
    sampler = DeviceBatchSampler(resident, batch, embed_P_mask=0.1, embed_p_zero=0.0, seed=0)
    state = dict()

    def iteration(advance):
        db = sampler.next_batch(advance_on_device=advance)
        y = (db["ratings"] - mean) / std
        preds, recons, gt = net.run(resident.set_batch_device(db), rating_targets=y, rating_scale=1.0 / y.numel())
        loss = M.star_gcn_loss(preds, recons, gt, y, recon_lambda=0.1)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0, foreach=True)
        state["opt"].step()
        state["opt"].zero_grad(set_to_none=True)
        return loss.detach()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        net.run(resident.set_batch_device(sampler.next_batch(advance_on_device=True)))      # materialises the parameters
        state["opt"] = torch.optim.Adam(net.parameters(), lr=0.002, capturable=True)
        for _ in range(3):
            iteration(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            loss = iteration(True)
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) / iters * 1e3
    torch.cuda.current_stream().wait_stream(side)
    out = {"workload": "%s-shaped synthetic graph (%d users x %d items, %d training ratings), batch %d ratings, 2-block "
                       "STAR-GCN with decoder (embed 64, AGG 250, OUT 75, dropout 0.5), rating + reconstruction loss, "
                       "clipping, Adam; resident plan, device edge removal / samplers / batch plans" %
                       (shape, graph[U, I].shape[0], graph[U, I].shape[1], it.train_graph[U, I].nnz, batch),
           "iters": iters, "ms_per_iteration_eager": eager_ms, "loss": float(loss)}
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            state["loss"] = iteration(True)
        torch.cuda.synchronize()
        for _ in range(2):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g.replay()
        torch.cuda.synchronize()
        out["ms_per_iteration_hipgraph"] = (time.perf_counter() - t0) / iters * 1e3
        out["loss_after_replays"] = float(state["loss"])
    except Exception as e:      # capture support is a property of the torch build
        out["ms_per_iteration_hipgraph"] = None
        out["hipgraph_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
    return out


def cpu_baseline(args):
    """The CPU baseline leg, in its OWN process: `bench.py --cpu-baseline-only` started with one OpenMP thread per physical
    core the container may use, pinned (OMP_NUM_THREADS / OMP_PROC_BIND=close / OMP_PLACES=cores in THAT process's
    environment only -- see the note at the top of this file); it regenerates the same seeded graph and prints one JSON
    object, which is returned."""
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = str(_host_threads())
    env["OMP_PROC_BIND"], env["OMP_PLACES"] = "close", "cores"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--shape", args.shape, "--dim", str(args.dim),
           "--cpu-sample-users", str(args.cpu_sample_users)]
    try:        # a reported baseline must never cost the headline: every failure becomes an "error" entry
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        if p.returncode != 0:
            return {"error": "cpu baseline process failed (%d): %s" % (p.returncode, p.stderr[-400:])}
        return json.loads(p.stdout.strip().splitlines()[-1])
    except (subprocess.TimeoutExpired, ValueError, IndexError, OSError) as e:
        return {"error": "cpu baseline: %s: %s" % (type(e).__name__, str(e)[:300])}


def cpu_baseline_main(args):
    """Body of the CPU-baseline process: oracle port (oracle/cpu_step.py) timed on this host: the same network on the same
    graph (the FULL graph by default), one OpenMP thread per physical core, pinned (environment given by the parent)."""
    import star_gcn_amd.synthetic as S
    from oracle import cpu_step as C
    from star_gcn_amd.mxgraph.graph import HeterGraph
    graph, _eu, _ei, _vals = S.make_graph(args.shape)
    D = args.dim
    csr = graph[U, I]
    n_u = max(1, int(csr.shape[0] * args.cpu_sample_users))
    if n_u < csr.shape[0]:
        sub = S.user_block(graph, U, I, 0, n_u)
        g = HeterGraph({U: np.arange(n_u, dtype=np.int32), I: np.arange(csr.shape[1], dtype=np.int32)}, {(U, I): sub})
    else:
        sub, g = csr, graph

    def levels_of(gr, s):
        lv = dict()
        for dst, a, b in (("user", U, I), ("item", I, U)):
            eps, _, ips, sps = gr[a, b].sample_neighbors(symm=True, use_multi_link=True, num_neighbors=-1)
            lv[dst] = ([np.ascontiguousarray(e, np.int32) for e in eps], ips,
                       [np.ascontiguousarray(x, np.float32) for x in sps])
        lv["pairs"] = (s.end_points, s.ind_ptr, None)
        return lv

    # warm-up on 1/32 of the users (page-in, BLAS and OpenMP thread pools), then ONE timed step of each variant
    n_w = max(1, csr.shape[0] // 32)
    wsub = S.user_block(graph, U, I, 0, n_w)
    wg = HeterGraph({U: np.arange(n_w, dtype=np.int32), I: np.arange(csr.shape[1], dtype=np.int32)}, {(U, I): wsub})
    C.run_cpu_step(levels_of(wg, wsub), n_w, csr.shape[1], D, steps=1)
    lv = levels_of(g, sub)
    ph, ph_fair = dict(), dict()
    sec = C.run_cpu_step(lv, n_u, csr.shape[1], D, steps=1, phases=ph)
    sec_fair = C.run_cpu_step(lv, n_u, csr.shape[1], D, steps=1, fair=True, phases=ph_fair)
    info = C.host_info()
    rnd = lambda d: {k: round(v, 3) for k, v in d.items()}
    out = {"value": sub.nnz / sec, "unit": "edges/s", "cores": int(os.environ.get("OMP_NUM_THREADS", info["physical_cores"])),
           "cpu_quota_cores": _cpu_quota(),
           "kind": "port", "seconds_per_step": round(sec, 3), "phases_s": rnd(ph),
           "fair_value": sub.nnz / sec_fair, "fair_seconds_per_step": round(sec_fair, 3), "fair_phases_s": rnd(ph_fair),
           "fair_note": "same port with the data-gradient kernel parallelised over destination rows through the "
                        "transposed CSR (the reference runs it serially for K = 1, seg_op.cc:232-233)",
           "sample": "users [0,%d) of %d x all %d items = %d of %d ratings of the same graph, 1 timed fwd+bwd step per "
                     "variant after a warm-up step on 1/32 of the users; seg ops = C restatement of reference seg_op.cc "
                     "CPU kernels (reference OpenMP placement: forward over rows, backward serial), dense = torch-CPU "
                     "BLAS standing in for MXNet FullyConnected; one OpenMP thread per physical core the container may "
                     "use (cgroup CPU quota respected; OMP_PROC_BIND=close, OMP_PLACES=cores), in a process of its own" % (
                         n_u, csr.shape[0], csr.shape[1], sub.nnz, csr.nnz),
           "host": info}
    print(json.dumps(out), flush=True)


def main():
    args = parse()
    if args.cpu_baseline_only:
        return cpu_baseline_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    run_rank(args)


if __name__ == "__main__":
    main()
