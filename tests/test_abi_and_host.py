"""CPU-runnable checks: the C-ABI library loads and exports every symbol include/stargcn.h declares, and the
host-side (`_cpu`) plan/graph helpers match the oracle.  No compute kernels are called (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import star_gcn_amd._lib as L
from oracle import seg as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "stargcn.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 47
    handle = ctypes.CDLL(L.SO_PATH)
    for n in names:
        assert hasattr(handle, n), "libstargcn_hip.so does not export %s" % n
    assert sorted(L.exported_symbols()) == names, "ctypes table and header disagree"
    assert L.lib().sg_version() >= 100


def test_errors_are_codes_not_exit():
    lib = L.lib()
    rc = lib.sg_build_transpose_cpu(None, None, None, None, None, -1, 0, 0)
    assert rc < 0 and b"negative" in lib.sg_last_error()


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_build_transpose_matches_stable_sort():
    rng = np.random.default_rng(3)
    S, T, nnz = 50, 37, 700
    cuts = np.sort(rng.integers(0, nnz - 20, S - 1))
    indptr = np.concatenate([[0], cuts, [nnz - 20]]).astype(np.int32)  # 20 padding edges past indptr[-1]
    indices = rng.integers(0, T, nnz).astype(np.int32)
    t_indptr, t_pos, t_seg = np.empty(T + 1, np.int32), np.empty(nnz, np.int32), np.empty(nnz, np.int32)
    L.check(L.lib().sg_build_transpose_cpu(_vp(t_indptr), _vp(t_pos), _vp(t_seg), _vp(indices), _vp(indptr), S, T, nnz))
    E = int(indptr[-1])
    order = np.argsort(indices[:E], kind="stable")
    assert t_indptr[-1] == E
    assert np.array_equal(t_pos[:E], order.astype(np.int32))
    seg_of = np.repeat(np.arange(S), np.diff(indptr)).astype(np.int32)
    assert np.array_equal(t_seg[:E], seg_of[order])
    assert np.array_equal(np.diff(t_indptr), np.bincount(indices[:E], minlength=T))


def test_get_support_and_multi_link_split_match_oracle():
    rng = np.random.default_rng(4)
    N, M, nnz = 40, 30, 500
    cuts = np.sort(rng.integers(0, nnz + 1, N - 1))
    ip = np.concatenate([[0], cuts, [nnz]]).astype(np.int32)
    ep = rng.integers(0, M, nnz).astype(np.int32)
    rd = np.diff(ip).astype(np.int32)
    cd = np.bincount(ep, minlength=M).astype(np.int32)
    cd[3] = 0  # force a zero-degree column: support must be 0 there
    for symm in (1, 0):
        got = np.empty(nnz, np.float32)
        L.check(L.lib().sg_get_support_cpu(_vp(got), _vp(rd), _vp(cd), _vp(ep), _vp(ip), N, symm))
        np.testing.assert_array_equal(got, O.get_support(rd, cd, ep, ip, symm=bool(symm)))
    levels = np.array([0.5, 1.0, 2.5, 4.0], np.float32)
    vals = levels[rng.integers(0, 4, nnz)]
    pos = np.empty(nnz, np.int32)
    ips = np.empty((4, N + 1), np.int32)
    off = np.empty(5, np.int64)
    L.check(L.lib().sg_multi_link_split_cpu(_vp(pos), _vp(ips), _vp(off), _vp(vals), _vp(ip), _vp(levels), N, 4))
    opos, oips = O.multi_link_split(vals, ip, levels)
    for l in range(4):
        assert np.array_equal(pos[off[l]:off[l + 1]], opos[l])
        assert np.array_equal(ips[l], oips[l])
    vals[7] = 3.0  # matches no level -> error code, not exit()
    assert L.lib().sg_multi_link_split_cpu(_vp(pos), _vp(ips), _vp(off), _vp(vals), _vp(ip), _vp(levels), N, 4) == -4


def make_multilink(rng, n_dst, n_src, nnz, R, pad_empty=True):
    """Random per-level CSR lists the way reference gen_plan/heter_sage hands them to the aggregator."""
    cuts = np.sort(rng.integers(0, nnz + 1, n_dst - 1))
    ip = np.concatenate([[0], cuts, [nnz]]).astype(np.int32)
    ep = rng.integers(0, n_src, nnz).astype(np.int32)
    sup = rng.uniform(0.05, 1.0, nnz).astype(np.float32)
    lev = rng.integers(0, R, nnz)
    if R > 2:
        lev[lev == R - 1] = 0  # make the last level empty (exercises empty_as_zero padding)
    eps, ips, sps = [], [], []
    for r in range(R):
        sel = lev == r
        cs = np.concatenate([[0], np.cumsum(sel.astype(np.int64))])
        cnt = cs[ip[1:]] - cs[ip[:-1]]
        ips.append(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32))
        e, s = ep[sel], sup[sel]
        if e.size == 0 and pad_empty:  # reference graph.py:221-222 empty_as_zero
            e, s = np.zeros(1, np.int32), np.zeros(1, np.float32)
        eps.append(e)
        sps.append(s)
    return eps, ips, sps


def test_multilink_fuse_structure():
    from star_gcn_amd.plan import MultiLinkPlan
    rng = np.random.default_rng(5)
    n_dst, n_src, nnz, R = 23, 17, 300, 4
    eps, ips, sps = make_multilink(rng, n_dst, n_src, nnz, R)
    plan = MultiLinkPlan(eps, ips, sps, n_src, "cpu")
    assert plan.nnz == nnz
    c_indptr, c_idx, c_w, c_q = (t.numpy() for t in (plan.c_indptr, plan.c_idx, plan.c_w, plan.c_q))
    for i in range(n_dst):
        for r in range(R):
            a, b = c_indptr[i * R + r], c_indptr[i * R + r + 1]
            assert np.array_equal(c_idx[a:b], eps[r][ips[r][i]:ips[r][i + 1]])
            assert np.array_equal(c_w[a:b], sps[r][ips[r][i]:ips[r][i + 1]])
            assert np.array_equal(c_q[a:b], c_idx[a:b] * R + r)
    # dense check of both CSRs against the same (n_dst*R, n_src) matrix
    A = np.zeros((n_dst * R, n_src))
    for s in range(n_dst * R):
        for j in range(c_indptr[s], c_indptr[s + 1]):
            A[s, c_idx[j]] += c_w[j]
    t_indptr, t_idx, t_w, t_q = (t.numpy() for t in (plan.t_indptr, plan.t_idx, plan.t_w, plan.t_q))
    B = np.zeros_like(A)
    for n in range(n_src):
        for r in range(R):
            seg = slice(t_indptr[n * R + r], t_indptr[n * R + r + 1])
            assert np.all(np.diff(t_idx[seg]) >= 0)  # destinations in increasing order = original CSR order
            assert np.array_equal(t_q[seg], t_idx[seg] * R + r)
            for i, w in zip(t_idx[seg], t_w[seg]):
                B[i * R + r, n] += w
    np.testing.assert_allclose(A, B, rtol=0, atol=1e-6)
    assert np.array_equal(plan.d_indptr.numpy(), c_indptr[::R])
    assert np.array_equal(plan.s_indptr.numpy(), t_indptr[::R])


def test_fused_order_routing_and_sizes_are_host_decisions(monkeypatch):
    """sg_multilink_agg_resolve_order2 (SG_ORDER_FUSED = 3, csrc/agg_fused.hip): 'auto' fuses only 256-wide 'sum' aggregations
    over graphs whose R-expanded matrix would travel through HBM; buffer sizes of the fused order; the plan-geometry helpers.
    Nothing is launched."""
    monkeypatch.delenv("SG_FUSED", raising=False)
    lib = L.lib()
    st = L.MultiLinkPlanStruct()
    ref = ctypes.cast(ctypes.pointer(st), ctypes.c_void_p)
    st.n_dst, st.n_src, st.nnz, st.num_links = 1_000_000, 1_250_000, 125_000_000, 16          # the config-5 shard
    assert lib.sg_multilink_agg_resolve_order2(ref, 0, 256, 256, 0) == 3
    assert lib.sg_multilink_agg_resolve_order2(ref, 0, 256, 256, 1) == 2          # 'stack': not fused, expansion on the smaller side
    assert lib.sg_multilink_agg_resolve_order2(ref, 0, 128, 256, 0) == 2          # other widths
    assert lib.sg_multilink_agg_resolve_order2(ref, 1, 256, 256, 0) == 1          # an explicit order is kept
    assert lib.sg_multilink_agg_resolve_order2(ref, 3, 64, 64, 0) == 3            # ... also 'fused' (the launch then refuses)
    st.n_dst, st.n_src, st.nnz, st.num_links = 69878, 10677, 10_000_054, 10       # MovieLens-10M: cache-resident, 167 item tiles
    assert lib.sg_multilink_agg_resolve_order2(ref, 0, 256, 256, 0) == 1
    # measured rule (multilink.hip, profiles/r5_fused_kernel.md section 7): node sides within a factor of two, >= 2^14 nodes on
    # the smaller one, its R-expanded matrix beyond 192 MB
    for shape, want in (((600_000, 500_000, 60_000_000, 5), 3), ((30_000, 25_000, 3_000_000, 10), 3), ((17_000, 16_500, 1_500_000, 16), 3),
                        ((300_000, 250_000, 30_000_000, 2), 3),
                        ((70_000, 35_000, 5_000_000, 5), 1),           # 171 MB: stays in the caches
                        ((156_250, 1_000_000, 15_600_000, 16), 2),     # one rank's user block of eight: lopsided
                        ((500_000, 20_000, 20_000_000, 16), 1), ((100_000, 40_000, 10_000_000, 10), 1),
                        ((300_000, 250_000, 30_000_000, 1), 1),        # one level: nothing is expanded
                        ((16_000, 16_000, 4_000_000, 32), 1)):         # fewer than 2^14 nodes: fewer tiles than CUs
        st.n_dst, st.n_src, st.nnz, st.num_links = shape
        assert lib.sg_multilink_agg_resolve_order2(ref, 0, 256, 256, 0) == want, shape
    assert lib.sg_multilink_agg_resolve_order2(ref, 7, 256, 256, 0) < 0
    # sizes: nothing saved by the fused forward; its backward workspace holds dpre, dH (n_src x R x 256) and the packed gradients
    st.n_dst, st.n_src, st.nnz, st.num_links = 1000, 3000, 50000, 4
    st.struct_bytes = ctypes.sizeof(L.MultiLinkPlanStruct)
    assert lib.sg_multilink_agg_workspace_bytes(ref, 256, 256, 3, 0, 1) == 0      # no level-major edge orders attached: invalid
    for which in range(2):                 # (size queries never dereference them)
        st.fused[which].f_ptr = st.fused[which].f_idx = st.fused[which].f_w = 4096
    # destination side smaller: the forward keeps Z = [A_r x]_r (n_dst x R x 256) for dW = dpre^T Z, the backward writes no dH
    assert lib.sg_multilink_agg_saved_bytes(ref, 256, 256, 3, 0) == 1000 * 4 * 256 * 4
    wb = lib.sg_multilink_agg_workspace_bytes(ref, 256, 256, 3, 0, 1)
    assert (1000 * 256 + 4 * 256 * 256) * 4 <= wb < (1000 * 256 + 3000 * 4 * 256) * 4 and wb >= lib.sg_agg_fused_workspace_bytes(4)
    # source side smaller: nothing saved by the forward, the data-gradient launch writes dH (n_src x R x 256) into the workspace
    st.n_dst, st.n_src = 3000, 1000
    assert lib.sg_multilink_agg_saved_bytes(ref, 256, 256, 3, 0) == 0
    wb = lib.sg_multilink_agg_workspace_bytes(ref, 256, 256, 3, 0, 1)
    assert wb >= (3000 * 256 + 1000 * 4 * 256 + 4 * 256 * 256) * 4
    assert lib.sg_multilink_agg_workspace_bytes(ref, 256, 256, 3, 0, 0) >= lib.sg_agg_fused_workspace_bytes(4)
    assert lib.sg_agg_fused_workspace_bytes(16) >= 16 * 256 * 256 * 4            # two f16 planes of sixteen 256 x 256 matrices
    assert lib.sg_agg_fused_tiles(0) == 0 and lib.sg_agg_fused_tiles(64) == 1 and lib.sg_agg_fused_tiles(65) == 2
    assert lib.sg_agg_fused_supported(256, 256, 32) == 1 and lib.sg_agg_fused_supported(256, 256, 33) == 0
    # round 6: rows of 4 .. 256 floats (multiple of 4), 1 .. 256 units per level -- the reference's 32 / 64 -> 250 among them
    assert lib.sg_agg_fused_supported(256, 128, 4) == 1 and lib.sg_agg_fused_supported(64, 250, 5) == 1
    assert lib.sg_agg_fused_supported(30, 64, 4) == 0 and lib.sg_agg_fused_supported(260, 64, 4) == 0 and lib.sg_agg_fused_supported(64, 257, 4) == 0
    # U = 250: every level of the backward's dpre / dH is padded to 252 floats (one float4 per gather lane)
    wb250 = lib.sg_multilink_agg_workspace_bytes(ref, 64, 250, 3, 0, 1)
    assert wb250 >= (3000 * 252 + 1000 * 4 * 252 + 4 * 252 * 64) * 4
    assert lib.sg_multilink_agg_workspace_bytes(ref, 64, 50, 3, 1, 1) >= (3000 * 4 * 52 + 1000 * 4 * 52) * 4      # 'stack'
    # a fused launch without the plan's level-major edge orders is refused with a message, not run
    for which in range(2):
        st.fused[which].f_ptr = None
    rc = lib.sg_multilink_agg_fwd_hip(None, None, None, None, None, ref, 256, 256, 3, 0, 0, 0.0, None, 0, None)
    assert rc < 0 and b"fused" in lib.sg_last_error().lower()


def test_fused_aggregator_entry_sizes_orders_and_errors():
    """sg_multilink_agg_* (the fused entry of SURVEY 8b): order resolution, buffer sizing and argument errors are
    host-side decisions -- checked here without launching anything."""
    lib = L.lib()
    st = L.MultiLinkPlanStruct()
    st.n_dst, st.n_src, st.nnz, st.num_links = 1000, 300, 20000, 5
    ref = ctypes.cast(ctypes.pointer(st), ctypes.c_void_p)
    assert lib.sg_multilink_agg_resolve_order(ref, 0) == 1          # n_src <= n_dst -> transform first
    st.n_dst, st.n_src = 300, 1000
    assert lib.sg_multilink_agg_resolve_order(ref, 0) == 2          # expansion on the smaller (destination) side
    assert lib.sg_multilink_agg_resolve_order(ref, 1) == 1
    D, U = 64, 50
    ld = 6 * 64                          # R*D + R rowsum columns = 325, row pitch rounded to 256 B because 4*D is (64 floats)
    assert lib.sg_multilink_agg_saved_bytes(ref, D, U, 2, 0) == 300 * ld * 4
    assert lib.sg_multilink_agg_saved_bytes(ref, 50, U, 2, 0) == 300 * (5 * 50 + 5 + 1) * 4   # otherwise to 16 B
    assert lib.sg_multilink_agg_saved_bytes(ref, D, U, 1, 0) == 0
    for order in (1, 2):
        for accum in (0, 1):
            f = lib.sg_multilink_agg_workspace_bytes(ref, D, U, order, accum, 0)
            b = lib.sg_multilink_agg_workspace_bytes(ref, D, U, order, accum, 1)
            assert f > 0 and b > f
    # transform-first forward must at least hold Wcat, bcat and H = (n_src, R*U)
    assert lib.sg_multilink_agg_workspace_bytes(ref, D, U, 1, 0, 0) >= (5 * U * D + 5 * U + 1000 * 5 * U) * 4
    buf = np.zeros(16, np.float32)
    ptrs = (ctypes.c_void_p * 5)(*[buf.ctypes.data] * 5)
    rc = lib.sg_multilink_agg_fwd_hip(_vp(buf), None, _vp(buf), ptrs, ptrs, ref, D, U, 1, 0, 0, 0.1, _vp(buf), 64, None)
    assert rc == -5 and b"workspace too small" in lib.sg_last_error()
    st.num_links = 33
    assert lib.sg_multilink_agg_fwd_hip(_vp(buf), None, _vp(buf), ptrs, ptrs, ref, D, U, 1, 0, 0, 0.1, None, 0, None) == -1
    assert lib.sg_multilink_agg_workspace_bytes(None, D, U, 1, 0, 0) == 0 and b"plan is null" in lib.sg_last_error()


def test_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from star_gcn_amd import contrib
    with pytest.raises(L.StarGCNError):
        contrib.seg_sum(torch.zeros(1, 4), torch.tensor([0, 4], dtype=torch.int32))


def test_drop_in_import_path():
    """PYTHONPATH=star-gcn_amd makes `import mxgraph.layers` / `mxgraph.graph` resolve to the mirror."""
    import subprocess
    import sys
    code = ("import mxgraph.layers as L, mxgraph.graph as G, mxgraph.iterators as I\n"
            "assert all(hasattr(L, n) for n in ('MultiLinkGCNAggregator', 'GCNAggregator', 'HeterGCNLayer',"
            " 'StackedHeterGCNLayers', 'InnerProductLayer', 'LayerDictionary', 'get_activation'))\n"
            "assert all(hasattr(G, n) for n in ('CSRMat', 'HeterGraph', 'merge_nodes', 'merge_node_ids_dict',"
            " 'unordered_unique', 'empty_as_zero'))\nprint('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "star-gcn_amd"))
    out = subprocess.check_output([sys.executable, "-c", code], env=env, cwd="/tmp").decode()
    assert out.strip().endswith("ok")


def test_bench_withholds_pmc_traffic_when_the_kernel_source_changed(monkeypatch):
    """roofline.traffic comes from committed PMC passes (profiles/pmc_traffic.json); every record carries the sha of the
    seg_gather.hip it was measured on and bench.py must not present it for another kernel source."""
    import bench
    rec = bench.profile_record("ml-10m:256")
    assert rec and not rec.get("stale") and rec["traffic_bytes_per_launch_mean"] > 0       # committed record matches the tree
    hbm = bench.profile_record("hbm-config5-shard:256")
    assert hbm and not hbm.get("stale")
    monkeypatch.setattr(bench, "_sha16", lambda path: "0123456789abcdef")
    stale = bench.profile_record("ml-10m:256")
    assert stale["stale"] and "traffic withheld" in stale["source"] and "traffic_bytes_per_launch_mean" not in stale
    assert bench.profile_record("no-such-shape:1") is None


@pytest.mark.parametrize("bad_rank", [0, 2])
def test_bench_launcher_fails_fast_when_any_rank_dies(tmp_path, bad_rank):
    """`python bench.py --gpus N` starts its ranks itself (bench.launch_ranks).  If ONE rank dies -- whichever -- while the
    others wait in a collective, the run must end at once with that rank's exit code instead of sitting in
    `procs[0].wait()` until the RCCL watchdog fires.  Stand-in ranks: the bad one exits 5 after 0.3 s,
    the others would sleep for two minutes."""
    import time
    import types
    import bench
    script = tmp_path / "rank.py"
    script.write_text("import os, sys, time\n"
                      "if int(os.environ['RANK']) == int(sys.argv[1]):\n    time.sleep(0.3)\n    sys.exit(5)\n"
                      "time.sleep(120)\n")
    t0 = time.time()
    rc = bench.launch_ranks(types.SimpleNamespace(gpus=4), script=str(script), argv=[str(bad_rank)])
    assert rc == 5 and time.time() - t0 < 20


def test_bench_launcher_returns_zero_when_all_ranks_finish(tmp_path):
    import types
    import bench
    script = tmp_path / "rank.py"
    script.write_text("import os, time\ntime.sleep(0.1 * int(os.environ['RANK']))\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n")
    assert bench.launch_ranks(types.SimpleNamespace(gpus=3), script=str(script), argv=[]) == 0


def test_importing_bench_and_calling_the_library_leave_the_process_environment_and_cpu_mask_alone():
    """Root cause of round 3's "freeze" (DESIGN section 5): `import bench` exported OMP_PROC_BIND / OMP_PLACES into the running
    process, the library's lazily initialised OpenMP runtime (LLVM libomp) picked them up at the next host builder call and
    pinned the caller's main thread to one core; threads created later inherited the mask.  Three facts keep it fixed:
    bench.py does not touch OMP_* at import, libstargcn_hip.so carries no OpenMP runtime, and a host builder call -- even
    with the binding variables set under the running process -- leaves the CPU mask of the calling thread as it was."""
    code = (
        "import os, ctypes, subprocess, sys\n"
        "import numpy as np\n"
        "import torch\n"
        "before = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OMP_PROC_BIND', 'OMP_PLACES')}\n"
        "mask = os.sched_getaffinity(0)\n"
        "import bench\n"
        "assert {k: os.environ.get(k) for k in before} == before, 'import bench changed the OpenMP environment'\n"
        "os.environ['OMP_PROC_BIND'], os.environ['OMP_PLACES'] = 'close', 'cores'    # what round 3's import did\n"
        "import star_gcn_amd._lib as L\n"
        "lib = L.lib()\n"
        "rng = np.random.default_rng(0)\n"
        "S, T, nnz = 5000, 700, 1500000\n"
        "idx = rng.integers(0, T, nnz).astype(np.int32)\n"
        "ip = np.concatenate([[0], np.sort(rng.integers(0, nnz + 1, S - 1)), [nnz]]).astype(np.int32)\n"
        "vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)\n"
        "rd, cd = np.diff(ip).astype(np.int32), np.bincount(idx, minlength=T).astype(np.int32)\n"
        "out, row = np.empty(nnz, np.float32), np.empty(nnz, np.int32)\n"
        "assert lib.sg_get_support_cpu(vp(out), vp(rd), vp(cd), vp(idx), vp(ip), S, 1) == 0      # > 2^20 edges: the threaded path\n"
        "assert lib.sg_gen_row_indices_cpu(vp(row), vp(ip), S, nnz) == 0\n"
        "assert np.array_equal(row, np.repeat(np.arange(S), np.diff(ip)))\n"
        "assert os.sched_getaffinity(0) == mask, 'a host builder changed the CPU mask of the calling thread'\n"
        "print('ok')\n")
    env = {k: v for k, v in os.environ.items() if not k.startswith("OMP_")}
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]
    deps = subprocess.run(["ldd", L.SO_PATH], capture_output=True, text=True).stdout
    assert "libomp" not in deps and "libgomp" not in deps, deps
