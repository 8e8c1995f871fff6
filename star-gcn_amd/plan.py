"""Aggregation plans: integer structure that is built ONCE per graph by native host code and kept resident in
HBM, instead of being re-sorted on every backward call (reference seg_op.cu:906-925) and re-uploaded on every
forward call (reference layers.py:366-377).
"""
import collections
import ctypes

import numpy as np
import torch

from . import _lib as L


def _np_i32(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.int32)


def _np_f32(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class TransposePlan(object):
    """Stable transpose of a CSR (indices, indptr): for every source row n the edges j with indices[j] == n in
    increasing j (t_pos) and the segment each belongs to (t_seg).  Same summation order as the reference's
    stable radix sort + run scan (seg_op.cu:906-925).

    Device index tensors are transposed ON the device (sg_build_transpose_hip: wave64 radix sort, no host round trip);
    host arrays go through the native host builder (sg_build_transpose_cpu) and one upload."""

    def __init__(self, indices, indptr, total_ind_num, device):
        T = int(total_ind_num)
        if isinstance(indices, torch.Tensor) and indices.is_cuda:
            idx, ip = L.i32c(indices), L.i32c(indptr)
            S, nnz = ip.shape[0] - 1, idx.shape[0]
            dev = idx.device
            self.t_indptr = torch.empty(T + 1, dtype=torch.int32, device=dev)
            self.t_pos = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
            self.t_seg = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)
            lib = L.lib()
            ws, wsn = L.workspace(lib.sg_build_transpose_workspace_bytes(S, T, nnz), dev)
            L.check(lib.sg_build_transpose_hip(L.ptr(self.t_indptr), L.ptr(self.t_pos), L.ptr(self.t_seg), L.ptr(idx),
                                               L.ptr(ip), S, T, nnz, L.ptr(ws), wsn, L.stream_ptr()),
                    "sg_build_transpose_hip")
            self.seg_num, self.nnz, self.total_ind_num = S, nnz, T
            self.covered = None          # = t_indptr[-1], on the device; nothing on the hot path needs it on the host
            return
        idx, ip = _np_i32(indices), _np_i32(indptr)
        S, nnz = ip.shape[0] - 1, idx.shape[0]
        t_indptr = np.empty(T + 1, np.int32)
        t_pos = np.empty(max(nnz, 1), np.int32)
        t_seg = np.empty(max(nnz, 1), np.int32)
        L.check(L.lib().sg_build_transpose_cpu(_vp(t_indptr), _vp(t_pos), _vp(t_seg), _vp(idx), _vp(ip), S, T, nnz),
                "sg_build_transpose_cpu")
        self.seg_num, self.nnz, self.total_ind_num = S, nnz, T
        self.covered = int(t_indptr[-1])
        self.t_indptr = torch.from_numpy(t_indptr).to(device)
        self.t_pos = torch.from_numpy(t_pos).to(device)
        self.t_seg = torch.from_numpy(t_seg).to(device)


_tplan_cache = collections.OrderedDict()
_TPLAN_CACHE_MAX = 64


def transpose_plan_for(indices, indptr, total_ind_num):
    """Cached TransposePlan for device index tensors (built on the device at first use: the torch autograd wrappers of
    `contrib` keep a graph's plan across steps instead of re-sorting per call as the reference does).  The cache holds
    references to the key tensors so their storage cannot be recycled under a stale entry.  The stateless C-ABI form
    for foreign runtimes is sg_seg_weighted_pool_bwd_data_dev_hip."""
    key = (indices.data_ptr(), indptr.data_ptr(), indices._version, indptr._version, indices.numel(),
           indptr.numel(), int(total_ind_num), str(indices.device))
    hit = _tplan_cache.get(key)
    if hit is not None:
        _tplan_cache.move_to_end(key)
        return hit[0]
    plan = TransposePlan(indices, indptr, total_ind_num, indices.device)
    _tplan_cache[key] = (plan, indices, indptr)
    while len(_tplan_cache) > _TPLAN_CACHE_MAX:
        _tplan_cache.popitem(last=False)
    return plan


class MultiLinkPlan(object):
    """The R per-rating-level CSRs of one (destination type, source type) aggregation (the `end_points_l`,
    `indptr_l`, `support_l` lists of reference aggregators.py:111-149) fused into ONE CSR over n_dst*R segments
    plus its transpose over n_src*R segments, resident on the device.

    c_*: segment i*R+r lists the level-r neighbours of destination i   (c_idx = source node, c_q = node*R+r)
    t_*: segment n*R+r lists the destinations reached from source n    (t_idx = dest node,   t_q = node*R+r)
    d_indptr / s_indptr: the same edge arrays viewed as un-split CSRs over n_dst / n_src rows.
    """

    def __init__(self, end_points_l, indptr_l, support_l, n_src, device):
        R = len(end_points_l)
        if R == 0 or len(indptr_l) != R or len(support_l) != R:
            raise L.StarGCNError("MultiLinkPlan needs equally long, non-empty per-level lists")
        if all(isinstance(a, torch.Tensor) and a.is_cuda for a in list(end_points_l) + list(indptr_l) + list(support_l)):
            self._init_from_device_lists(end_points_l, indptr_l, support_l, int(n_src))
            return
        ips = [_np_i32(a) for a in indptr_l]
        n_dst = ips[0].shape[0] - 1
        for a in ips:
            if a.shape[0] != n_dst + 1:
                raise L.StarGCNError("every level must carry a full-length indptr (n_dst+1)")
        eps = [_np_i32(a) for a in end_points_l]
        sps = [_np_f32(a) for a in support_l]
        nnz = int(sum(int(a[-1]) for a in ips))
        for r in range(R):  # padding (empty_as_zero, reference graph.py:221-222) may make arrays longer, never shorter
            if eps[r].shape[0] < ips[r][-1] or sps[r].shape[0] < ips[r][-1]:
                raise L.StarGCNError("level %d: end_points/support shorter than indptr[-1]" % r)
        n_src = int(n_src)
        m = max(nnz, 1)
        c_indptr = np.empty(n_dst * R + 1, np.int32)
        t_indptr = np.empty(n_src * R + 1, np.int32)
        c_idx, c_q, t_idx, t_q = (np.zeros(m, np.int32) for _ in range(4))
        c_w, t_w = np.zeros(m, np.float32), np.zeros(m, np.float32)
        arr = ctypes.c_void_p * R
        L.check(L.lib().sg_multilink_fuse_cpu(
            _vp(c_indptr), _vp(c_idx), _vp(c_q), _vp(c_w), _vp(t_indptr), _vp(t_idx), _vp(t_q), _vp(t_w),
            ctypes.cast(arr(*[a.ctypes.data for a in eps]), ctypes.c_void_p),
            ctypes.cast(arr(*[a.ctypes.data for a in ips]), ctypes.c_void_p),
            ctypes.cast(arr(*[a.ctypes.data for a in sps]), ctypes.c_void_p), R, n_dst, n_src),
            "sg_multilink_fuse_cpu")
        self.R, self.n_dst, self.n_src, self.nnz, self.device = R, n_dst, n_src, nnz, torch.device(device)
        # one host->device copy for the ten arrays (the two fp32 weight arrays travel as their bit patterns)
        (self.c_indptr, self.c_idx, self.c_q, cw, self.t_indptr, self.t_idx, self.t_q, tw, self.d_indptr,
         self.s_indptr) = upload_packed([c_indptr, c_idx, c_q, c_w.view(np.int32), t_indptr, t_idx, t_q, t_w.view(np.int32),
                                         np.ascontiguousarray(c_indptr[::R]), np.ascontiguousarray(t_indptr[::R])], device)
        self.c_w, self.t_w = cw.view(torch.float32), tw.view(torch.float32)
        self._rowsum = None
        self._struct = None
        self.c_from = self.t_from = None

    def _alloc(self, R, n_dst, n_src, nnz, dev, with_from=False):
        self.R, self.n_dst, self.n_src, self.nnz, self.device = int(R), int(n_dst), int(n_src), int(nnz), dev
        m = max(self.nnz, 1)
        i32 = lambda n: torch.zeros(n, dtype=torch.int32, device=dev)
        self.c_indptr, self.t_indptr = i32(self.n_dst * self.R + 1), i32(self.n_src * self.R + 1)
        self.c_idx, self.c_q, self.t_idx, self.t_q = i32(m), i32(m), i32(m), i32(m)
        self.c_w = torch.zeros(m, dtype=torch.float32, device=dev)
        self.t_w = torch.zeros(m, dtype=torch.float32, device=dev)
        self.d_indptr, self.s_indptr = i32(self.n_dst + 1), i32(self.n_src + 1)
        self.c_from, self.t_from = (i32(m), i32(m)) if with_from else (None, None)
        self._rowsum = None
        self._struct = None

    def _init_from_device_lists(self, end_points_l, indptr_l, support_l, n_src):
        """Per-level lists that already live in HBM: fused on the device (sg_multilink_fuse_hip), no host round trip
        except the R level sizes."""
        R = len(end_points_l)
        eps, ips = [L.i32c(a) for a in end_points_l], [L.i32c(a) for a in indptr_l]
        sps = [L.f32c(a) for a in support_l]
        n_dst = ips[0].shape[0] - 1
        for a in ips:
            if a.shape[0] != n_dst + 1:
                raise L.StarGCNError("every level must carry a full-length indptr (n_dst+1)")
        nnz = int(torch.stack([a[-1] for a in ips]).sum().item())
        dev = eps[0].device
        self._alloc(R, n_dst, n_src, nnz, dev)
        lib = L.lib()
        ws, wsn = L.workspace(lib.sg_multilink_fuse_workspace_bytes(R, n_dst, n_src, nnz), dev)
        arr = ctypes.c_void_p * R
        L.check(lib.sg_multilink_fuse_hip(
            L.ptr(self.c_indptr), L.ptr(self.c_idx), L.ptr(self.c_q), L.ptr(self.c_w), L.ptr(self.t_indptr),
            L.ptr(self.t_idx), L.ptr(self.t_q), L.ptr(self.t_w), L.ptr(self.d_indptr), L.ptr(self.s_indptr),
            ctypes.cast(arr(*[a.data_ptr() for a in eps]), ctypes.c_void_p),
            ctypes.cast(arr(*[a.data_ptr() for a in ips]), ctypes.c_void_p),
            ctypes.cast(arr(*[a.data_ptr() for a in sps]), ctypes.c_void_p), R, n_dst, n_src, nnz, L.ptr(ws), wsn,
            L.stream_ptr()), "sg_multilink_fuse_hip")

    @classmethod
    def from_device_csr(cls, indptr, end_points, level, support, n_src, num_links, with_from=False):
        """Plan of the full-neighbourhood aggregation straight from a device-resident CSR (rows = destinations,
        end_points = sources, level[j] in [0, R), support[j]) -- sg_multilink_fuse_csr_hip.  with_from: also keep, for
        every slot of the two edge orders, the CSR edge id it holds (c_from / t_from)."""
        self = cls.__new__(cls)
        indptr, end_points, level = L.i32c(indptr), L.i32c(end_points), L.i32c(level)
        support = None if support is None else L.f32c(support)
        n_dst, nnz, dev = indptr.shape[0] - 1, end_points.shape[0], indptr.device
        self._alloc(num_links, n_dst, n_src, nnz, dev, with_from)
        lib = L.lib()
        ws, wsn = L.workspace(lib.sg_multilink_fuse_workspace_bytes(self.R, n_dst, int(n_src), nnz), dev)
        L.check(lib.sg_multilink_fuse_csr_hip(
            L.ptr(self.c_indptr), L.ptr(self.c_idx), L.ptr(self.c_q), L.ptr(self.c_w), L.ptr(self.t_indptr),
            L.ptr(self.t_idx), L.ptr(self.t_q), L.ptr(self.t_w), L.ptr(self.d_indptr), L.ptr(self.s_indptr),
            L.ptr(self.c_from), L.ptr(self.t_from), L.ptr(indptr), L.ptr(end_points), L.ptr(level), L.ptr(support),
            self.R, n_dst, int(n_src), nnz, L.ptr(ws), wsn, L.stream_ptr()), "sg_multilink_fuse_csr_hip")
        return self

    def c_struct(self, need_rowsum):
        """ctypes `sg_multilink_plan` view of the resident arrays (kept alive by this object)."""
        if self._struct is None or (need_rowsum and not self._struct.rowsum):
            st = L.MultiLinkPlanStruct()
            for name in ("c_indptr", "c_idx", "c_q", "c_w", "t_indptr", "t_idx", "t_q", "t_w", "d_indptr", "s_indptr"):
                setattr(st, name, getattr(self, name).data_ptr())
            st.rowsum = self.rowsum.data_ptr() if need_rowsum else None
            st.n_dst, st.n_src, st.nnz, st.num_links = self.n_dst, self.n_src, self.nnz, self.R
            st.struct_bytes = ctypes.sizeof(L.MultiLinkPlanStruct)
            for which, fp in enumerate(getattr(self, "_fused", None) or ()):
                e = st.fused[which]
                e.f_ptr, e.f_idx, e.f_w, e.tile_order = (t.data_ptr() for t in fp[:4])
            for view, ph in getattr(self, "_phases", {}).items():
                if ph is None:
                    continue
                idx_p, wpos_p, indptr_p, n0, n1 = ph
                e = st.phases[view]
                e.num_phases, e.idx, e.wpos, e.indptr = 2, idx_p.data_ptr(), wpos_p.data_ptr(), indptr_p.data_ptr()
                e.nnz_p[0], e.nnz_p[1] = n0, n1
            self._struct = st
        return self._struct

    # (index array, CSR pointers, segments, source rows) of the six gather views, by SG_VIEW_* (include/stargcn.h)
    _VIEWS = {
        L.VIEW_C_Q_D: ("c_q", "d_indptr", lambda p: p.n_dst, lambda p: p.n_src * p.R),
        L.VIEW_C_Q_C: ("c_q", "c_indptr", lambda p: p.n_dst * p.R, lambda p: p.n_src * p.R),
        L.VIEW_C_IDX_C: ("c_idx", "c_indptr", lambda p: p.n_dst * p.R, lambda p: p.n_src),
        L.VIEW_T_IDX_T: ("t_idx", "t_indptr", lambda p: p.n_src * p.R, lambda p: p.n_dst),
        L.VIEW_T_Q_T: ("t_q", "t_indptr", lambda p: p.n_src * p.R, lambda p: p.n_dst * p.R),
        L.VIEW_T_Q_S: ("t_q", "s_indptr", lambda p: p.n_src, lambda p: p.n_dst * p.R),
    }
    PHASE_MIN_EDGES = 1 << 20          # below this a launch is too short for two

    def ensure_phases(self, view):
        """Source-range phases of one gather view (sg_gather_phases_build_hip), built on the device and kept with the plan:
        the library then issues that view's gather as two launches over one half of the source rows each.
        WHICH view is worth building is the library's decision (ops._ensure_phases asks sg_multilink_agg_phased_view with
        the call's feature widths: the same width / footprint rule the launch applies), so a view that can never be phased
        costs no resident memory and no build.  Returns True when the view is settled (built, or already there), False when
        the build had to be skipped because a hipGraph is being captured on this stream (it allocates and reads two counters
        back): a captured step must have run eagerly once before -- bench.py and examples/ do -- or it replays single
        launches; a warning says so."""
        ph = self.__dict__.setdefault("_phases", {})
        if view in ph:
            return True
        if torch.cuda.is_current_stream_capturing():
            import warnings
            warnings.warn("MultiLinkPlan: source-range phases of gather view %d are not built yet and cannot be built during "
                          "stream capture; this graph will replay single launches (run one eager step first)" % view)
            return False
        idx_name, ip_name, segs, rows = self._VIEWS[view]
        n_seg, n_rows = int(segs(self)), int(rows(self))
        if self.nnz < self.PHASE_MIN_EDGES or n_rows < 2 or not self.c_idx.is_cuda:
            ph[view] = None
            return True
        dev = self.c_idx.device
        idx_p = torch.empty(self.nnz, dtype=torch.int32, device=dev)
        wpos_p = torch.empty(self.nnz, dtype=torch.int32, device=dev)
        indptr_p = torch.empty(2 * (n_seg + 1), dtype=torch.int32, device=dev)
        nnz_p = torch.empty(2, dtype=torch.int32, device=dev)
        lib = L.lib()
        ws, wsn = L.workspace(lib.sg_gather_phases_workspace_bytes(self.nnz), dev)
        L.check(lib.sg_gather_phases_build_hip(L.ptr(idx_p), L.ptr(wpos_p), L.ptr(indptr_p), L.ptr(nnz_p),
                                               L.ptr(getattr(self, idx_name)), L.ptr(getattr(self, ip_name)), n_seg,
                                               self.nnz, n_rows, L.ptr(ws), wsn, L.stream_ptr()),
                "sg_gather_phases_build_hip")
        n0, n1 = (int(v) for v in nnz_p.cpu())
        ph[view] = (idx_p, wpos_p, indptr_p, n0, n1)
        self._struct = None
        return True

    def ensure_fused(self, rebuild=False):
        """The two level-major edge orders the fused aggregate -> contract kernel walks (csrc/agg_fused.hip,
        sg_agg_fused_plan_build_hip): [0] over (c_indptr, c_idx, c_w) for the forward, [1] over (t_indptr, t_idx, t_w) for
        the data gradient; 12 bytes per edge each (index, weight, source position), resident with the plan.  The launch order of the 64-row tiles is by
        descending edge count (the persistent workgroups take the heavy tiles first).  Returns False when they are missing and
        cannot be built now (stream capture)."""
        if getattr(self, "_fused", None) is not None and not rebuild:
            return True
        old = getattr(self, "_fused", None)
        if old is None and torch.cuda.is_current_stream_capturing():
            return False        # the first build allocates and sorts; a rebuild (below) is one kernel launch per view into the same
        lib = L.lib()           # buffers and MUST run inside a captured iteration: the weights it copies were just rewritten
        built = []
        for which, (ip, idx, w, rows) in enumerate(((self.c_indptr, self.c_idx, self.c_w, self.n_dst),
                                                    (self.t_indptr, self.t_idx, self.t_w, self.n_src))):
            dev = idx.device
            tiles = int(lib.sg_agg_fused_tiles(rows))
            if old is not None:       # same graph, new weights (per-batch edge masking): one pass over the weights, not a re-plan
                f_ptr, f_idx, f_w, order, f_pos = old[which]
                L.check(lib.sg_agg_fused_refresh_hip(L.ptr(f_w), L.ptr(f_pos), L.ptr(w), self.nnz, L.stream_ptr()),
                        "sg_agg_fused_refresh_hip")
                built.append((f_ptr, f_idx, f_w, order, f_pos))
                continue
            else:
                R = self.R
                per_row = (ip[R::R] - ip[:-1:R]).to(torch.int64) if rows > 0 else torch.zeros(0, dtype=torch.int64, device=dev)
                work = torch.zeros(max(tiles, 1) * 64, dtype=torch.int64, device=dev)
                work[:rows] = per_row
                order = torch.argsort(work.view(-1, 64).sum(1), descending=True, stable=True).to(torch.int32)
                f_ptr = torch.empty(max(tiles, 1) * R * 65, dtype=torch.int32, device=dev)
                f_idx, f_w = torch.empty_like(idx), torch.empty_like(w)
                f_pos = torch.empty_like(idx)          # source position of every edge: what a later weight refresh reads through
            L.check(lib.sg_agg_fused_plan_build_hip(L.ptr(f_ptr), L.ptr(f_idx), L.ptr(f_w), L.ptr(f_pos), L.ptr(order), L.ptr(ip),
                                                    L.ptr(idx), L.ptr(w), rows, self.R, self.nnz, L.stream_ptr()),
                    "sg_agg_fused_plan_build_hip")
            built.append((f_ptr, f_idx, f_w, order, f_pos))
        self._fused = tuple(built)
        self._struct = None
        return True

    def prepare_phases(self, in_dim, units_per_level, order="auto", accum="sum"):
        """Build, NOW, the phases the forward and the backward call of an aggregator with these widths will use -- at plan
        construction or warm-up instead of inside the first timed / captured step (the build reads two counters back)."""
        from . import ops
        o, a = ops._ORDER[order], ops._ACCUM[accum]
        for backward in (0, 1):
            ops._ensure_phases(self, int(in_dim), int(units_per_level), o, a, backward)

    def refresh_rowsum(self):
        """Recompute `rowsum` IN PLACE after the weights were rewritten (resident edge masking); the tensor -- and
        the pointer inside the cached C struct -- stays the same."""
        if self._rowsum is not None and self.nnz > 0:
            from . import ops
            ops.seg_sum(self.c_w.view(1, -1), self.c_indptr, out=self._rowsum.view(1, -1))
        if getattr(self, "_fused", None) is not None:      # the fused kernel reads its own copy of the weights: same buffers, new values
            self.ensure_fused(rebuild=True)

    @property
    def rowsum(self):
        """(n_dst, R): sum of the support over each (node, level) segment -- the factor that scales the level bias
        when the contraction is applied after aggregation (A_r (X W^T + 1 b^T) = (A_r X) W^T + (A_r 1) b^T)."""
        if self._rowsum is None:
            from . import ops
            if self.nnz == 0:
                self._rowsum = torch.zeros((self.n_dst, self.R), dtype=torch.float32, device=self.device)
            else:
                self._rowsum = ops.seg_sum(self.c_w.view(1, -1), self.c_indptr).view(self.n_dst, self.R)
        return self._rowsum


class SourcePartition(object):
    """Edges of a gather plan (`indptr` over n_seg segments, `src_ids` = gathered row of every edge, `pos` = slot of the
    edge's weight, None = its own position) re-ordered by source-row range first: `parts` ranges holding about the same
    number of edges each.  Consumed by ops.gather_sum_parts (sg_seg_gather_sum_parts_hip): with 8 parts every XCD gathers
    from ONE range, which fits its private L2.  Built on the device: degree histogram + running sum + searchsorted for the
    range bounds, sg_part_keys_hip, one stable radix sort, sg_bounds_from_sorted_hip, two index gathers."""

    def __init__(self, indptr, src_ids, n_src_rows, pos=None, parts=8):
        indptr, src_ids = L.i32c(indptr), L.i32c(src_ids)
        L.require_gpu(indptr, src_ids)
        dev = indptr.device
        lib = L.lib()
        n, n_seg = int(src_ids.numel()), int(indptr.numel() - 1)
        self.parts, self.n_seg, self.nnz = int(parts), n_seg, n
        counts = torch.zeros(int(n_src_rows), dtype=torch.int32, device=dev)
        L.check(lib.sg_count_indices_hip(L.ptr(counts), L.ptr(src_ids), n, int(n_src_rows), L.stream_ptr()),
                "sg_count_indices_hip")
        run = torch.cumsum(counts, 0)
        want = (torch.arange(parts + 1, device=dev, dtype=torch.int64) * n) // parts
        self.bounds = torch.searchsorted(run, want.to(run.dtype), right=False).to(torch.int32).contiguous()
        keys = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        L.check(lib.sg_part_keys_hip(L.ptr(keys), L.ptr(src_ids), L.ptr(indptr), L.ptr(self.bounds), parts, n_seg, n,
                                     L.stream_ptr()), "sg_part_keys_hip")
        skeys, order = torch.empty_like(keys), torch.empty_like(keys)
        ws, wsn = L.workspace(lib.sg_sort_i32_workspace_bytes(n), dev)
        L.check(lib.sg_sort_i32_hip(L.ptr(skeys), L.ptr(order), L.ptr(keys), None, n, parts * n_seg, L.ptr(ws), wsn,
                                    L.stream_ptr()), "sg_sort_i32_hip")
        self.indptr = torch.empty(parts * n_seg + 1, dtype=torch.int32, device=dev)
        L.check(lib.sg_bounds_from_sorted_hip(L.ptr(self.indptr), L.ptr(skeys), n, parts * n_seg, L.stream_ptr()),
                "sg_bounds_from_sorted_hip")
        self.src = torch.empty_like(keys)
        L.check(lib.sg_gather_i32_hip(L.ptr(self.src), L.ptr(src_ids), L.ptr(order), n, L.stream_ptr()), "sg_gather_i32_hip")
        if pos is None:
            self.pos = order
        else:
            self.pos = torch.empty_like(keys)
            L.check(lib.sg_gather_i32_hip(L.ptr(self.pos), L.ptr(L.i32c(pos)), L.ptr(order), n, L.stream_ptr()),
                    "sg_gather_i32_hip")


def upload_packed(arrays, device):
    """ONE host->device copy for several int32 arrays (each padded to a 16-byte multiple); returns device views.
    A per-batch plan used to cost one ~0.2 ms pageable copy per array."""
    sizes = [int(a.size) for a in arrays]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 3) & ~3
    buf = np.zeros(max(total, 1), np.int32)
    for a, o, n in zip(arrays, offs, sizes):
        buf[o:o + n] = a.reshape(-1)
    dev = torch.from_numpy(buf).to(device)
    return [dev[o:o + n] for o, n in zip(offs, sizes)]


class TakePlan(object):
    """Row gather `out[i] = table[ids[i]]` (ids == -1 -> zero row) with an atomic-free gradient: when every row is
    taken at most once (permutations, subsets) the gradient is a row copy through the inverse index `inv_ids`
    (-1 = row not taken -> zero); otherwise the transposed plan groups the positions i by id, so
    d table[n] = sum of dout rows in segment n (gather kernel again).  Built by native code (sg_take_plan_cpu),
    uploaded with one copy."""

    @classmethod
    def identity_plan(cls, n):
        """take of rows 0..n-1 in order: costs nothing forward or backward (no index arrays at all)."""
        self = cls.__new__(cls)
        self.n = self.n_rows = self.covered = int(n)
        self.identity = True
        self.ids = self.inv_ids = self.t_indptr = self.t_pos = None
        return self

    @classmethod
    def from_device_unique(cls, ids, n_rows):
        """ids: int32 DEVICE tensor whose non-negative entries are distinct (-1 = zero row): the gradient is a row copy
        through the inverse index, built on the device (sg_inverse_index_hip) -- no host round trip."""
        self = cls.__new__(cls)
        ids = L.i32c(ids).reshape(-1)
        self.n, self.n_rows, self.covered = int(ids.shape[0]), int(n_rows), None
        self.identity = False
        self.ids = ids
        self.inv_ids = torch.empty(max(self.n_rows, 1), dtype=torch.int32, device=ids.device)
        L.check(L.lib().sg_inverse_index_hip(L.ptr(self.inv_ids), L.ptr(ids), self.n, self.n_rows, L.stream_ptr()),
                "sg_inverse_index_hip")
        self.inv_ids = self.inv_ids[:self.n_rows]
        self.t_indptr = self.t_pos = None
        return self

    def __init__(self, ids, n_rows, device):
        ids = _np_i32(ids).reshape(-1)
        n = ids.shape[0]
        self.n, self.n_rows = n, int(n_rows)
        t_indptr = np.empty(self.n_rows + 1, np.int32)
        t_pos = np.empty(max(n, 1), np.int32)
        inv = np.empty(max(self.n_rows, 1), np.int32)
        flags, covered = ctypes.c_int32(0), ctypes.c_int64(0)
        L.check(L.lib().sg_take_plan_cpu(_vp(t_indptr), _vp(t_pos), _vp(inv), ctypes.byref(flags), ctypes.byref(covered),
                                         _vp(ids), n, self.n_rows), "sg_take_plan_cpu")
        # identity takes (full-graph plans: unique ids are already 0..n-1 in order) cost nothing
        self.identity = bool(flags.value & 1)
        self.covered = int(covered.value)
        self.inv_ids = self.t_indptr = self.t_pos = None
        if flags.value & 2:
            self.ids, self.inv_ids = upload_packed([ids, inv[:self.n_rows]], device)
        else:
            self.ids, self.t_indptr, self.t_pos = upload_packed([ids, t_indptr, t_pos], device)
