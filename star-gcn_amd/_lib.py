"""ctypes binding of libstargcn_hip.so (C ABI: include/stargcn.h).

There is deliberately NO fallback: if the shared library is missing, or a HIP device is not available when
an operator is called, an exception is raised.  PyTorch is used only as plumbing (device memory, streams).
"""
import collections
import ctypes
import os
import subprocess

import torch  # imported first so libamdhip64 (SONAME libamdhip64.so.7) is the one torch ships

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("SG_LIB_OVERRIDE") or os.path.join(CSRC, "libstargcn_hip.so")   # override: ablation builds (tools/)

REQ_NULL, REQ_WRITE, REQ_ADD = 0, 1, 3
POOL = {"sum": 0, "avg": 1, "max": 2}
ACT = {None: 0, "identity": 0, "none": 0, "leaky": 1, "relu": 2, "sigmoid": 3, "tanh": 4}


class StarGCNError(RuntimeError):
    """Raised when a libstargcn_hip entry point returns a negative status (message = sg_last_error())."""


def build(force=False, verbose=False):
    """Compile csrc/*.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j8"]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=None if verbose else subprocess.DEVNULL)
    return SO_PATH


_c = ctypes
_P, _I64, _INT, _F32, _SZ = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_float, _c.c_size_t

# name -> (restype, argtypes)   -- kept in the order of include/stargcn.h
_SIGS = {
    "sg_last_error": (_c.c_char_p, []),
    "sg_version": (_INT, []),
    "sg_device_count": (_INT, []),
    "sg_seg_weighted_pool_workspace_bytes": (_SZ, [_I64] * 4),
    "sg_seg_weighted_pool_hip": (_INT, [_P] * 5 + [_I64] * 5 + [_INT, _P, _SZ, _P]),
    "sg_seg_gather_sum_hip": (_INT, [_P, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _INT, _INT, _F32,
                                     _P, _SZ, _P]),
    "sg_seg_gather_sum_hinted_hip": (_INT, [_P, _I64, _I64, _P, _I64, _I64, _P, _P, _P, _I64, _I64, _I64, _INT, _INT,
                                            _F32, _P, _SZ, _P, _I64]),
    "sg_seg_weighted_pool_bwd_data_workspace_bytes": (_SZ, [_I64] * 4),
    "sg_seg_weighted_pool_bwd_data_hip": (_INT, [_P] * 6 + [_I64] * 5 + [_INT, _P, _SZ, _P]),
    "sg_build_transpose_cpu": (_INT, [_P] * 5 + [_I64] * 3),
    "sg_seg_take_k_corr_hip": (_INT, [_P] * 5 + [_I64] * 5 + [_INT, _P]),
    "sg_seg_sum_hip": (_INT, [_P] * 3 + [_I64] * 3 + [_INT, _P]),
    "sg_seg_broadcast_hip": (_INT, [_P] * 4 + [_I64] * 3 + [_INT, _INT, _P]),
    "sg_seg_softmax_hip": (_INT, [_P] * 3 + [_I64] * 3 + [_INT, _P]),
    "sg_seg_softmax_bwd_hip": (_INT, [_P] * 4 + [_I64] * 3 + [_INT, _P]),
    "sg_seg_pool_workspace_bytes": (_SZ, [_I64] * 4),
    "sg_seg_pool_hip": (_INT, [_P] * 5 + [_I64] * 5 + [_INT, _INT, _P, _SZ, _P]),
    "sg_seg_pool_bwd_workspace_bytes": (_SZ, [_I64] * 4),
    "sg_seg_pool_bwd_hip": (_INT, [_P] * 7 + [_I64] * 5 + [_INT, _INT, _P, _SZ, _P]),
    "sg_gemm_f32_workspace_bytes": (_SZ, [_I64, _I64, _I64, _INT]),
    "sg_gemm_f32_hip": (_INT, [_P, _I64, _P, _I64, _INT, _P, _I64, _INT, _I64, _I64, _I64, _P, _INT, _F32, _INT,
                               _P, _SZ, _P]),
    "sg_act_hip": (_INT, [_P, _P, _I64, _INT, _F32, _P]),
    "sg_act_bwd_hip": (_INT, [_P, _P, _P, _I64, _INT, _F32, _P]),
    "sg_colsum_workspace_bytes": (_SZ, [_I64, _I64]),
    "sg_colsum_hip": (_INT, [_P, _P, _I64, _I64, _I64, _INT, _P, _SZ, _P]),
    "sg_act_bwd_colsum_hip": (_INT, [_P, _P, _P, _P, _I64, _I64, _INT, _F32, _INT, _P, _SZ, _P]),
    "sg_masked_embed_hip": (_INT, [_P] * 4 + [_I64] * 3 + [_P]),
    "sg_resolve_ids_hip": (_INT, [_P] * 3 + [_I64, _P]),
    "sg_get_support_cpu": (_INT, [_P] * 5 + [_I64, _INT]),
    "sg_multi_link_split_cpu": (_INT, [_P] * 6 + [_I64, _I64]),
    "sg_multilink_fuse_cpu": (_INT, [_P] * 11 + [_I64] * 3),
    "sg_unique_inverse_cpu": (_INT, [_P] * 5 + [_I64, _I64]),
    "sg_remove_edges_cpu": (_INT, [_P] * 7 + [_I64, _P, _P, _I64]),
    "sg_csr_submat_cpu": (_INT, [_P] * 7 + [_I64, _P, _I64, _P]),
    "sg_sample_fix_neighbor_cpu": (_INT, [_P] * 4 + [_I64, _I64, _c.c_uint64]),
    "sg_gen_row_indices_cpu": (_INT, [_P, _P, _I64, _I64]),
    "sg_edge_positions_cpu": (_INT, [_P] * 3 + [_I64, _P, _P, _I64]),
    "sg_pair_plan_cpu": (_INT, [_P] * 10 + [_I64] * 3),
    "sg_take_plan_cpu": (_INT, [_P] * 6 + [_I64] * 2),
    "sg_mask_edges_workspace_bytes": (_SZ, [_I64] * 3),
    "sg_mask_edges_hip": (_INT, [_P, _P, _P, _c.c_int32] + [_P] * 5 + [_I64] * 4 + [_INT, _P, _SZ, _P]),
    "sg_l2_loss_workspace_bytes": (_SZ, [_I64]),
    "sg_l2_loss_hip": (_INT, [_P] * 4 + [_I64, _F32, _P, _SZ, _P]),
    "sg_gather_profile_enable": (_INT, [_INT]),
    "sg_gather_profile_read": (_I64, [_P, _P, _P, _I64]),
    "sg_gather_profile_read2": (_I64, [_P, _P, _P, _P, _I64]),
    "sg_gather_tuning": (_INT, [_INT, _INT]),
    "sg_gemm_backend": (_INT, [_INT]),
    "sg_gemm_x3_variant": (_INT, [_INT]),
    "sg_gemm_profile_enable": (_INT, [_INT]),
    "sg_gemm_profile_read": (_I64, [_P, _P, _P, _I64]),
    "sg_stream_read_hip": (_INT, [_P, _I64, _INT, _I64, _P, _P]),
    "sg_stream_read_strided_hip": (_INT, [_P, _I64, _INT, _I64, _I64, _P, _P]),
    "sg_build_transpose_workspace_bytes": (_SZ, [_I64] * 3),
    "sg_build_transpose_hip": (_INT, [_P] * 5 + [_I64] * 3 + [_P, _SZ, _P]),
    "sg_seg_weighted_pool_bwd_data_dev_workspace_bytes": (_SZ, [_I64] * 5),
    "sg_seg_weighted_pool_bwd_data_dev_hip": (_INT, [_P] * 5 + [_I64] * 5 + [_INT, _P, _SZ, _P]),
    "sg_multilink_fuse_workspace_bytes": (_SZ, [_I64] * 4),
    "sg_multilink_fuse_hip": (_INT, [_P] * 13 + [_I64] * 4 + [_P, _SZ, _P]),
    "sg_multilink_fuse_csr_hip": (_INT, [_P] * 16 + [_I64] * 4 + [_P, _SZ, _P]),
    "sg_pair_l2_workspace_bytes": (_SZ, [_I64] * 4),
    "sg_pair_l2_hip": (_INT, [_P] * 7 + [_I64] * 4 + [_F32, _P, _F32, _INT, _P, _SZ, _P, _I64]),
    "sg_part_keys_hip": (_INT, [_P] * 4 + [_I64] * 3 + [_P]),
    "sg_seg_gather_sum_parts_workspace_bytes": (_SZ, [_I64] * 4),
    "sg_seg_gather_sum_parts_hip": (_INT, [_P, _I64, _I64, _P, _I64, _I64] + [_P] * 4 + [_I64] * 4
                                    + [_INT, _INT, _F32, _P, _SZ, _P, _I64]),
    "sg_gen_row_indices_hip": (_INT, [_P, _P, _I64, _I64, _P]),
    "sg_count_indices_hip": (_INT, [_P, _P, _I64, _I64, _P]),
    "sg_get_support_hip": (_INT, [_P] * 5 + [_I64, _INT, _P]),
    "sg_level_index_hip": (_INT, [_P] * 3 + [_I64, _I64, _P]),
    "sg_sample_distinct_hip": (_INT, [_P, _I64, _I64, _c.c_uint64, _c.c_uint64, _P]),
    "sg_recon_mask_hip": (_INT, [_P, _P, _I64, _I64, _F32, _c.c_uint64, _c.c_uint64, _P]),
    "sg_sample_distinct_dev_hip": (_INT, [_P, _I64, _I64, _c.c_uint64, _c.c_uint64, _P, _P]),
    "sg_recon_mask_dev_hip": (_INT, [_P, _P, _I64, _I64, _F32, _c.c_uint64, _c.c_uint64, _P, _P]),
    "sg_recon_mask_cand_dev_hip": (_INT, [_P, _P, _I64, _P, _I64, _I64, _F32, _c.c_uint64, _c.c_uint64, _P, _P]),
    "sg_counter_add_hip": (_INT, [_P, _c.c_uint64, _P]),
    "sg_sort_i32_workspace_bytes": (_SZ, [_I64]),
    "sg_sort_i32_hip": (_INT, [_P] * 4 + [_I64, _I64, _P, _SZ, _P]),
    "sg_bounds_from_sorted_hip": (_INT, [_P, _P, _I64, _I64, _P]),
    "sg_gather_i32_hip": (_INT, [_P, _P, _P, _I64, _P]),
    "sg_inverse_index_hip": (_INT, [_P, _P, _I64, _I64, _P]),
    "sg_gather_phases_workspace_bytes": (_SZ, [_I64]),
    "sg_gather_phases_build_hip": (_INT, [_P] * 6 + [_I64] * 3 + [_P, _SZ, _P]),
    "sg_seg_gather_sum_phased_hip": (_INT, [_P, _I64, _I64, _P, _I64, _I64, _P, _P, _I64, _I64, _INT, _INT, _F32, _P, _SZ,
                                            _P, _I64]),
    "sg_unique_inverse_workspace_bytes": (_SZ, [_I64, _I64]),
    "sg_unique_inverse_hip": (_INT, [_P] * 6 + [_I64, _I64, _P, _SZ, _P]),
    "sg_sample_fix_neighbor_workspace_bytes": (_SZ, [_I64]),
    "sg_sample_fix_neighbor_hip": (_INT, [_P] * 4 + [_I64, _I64, _c.c_uint64, _P, _SZ, _P]),
    "sg_multilink_agg_resolve_order": (_INT, [_P, _INT]),
    "sg_multilink_agg_resolve_order2": (_INT, [_P, _INT, _I64, _I64, _INT]),
    "sg_multilink_agg_phased_view": (_INT, [_P, _I64, _I64, _INT, _INT, _INT]),
    "sg_multilink_agg_saved_bytes": (_SZ, [_P, _I64, _I64, _INT, _INT]),
    "sg_multilink_agg_workspace_bytes": (_SZ, [_P, _I64, _I64, _INT, _INT, _INT]),
    "sg_multilink_agg_fwd_hip": (_INT, [_P] * 6 + [_I64, _I64, _INT, _INT, _INT, _F32, _P, _SZ, _P]),
    "sg_multilink_agg_bwd_hip": (_INT, [_P] * 9 + [_I64, _I64, _INT, _INT, _INT, _F32, _P, _SZ, _P]),
    "sg_agg_fused_tiles": (_I64, [_I64]),
    "sg_agg_fused_supported": (_INT, [_I64, _I64, _c.c_int32]),
    "sg_agg_fused_plan_build_hip": (_INT, [_P] * 8 + [_I64, _c.c_int32, _I64, _P]),
    "sg_agg_fused_refresh_hip": (_INT, [_P, _P, _P, _I64, _P]),
    "sg_agg_fused_workspace_bytes": (_SZ, [_c.c_int32]),
    "sg_agg_fused_hip": (_INT, [_P, _I64, _P, _I64, _P, _I64, _P, _I64, _INT, _P, _P, _P, _P, _P, _P, _I64, _I64, _c.c_int32,
                                _I64, _I64, _I64, _INT, _F32, _INT, _P, _SZ, _P]),
    # out ldo zsave ldz x ldx x_level_stride weights ldw trans_w k_valid biases rowsum f_ptr f_idx f_w tile_order n_dst n_src R nnz
    # in_dim out_dim accum act slope nt_loads workspace bytes stream
    "sg_agg_fused2_hip": (_INT, [_P, _I64, _P, _I64, _P, _I64, _I64, _P, _I64, _INT, _I64, _P, _P, _P, _P, _P, _P, _I64, _I64,
                                 _c.c_int32, _I64, _I64, _I64, _INT, _INT, _F32, _INT, _P, _SZ, _P]),
    "sg_agg_fused_profile_enable": (_INT, [_INT]),
    "sg_agg_fused_profile_read": (_I64, [_P, _P, _P, _I64]),
}


NUM_VIEWS = 6      # SG_VIEW_* of include/stargcn.h
VIEW_C_Q_D, VIEW_C_Q_C, VIEW_C_IDX_C, VIEW_T_IDX_T, VIEW_T_Q_T, VIEW_T_Q_S = range(NUM_VIEWS)


class GatherPhasesStruct(_c.Structure):
    """`sg_gather_phases` of include/stargcn.h."""
    _fields_ = [("num_phases", _c.c_int32), ("reserved", _c.c_int32), ("idx", _P), ("wpos", _P), ("indptr", _P),
                ("nnz_p", _I64 * 2)]


class FusedPlanStruct(_c.Structure):
    """`sg_fused_plan` of include/stargcn.h."""
    _fields_ = [("f_ptr", _P), ("f_idx", _P), ("f_w", _P), ("tile_order", _P)]


class MultiLinkPlanStruct(_c.Structure):
    """`sg_multilink_plan` of include/stargcn.h (device pointers of a resident MultiLinkPlan)."""
    _fields_ = [(n, _P) for n in ("c_indptr", "c_idx", "c_q", "c_w", "t_indptr", "t_idx", "t_q", "t_w", "d_indptr",
                                  "s_indptr", "rowsum")] + \
               [("n_dst", _I64), ("n_src", _I64), ("nnz", _I64), ("num_links", _c.c_int32), ("struct_bytes", _c.c_int32),
                ("phases", GatherPhasesStruct * NUM_VIEWS), ("fused", FusedPlanStruct * 2)]

_lib = None


def lib():
    """The loaded library (loads on first use; raises ImportError with the build hint if absent)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError("libstargcn_hip.so is missing (%s). Build it with `python -c 'import "
                              "__graft_entry__ as g; g.build()'` or `make -C star-gcn_amd/csrc`. There is no "
                              "CPU fallback." % SO_PATH)
        handle = ctypes.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def exported_symbols():
    return list(_SIGS)


def check(rc, what=""):
    if rc != 0:
        msg = lib().sg_last_error()
        raise StarGCNError("%s failed (%d): %s" % (what or "libstargcn_hip", rc, msg.decode() if msg else "?"))


def require_gpu(*tensors):
    if not torch.cuda.is_available():
        raise StarGCNError("no HIP device available: the STAR-GCN hot path only runs on the GPU (no CPU fallback)")
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise StarGCNError("expected a CUDA/HIP tensor, got a %s tensor" % t.device)


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_ws_cache = collections.OrderedDict()
_WS_STREAMS_PER_DEVICE = 4       # (stream) scratch buffers kept per DEVICE; older ones of that device are released


def workspace(nbytes, device):
    """Per-(device, stream) scratch buffer; kernels are stream-ordered so reuse across calls is safe.  At most
    _WS_STREAMS_PER_DEVICE buffers are kept per device (least recently used first out): a process that creates many
    streams -- or one that ran a config-5-sized step on a side stream once -- must not pin a multi-GB buffer per stream
    for ever, and a process that drives eight GPUs must not evict seven of them on every call (the cap used
    to be global).  Dropping an entry only drops this module's reference: the block goes back to torch's allocator pool
    of the stream it was allocated on, which re-issues it in that stream's order.
    While a hipGraph is being captured on the current stream nothing is evicted or replaced-and-dropped: a captured kernel
    may hold the address of a buffer handed out earlier in the capture, and a block returned to the pool could be handed
    out again before the graph is replayed.  Buffers that grow during a capture are kept alive in `_ws_pinned`."""
    if nbytes <= 0:
        return None, 0
    dev_index = device.index if device.index is not None else torch.cuda.current_device()
    key = (dev_index, torch.cuda.current_stream().cuda_stream)
    capturing = torch.cuda.is_current_stream_capturing()
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        if buf is not None and capturing:
            _ws_pinned.append(buf)          # still referenced by already-captured kernels
        buf = None
        _ws_cache.pop(key, None)            # release the smaller buffer BEFORE allocating the larger one
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
        if not capturing:
            mine = [k for k in _ws_cache if k[0] == dev_index]
            for k in mine[:max(0, len(mine) - _WS_STREAMS_PER_DEVICE)]:
                del _ws_cache[k]
    else:
        _ws_cache.move_to_end(key)
    return buf, buf.numel()


_ws_pinned = []


def release_workspaces():
    """Drop every cached scratch buffer (after a one-off large problem: the next call allocates what IT needs).  Not while a
    captured hipGraph that used them is still going to be replayed."""
    _ws_cache.clear()
    del _ws_pinned[:]


def f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def i32c(t):
    if t.dtype != torch.int32:
        t = t.to(torch.int32)
    return t if t.is_contiguous() else t.contiguous()
