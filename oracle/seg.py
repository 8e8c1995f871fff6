"""ctypes binding of oracle/libseg_oracle.so (see seg_oracle.c).  TEST INFRASTRUCTURE ONLY.

All functions take/return numpy arrays (float32 / int32, C-contiguous) and mirror the operator
names of the reference (`mx.nd.contrib.seg_*`, reference seg_op.cc:339-861).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libseg_oracle.so")
REQ_NULL, REQ_WRITE, REQ_ADD = 0, 1, 3

_lib = None


def build(force=False):
    """Compile seg_oracle.c with gcc (called by __graft_entry__.build())."""
    src = os.path.join(_HERE, "seg_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libseg_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _i64(x):
    return ctypes.c_int64(int(x))


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError("oracle %s failed with code %d" % (name, rc))


def seg_reduce(data, indptr, reduce_type=0, out=None, req=REQ_WRITE):
    data, indptr = _f(data), _i(indptr)
    B, nnz = data.shape
    S = indptr.shape[0] - 1
    dst = np.zeros((B, S), np.float32) if out is None else out
    _chk(lib().oracle_seg_reduce(_p(dst), _p(data), _p(indptr), _i64(B), _i64(S), _i64(nnz),
                                 int(reduce_type), int(req)), "seg_reduce")
    return dst


def seg_sum(data, indptr, **kw):
    return seg_reduce(data, indptr, 0, **kw)


def seg_broadcast(lhs, rhs, indptr, op, nnz=None, out=None, req=REQ_WRITE):
    rhs, indptr = _f(rhs), _i(indptr)
    if lhs is not None:
        lhs = _f(lhs)
        nnz = lhs.shape[1]
    B, S = rhs.shape
    dst = np.zeros((B, nnz), np.float32) if out is None else out
    _chk(lib().oracle_seg_broadcast(_p(dst), _p(lhs), _p(rhs), _p(indptr), _i64(B), _i64(S), _i64(nnz),
                                    int(op), int(req)), "seg_broadcast")
    return dst


def seg_broadcast_add(lhs, rhs, indptr, **kw):
    return seg_broadcast(lhs, rhs, indptr, 0, **kw)


def seg_broadcast_mul(lhs, rhs, indptr, **kw):
    return seg_broadcast(lhs, rhs, indptr, 1, **kw)


def seg_broadcast_to(rhs, indptr, nnz, **kw):
    return seg_broadcast(None, rhs, indptr, 2, nnz=nnz, **kw)


def seg_softmax(data, indptr):
    data, indptr = _f(data), _i(indptr)
    B, nnz = data.shape
    S = indptr.shape[0] - 1
    dst = np.empty((B, nnz), np.float32)
    _chk(lib().oracle_seg_softmax(_p(dst), _p(data), _p(indptr), _i64(B), _i64(S), _i64(nnz), REQ_WRITE),
         "seg_softmax")
    return dst


def seg_softmax_bwd(ograd, val, indptr, out=None, req=REQ_WRITE):
    ograd, val, indptr = _f(ograd), _f(val), _i(indptr)
    B, nnz = ograd.shape
    S = indptr.shape[0] - 1
    dst = np.zeros((B, nnz), np.float32) if out is None else out
    _chk(lib().oracle_seg_softmax_bwd(_p(dst), _p(ograd), _p(val), _p(indptr), _i64(B), _i64(S), _i64(nnz),
                                      int(req)), "seg_softmax_bwd")
    return dst


def seg_take_k_corr(embed1, embed2, neighbor_ids, neighbor_indptr, out=None, req=REQ_WRITE):
    embed1, embed2 = _f(embed1), _f(embed2)
    ids, indptr = _i(neighbor_ids), _i(neighbor_indptr)
    K, N, C = embed1.shape
    M = embed2.shape[1]
    nnz = ids.shape[0]
    dst = np.zeros((K, nnz), np.float32) if out is None else out
    _chk(lib().oracle_seg_take_k_corr(_p(dst), _p(embed1), _p(embed2), _p(ids), _p(indptr), _i64(K), _i64(N),
                                      _i64(M), _i64(nnz), _i64(C), int(req)), "seg_take_k_corr")
    return dst


def seg_weighted_pool(data, weights, indices, indptr, out=None, req=REQ_WRITE):
    data, weights = _f(data), _f(weights)
    indices, indptr = _i(indices), _i(indptr)
    B, T, C = data.shape
    nnz = indices.shape[0]
    S = indptr.shape[0] - 1
    dst = np.zeros((B, S, C), np.float32) if out is None else out
    _chk(lib().oracle_seg_weighted_pool(_p(dst), _p(data), _p(weights), _p(indices), _p(indptr), _i64(B),
                                        _i64(S), _i64(T), _i64(nnz), _i64(C), int(req)), "seg_weighted_pool")
    return dst


def seg_weighted_pool_bwd_data(weights, ograd, indices, indptr, total_ind_num, out=None, req=REQ_WRITE,
                               fair=False):
    weights, ograd = _f(weights), _f(ograd)
    indices, indptr = _i(indices), _i(indptr)
    B, S, C = ograd.shape
    nnz = indices.shape[0]
    T = int(total_ind_num)
    dst = np.zeros((B, T, C), np.float32) if out is None else out
    fn = lib().oracle_seg_weighted_pool_bwd_data_fair if fair else lib().oracle_seg_weighted_pool_bwd_data
    _chk(fn(_p(dst), _p(weights), _p(ograd), _p(indices), _p(indptr), _i64(B), _i64(S), _i64(T), _i64(nnz),
            _i64(C), int(req)), "seg_weighted_pool_bwd_data")
    return dst


_POOL = {"sum": 0, "avg": 1, "max": 2}


def seg_pool(data, indices, indptr, pool_type):
    data = _f(data)
    indices, indptr = _i(indices), _i(indptr)
    B, T, C = data.shape
    nnz = indices.shape[0]
    S = indptr.shape[0] - 1
    dst = np.empty((B, S, C), np.float32)
    arg = np.empty((B, S, C), np.int32) if pool_type == "max" else None
    _chk(lib().oracle_seg_pool(_p(dst), _p(arg), _p(data), _p(indices), _p(indptr), _i64(B), _i64(S), _i64(T),
                               _i64(nnz), _i64(C), _POOL[pool_type], REQ_WRITE), "seg_pool")
    return (dst, arg) if pool_type == "max" else dst


def seg_pool_bwd(ograd, pool_indices, indices, indptr, total_ind_num, pool_type, out=None, req=REQ_WRITE):
    ograd = _f(ograd)
    indices, indptr = _i(indices), _i(indptr)
    B, S, C = ograd.shape
    nnz = indices.shape[0]
    T = int(total_ind_num)
    dst = np.zeros((B, T, C), np.float32) if out is None else out
    pi = _i(pool_indices) if pool_indices is not None else None
    _chk(lib().oracle_seg_pool_bwd(_p(dst), _p(ograd), _p(pi), _p(indices), _p(indptr), _i64(B), _i64(S),
                                   _i64(T), _i64(nnz), _i64(C), _POOL[pool_type], int(req)), "seg_pool_bwd")
    return dst


def get_support(row_degrees, col_degrees, end_points, ind_ptr, symm=True):
    rd, cd, ep, ip = _i(row_degrees), _i(col_degrees), _i(end_points), _i(ind_ptr)
    out = np.empty(ep.shape[0], np.float32)
    _chk(lib().oracle_get_support(_p(out), _p(rd), _p(cd), _p(ep), _p(ip), _i64(ip.shape[0] - 1), int(symm)),
         "get_support")
    return out


def multi_link_split(values, ind_ptr, multi_link):
    """-> (list of per-level edge-position arrays, list of per-level indptr arrays)."""
    values, multi_link, ip = _f(values), _f(multi_link), _i(ind_ptr)
    N, L = ip.shape[0] - 1, multi_link.shape[0]
    nnz = int(ip[-1])
    pos = np.empty(max(nnz, 1), np.int32)
    indptrs = np.empty((L, N + 1), np.int32)
    off = np.empty(L + 1, np.int64)
    _chk(lib().oracle_multi_link_split(_p(pos), _p(indptrs), _p(off), _p(values), _p(ip), _p(multi_link),
                                       _i64(N), _i64(L)), "multi_link_split")
    return [pos[off[l]:off[l + 1]].copy() for l in range(L)], [indptrs[l].copy() for l in range(L)]
