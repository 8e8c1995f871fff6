"""ctypes binding of oracle/_ref/libgs_ref.so = the REFERENCE's GraphSampler C++ core compiled from its own sources
(oracle/Makefile target `_ref`, oracle/gs_ref_wrap.cpp).  TEST INFRASTRUCTURE ONLY: imported by tests/, by the golden
generators under tests/golden/ and by __graft_entry__ -- never by the product.

`GraphSamplerRef` plays the role of the reference's CPython extension module `mxgraph._graph_sampler`: the 13 functions
of its method table (py_ext.cpp:612-627) with the same positional arguments, dtype checks and return tuples, so the
reference's own `mxgraph/graph.py` runs on top of it unmodified (tests/golden/make_graph_golden.py).

`available()` is False where the library was not built (no /root/reference and no prebuilt file in the snapshot); the
committed fixtures under tests/golden/ carry its outputs there.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "_ref")
SO_IEEE = os.path.join(_DIR, "libgs_ref.so")
SO_FASTMATH = os.path.join(_DIR, "libgs_ref_fastmath.so")

_libs = {}


def build():
    """Run the committed recipe (a no-op where /root/reference is absent)."""
    subprocess.check_call(["make", "-C", _HERE, "-s", "_ref"])
    return SO_IEEE if os.path.exists(SO_IEEE) else None


def available(fastmath=False):
    return os.path.exists(SO_FASTMATH if fastmath else SO_IEEE)


def _lib(fastmath=False):
    path = SO_FASTMATH if fastmath else SO_IEEE
    if path not in _libs:
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref is not built (run `make -C oracle _ref` where /root/reference exists)")
        L = ctypes.CDLL(path)
        L.gsr_result_int_size.restype = ctypes.c_longlong
        L.gsr_result_float_size.restype = ctypes.c_longlong
        for name in ("gsr_random_sample_fix_neighbor", "gsr_csr_submat", "gsr_unique_cnt", "gsr_unique_inverse",
                     "gsr_remove_edges", "gsr_multi_link_split"):
            getattr(L, name).restype = ctypes.c_void_p
        _libs[path] = L
    return _libs[path]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _need(a, dtype, name):
    """py_ext.cpp's PY_CHECK_EQUAL(PyArray_TYPE(x), NPY_*) / PY_CHECK_CONTIGUOUS."""
    if not isinstance(a, np.ndarray) or a.dtype != dtype:
        raise TypeError("%s must be a numpy array of dtype %s (got %r)" % (name, np.dtype(dtype).name,
                                                                            getattr(a, "dtype", type(a))))
    if not a.flags.c_contiguous:
        raise ValueError("%s must be C-contiguous" % name)
    return a


class GraphSamplerRef(object):
    def __init__(self, fastmath=False):
        self._L = _lib(fastmath)

    # -- result handles ---------------------------------------------------------------------------------------------
    def _take(self, handle):
        L, h = self._L, ctypes.c_void_p(handle)
        ints, floats = [], []
        for k in range(L.gsr_result_num_int(h)):
            a = np.empty(L.gsr_result_int_size(h, k), np.int32)
            L.gsr_result_int_copy(h, k, _p(a))
            ints.append(a)
        for k in range(L.gsr_result_num_float(h)):
            a = np.empty(L.gsr_result_float_size(h, k), np.float32)
            L.gsr_result_float_copy(h, k, _p(a))
            floats.append(a)
        L.gsr_result_free(h)
        return ints, floats

    # -- the method table of py_ext.cpp:612-627 ---------------------------------------------------------------------
    def set_seed(self, seed):
        self._L.gsr_set_seed(int(seed))
        return 1

    def random_sample_fix_neighbor(self, src_ind_ptr, sel_indices, neighbor_num):
        _need(src_ind_ptr, np.int32, "src_ind_ptr"), _need(sel_indices, np.int32, "sel_indices")
        (pos, ptr), _ = self._take(self._L.gsr_random_sample_fix_neighbor(_p(src_ind_ptr), _p(sel_indices),
                                                                          int(sel_indices.size), int(neighbor_num)))
        return pos, ptr

    def csr_submat(self, src_end_points, src_values, src_ind_ptr, src_row_ids, src_col_ids, sel_row_indices,
                   sel_col_indices):
        _need(src_end_points, np.int32, "src_end_points"), _need(src_ind_ptr, np.int32, "src_ind_ptr")
        _need(src_row_ids, np.int32, "src_row_ids"), _need(src_col_ids, np.int32, "src_col_ids")
        if src_values is not None:
            _need(src_values, np.float32, "src_values")
        if sel_row_indices is not None:
            _need(sel_row_indices, np.int32, "sel_row_indices")
        if sel_col_indices is not None:
            _need(sel_col_indices, np.int32, "sel_col_indices")
        ints, floats = self._take(self._L.gsr_csr_submat(
            _p(src_end_points), _p(src_values), _p(src_ind_ptr), _p(src_row_ids), _p(src_col_ids), int(src_row_ids.size),
            int(src_col_ids.size), int(src_end_points.size), _p(sel_row_indices),
            0 if sel_row_indices is None else int(sel_row_indices.size), _p(sel_col_indices),
            0 if sel_col_indices is None else int(sel_col_indices.size)))
        return ints[0], (floats[0] if floats else None), ints[1], ints[2], ints[3]

    def _seg2(self, name, lhs, ind_ptr, rhs):
        _need(ind_ptr, np.int32, "ind_ptr")
        if lhs.dtype == np.int32:
            _need(rhs, np.int32, "rhs")
            out, fn = np.empty(lhs.size, np.int32), getattr(self._L, name + "_i")
        elif lhs.dtype == np.float32:
            _need(rhs, np.float32, "rhs")
            out, fn = np.empty(lhs.size, np.float32), getattr(self._L, name + "_f")
        else:
            raise TypeError("UnImplemented!")
        fn(_p(np.ascontiguousarray(lhs)), _p(ind_ptr), _p(np.ascontiguousarray(rhs)), int(ind_ptr.size - 1),
           int(lhs.size), _p(out))
        return out

    def seg_mul(self, lhs, ind_ptr, rhs):
        return self._seg2("gsr_seg_mul", lhs, ind_ptr, rhs)

    def seg_add(self, lhs, ind_ptr, rhs):
        return self._seg2("gsr_seg_add", lhs, ind_ptr, rhs)

    def seg_sum(self, data, ind_ptr):
        _need(ind_ptr, np.int32, "ind_ptr")
        n = int(ind_ptr.size - 1)
        if data.dtype == np.int32:
            out, fn = np.empty(n, np.int32), self._L.gsr_seg_sum_i
        elif data.dtype == np.float32:
            out, fn = np.empty(n, np.float32), self._L.gsr_seg_sum_f
        else:
            raise TypeError("UnImplemented!")
        fn(_p(np.ascontiguousarray(data)), _p(ind_ptr), n, int(data.size), _p(out))
        return out

    def unique_cnt(self, data):
        _need(data, np.int32, "data")
        (u, c), _ = self._take(self._L.gsr_unique_cnt(_p(data), int(data.size)))
        return u, c

    def unique_inverse(self, data):
        _need(data, np.int32, "data")
        (u, i), _ = self._take(self._L.gsr_unique_inverse(_p(data), int(data.size)))
        return u, i

    def remove_edges_by_indices(self, end_points, values, ind_ptr, row_indices, col_indices, omp=False):
        _need(end_points, np.int32, "end_points"), _need(values, np.float32, "values")
        _need(ind_ptr, np.int32, "ind_ptr"), _need(row_indices, np.int32, "row_indices")
        _need(col_indices, np.int32, "col_indices")
        if row_indices.size != col_indices.size:
            raise ValueError("edge_num == PyArray_SIZE(col_indices) failed")
        ints, floats = self._take(self._L.gsr_remove_edges(
            _p(end_points), _p(values), _p(ind_ptr), _p(row_indices), _p(col_indices), int(ind_ptr.size - 1),
            int(end_points.size), int(row_indices.size), int(bool(omp))))
        return ints[0], floats[0], ints[1]

    def multi_link_split(self, edge_values, ind_ptr, possible_edge_values, omp=False):
        _need(edge_values, np.float32, "edge_values"), _need(ind_ptr, np.int32, "ind_ptr")
        _need(possible_edge_values, np.float32, "possible_edge_values")
        n = int(possible_edge_values.size)
        ints, _ = self._take(self._L.gsr_multi_link_split(_p(edge_values), _p(ind_ptr), _p(possible_edge_values),
                                                          int(ind_ptr.size - 1), int(edge_values.size), n,
                                                          int(bool(omp))))
        return ints[:n], ints[n:]

    def take_1d_omp(self, data, sel):
        _need(sel, np.int32, "sel")
        if data.dtype == np.int32:
            out, fn = np.empty(sel.size, np.int32), self._L.gsr_take_1d_i
        elif data.dtype == np.float32:
            out, fn = np.empty(sel.size, np.float32), self._L.gsr_take_1d_f
        else:
            raise TypeError("UnImplemented!")
        fn(_p(np.ascontiguousarray(data)), _p(sel), int(data.size), int(sel.size), _p(out))
        return out

    def gen_row_indices_by_indptr(self, ind_ptr, nnz):
        _need(ind_ptr, np.int32, "ind_ptr")
        out = np.empty(int(nnz), np.int32)
        self._L.gsr_gen_row_indices_by_indptr(_p(ind_ptr), int(ind_ptr.size - 1), int(nnz), _p(out))
        return out

    def get_support(self, row_degrees, col_degrees, ind_ptr, end_points, symm):
        _need(row_degrees, np.int32, "row_degrees"), _need(col_degrees, np.int32, "col_degrees")
        _need(ind_ptr, np.int32, "ind_ptr"), _need(end_points, np.int32, "end_points")
        out = np.empty(end_points.size, np.float32)
        self._L.gsr_get_support(_p(row_degrees), _p(col_degrees), _p(ind_ptr), _p(end_points), int(ind_ptr.size - 1),
                                int(end_points.size), int(bool(symm)), _p(out))
        return out
