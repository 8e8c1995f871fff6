// TEST INFRASTRUCTURE ONLY.  C-ABI wrapper around the REFERENCE's GraphSampler core, compiled from the reference's own
// sources where they lie (/root/reference/GraphSampler/graph_sampler.{h,cpp}) by oracle/Makefile target `_ref`
// -> oracle/_ref/libgs_ref.so.  No reference text is in this file: it only calls the functions the reference's CPython
// binding calls (py_ext.cpp:65-610, method table :612-627) with the same argument order, and hands the results back as
// plain buffers so that ctypes can play the role of that binding (which does not compile against NumPy 2).
//
// How the core builds without google/sparsehash: graph_sampler.h:4-6 enables sparsehash only `#if !defined(_WIN32)`, and
// every use has an `#else` branch on std::unordered_{map,set} (graph_sampler.cpp:165-177, 287-292; graph_sampler.h:446,
// 472, 516).  The recipe passes -D_WIN32 (nothing else in the two files tests that macro) and -Dmain=gs_selftest_main
// (graph_sampler.cpp ends in a self-test `main`).  No stand-in header, library or generated code is involved.
//
// Only tests/, tests/golden/make_*.py, __graft_entry__ and bench.py's cpu_baseline leg may load the library built here.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "graph_sampler.h"

namespace {
struct Result {
  std::vector<std::vector<int>> iv;
  std::vector<std::vector<float>> fv;
};
graph_sampler::GraphSampler g_handle;   // py_ext.cpp:11 keeps one global sampler too
}  // namespace

extern "C" {

// ---- variable-size results -------------------------------------------------------------------------------------------
int gsr_result_num_int(void* r) { return (int)static_cast<Result*>(r)->iv.size(); }
int gsr_result_num_float(void* r) { return (int)static_cast<Result*>(r)->fv.size(); }
long long gsr_result_int_size(void* r, int k) { return (long long)static_cast<Result*>(r)->iv[k].size(); }
long long gsr_result_float_size(void* r, int k) { return (long long)static_cast<Result*>(r)->fv[k].size(); }
void gsr_result_int_copy(void* r, int k, int* dst) {
  const std::vector<int>& v = static_cast<Result*>(r)->iv[k];
  if (!v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(int));
}
void gsr_result_float_copy(void* r, int k, float* dst) {
  const std::vector<float>& v = static_cast<Result*>(r)->fv[k];
  if (!v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(float));
}
void gsr_result_free(void* r) { delete static_cast<Result*>(r); }

// ---- py_ext.cpp:102-107 ----------------------------------------------------------------------------------------------
void gsr_set_seed(int seed) { g_handle.set_seed(seed); }

// ---- py_ext.cpp:65-92: (sampled_indices, dst_ind_ptr) ------------------------------------------------------------------
void* gsr_random_sample_fix_neighbor(const int* src_ind_ptr, const int* sel_indices, int sel_node_num, int neighbor_num) {
  Result* r = new Result();
  r->iv.resize(2);
  g_handle.random_sample_fix_neighbor(src_ind_ptr, sel_indices, sel_node_num, neighbor_num, &r->iv[0], &r->iv[1]);
  return r;
}

// ---- py_ext.cpp:129-218: (end_points, values | none, ind_ptr, row_ids, col_ids) ------------------------------------------
void* gsr_csr_submat(const int* src_end_points, const float* src_values, const int* src_ind_ptr, const int* src_row_ids,
                     const int* src_col_ids, int src_row_num, int src_col_num, int src_nnz, const int* sel_row_indices,
                     int n_sel_rows, const int* sel_col_indices, int n_sel_cols) {
  int dst_row_num = sel_row_indices ? n_sel_rows : src_row_num;
  int dst_col_num = sel_col_indices ? n_sel_cols : src_col_num;
  int *ep = nullptr, *ip = nullptr, *rid = nullptr, *cid = nullptr;
  float* val = nullptr;
  int dst_nnz = 0;
  graph_sampler::slice_csr_mat(src_end_points, src_values, src_ind_ptr, src_row_ids, src_col_ids, src_row_num, src_col_num,
                               src_nnz, sel_row_indices, sel_col_indices, dst_row_num, dst_col_num, &ep, &val, &ip, &rid,
                               &cid, &dst_nnz);
  Result* r = new Result();
  r->iv.resize(4);
  r->iv[0].assign(ep, ep + dst_nnz);
  r->iv[1].assign(ip, ip + dst_row_num + 1);
  r->iv[2].assign(rid, rid + dst_row_num);
  r->iv[3].assign(cid, cid + dst_col_num);
  if (val && src_values) {   // the all-copy branch allocates (and leaves unwritten) a value array even without values
    r->fv.resize(1);
    r->fv[0].assign(val, val + dst_nnz);
  }
  delete[] val;
  delete[] ep;
  delete[] ip;
  delete[] rid;
  delete[] cid;
  return r;
}

// ---- py_ext.cpp:230-372 (int32 and float32 forms) ---------------------------------------------------------------------
void gsr_seg_mul_f(const float* lhs, const int* ind_ptr, const float* rhs, int seg_num, int nnz, float* out) {
  float* ret = nullptr;
  graph_sampler::seg_mul(lhs, ind_ptr, rhs, seg_num, nnz, &ret);
  std::memcpy(out, ret, sizeof(float) * nnz);
  delete[] ret;
}
void gsr_seg_mul_i(const int* lhs, const int* ind_ptr, const int* rhs, int seg_num, int nnz, int* out) {
  int* ret = nullptr;
  graph_sampler::seg_mul(lhs, ind_ptr, rhs, seg_num, nnz, &ret);
  std::memcpy(out, ret, sizeof(int) * nnz);
  delete[] ret;
}
void gsr_seg_add_f(const float* lhs, const int* ind_ptr, const float* rhs, int seg_num, int nnz, float* out) {
  float* ret = nullptr;
  graph_sampler::seg_add(lhs, ind_ptr, rhs, seg_num, nnz, &ret);
  std::memcpy(out, ret, sizeof(float) * nnz);
  delete[] ret;
}
void gsr_seg_add_i(const int* lhs, const int* ind_ptr, const int* rhs, int seg_num, int nnz, int* out) {
  int* ret = nullptr;
  graph_sampler::seg_add(lhs, ind_ptr, rhs, seg_num, nnz, &ret);
  std::memcpy(out, ret, sizeof(int) * nnz);
  delete[] ret;
}
void gsr_seg_sum_f(const float* data, const int* ind_ptr, int seg_num, int nnz, float* out) {
  float* ret = nullptr;
  graph_sampler::seg_sum(data, ind_ptr, seg_num, nnz, &ret);
  std::memcpy(out, ret, sizeof(float) * seg_num);
  delete[] ret;
}
void gsr_seg_sum_i(const int* data, const int* ind_ptr, int seg_num, int nnz, int* out) {
  int* ret = nullptr;
  graph_sampler::seg_sum(data, ind_ptr, seg_num, nnz, &ret);
  std::memcpy(out, ret, sizeof(int) * seg_num);
  delete[] ret;
}

// ---- py_ext.cpp:380-430: (unique, counts) / (unique, inverse) ----------------------------------------------------------
void* gsr_unique_cnt(const int* data, int num) {
  Result* r = new Result();
  r->iv.resize(2);
  graph_sampler::unique_cnt(data, num, &r->iv[0], &r->iv[1]);
  return r;
}
void* gsr_unique_inverse(const int* data, int num) {
  Result* r = new Result();
  r->iv.resize(2);
  graph_sampler::unique_inverse(data, num, &r->iv[0], &r->iv[1]);
  return r;
}

// ---- py_ext.cpp:441-486: (end_points, values, ind_ptr); `omp` selects the reference's second implementation -------------
void* gsr_remove_edges(const int* end_points, const float* values, const int* ind_ptr, const int* row_indices,
                       const int* col_indices, int row_num, int nnz, int edge_num, int omp) {
  Result* r = new Result();
  r->iv.resize(2);
  r->fv.resize(1);
  if (omp)
    graph_sampler::remove_edges_omp(end_points, values, ind_ptr, row_indices, col_indices, row_num, nnz, edge_num,
                                    &r->iv[0], &r->fv[0], &r->iv[1]);
  else
    graph_sampler::remove_edges(end_points, values, ind_ptr, row_indices, col_indices, row_num, nnz, edge_num, &r->iv[0],
                                &r->fv[0], &r->iv[1]);
  return r;
}

// ---- py_ext.cpp:498-533: val_num position lists then val_num row pointers; `omp`: 0 = dispatching entry (as the
//      binding calls it: > 10 000 nnz goes to the _omp form, graph_sampler.cpp:285-286), 1 = force the _omp form ------------
void* gsr_multi_link_split(const float* edge_values, const int* ind_ptr, const float* possible_edge_values, int node_num,
                           int nnz, int val_num, int omp) {
  std::vector<std::vector<int>> idx, ptr;
  if (omp)
    graph_sampler::multi_link_split_by_value_omp(edge_values, ind_ptr, possible_edge_values, node_num, nnz, val_num, &idx,
                                                 &ptr);
  else
    graph_sampler::multi_link_split_by_value(edge_values, ind_ptr, possible_edge_values, node_num, nnz, val_num, &idx, &ptr);
  Result* r = new Result();
  for (int i = 0; i < val_num; i++) r->iv.push_back(idx[i]);
  for (int i = 0; i < val_num; i++) r->iv.push_back(ptr[i]);
  return r;
}

// ---- py_ext.cpp:535-563 ------------------------------------------------------------------------------------------------
void gsr_take_1d_i(const int* data, const int* sel, int data_num, int sel_num, int* out) {
  std::vector<int> ret;
  graph_sampler::take_1d_omp(data, sel, data_num, sel_num, &ret);
  if (sel_num) std::memcpy(out, ret.data(), sizeof(int) * sel_num);
}
void gsr_take_1d_f(const float* data, const int* sel, int data_num, int sel_num, float* out) {
  std::vector<float> ret;
  graph_sampler::take_1d_omp(data, sel, data_num, sel_num, &ret);
  if (sel_num) std::memcpy(out, ret.data(), sizeof(float) * sel_num);
}

// ---- py_ext.cpp:565-578 ------------------------------------------------------------------------------------------------
void gsr_gen_row_indices_by_indptr(const int* ind_ptr, int num, int nnz, int* out) {
  std::vector<int> ret;
  graph_sampler::gen_row_indices_by_indptr(ind_ptr, num, nnz, &ret);
  if (nnz) std::memcpy(out, ret.data(), sizeof(int) * nnz);
}

// ---- py_ext.cpp:580-610 ------------------------------------------------------------------------------------------------
void gsr_get_support(const int* row_degrees, const int* col_degrees, const int* ind_ptr, const int* end_points, int num,
                     int nnz, int symm, float* out) {
  std::vector<float> ret;
  graph_sampler::get_support(row_degrees, col_degrees, ind_ptr, end_points, num, nnz, symm != 0, &ret);
  if (nnz) std::memcpy(out, ret.data(), sizeof(float) * nnz);
}

}  // extern "C"
