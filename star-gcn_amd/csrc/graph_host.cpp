// graph_host.cpp -- host-side plan / graph helpers of libstargcn_hip.so (C++17, no GPU work).
//
// Counterparts of the reference's native `mxgraph._graph_sampler` helpers (reference
// GraphSampler/graph_sampler.cpp, exposed by py_ext.cpp:612-627) that produce the integer inputs of the hot
// path, plus the plan builders the reference does not have (it re-sorts edges on every backward call,
// seg_op.cu:906-925, and rebuilds/uploads every index array on every iteration, layers.py:366-377).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

// Host helpers run once per training iteration on arrays of 1e4..1e7 entries.  Waking the OpenMP pool of a 256-thread
// host costs ~10 ms (measured: sg_get_support_cpu on 72 k edges took 14 ms with the default team, <0.1 ms serial),
// so small inputs stay serial and large ones use a bounded team.
#define SG_HOST_THREADS 16
#define SG_OMP_MIN_WORK (1 << 20)
#include <system_error>
#include <thread>
#include <vector>

#include "common.hpp"

namespace sg {
// Host-side loops over rows run on plain std::threads, NOT on OpenMP (round 4).  hipcc's -fopenmp brings LLVM's libomp into
// the process next to the libgomp a PyTorch host already runs; libomp initialises lazily, at the first parallel region, from
// whatever OMP_* the environment holds AT THAT MOMENT -- with OMP_PROC_BIND / OMP_PLACES set it pins the calling thread (the
// host application's main thread) to one core, and every thread created afterwards inherits the one-core mask.  That was the
// "freeze" of round 3 (a test imported bench.py, which exported those variables for its CPU baseline).  A library must not
// change its caller's affinity, whatever the environment says: std::thread workers inherit the caller's mask and leave it alone.
template <class F>
static void parallel_blocks(int64_t n, bool parallel, F body) {      // body(begin, end) over a static partition of [0, n)
  int t = 1;
  if (parallel) {
    const unsigned hw = std::thread::hardware_concurrency();
    t = static_cast<int>(hw ? (hw < SG_HOST_THREADS ? hw : SG_HOST_THREADS) : 1);
    if (n < t) t = 1;
  }
  if (t <= 1) { body(static_cast<int64_t>(0), n); return; }
  std::vector<std::thread> team;
  team.reserve(t - 1);
  const int64_t per = (n + t - 1) / t;
  int64_t serial_from = n;      // ranges [serial_from, n) could not get a thread and are run by the caller
  for (int k = 1; k < t; ++k) {
    const int64_t b = per * k, e = (b + per < n) ? b + per : n;
    if (b >= e) continue;
    // std::thread's constructor throws std::system_error on EAGAIN (pids cgroup, RLIMIT_NPROC): an exception leaving an
    // extern "C" entry point past joinable threads would be std::terminate -- the host process gone.  Fall back to
    // running the remaining ranges serially instead.
    try {
      team.emplace_back([=] { body(b, e); });
    } catch (const std::system_error&) {
      serial_from = b;
      break;
    }
  }
  body(static_cast<int64_t>(0), per < n ? per : n);
  if (serial_from < n) body(serial_from, n);
  for (auto& th : team) th.join();
}

static thread_local std::string g_last_error;
void set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}
}  // namespace sg

using namespace sg;

SG_API const char* sg_last_error(void) { return g_last_error.c_str(); }
SG_API int sg_version(void) { return 100; }  // 0.1.0

SG_API int sg_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(SG_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

// Stable counting sort of the covered edges by destination index (reference semantics: stable radix sort
// of (indices, iota), seg_op.cu:906-912) -> transposed CSR.  t_pos[p] = original edge position (increasing
// inside a row), t_seg[p] = segment that edge belongs to.
SG_API int sg_build_transpose_cpu(int32_t* t_indptr, int32_t* t_pos, int32_t* t_seg, const int32_t* indices,
                                  const int32_t* indptr, int64_t seg_num, int64_t total_ind_num, int64_t nnz) {
  if (seg_num < 0 || total_ind_num < 0 || nnz < 0) return fail(SG_ERR_INVALID, "negative dimension");
  if (!t_indptr || !indptr) return fail(SG_ERR_INVALID, "null pointer argument");
  const int64_t E = seg_num > 0 ? indptr[seg_num] : 0;
  if (E < 0 || E > nnz) return fail(SG_ERR_INVALID, "indptr[-1]=%lld exceeds nnz=%lld", (long long)E, (long long)nnz);
  std::vector<int64_t> cnt(static_cast<size_t>(total_ind_num) + 1, 0);
  for (int64_t j = 0; j < E; ++j) {
    const int32_t n = indices[j];
    if (n < 0 || n >= total_ind_num) return fail(SG_ERR_VALUE, "indices[%lld]=%d out of range [0,%lld)", (long long)j, n, (long long)total_ind_num);
    cnt[static_cast<size_t>(n) + 1]++;
  }
  for (int64_t n = 0; n < total_ind_num; ++n) cnt[n + 1] += cnt[n];
  for (int64_t n = 0; n <= total_ind_num; ++n) t_indptr[n] = static_cast<int32_t>(cnt[n]);
  for (int64_t i = 0; i < seg_num; ++i) {
    if (indptr[i + 1] < indptr[i]) return fail(SG_ERR_VALUE, "indptr is not non-decreasing at %lld", (long long)i);
    for (int64_t j = indptr[i]; j < indptr[i + 1]; ++j) {
      const int64_t p = cnt[indices[j]]++;
      t_pos[p] = static_cast<int32_t>(j);
      t_seg[p] = static_cast<int32_t>(i);
    }
  }
  return SG_OK;
}

// reference graph_sampler.cpp:393-420
SG_API int sg_get_support_cpu(float* support, const int32_t* row_degrees, const int32_t* col_degrees,
                              const int32_t* end_points, const int32_t* ind_ptr, int64_t row_num, int symm) {
  if (row_num < 0) return fail(SG_ERR_INVALID, "negative row_num");
  sg::parallel_blocks(row_num, row_num > 0 && ind_ptr[row_num] > SG_OMP_MIN_WORK, [=](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const int32_t dr = row_degrees[i];
      for (int64_t j = ind_ptr[i]; j < ind_ptr[i + 1]; ++j) {
        if (symm) {
          const int32_t dc = col_degrees[end_points[j]];
          support[j] = (dr == 0 || dc == 0) ? 0.0f : std::sqrt(1.0f / static_cast<float>(dr) / static_cast<float>(dc));
        } else {
          support[j] = (dr == 0) ? 0.0f : 1.0f / static_cast<float>(dr);
        }
      }
    }
  });
  return SG_OK;
}

// reference graph_sampler.cpp:277-376 (levels matched by exact float equality, edges kept in CSR order,
// a full-length indptr per level).
SG_API int sg_multi_link_split_cpu(int32_t* out_pos, int32_t* out_indptr, int64_t* level_off, const float* values,
                                   const int32_t* ind_ptr, const float* multi_link, int64_t row_num,
                                   int64_t num_links) {
  if (row_num < 0 || num_links <= 0) return fail(SG_ERR_INVALID, "bad row_num/num_links");
  const int64_t nnz = ind_ptr[row_num];
  std::vector<int32_t> level(static_cast<size_t>(nnz));
  std::vector<int64_t> cnt(static_cast<size_t>(num_links) + 1, 0);
  for (int64_t j = 0; j < nnz; ++j) {
    int64_t l = 0;
    while (l < num_links && values[j] != multi_link[l]) ++l;
    if (l == num_links) return fail(SG_ERR_VALUE, "edge value %g at position %lld matches no link level", (double)values[j], (long long)j);
    level[j] = static_cast<int32_t>(l);
    cnt[l + 1]++;
  }
  level_off[0] = 0;
  for (int64_t l = 0; l < num_links; ++l) level_off[l + 1] = level_off[l] + cnt[l + 1];
  std::vector<int64_t> w(level_off, level_off + num_links);
  for (int64_t l = 0; l < num_links; ++l) out_indptr[l * (row_num + 1)] = 0;
  for (int64_t i = 0; i < row_num; ++i) {
    for (int64_t j = ind_ptr[i]; j < ind_ptr[i + 1]; ++j) out_pos[w[level[j]]++] = static_cast<int32_t>(j);
    for (int64_t l = 0; l < num_links; ++l) out_indptr[l * (row_num + 1) + i + 1] = static_cast<int32_t>(w[l] - level_off[l]);
  }
  return SG_OK;
}

// Fuses the R per-level CSRs into one CSR over n_dst*R segments and builds its transpose over n_src*R
// segments (edges of a transposed segment in increasing destination order == original CSR order).
SG_API int sg_multilink_fuse_cpu(int32_t* c_indptr, int32_t* c_idx, int32_t* c_q, float* c_w, int32_t* t_indptr,
                                 int32_t* t_idx, int32_t* t_q, float* t_w, const int32_t* const* end_points_l,
                                 const int32_t* const* indptr_l, const float* const* support_l, int64_t num_links,
                                 int64_t n_dst, int64_t n_src) {
  if (num_links <= 0 || n_dst < 0 || n_src < 0) return fail(SG_ERR_INVALID, "bad multilink dimensions");
  const int64_t R = num_links;
  if (n_dst * R >= (1ll << 31) - 1 || n_src * R >= (1ll << 31) - 1) return fail(SG_ERR_INVALID, "n*R overflows int32");
  int64_t total = 0;
  for (int64_t r = 0; r < R; ++r) {
    if (indptr_l[r][0] != 0) return fail(SG_ERR_VALUE, "indptr_l[%lld][0] != 0", (long long)r);
    total += indptr_l[r][n_dst];
  }
  if (total >= (1ll << 31) - 1) return fail(SG_ERR_INVALID, "edge count overflows int32");
  std::vector<int64_t> tcnt(static_cast<size_t>(n_src * R) + 1, 0);
  int64_t w = 0;
  c_indptr[0] = 0;
  for (int64_t i = 0; i < n_dst; ++i) {
    for (int64_t r = 0; r < R; ++r) {
      const int32_t* ep = end_points_l[r];
      const float* sp = support_l[r];
      for (int64_t j = indptr_l[r][i]; j < indptr_l[r][i + 1]; ++j) {
        const int32_t n = ep[j];
        if (n < 0 || n >= n_src) return fail(SG_ERR_VALUE, "end point %d out of range [0,%lld)", n, (long long)n_src);
        c_idx[w] = n;
        if (c_q) c_q[w] = static_cast<int32_t>(static_cast<int64_t>(n) * R + r);
        c_w[w] = sp[j];
        ++w;
        tcnt[static_cast<size_t>(n) * R + r + 1]++;
      }
      c_indptr[i * R + r + 1] = static_cast<int32_t>(w);
    }
  }
  for (int64_t s = 0; s < n_src * R; ++s) tcnt[s + 1] += tcnt[s];
  for (int64_t s = 0; s <= n_src * R; ++s) t_indptr[s] = static_cast<int32_t>(tcnt[s]);
  for (int64_t i = 0; i < n_dst; ++i) {
    for (int64_t r = 0; r < R; ++r) {
      for (int64_t j = c_indptr[i * R + r]; j < c_indptr[i * R + r + 1]; ++j) {
        const int64_t p = tcnt[static_cast<size_t>(c_idx[j]) * R + r]++;
        t_idx[p] = static_cast<int32_t>(i);
        if (t_q) t_q[p] = static_cast<int32_t>(i * R + r);
        t_w[p] = c_w[j];
      }
    }
  }
  return SG_OK;
}

// unique_inverse / unique_cnt of the reference (graph_sampler.h:465-534, py_ext.cpp unique_inverse / unique_cnt) for
// non-negative ids: first-occurrence order, O(n) with a direct-address table instead of a hash map (node ids of a
// HeterGraph are small contiguous integers).  uniq (n), inverse (n, may be NULL), counts (n, may be NULL).
SG_API int sg_unique_inverse_cpu(int32_t* uniq, int32_t* inverse, int32_t* counts, int64_t* n_uniq, const int32_t* ids,
                                 int64_t n, int64_t max_id) {
  if (n < 0 || max_id < -1) return fail(SG_ERR_INVALID, "bad unique_inverse arguments");
  std::vector<int32_t> slot(static_cast<size_t>(max_id) + 1, -1);
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t v = ids[i];
    if (v < 0 || v > max_id) return fail(SG_ERR_VALUE, "id %d at position %lld outside [0,%lld]", v, (long long)i, (long long)max_id);
    int32_t s = slot[v];
    if (s < 0) {
      s = static_cast<int32_t>(m);
      slot[v] = s;
      uniq[m] = v;
      if (counts) counts[m] = 0;
      ++m;
    }
    if (inverse) inverse[i] = s;
    if (counts) counts[s]++;
  }
  *n_uniq = m;
  return SG_OK;
}

// remove_edges_by_indices of the reference (graph_sampler.cpp:154-201): drop the listed (row index, col index) pairs
// from a CSR whose rows are sorted by column.  Pairs that are not edges are ignored.  Outputs: new end_points /
// values (nnz entries allocated by the caller, *new_nnz used) and new ind_ptr (row_num+1).
SG_API int sg_remove_edges_cpu(int32_t* out_end_points, float* out_values, int32_t* out_ind_ptr, int64_t* new_nnz,
                               const int32_t* end_points, const float* values, const int32_t* ind_ptr, int64_t row_num,
                               const int32_t* rm_rows, const int32_t* rm_cols, int64_t rm_num) {
  if (row_num < 0 || rm_num < 0) return fail(SG_ERR_INVALID, "negative dimension");
  const int64_t nnz = ind_ptr[row_num];
  std::vector<uint8_t> drop(static_cast<size_t>(nnz), 0);
  for (int64_t k = 0; k < rm_num; ++k) {
    const int32_t r = rm_rows[k];
    if (r < 0 || r >= row_num) return fail(SG_ERR_VALUE, "row index %d out of range", r);
    const int32_t* b = end_points + ind_ptr[r];
    const int32_t* e = end_points + ind_ptr[r + 1];
    const int32_t* p = std::lower_bound(b, e, rm_cols[k]);
    if (p != e && *p == rm_cols[k]) drop[p - end_points] = 1;
  }
  int64_t w = 0;
  out_ind_ptr[0] = 0;
  for (int64_t i = 0; i < row_num; ++i) {
    for (int64_t j = ind_ptr[i]; j < ind_ptr[i + 1]; ++j) {
      if (!drop[j]) {
        out_end_points[w] = end_points[j];
        if (values) out_values[w] = values[j];
        ++w;
      }
    }
    out_ind_ptr[i + 1] = static_cast<int32_t>(w);
  }
  *new_nnz = w;
  return SG_OK;
}

// slice_csr_mat of the reference (graph_sampler.cpp:31-152): rows `sel_rows` (in the given order; NULL = all rows) and
// the columns with col_map[c] >= 0 (col_map[c] = new column index; NULL = all columns, unchanged).  Entries of
// unselected columns are dropped, the remaining ones keep their order inside the row -- so rows are NOT column-sorted
// any more when col_map is not monotone (the reference has the same property).  The caller sizes the outputs for the
// total length of the selected rows; *out_nnz entries are used.  Two passes (count, fill), rows in parallel.
SG_API int sg_csr_submat_cpu(int32_t* out_end_points, float* out_values, int32_t* out_ind_ptr, int64_t* out_nnz,
                             const int32_t* end_points, const float* values, const int32_t* ind_ptr, int64_t row_num,
                             const int32_t* sel_rows, int64_t sel_num, const int32_t* col_map) {
  if (row_num < 0 || sel_num < 0) return fail(SG_ERR_INVALID, "negative dimension");
  if (!out_ind_ptr || !out_nnz || !ind_ptr) return fail(SG_ERR_INVALID, "null pointer argument");
  const int64_t n = sel_rows ? sel_num : row_num;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t r = sel_rows ? sel_rows[i] : i;
    if (r < 0 || r >= row_num) return fail(SG_ERR_VALUE, "row index %lld out of range", static_cast<long long>(r));
  }
  out_ind_ptr[0] = 0;
  sg::parallel_blocks(n, n > SG_OMP_MIN_WORK, [=](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t r = sel_rows ? sel_rows[i] : i;
      int32_t cnt = 0;
      if (!col_map) cnt = ind_ptr[r + 1] - ind_ptr[r];
      else
        for (int64_t j = ind_ptr[r]; j < ind_ptr[r + 1]; ++j) cnt += col_map[end_points[j]] >= 0;
      out_ind_ptr[i + 1] = cnt;
    }
  });
  for (int64_t i = 0; i < n; ++i) out_ind_ptr[i + 1] += out_ind_ptr[i];
  *out_nnz = out_ind_ptr[n];
  sg::parallel_blocks(n, n > SG_OMP_MIN_WORK, [=](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t r = sel_rows ? sel_rows[i] : i;
      int64_t w = out_ind_ptr[i];
      for (int64_t j = ind_ptr[r]; j < ind_ptr[r + 1]; ++j) {
        const int32_t c = col_map ? col_map[end_points[j]] : end_points[j];
        if (c < 0) continue;
        out_end_points[w] = c;
        if (values && out_values) out_values[w] = values[j];
        ++w;
      }
    }
  });
  return SG_OK;
}

// random_sample_fix_neighbor of the reference (graph_sampler.cpp:742-779): for every selected row keep all of its
// edges when it has <= neighbor_num of them (or neighbor_num < 0), otherwise draw neighbor_num edge positions
// uniformly WITHOUT replacement.  Differences by design: the draw of row i depends only on (seed, i) -- the
// reference indexes its RNG by OpenMP thread id, so its output changes with the thread count -- and the positions
// of a row are returned in increasing order (CSR column order is preserved for the level split that follows).
// Call with sampled == NULL to obtain dst_ind_ptr (sel_num+1) first; sampled holds dst_ind_ptr[sel_num] entries.
static inline uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
SG_API int sg_sample_fix_neighbor_cpu(int32_t* sampled, int32_t* dst_ind_ptr, const int32_t* src_ind_ptr,
                                      const int32_t* sel_indices, int64_t sel_num, int64_t neighbor_num,
                                      uint64_t seed) {
  if (sel_num < 0) return fail(SG_ERR_INVALID, "negative dimension");
  if (!dst_ind_ptr || !src_ind_ptr || (sel_num > 0 && !sel_indices)) return fail(SG_ERR_INVALID, "null argument");
  int64_t total = 0;
  dst_ind_ptr[0] = 0;
  for (int64_t i = 0; i < sel_num; ++i) {
    const int64_t len = src_ind_ptr[sel_indices[i] + 1] - src_ind_ptr[sel_indices[i]];
    total += (neighbor_num < 0) ? len : std::min<int64_t>(neighbor_num, len);
    if (total > INT32_MAX) return fail(SG_ERR_VALUE, "sampled edge count exceeds int32");
    dst_ind_ptr[i + 1] = static_cast<int32_t>(total);
  }
  if (!sampled) return SG_OK;
  sg::parallel_blocks(sel_num, total > SG_OMP_MIN_WORK, [=](int64_t lo, int64_t hi) {
  for (int64_t i = lo; i < hi; ++i) {
    const int32_t b = src_ind_ptr[sel_indices[i]], e = src_ind_ptr[sel_indices[i] + 1];
    const int32_t k = dst_ind_ptr[i + 1] - dst_ind_ptr[i], len = e - b;
    int32_t* out = sampled + dst_ind_ptr[i];
    if (k == len) {
      for (int32_t j = 0; j < len; ++j) out[j] = b + j;
      continue;
    }
    uint64_t st = seed ^ (0xD1B54A32D192ED03ull * static_cast<uint64_t>(i + 1));
    // Floyd's algorithm: k distinct values of [0, len) in O(k^2) worst case on tiny k, O(k log k) via the sorted insert
    int32_t n = 0;
    for (int32_t j = len - k; j < len; ++j) {
      const int32_t t = static_cast<int32_t>(splitmix64(st) % static_cast<uint64_t>(j + 1));
      int32_t* pos = std::lower_bound(out, out + n, b + t);
      int32_t v = b + t;
      if (pos != out + n && *pos == v) {   // already chosen -> take j itself (larger than everything chosen so far)
        v = b + j;
        pos = out + n;
      }
      std::copy_backward(pos, out + n, out + n + 1);
      *pos = v;
      ++n;
    }
  }
  });
  return SG_OK;
}

// gen_row_indices_by_indptr of the reference (graph_sampler.h / py_ext.cpp:612-627): COO row index of every edge
SG_API int sg_gen_row_indices_cpu(int32_t* row_indices, const int32_t* ind_ptr, int64_t row_num, int64_t nnz) {
  if (row_num < 0 || nnz < 0) return fail(SG_ERR_INVALID, "negative dimension");
  if (row_num > 0 && ind_ptr[row_num] != nnz) return fail(SG_ERR_VALUE, "ind_ptr[-1] = %d but nnz = %lld", ind_ptr[row_num],
                                                         static_cast<long long>(nnz));
  sg::parallel_blocks(row_num, nnz > SG_OMP_MIN_WORK, [=](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i)
      for (int32_t j = ind_ptr[i]; j < ind_ptr[i + 1]; ++j) row_indices[j] = static_cast<int32_t>(i);
  });
  return SG_OK;
}

// ---- per-batch index plans (host side of the resident training loop) -------------------------------------------------
// Position (edge id) of every (row index, col index) pair in a CSR whose rows are sorted by column; -1 when the pair
// is not an edge.  One binary search inside the row per pair (the numpy formulation searched a 64-bit key array of all
// E edges per pair: 100 ms per 100 k pairs at MovieLens-1M).
SG_API int sg_edge_positions_cpu(int32_t* pos, const int32_t* end_points, const int32_t* ind_ptr, int64_t row_num,
                                 const int32_t* rows, const int32_t* cols, int64_t n) {
  if (row_num < 0 || n < 0) return fail(SG_ERR_INVALID, "negative dimension");
  sg::parallel_blocks(n, n > SG_OMP_MIN_WORK, [=](int64_t lo, int64_t hi) {
    for (int64_t k = lo; k < hi; ++k) {
      const int32_t r = rows[k];
      if (r < 0 || r >= row_num) { pos[k] = -1; continue; }
      const int32_t* b = end_points + ind_ptr[r];
      const int32_t* e = end_points + ind_ptr[r + 1];
      const int32_t* p = std::lower_bound(b, e, cols[k]);
      pos[k] = (p != e && *p == cols[k]) ? static_cast<int32_t>(p - end_points) : -1;
    }
  });
  return SG_OK;
}

// Rating-head plan (star_gcn_amd.model.PairPlan): the (user, item) index pairs of a batch grouped by user into a CSR
// (stable counting sort: order / inv_order / indptr / items) plus its stable transpose by item (t_indptr / t_pos /
// t_seg, as sg_build_transpose_cpu).  *identity = 1 when the pairs already are in user order.
SG_API int sg_pair_plan_cpu(int32_t* order, int32_t* inv_order, int32_t* indptr, int32_t* items, int32_t* t_indptr,
                            int32_t* t_pos, int32_t* t_seg, int32_t* identity, const int32_t* user_idx,
                            const int32_t* item_idx, int64_t n_pairs, int64_t n_user, int64_t n_item) {
  if (n_pairs < 0 || n_user < 0 || n_item < 0) return fail(SG_ERR_INVALID, "negative dimension");
  std::vector<int32_t> cur(static_cast<size_t>(std::max(n_user, n_item)) + 1, 0);
  for (int64_t u = 0; u <= n_user; ++u) indptr[u] = 0;
  for (int64_t k = 0; k < n_pairs; ++k) {
    const int32_t u = user_idx[k], i = item_idx[k];
    if (u < 0 || u >= n_user || i < 0 || i >= n_item) return fail(SG_ERR_VALUE, "pair %lld (%d, %d) out of range", static_cast<long long>(k), u, i);
    ++indptr[u + 1];
  }
  for (int64_t u = 0; u < n_user; ++u) indptr[u + 1] += indptr[u];
  for (int64_t u = 0; u < n_user; ++u) cur[u] = indptr[u];
  int ident = 1;
  for (int64_t k = 0; k < n_pairs; ++k) {
    const int32_t slot = cur[user_idx[k]]++;
    order[slot] = static_cast<int32_t>(k);
    inv_order[k] = slot;
    items[slot] = item_idx[k];
    if (slot != k) ident = 0;
  }
  *identity = ident;
  // transpose: positions of the user-grouped list, grouped by item, increasing position inside an item
  for (int64_t i = 0; i <= n_item; ++i) t_indptr[i] = 0;
  for (int64_t p = 0; p < n_pairs; ++p) ++t_indptr[items[p] + 1];
  for (int64_t i = 0; i < n_item; ++i) t_indptr[i + 1] += t_indptr[i];
  for (int64_t i = 0; i < n_item; ++i) cur[i] = t_indptr[i];
  int64_t u = 0;
  for (int64_t p = 0; p < n_pairs; ++p) {
    while (p >= indptr[u + 1]) ++u;
    const int32_t slot = cur[items[p]]++;
    t_pos[slot] = static_cast<int32_t>(p);
    t_seg[slot] = static_cast<int32_t>(u);
  }
  return SG_OK;
}

// Row-take plan (star_gcn_amd.plan.TakePlan): ids (n, -1 = zero row) into a table of n_rows rows.
// flags bit 0: identity take (ids == 0..n-1 and n == n_rows); bit 1: every row taken at most once -> inv_ids
// (n_rows, -1 = not taken) is valid and the gradient is a row copy.  t_indptr (n_rows+1) / t_pos (n): positions
// grouped by id, increasing (the gradient's segment plan otherwise).
SG_API int sg_take_plan_cpu(int32_t* t_indptr, int32_t* t_pos, int32_t* inv_ids, int32_t* flags, int64_t* covered,
                            const int32_t* ids, int64_t n, int64_t n_rows) {
  if (n < 0 || n_rows < 0) return fail(SG_ERR_INVALID, "negative dimension");
  for (int64_t r = 0; r <= n_rows; ++r) t_indptr[r] = 0;
  int ident = (n == n_rows), once = 1;
  int64_t cov = 0;
  for (int64_t k = 0; k < n; ++k) {
    const int32_t id = ids[k];
    if (id != k) ident = 0;
    if (id < 0) continue;
    if (id >= n_rows) return fail(SG_ERR_VALUE, "id %d at %lld outside the table (%lld rows)", id, static_cast<long long>(k), static_cast<long long>(n_rows));
    if (++t_indptr[id + 1] > 1) once = 0;
    ++cov;
  }
  for (int64_t r = 0; r < n_rows; ++r) t_indptr[r + 1] += t_indptr[r];
  std::vector<int32_t> cur(t_indptr, t_indptr + n_rows);
  for (int64_t r = 0; r < n_rows; ++r) inv_ids[r] = -1;
  for (int64_t k = 0; k < n; ++k) {
    const int32_t id = ids[k];
    if (id < 0) continue;
    t_pos[cur[id]++] = static_cast<int32_t>(k);
    inv_ids[id] = static_cast<int32_t>(k);
  }
  *flags = (ident ? 1 : 0) | (once ? 2 : 0);
  *covered = cov;
  return SG_OK;
}
