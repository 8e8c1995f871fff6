/*
 * oracle/seg_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the STAR-GCN segment operators, used ONLY as the
 * checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * The product path (star-gcn_amd/) never links, imports or calls this file.
 *
 * Every function follows the loop order of the reference CPU implementation it
 * cites (reference/seg_ops_cuda/mxnet_op/seg_op.cc), so fp32 sums are formed in
 * the same order as the reference's FCompute<cpu> path.  Offsets are 64-bit
 * (the reference is int32 and caps K*N*C < 2^31, seg_op.cu:825-831).
 *
 * PIN STATUS: pinned against the reference's own numpy models
 * (reference/seg_ops_cuda/mxnet_op/test_seg_ops.py:11-99) executed from
 * /root/reference by tests/golden/make_golden.py -> tests/golden/ (npz files).
 * The reference C++ (seg_op.cc) needs MXNet/mshadow/dmlc headers that are not in
 * this image, so it is unbuildable here and is NOT compiled (oracle/_ref holds the
 * reference's GraphSampler core only, see oracle/Makefile).
 *
 * req follows MXNet OpReqType values: 0 = kNullOp, 1 = kWriteTo, 3 = kAddTo.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REQ_NULL 0
#define REQ_WRITE 1
#define REQ_ADD 3

#define EXPORT __attribute__((visibility("default")))

/* ---- seg_sum / seg_max / seg_min : seg_op.cc:7-50 (SegReduceImpl) ----------------- */
/* reduce_type: 0 sum, 1 max, 2 min.  data (B,nnz) -> dst (B,S). */
EXPORT int oracle_seg_reduce(float *dst, const float *data, const int32_t *indptr,
                             int64_t batch, int64_t seg_num, int64_t nnz, int reduce_type, int req) {
  if (req == REQ_NULL) return 0;
  for (int64_t k = 0; k < batch; k++) {
    for (int64_t i = 0; i < seg_num; i++) {
      float res = 0.0f;
      if (reduce_type == 1) res = -FLT_MAX;       /* numeric_limits<float>::lowest() */
      else if (reduce_type == 2) res = FLT_MAX;
      else if (reduce_type != 0) return -1;
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) {
        float v = data[k * nnz + j];
        if (reduce_type == 0) res += v;
        else if (reduce_type == 1) res = res > v ? res : v;
        else res = res < v ? res : v;
      }
      if (req == REQ_ADD) dst[k * seg_num + i] += res;
      else dst[k * seg_num + i] = res;
    }
  }
  return 0;
}

/* ---- seg_broadcast_{add,mul,to} : seg_op.cc:52-78 (SegBroadcastBinaryImpl) -------- */
/* op: 0 plus, 1 mul, 2 right (broadcast_to; lhs may be NULL).  dst (B,nnz). */
EXPORT int oracle_seg_broadcast(float *dst, const float *lhs, const float *rhs, const int32_t *indptr,
                                int64_t batch, int64_t seg_num, int64_t nnz, int op, int req) {
  if (req == REQ_NULL) return 0;
  if (req != REQ_ADD) memset(dst, 0, sizeof(float) * (size_t)(batch * nnz));
  for (int64_t k = 0; k < batch; k++) {
    for (int64_t i = 0; i < seg_num; i++) {
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) {
        float r = rhs[k * seg_num + i];
        float v;
        if (op == 0) v = lhs[k * nnz + j] + r;
        else if (op == 1) v = lhs[k * nnz + j] * r;
        else v = r;
        if (req == REQ_ADD) dst[k * nnz + j] += v;
        else dst[k * nnz + j] = v;
      }
    }
  }
  return 0;
}

/* ---- seg_softmax fwd : seg_op.cc:80-114 (SegSoftmaxImpl) -------------------------- */
EXPORT int oracle_seg_softmax(float *dst, const float *data, const int32_t *indptr,
                              int64_t batch, int64_t seg_num, int64_t nnz, int req) {
  if (req == REQ_NULL) return 0;
  if (req == REQ_ADD) return -2; /* reference: CHECK_NE(req, kAddTo), seg_op.cc:86 */
  for (int64_t k = 0; k < batch; k++)
    for (int64_t i = 0; i < nnz; i++) dst[k * nnz + i] = 0;
  for (int64_t k = 0; k < batch; k++) {
    for (int64_t i = 0; i < seg_num; i++) {
      float sum_val = 0.0f;
      float max_val = -FLT_MAX; /* red::maximum::SetInitValue = lowest (mshadow MinValue<float>) */
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) {
        float v = data[k * nnz + j];
        if (v > max_val) max_val = v;
      }
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) dst[k * nnz + j] = expf(data[k * nnz + j] - max_val);
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) sum_val += dst[k * nnz + j];
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) dst[k * nnz + j] /= sum_val;
    }
  }
  return 0;
}

/* ---- seg_softmax bwd : seg_op.cc:120-148 (SegSoftmaxBackwardImpl) ----------------- */
/* The reference reads seg_num from indptr.shape_[1] of a 1-D tensor (seg_op.cc:131,
 * defect #2 in SURVEY appendix A); the intended value shape_[0]-1 is used here. */
EXPORT int oracle_seg_softmax_bwd(float *dst, const float *ograd, const float *val, const int32_t *indptr,
                                  int64_t batch, int64_t seg_num, int64_t nnz, int req) {
  if (req == REQ_NULL) return 0;
  for (int64_t k = 0; k < batch; k++) {
    for (int64_t i = 0; i < seg_num; i++) {
      float sum_val = 0;
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) sum_val += ograd[k * nnz + j] * val[k * nnz + j];
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) {
        float g = val[k * nnz + j] * (ograd[k * nnz + j] - sum_val);
        if (req == REQ_ADD) dst[k * nnz + j] += g;
        else dst[k * nnz + j] = g;
      }
    }
  }
  return 0;
}

/* ---- seg_take_k_corr : seg_op.cc:150-178 (SegTakeKCorrImpl) ----------------------- */
/* dst (K,nnz); embed1 (K,N,C); embed2 (K,M,C).  Also = grad of seg_weighted_pool wrt weights
 * (seg_op.cc:703). */
EXPORT int oracle_seg_take_k_corr(float *dst, const float *embed1, const float *embed2,
                                  const int32_t *neighbor_ids, const int32_t *neighbor_indptr,
                                  int64_t K, int64_t node_num, int64_t neighbor_node_num, int64_t nnz,
                                  int64_t feat_dim, int req) {
  if (req == REQ_NULL) return 0;
  if (req != REQ_ADD) memset(dst, 0, sizeof(float) * (size_t)(K * nnz));
  for (int64_t k = 0; k < K; k++) {
#pragma omp parallel for
    for (int64_t i = 0; i < node_num; i++) {
      for (int64_t j = neighbor_indptr[i]; j < neighbor_indptr[i + 1]; j++) {
        const float *e1 = embed1 + (k * node_num + i) * feat_dim;
        const float *e2 = embed2 + (k * neighbor_node_num + neighbor_ids[j]) * feat_dim;
        for (int64_t c = 0; c < feat_dim; c++) dst[k * nnz + j] += e1[c] * e2[c];
      }
    }
  }
  return 0;
}

/* ---- seg_weighted_pool fwd : seg_op.h:460-476 -> seg_op.cc:180-207 ----------------
 * (SegTakeKCorrBackwardEmbed1Impl(dst, weights, data, indices, indptr)).
 * dst (B,S,C); data (B,T,C); weights (B,nnz).  omp placement as the reference (over rows). */
EXPORT int oracle_seg_weighted_pool(float *dst, const float *data, const float *weights,
                                    const int32_t *indices, const int32_t *indptr,
                                    int64_t batch, int64_t seg_num, int64_t total_ind_num, int64_t nnz,
                                    int64_t feat_dim, int req) {
  if (req == REQ_NULL) return 0;
  if (req != REQ_ADD) memset(dst, 0, sizeof(float) * (size_t)(batch * seg_num * feat_dim));
  for (int64_t k = 0; k < batch; k++) {
#pragma omp parallel for
    for (int64_t i = 0; i < seg_num; i++) {
      float *d = dst + (k * seg_num + i) * feat_dim;
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) {
        const float g = weights[k * nnz + j];
        const float *e2 = data + (k * total_ind_num + indices[j]) * feat_dim;
        for (int64_t c = 0; c < feat_dim; c++) d[c] += g * e2[c];
      }
    }
  }
  return 0;
}

/* ---- grad of seg_weighted_pool wrt data : seg_op.cc:700-703, kernel :209-240 ------
 * (_backward_seg_take_k_corr_embed2(weights, ograd, indices, indptr)).
 * dst (B,T,C) += g[b,j] * ograd_rows[b, seg(j), :], serial over edges in increasing j,
 * omp only over the batch axis exactly as the reference (so serial for B = 1). */
EXPORT int oracle_seg_weighted_pool_bwd_data(float *dst, const float *weights, const float *ograd_rows,
                                             const int32_t *indices, const int32_t *indptr,
                                             int64_t batch, int64_t seg_num, int64_t total_ind_num,
                                             int64_t nnz, int64_t feat_dim, int req) {
  if (req == REQ_NULL) return 0;
  if (req != REQ_ADD) memset(dst, 0, sizeof(float) * (size_t)(batch * total_ind_num * feat_dim));
  /* std::vector<int> seg_ids(nnz) is zero-initialised in the reference (seg_op.cc:225), so edges
   * past indptr[seg_num] ("padding", e.g. the 1-element empty_as_zero arrays of graph.py:221-222)
   * are attributed to segment 0 and DO contribute g*ograd[0] (the reference GPU path attributes
   * them to the last non-empty segment instead, seg_op.cu:91-110).  The model only ever pads with
   * weight 0, where every variant agrees. */
  int32_t *seg_ids = (int32_t *)calloc((size_t)(nnz > 0 ? nnz : 1), sizeof(int32_t));
  if (!seg_ids) return -3;
  for (int64_t i = 0; i < seg_num; i++)
    for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) seg_ids[j] = (int32_t)i;
#pragma omp parallel for
  for (int64_t k = 0; k < batch; k++) {
    for (int64_t i = 0; i < nnz; i++) {
      float *d = dst + (k * total_ind_num + indices[i]) * feat_dim;
      const float g = weights[k * nnz + i];
      const float *e1 = ograd_rows + (k * seg_num + seg_ids[i]) * feat_dim;
      for (int64_t c = 0; c < feat_dim; c++) d[c] += g * e1[c];
    }
  }
  free(seg_ids);
  return 0;
}

/* "fair" CPU variant of the same gradient: parallel over destination rows through a
 * transposed CSR built with a stable counting sort (same per-row summation order as above).
 * Not a reference function; used only as the labelled "fair" leg of bench.py's cpu_baseline. */
EXPORT int oracle_seg_weighted_pool_bwd_data_fair(float *dst, const float *weights, const float *ograd_rows,
                                                  const int32_t *indices, const int32_t *indptr,
                                                  int64_t batch, int64_t seg_num, int64_t total_ind_num,
                                                  int64_t nnz, int64_t feat_dim, int req) {
  if (req == REQ_NULL) return 0;
  if (req != REQ_ADD) memset(dst, 0, sizeof(float) * (size_t)(batch * total_ind_num * feat_dim));
  const int64_t covered = seg_num > 0 ? indptr[seg_num] : 0;
  int64_t *tptr = (int64_t *)calloc((size_t)total_ind_num + 2, sizeof(int64_t));
  int32_t *tpos = (int32_t *)malloc(sizeof(int32_t) * (size_t)(covered > 0 ? covered : 1));
  int32_t *tseg = (int32_t *)malloc(sizeof(int32_t) * (size_t)(covered > 0 ? covered : 1));
  if (!tptr || !tpos || !tseg) { free(tptr); free(tpos); free(tseg); return -3; }
  for (int64_t j = 0; j < covered; j++) tptr[indices[j] + 2]++;
  for (int64_t n = 0; n < total_ind_num; n++) tptr[n + 2] += tptr[n + 1];
  for (int64_t i = 0; i < seg_num; i++)
    for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) {
      int64_t p = tptr[indices[j] + 1]++;
      tpos[p] = (int32_t)j; tseg[p] = (int32_t)i;
    }
  for (int64_t k = 0; k < batch; k++) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t n = 0; n < total_ind_num; n++) {
      float *d = dst + (k * total_ind_num + n) * feat_dim;
      for (int64_t p = tptr[n]; p < tptr[n + 1]; p++) {
        const float g = weights[k * nnz + tpos[p]];
        const float *e1 = ograd_rows + (k * seg_num + tseg[p]) * feat_dim;
        for (int64_t c = 0; c < feat_dim; c++) d[c] += g * e1[c];
      }
    }
  }
  free(tptr); free(tpos); free(tseg);
  return 0;
}

/* ---- seg_pool fwd : seg_op.cc:242-297 (SegPoolImpl) -------------------------------
 * pool_type: 0 sum, 1 avg(mean), 2 max.  pool_indices (B,S,C) int32 only for max; argmax is the
 * EDGE POSITION j, strict '>' so the first maximum wins; empty segment -> value 0, index -1. */
EXPORT int oracle_seg_pool(float *dst_value, int32_t *pool_indices, const float *data,
                           const int32_t *indices, const int32_t *indptr,
                           int64_t batch, int64_t seg_num, int64_t total_ind_num, int64_t nnz,
                           int64_t feat_dim, int pool_type, int req) {
  (void)nnz;
  if (req == REQ_NULL) return 0;
  if (req == REQ_ADD) return -2; /* seg_op.cc:252 */
  if (pool_type < 0 || pool_type > 2) return -1;
  for (int64_t k = 0; k < batch; k++) {
#pragma omp parallel for
    for (int64_t i = 0; i < seg_num; i++) {
      float *d = dst_value + (k * seg_num + i) * feat_dim;
      int32_t *pi = pool_indices ? pool_indices + (k * seg_num + i) * feat_dim : NULL;
      for (int64_t c = 0; c < feat_dim; c++) {
        if (pool_type != 2) d[c] = 0;
        else {
          d[c] = -FLT_MAX;
          if (indptr[i + 1] == indptr[i]) d[c] = 0;
          pi[c] = -1;
        }
      }
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) {
        const float *row = data + (k * total_ind_num + indices[j]) * feat_dim;
        for (int64_t c = 0; c < feat_dim; c++) {
          if (pool_type != 2) d[c] += row[c];
          else if (row[c] > d[c]) { d[c] = row[c]; pi[c] = (int32_t)j; }
        }
      }
      if (pool_type == 1 && indptr[i + 1] - indptr[i] > 0)
        for (int64_t c = 0; c < feat_dim; c++) d[c] /= (float)(indptr[i + 1] - indptr[i]);
    }
  }
  return 0;
}

/* ---- seg_pool bwd : seg_op.cc:299-332 (SegPoolBackwardImpl) ----------------------- */
EXPORT int oracle_seg_pool_bwd(float *dst, const float *ograd, const int32_t *pool_indices,
                               const int32_t *indices, const int32_t *indptr,
                               int64_t batch, int64_t seg_num, int64_t total_ind_num, int64_t nnz,
                               int64_t feat_dim, int pool_type, int req) {
  (void)nnz;
  if (req == REQ_NULL) return 0;
  if (req != REQ_ADD) memset(dst, 0, sizeof(float) * (size_t)(batch * total_ind_num * feat_dim));
#pragma omp parallel for
  for (int64_t k = 0; k < batch; k++) {
    for (int64_t i = 0; i < seg_num; i++) {
      const float *g = ograd + (k * seg_num + i) * feat_dim;
      for (int64_t j = indptr[i]; j < indptr[i + 1]; j++) {
        float *d = dst + (k * total_ind_num + indices[j]) * feat_dim;
        for (int64_t c = 0; c < feat_dim; c++) {
          if (pool_type == 1) d[c] += g[c] / (float)(indptr[i + 1] - indptr[i]);
          else if (pool_type == 0) d[c] += g[c];
          else d[c] += g[c] * (float)(pool_indices[(k * seg_num + i) * feat_dim + c] == (int32_t)j);
        }
      }
    }
  }
  return 0;
}

/* ======================================================================================
 * Host graph helpers (reference/GraphSampler/graph_sampler.cpp).  PINNED (round 4) against the
 * reference core itself, compiled from its own sources into oracle/_ref (oracle/Makefile `_ref`,
 * -D_WIN32 = the std::unordered_* branches the reference carries; no stand-in headers):
 * tests/golden/graph_primitives_golden.npz, tests/test_graph_primitives_ref.py -- bit-exact,
 * serial and _omp forms.
 * ==================================================================================== */

/* get_support : graph_sampler.cpp:393-420.  symm: sqrt(1/deg_row/deg_col) as
 * sqrt(1.0f/float(r)/float(c)); else 1/deg_row; 0 when a degree is 0. */
EXPORT int oracle_get_support(float *support, const int32_t *row_degrees, const int32_t *col_degrees,
                              const int32_t *end_points, const int32_t *ind_ptr,
                              int64_t row_num, int symm) {
  for (int64_t i = 0; i < row_num; i++) {
    for (int64_t j = ind_ptr[i]; j < ind_ptr[i + 1]; j++) {
      int32_t dr = row_degrees[i];
      int32_t dc = col_degrees[end_points[j]];
      if (symm) support[j] = (dr == 0 || dc == 0) ? 0.0f : sqrtf(1.0f / (float)dr / (float)dc);
      else support[j] = (dr == 0) ? 0.0f : 1.0f / (float)dr;
    }
  }
  return 0;
}

/* multi_link_split_by_value : graph_sampler.cpp:277-376.  For each level l (matched by exact
 * float equality against multi_link[l]) emits the edge positions of that level in original CSR
 * order and a full-length indptr (row_num+1).  Outputs are written level after level:
 *   out_pos   : nnz int32, level l occupies [level_off[l], level_off[l+1])
 *   out_indptr: num_links*(row_num+1) int32
 *   level_off : num_links+1 int64
 * Returns -4 if a value matches no level. */
EXPORT int oracle_multi_link_split(int32_t *out_pos, int32_t *out_indptr, int64_t *level_off,
                                   const float *values, const int32_t *ind_ptr,
                                   const float *multi_link, int64_t row_num, int64_t num_links) {
  int64_t nnz = ind_ptr[row_num];
  int64_t *cnt = (int64_t *)calloc((size_t)num_links + 1, sizeof(int64_t));
  if (!cnt) return -3;
  for (int64_t j = 0; j < nnz; j++) {
    int64_t l = 0;
    while (l < num_links && values[j] != multi_link[l]) l++;
    if (l == num_links) { free(cnt); return -4; }
    cnt[l + 1]++;
  }
  level_off[0] = 0;
  for (int64_t l = 0; l < num_links; l++) level_off[l + 1] = level_off[l] + cnt[l + 1];
  for (int64_t l = 0; l < num_links; l++) {
    int64_t w = level_off[l];
    int32_t *ip = out_indptr + l * (row_num + 1);
    ip[0] = 0;
    for (int64_t i = 0; i < row_num; i++) {
      for (int64_t j = ind_ptr[i]; j < ind_ptr[i + 1]; j++)
        if (values[j] == multi_link[l]) out_pos[w++] = (int32_t)j;
      ip[i + 1] = (int32_t)(w - level_off[l]);
    }
  }
  free(cnt);
  return 0;
}
