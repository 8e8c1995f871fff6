/*
 * stargcn.h -- C ABI of libstargcn_hip.so: the MI355X (gfx950) replacement for the STAR-GCN
 * multi-link graph-conv hot path.
 *
 * Boundary being replaced: the MXNet NNVM operators `_contrib_seg_*` registered in
 * reference/seg_ops_cuda/mxnet_op/seg_op.cc:339-861 (FCompute signature seg_op.h:461-465; GPU attach
 * seg_op.cu:1353-1390) and the `mxgraph._graph_sampler` CPython helpers
 * (reference/GraphSampler/py_ext.cpp:612-627) that produce their integer inputs.
 *
 * Conventions (all entry points):
 *   - plain pointers + int64 sizes, no framework types.  `_hip` functions take DEVICE pointers and
 *     enqueue on `stream` (a hipStream_t, NULL = default stream) WITHOUT synchronising; `_cpu`
 *     functions take HOST pointers and are host-side plan/graph helpers (no floating-point hot path
 *     has a CPU implementation in this library: a missing GPU is an error, never a fallback).
 *   - data is fp32, indices are int32 (reference seg_op.h:232-233, 410-413); element offsets are
 *     64-bit internally (the reference is int32 and caps K*N*C < 2^31, seg_op.cu:825-831).
 *   - `req` uses MXNet OpReqType values (reference seg_op.cc:188-196): 0 kNullOp = do nothing,
 *     1 kWriteTo = overwrite dst, 3 kAddTo = accumulate into dst.
 *   - return 0 on success or a negative SG_ERR_* code; sg_last_error() gives the thread-local message.
 *     Nothing in this library calls exit() (the reference does: seg_op.cu:14-18).
 *   - the caller owns every buffer; `workspace` may be NULL when sg_*_workspace_bytes() returns 0.
 */
#ifndef STARGCN_H_
#define STARGCN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG_REQ_NULL 0
#define SG_REQ_WRITE 1
#define SG_REQ_ADD 3

#define SG_OK 0
#define SG_ERR_INVALID (-1)   /* bad argument (shape / req / enum) */
#define SG_ERR_UNSUPPORTED (-2) /* e.g. kAddTo on seg_pool / seg_softmax forward, as the reference */
#define SG_ERR_ALLOC (-3)
#define SG_ERR_VALUE (-4)     /* data-dependent failure (e.g. rating value matches no link) */
#define SG_ERR_WORKSPACE (-5) /* workspace too small */
#define SG_ERR_HIP (-6)       /* HIP runtime / launch error */

#define SG_POOL_SUM 0
#define SG_POOL_AVG 1
#define SG_POOL_MAX 2

#define SG_ACT_NONE 0
#define SG_ACT_LEAKY 1 /* x>0 ? x : slope*x ; reference common.py:47 uses slope 0.1 */
#define SG_ACT_RELU 2
#define SG_ACT_SIGMOID 3
#define SG_ACT_TANH 4

const char* sg_last_error(void);
int sg_version(void);
/* number of visible HIP devices, or a negative SG_ERR_HIP */
int sg_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * (1) seg_weighted_pool forward  == reference `_contrib_seg_weighted_pool`
 *     seg_op.cc:665-716, adapter seg_op.h:460-476, CPU kernel seg_op.cc:180-207, GPU seg_op.cu:682-722.
 *       dst[b,i,:] (+)= sum_{j=indptr[i]}^{indptr[i+1]-1} weights[b,j] * data[b, indices[j], :]
 *     data (batch,total_ind_num,feat_dim)  weights (batch,nnz)  indices (nnz)  indptr (seg_num+1)
 *     dst (batch,seg_num,feat_dim).  Edges at positions >= indptr[seg_num] are ignored.
 * ---------------------------------------------------------------------------------------------- */
size_t sg_seg_weighted_pool_workspace_bytes(int64_t batch, int64_t seg_num, int64_t nnz, int64_t feat_dim);
int sg_seg_weighted_pool_hip(float* dst, const float* data, const float* weights, const int32_t* indices,
                             const int32_t* indptr, int64_t batch, int64_t seg_num, int64_t total_ind_num,
                             int64_t nnz, int64_t feat_dim, int req, void* workspace, size_t workspace_bytes,
                             void* stream);

/* Strided / grouped form of (1) used by the fused multi-link aggregator (batch = 1):
 *   segment s writes   dst + (s / dst_group) * dst_ld + (s % dst_group) * feat_dim
 *   index q reads      src + (q / src_group) * src_ld + (q % src_group) * feat_dim
 * so R per-rating-level aggregates of one destination node land side by side in one row of a
 * (N_dst, R*feat_dim [+pad]) matrix that feeds the MFMA contraction directly.  weights may be NULL
 * (all ones = seg_pool 'sum').  `act`/`slope` (SG_ACT_*) fuse the aggregator activation
 * (reference aggregators.py:160) into the row store: dst = act(sum (+ dst if req = add)).  */
int sg_seg_gather_sum_hip(float* dst, int64_t dst_group, int64_t dst_ld, const float* src, int64_t src_group,
                          int64_t src_ld, const float* weights, const int32_t* indices, const int32_t* indptr,
                          int64_t seg_num, int64_t nnz, int64_t feat_dim, int req, int act, float slope,
                          void* workspace, size_t workspace_bytes, void* stream);
/* Same, with the byte size of the source matrix as a scheduling hint (0 = unknown): when the source fits the 256 MB
 * Infinity Cache but not one XCD's 4 MB L2, the launch is column-sliced across the XCDs (each XCD only touches one
 * 256-byte slice of every source row), which raises the L2 hit rate.  Deterministic either way; a sliced launch
 * sums each row's edges as 4 interleaved partial sums combined by a fixed shuffle tree (same fp32 error class). */
int sg_seg_gather_sum_hinted_hip(float* dst, int64_t dst_group, int64_t dst_ld, const float* src, int64_t src_group,
                                 int64_t src_ld, const float* weights, const int32_t* indices, const int32_t* indptr,
                                 int64_t seg_num, int64_t nnz, int64_t feat_dim, int req, int act, float slope,
                                 void* workspace, size_t workspace_bytes, void* stream, int64_t src_bytes);
/* Source-partitioned form, for gathers whose OUTPUT is small next to what they read (the item-side gradient of the rating
 * head: 10 M edges x 256 B gathered from an 18 MB matrix into 10 677 rows).  The plan's edges are ordered by source-row
 * range first (sg_part_keys_hip + sg_sort_i32_hip + sg_bounds_from_sorted_hip): indptr_p has parts * seg_num + 1
 * entries, indices_p the source rows, wpos the ORIGINAL slot of every edge (weights are read through it).  With
 * parts = 8 each XCD takes one contiguous eighth of the edge list, so its private 4 MB L2 only ever sees one source
 * range; the parts * seg_num partial rows go to the workspace and are summed in part order: dst = act(sum (+ dst)).
 * No reference counterpart (a scheduling transformation of seg_weighted_pool / its data gradient); deterministic. */
size_t sg_seg_gather_sum_parts_workspace_bytes(int64_t seg_num, int64_t parts, int64_t nnz, int64_t feat_dim);
int sg_seg_gather_sum_parts_hip(float* dst, int64_t dst_group, int64_t dst_ld, const float* src, int64_t src_group,
                                int64_t src_ld, const float* weights, const int32_t* wpos, const int32_t* indices_p,
                                const int32_t* indptr_p, int64_t seg_num, int64_t parts, int64_t nnz, int64_t feat_dim,
                                int req, int act, float slope, void* workspace, size_t workspace_bytes, void* stream,
                                int64_t src_bytes);

/* Rating head in two passes (reference: InnerProductLayer layers.py:210-222 + gluon L2Loss, STAR-GCN.py:428-438, 612):
 * one gather whose per-edge weight is formed from the row it has just loaded,
 *     r_j = < src[indices[j]], other[seg(j)] > - y[j],     rows[seg] (+)= sum_j scale * (*scale_dev) * r_j * src[indices[j]],
 *     loss = loss_scale * sum_j r_j^2   (written when loss != NULL)
 * Pass A, edges grouped by user: src = item projections, other = user projections -> d(user projections) and the loss;
 * pass B, edges grouped by item (optionally as a source-partitioned plan with `parts` ranges, indptr then has
 * parts * seg_num + 1 entries): src = user projections, other = item projections -> d(item projections).  y is in the
 * pass's edge order.  Replaces seg_take_k_corr + two weighted gathers + the loss kernels: the projections are read
 * twice instead of four times and no per-pair array is written or re-read.  feat_dim % 4 == 0, feat_dim <= 128. */
size_t sg_pair_l2_workspace_bytes(int64_t seg_num, int64_t parts, int64_t nnz, int64_t feat_dim);
int sg_pair_l2_hip(float* rows, float* loss, const float* src, const float* other, const float* y, const int32_t* indices,
                   const int32_t* indptr, int64_t seg_num, int64_t parts, int64_t nnz, int64_t feat_dim, float scale,
                   const float* scale_dev, float loss_scale, int req, void* workspace, size_t workspace_bytes, void* stream,
                   int64_t src_bytes);

/* ------------------------------------------------------------------------------------------------
 * (2) gradient of (1) w.r.t. data == reference `_contrib__backward_seg_take_k_corr_embed2(weights,
 *     ograd, indices, indptr)` seg_op.cc:700-703,718-752; CPU kernel :209-240; GPU seg_op.cu:747-790,
 *     882-926 (which radix-sorts the edges on EVERY call).
 *       ddata[b, indices[j], :] (+)= weights[b,j] * ograd[b, seg(j), :]
 *     Here the sort is hoisted into a reusable transposed plan (stable => same per-row summation
 *     order as the reference): t_indptr (total_ind_num+1), t_pos (nnz_t: original edge position j,
 *     increasing within a row), t_seg (nnz_t: segment of that edge).  Build it once with
 *     sg_build_transpose_cpu (plan construction is host-side, as gen_plan is in the reference,
 *     layers.py:260-337) and pass it on every call.
 *     Padding edges (j >= indptr[seg_num]) are ignored (true gradient of (1)); the reference CPU path
 *     charges them to segment 0 and its GPU path to the last non-empty segment.
 * ---------------------------------------------------------------------------------------------- */
size_t sg_seg_weighted_pool_bwd_data_workspace_bytes(int64_t batch, int64_t total_ind_num, int64_t nnz,
                                                     int64_t feat_dim);
int sg_seg_weighted_pool_bwd_data_hip(float* ddata, const float* weights, const float* ograd,
                                      const int32_t* t_indptr, const int32_t* t_pos, const int32_t* t_seg,
                                      int64_t batch, int64_t seg_num, int64_t total_ind_num, int64_t nnz,
                                      int64_t feat_dim, int req, void* workspace, size_t workspace_bytes,
                                      void* stream);
/* host: stable counting sort of the covered edges by indices[j]; outputs sized total_ind_num+1 / nnz / nnz */
int sg_build_transpose_cpu(int32_t* t_indptr, int32_t* t_pos, int32_t* t_seg, const int32_t* indices,
                           const int32_t* indptr, int64_t seg_num, int64_t total_ind_num, int64_t nnz);

/* ------------------------------------------------------------------------------------------------
 * (3) seg_take_k_corr == reference `_contrib_seg_take_k_corr` seg_op.cc:602-663, CPU :150-178,
 *     GPU seg_op.cu:573-664.  Also the gradient of (1) w.r.t. weights (seg_op.cc:703).
 *       dst[k,j] (+)= sum_c embed1[k, seg(j), c] * embed2[k, neighbor_ids[j], c]
 *     embed1 (K,node_num,C) embed2 (K,neighbor_node_num,C) -> dst (K,nnz); uncovered j get 0 on write.
 * ---------------------------------------------------------------------------------------------- */
int sg_seg_take_k_corr_hip(float* dst, const float* embed1, const float* embed2, const int32_t* neighbor_ids,
                           const int32_t* neighbor_indptr, int64_t K, int64_t node_num,
                           int64_t neighbor_node_num, int64_t nnz, int64_t feat_dim, int req, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (4) remaining segment-operator surface (reference seg_op.cc:339-600, 754-861)
 * ---------------------------------------------------------------------------------------------- */
/* seg_sum: data (B,nnz) -> dst (B,S)   [seg_op.cc:7-50] */
int sg_seg_sum_hip(float* dst, const float* data, const int32_t* indptr, int64_t batch, int64_t seg_num,
                   int64_t nnz, int req, void* stream);
/* seg_broadcast_{add,mul,to}: op 0 add, 1 mul, 2 to (lhs ignored)  [seg_op.cc:52-78]
 * dst[b,j] (+)= OP(lhs[b,j], rhs[b,seg(j)]) for covered j; on write, uncovered j get 0. */
int sg_seg_broadcast_hip(float* dst, const float* lhs, const float* rhs, const int32_t* indptr, int64_t batch,
                         int64_t seg_num, int64_t nnz, int op, int req, void* stream);
/* seg_softmax forward / backward  [seg_op.cc:80-148]; forward rejects kAddTo like the reference */
int sg_seg_softmax_hip(float* dst, const float* data, const int32_t* indptr, int64_t batch, int64_t seg_num,
                       int64_t nnz, int req, void* stream);
int sg_seg_softmax_bwd_hip(float* dst, const float* ograd, const float* val, const int32_t* indptr,
                           int64_t batch, int64_t seg_num, int64_t nnz, int req, void* stream);
/* seg_pool forward (sum/avg/max) [seg_op.cc:242-297]; pool_indices (B,S,C) int32 required for max:
 * argmax is the EDGE POSITION j, first maximum wins, empty segment -> value 0 / index -1. */
size_t sg_seg_pool_workspace_bytes(int64_t batch, int64_t seg_num, int64_t nnz, int64_t feat_dim);
int sg_seg_pool_hip(float* dst, int32_t* pool_indices, const float* data, const int32_t* indices,
                    const int32_t* indptr, int64_t batch, int64_t seg_num, int64_t total_ind_num, int64_t nnz,
                    int64_t feat_dim, int pool_type, int req, void* workspace, size_t workspace_bytes,
                    void* stream);
/* seg_pool backward [seg_op.cc:299-332], through the transposed plan of (2) */
size_t sg_seg_pool_bwd_workspace_bytes(int64_t batch, int64_t total_ind_num, int64_t nnz, int64_t feat_dim);
int sg_seg_pool_bwd_hip(float* ddata, const float* ograd, const int32_t* pool_indices, const int32_t* indptr,
                        const int32_t* t_indptr, const int32_t* t_pos, const int32_t* t_seg, int64_t batch,
                        int64_t seg_num, int64_t total_ind_num, int64_t nnz, int64_t feat_dim, int pool_type,
                        int req, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (5) per-rating-level dense mix: fp32 MFMA GEMM with fused bias / activation epilogue.
 *     Replaces MXNet FullyConnected/Dense at reference aggregators.py:141-145, layers.py:120,183,
 *     STAR-GCN.py:241-245,257.  Row-major:
 *       C[M,N] = act( opA(A)[M,K] * opB(B)[K,N] + bias[N] (+ C if accumulate) )
 *     transA = 0: A is (M,K) with leading dim lda;  1: A is stored (K,M).
 *     transB = 0: B is (K,N) with leading dim ldb;  1: B is stored (N,K)  (Linear weight layout).
 *     v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (no TF32 on gfx950).
 * ---------------------------------------------------------------------------------------------- */
size_t sg_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K, int transA);
int sg_gemm_f32_hip(float* C, int64_t ldc, const float* A, int64_t lda, int transA, const float* B,
                    int64_t ldb, int transB, int64_t M, int64_t N, int64_t K, const float* bias, int act,
                    float slope, int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* out = act(in) elementwise (reference common.py:32-57: leaky / relu / sigmoid / tanh), for the places where the
 * activation cannot ride on a GEMM / gather epilogue (after the all-reduce of a partitioned aggregate); out may alias in */
int sg_act_hip(float* out, const float* in, int64_t n, int act, float slope, void* stream);
/* dpre = dout * act'(.) evaluated from the activation OUTPUT `out` (all supported activations allow it) */
int sg_act_bwd_hip(float* dpre, const float* dout, const float* out, int64_t n, int act, float slope,
                   void* stream);
/* dst[N] (+)= column sums of X (M,N) with leading dim ldx (bias gradient) */
size_t sg_colsum_workspace_bytes(int64_t M, int64_t N);
int sg_colsum_hip(float* dst, const float* X, int64_t ldx, int64_t M, int64_t N, int req, void* workspace,
                  size_t workspace_bytes, void* stream);
/* Both at once for a Dense layer's backward (dpre = dout * act'(out) AND dbias = column sums of dpre, one pass over
 * dout / out; all three matrices dense M x N).  Same partial-sum order as sg_colsum_hip(dpre): identical results. */
int sg_act_bwd_colsum_hip(float* dpre, float* dbias, const float* dout, const float* out, int64_t M, int64_t N, int act,
                          float slope, int req, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (6) masked embedding gather (reference STAR-GCN.py:264-300 Net.get_embed) and row take / its grad
 *       id' = noise ? noise[ids[i]] : ids[i];  out[i,:] = (id' == -1) ? 0 : table[id', :]
 * ---------------------------------------------------------------------------------------------- */
int sg_masked_embed_hip(float* out, const float* table, const int32_t* ids, const int32_t* noise,
                        int64_t n_ids, int64_t n_rows, int64_t dim, void* stream);
/* dtable[id', :] += dout[i,:] (atomic-free: through a transposed plan built on (ids -> id')) is done by
 * calling sg_seg_gather_sum_hip on the plan; this helper resolves id' for plan building. */
int sg_resolve_ids_hip(int32_t* resolved, const int32_t* ids, const int32_t* noise, int64_t n_ids,
                       void* stream);

/* rating loss of the reference (gluon L2Loss, STAR-GCN.py:550,612): *loss = scale * sum_i 0.5 (pred_i - target_i)^2 and,
 * in the same pass, grad_i = scale * (pred_i - target_i) (grad may be NULL).  Two-pass fixed-order reduction. */
size_t sg_l2_loss_workspace_bytes(int64_t n);
int sg_l2_loss_hip(float* loss, float* grad, const float* pred, const float* target, int64_t n, float scale,
                   void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (7) host graph helpers (reference GraphSampler/graph_sampler.cpp; `_cpu` = host pointers)
 * ---------------------------------------------------------------------------------------------- */
/* get_support  graph_sampler.cpp:393-420 */
int sg_get_support_cpu(float* support, const int32_t* row_degrees, const int32_t* col_degrees,
                       const int32_t* end_points, const int32_t* ind_ptr, int64_t row_num, int symm);
/* multi_link_split_by_value  graph_sampler.cpp:277-376: out_pos (nnz) holds, level after level, the edge
 * positions of each level in CSR order; out_indptr (num_links*(row_num+1)); level_off (num_links+1). */
int sg_multi_link_split_cpu(int32_t* out_pos, int32_t* out_indptr, int64_t* level_off, const float* values,
                            const int32_t* ind_ptr, const float* multi_link, int64_t row_num,
                            int64_t num_links);
/* Fuse R per-level CSRs (the end_points_l / indptr_l / support_l lists of reference
 * aggregators.py:111-149) into ONE CSR over n_dst*R segments (segment i*R+r = level r of node i), and
 * its transpose over n_src*R segments whose indices are destination NODES i.  Sizes:
 * c_indptr n_dst*R+1, t_indptr n_src*R+1, c_idx/c_q/c_w/t_idx/t_q/t_w: sum_r indptr_l[r][n_dst].
 * c_q = c_idx*R + r and t_q = t_idx*R + r (may be NULL) address the row of level r inside an R-expanded
 * (N, R*width) matrix, so the same edge arrays also serve the un-split CSRs c_indptr[::R] / t_indptr[::R]. */
int sg_multilink_fuse_cpu(int32_t* c_indptr, int32_t* c_idx, int32_t* c_q, float* c_w, int32_t* t_indptr,
                          int32_t* t_idx, int32_t* t_q, float* t_w, const int32_t* const* end_points_l,
                          const int32_t* const* indptr_l, const float* const* support_l, int64_t num_links,
                          int64_t n_dst, int64_t n_src);

/* unique_inverse / unique_cnt  graph_sampler.h:465-534 (py_ext.cpp:612-627): unique ids in FIRST-OCCURRENCE order for
 * ids in [0, max_id]; inverse / counts may be NULL.  uniq, inverse, counts hold n entries; *n_uniq are used. */
int sg_unique_inverse_cpu(int32_t* uniq, int32_t* inverse, int32_t* counts, int64_t* n_uniq, const int32_t* ids,
                          int64_t n, int64_t max_id);
/* remove_edges_by_indices  graph_sampler.cpp:154-201: drop (row index, col index) pairs from a CSR with column-sorted
 * rows; out arrays hold ind_ptr[row_num] entries, *new_nnz are used. */
int sg_remove_edges_cpu(int32_t* out_end_points, float* out_values, int32_t* out_ind_ptr, int64_t* new_nnz,
                        const int32_t* end_points, const float* values, const int32_t* ind_ptr, int64_t row_num,
                        const int32_t* rm_rows, const int32_t* rm_cols, int64_t rm_num);
/* csr_submat / slice_csr_mat  graph_sampler.cpp:31-152 (py_ext.cpp:129-200): rows sel_rows (given order; NULL = all) and
 * columns with col_map[c] >= 0 (= new column index; NULL = all); surviving entries keep their order inside the row.
 * Outputs sized for the total length of the selected rows; out_ind_ptr has (sel_num | row_num) + 1 entries. */
int sg_csr_submat_cpu(int32_t* out_end_points, float* out_values, int32_t* out_ind_ptr, int64_t* out_nnz,
                      const int32_t* end_points, const float* values, const int32_t* ind_ptr, int64_t row_num,
                      const int32_t* sel_rows, int64_t sel_num, const int32_t* col_map);
/* random_sample_fix_neighbor  graph_sampler.cpp:742-779: per selected row all edges (<= neighbor_num of them, or
 * neighbor_num < 0) or neighbor_num positions drawn without replacement; row i's draw depends only on (seed, i)
 * and positions come back in increasing order.  sampled == NULL: only dst_ind_ptr (sel_num+1) is filled. */
int sg_sample_fix_neighbor_cpu(int32_t* sampled, int32_t* dst_ind_ptr, const int32_t* src_ind_ptr,
                               const int32_t* sel_indices, int64_t sel_num, int64_t neighbor_num, uint64_t seed);
/* gen_row_indices_by_indptr  py_ext.cpp:612-627 / graph.py:83-99: COO row index of every edge */
int sg_gen_row_indices_cpu(int32_t* row_indices, const int32_t* ind_ptr, int64_t row_num, int64_t nnz);
/* host side of the per-batch index plans (resident training loop) */
/* edge id (CSR position) of every (row index, col index) pair, -1 when it is not an edge; rows sorted by column.
 * Reference counterpart: the scipy fancy-index lookup of CSRMat.fetch_edges_by_ind (graph.py:595-613). */
int sg_edge_positions_cpu(int32_t* pos, const int32_t* end_points, const int32_t* ind_ptr, int64_t row_num,
                          const int32_t* rows, const int32_t* cols, int64_t n);
/* rating-head plan: batch pairs grouped by user (stable) + the stable transpose by item; arrays of n_pairs entries,
 * indptr n_user+1, t_indptr n_item+1 (reference STAR-GCN.py:428-438 takes both rows per pair instead) */
int sg_pair_plan_cpu(int32_t* order, int32_t* inv_order, int32_t* indptr, int32_t* items, int32_t* t_indptr,
                     int32_t* t_pos, int32_t* t_seg, int32_t* identity, const int32_t* user_idx,
                     const int32_t* item_idx, int64_t n_pairs, int64_t n_user, int64_t n_item);
/* row-take plan for the `take` ops of the path (reference layers.py:360,382 heter_sage; STAR-GCN.py:264-300,440-459)
 * and their atomic-free gradients: flags bit 0 identity, bit 1 every row taken at most once (inv_ids valid);
 * t_indptr n_rows+1, t_pos n, inv_ids n_rows */
int sg_take_plan_cpu(int32_t* t_indptr, int32_t* t_pos, int32_t* inv_ids, int32_t* flags, int64_t* covered,
                     const int32_t* ids, int64_t n, int64_t n_rows);

/* ------------------------------------------------------------------------------------------------
 * (8) the fused multi-link aggregator: reference MultiLinkGCNAggregator.hybrid_forward
 *     (aggregators.py:111-163: R x FullyConnected + R x seg_weighted_pool + add_n/concat + activation,
 *     and the FGradient graph MXNet builds from it) as ONE forward and ONE backward entry point:
 *        out = act( accum_r  A_r ( x W_r^T + 1 b_r^T ) )        accum = add_n ('sum') | concat ('stack')
 *     Parameters keep the reference layout (aggregators.py:86-97): `weights[r]` (units_per_level, in_dim)
 *     and `biases[r]` (units_per_level) are HOST arrays of R DEVICE pointers.  The graph structure is the
 *     resident plan built by sg_multilink_fuse_cpu (uploaded once by the caller).
 *     order: SG_ORDER_TRANSFORM_FIRST  H = x Wcat^T + bcat on the source side, one grouped gather;
 *            SG_ORDER_AGGREGATE_FIRST  Zext = [A_0 x | .. | A_{R-1} x | A_r 1 | 0] on the destination side,
 *                                      one GEMM with the packed [W_0 | .. | b | 0] matrix.
 *            SG_ORDER_AUTO             expansion on the smaller side (n_src <= n_dst -> transform first).
 *     out: (n_dst, units_per_level) for 'sum', (n_dst, R*units_per_level) for 'stack'.
 *     `saved` (sg_multilink_agg_saved_bytes; 0 for transform-first) is written by fwd and must be handed
 *     unchanged to bwd: it holds Zext, the only intermediate the backward needs besides x and out.
 *     bwd: dx (n_src,in_dim), dweights[r], dbiases[r] (host arrays of device pointers) may each be NULL;
 *     gradients are WRITTEN (kWriteTo).  Deterministic: no atomics anywhere on the path.
 * ---------------------------------------------------------------------------------------------- */
#define SG_ORDER_AUTO 0
#define SG_ORDER_TRANSFORM_FIRST 1
#define SG_ORDER_AGGREGATE_FIRST 2
#define SG_ORDER_FUSED 3                /* aggregate and contract in one kernel (8b); needs plan->fused, 256-wide in / out, 'sum' */
#define SG_ACCUM_SUM 0
#define SG_ACCUM_STACK 1
#define SG_MAX_LINKS 32
/* Source-range phases of one gather view of the plan (optional).  A gather over a source matrix a few times
 * the aggregate L2 is issued as num_phases = 2 launches, the second accumulating: phase 0 holds the view's edges whose
 * source row lies in the first half of the rows, phase 1 the rest, each a CSR over the same segments.  Built by
 * sg_gather_phases_build_hip; used by sg_multilink_agg_{fwd,bwd}_hip when present and the launch's source footprint is
 * cache-resident (24 MB .. 256 MB).  The sum of a segment is then (low rows) + (high rows): deterministic, and equal to the
 * single-launch value up to fp32 association. */
typedef struct sg_gather_phases {
  int32_t num_phases;                   /* 0: not phased; 2: phased */
  int32_t reserved;
  const int32_t* idx;                   /* (nnz) the view's source indices, phase 0's edges first */
  const int32_t* wpos;                  /* (nnz) position of each of those edges in the view's weight array (c_w / t_w) */
  const int32_t* indptr;                /* (2, segments + 1) CSR pointers of either phase, relative to its first edge */
  int64_t nnz_p[2];                     /* edges of either phase */
} sg_gather_phases;
#define SG_VIEW_C_Q_D 0                 /* (c_q, d_indptr)   transform-first forward, 'sum'   */
#define SG_VIEW_C_Q_C 1                 /* (c_q, c_indptr)   transform-first forward, 'stack' */
#define SG_VIEW_C_IDX_C 2               /* (c_idx, c_indptr) aggregate-first forward          */
#define SG_VIEW_T_IDX_T 3               /* (t_idx, t_indptr) transform-first backward, 'sum'  */
#define SG_VIEW_T_Q_T 4                 /* (t_q, t_indptr)   transform-first backward, 'stack' */
#define SG_VIEW_T_Q_S 5                 /* (t_q, s_indptr)   aggregate-first backward         */
#define SG_NUM_VIEWS 6
size_t sg_gather_phases_workspace_bytes(int64_t nnz);
int sg_gather_phases_build_hip(int32_t* idx_p, int32_t* wpos_p, int32_t* indptr_p, int32_t* nnz_p, const int32_t* indices,
                               const int32_t* indptr, int64_t seg_num, int64_t nnz, int64_t n_rows, void* workspace,
                               size_t workspace_bytes, void* stream);
/* dst (+)= act(sum) as sg_seg_gather_sum_hinted_hip, issued phase by phase (weights are read through ph->wpos) */
int sg_seg_gather_sum_phased_hip(float* dst, int64_t dst_group, int64_t dst_ld, const float* src, int64_t src_group,
                                 int64_t src_ld, const float* weights, const sg_gather_phases* ph, int64_t seg_num,
                                 int64_t feat_dim, int req, int act, float slope, void* workspace, size_t workspace_bytes,
                                 void* stream, int64_t src_bytes);
/* Level-major edge order of one CSR of the plan for the fused kernel of (8b) (sg_agg_fused_plan_build_hip); zero = absent */
typedef struct sg_fused_plan {
  const int32_t* f_ptr;                 /* tiles * R * 65 */
  const int32_t* f_idx;                 /* nnz */
  const float* f_w;                     /* nnz */
  const int32_t* tile_order;            /* tiles, or NULL */
} sg_fused_plan;
typedef struct sg_multilink_plan {      /* all pointers are DEVICE pointers; see sg_multilink_fuse_cpu */
  const int32_t* c_indptr;              /* n_dst*R+1 */
  const int32_t* c_idx;                 /* nnz: source node */
  const int32_t* c_q;                   /* nnz: source node*R + r */
  const float* c_w;                     /* nnz: support */
  const int32_t* t_indptr;              /* n_src*R+1 */
  const int32_t* t_idx;                 /* nnz: destination node */
  const int32_t* t_q;                   /* nnz: destination node*R + r */
  const float* t_w;                     /* nnz */
  const int32_t* d_indptr;              /* n_dst+1 = c_indptr[::R] */
  const int32_t* s_indptr;              /* n_src+1 = t_indptr[::R] */
  const float* rowsum;                  /* (n_dst, R) support sums per (node, level); needed by aggregate-first */
  int64_t n_dst, n_src, nnz;
  int32_t num_links;
  int32_t struct_bytes;                 /* sizeof(sg_multilink_plan) of the CALLER's header, or 0.  The members below this
                                         * line were added after the first ABI: the library reads them only when struct_bytes
                                         * says the caller's struct has them, so a caller built against the older header (this
                                         * field was `reserved`, documented as 0) or one that does not zero the struct can
                                         * never make it read garbage pointers. */
  sg_gather_phases phases[SG_NUM_VIEWS]; /* optional (zero = absent): source-range phases of the views, by SG_VIEW_* */
  sg_fused_plan fused[2];               /* optional: [0] over (c_indptr, c_idx, c_w) -- SG_ORDER_FUSED forward; [1] over
                                         * (t_indptr, t_idx, t_w) -- its data gradient */
} sg_multilink_plan;
int sg_multilink_agg_resolve_order(const sg_multilink_plan* plan, int order);
/* The order sg_multilink_agg_{fwd,bwd}_hip run for these sizes.  SG_ORDER_AUTO resolves to SG_ORDER_FUSED when the fused
 * kernel handles the widths ('sum', in_dim = units_per_level = 256) and the measured rule says it is the faster one (>= 2 levels,
 * >= 2^14 nodes on the smaller side, the larger side at most twice the smaller, the smaller side's R-expanded matrix beyond
 * 192 MB: multilink.hip, profiles/r5_fused_kernel.md section 7; SG_FUSED=0 / 1 in the environment forces never / whenever
 * supported), and otherwise to the rule of sg_multilink_agg_resolve_order.
 * The caller then attaches plan->fused (both entries) before the launch.
 * SG_ORDER_AUTO callers of the fwd / bwd / size entries: resolve ONCE with this function and pass the explicit order to
 * sg_multilink_agg_saved_bytes, _workspace_bytes, _fwd_hip and _bwd_hip of the same layer call.  Inside those entries AUTO also
 * looks at plan->fused (a plan without the level-major orders falls back to the unfused rule) and at SG_FUSED / SG_FUSED_SAVEZ:
 * a plan whose `fused` entries change between the forward and the backward would make the backward read `saved` in another
 * layout (Z (n_dst, R D) of the fused order vs Zext (n_dst, ld) of aggregate-first).  The Python host side does exactly this
 * (ops.multilink_resolve_order, then the order's name everywhere). */
int sg_multilink_agg_resolve_order2(const sg_multilink_plan* plan, int order, int64_t in_dim, int64_t units_per_level,
                                    int accum);
/* Which gather view (SG_VIEW_*) the fused entries would issue as two source-range phases for these sizes (backward = 0 / 1),
 * or -1 when they would not -- feature width below one 256-byte column slice, gathered matrix outside 24 MB .. 256 MB,
 * SG_GATHER_PHASES=0.  The caller builds phases (sg_gather_phases_build_hip) for THAT view only: a view that can never be
 * phased costs neither 8 bytes per edge of resident memory nor a build.  Negative codes below -1: invalid arguments. */
int sg_multilink_agg_phased_view(const sg_multilink_plan* plan, int64_t in_dim, int64_t units_per_level, int order, int accum,
                                 int backward);
size_t sg_multilink_agg_saved_bytes(const sg_multilink_plan* plan, int64_t in_dim, int64_t units_per_level,
                                    int order, int accum);
size_t sg_multilink_agg_workspace_bytes(const sg_multilink_plan* plan, int64_t in_dim, int64_t units_per_level,
                                        int order, int accum, int backward);
int sg_multilink_agg_fwd_hip(float* out, void* saved, const float* x, const float* const* weights,
                             const float* const* biases, const sg_multilink_plan* plan, int64_t in_dim,
                             int64_t units_per_level, int order, int accum, int act, float slope,
                             void* workspace, size_t workspace_bytes, void* stream);
int sg_multilink_agg_bwd_hip(float* dx, float* const* dweights, float* const* dbiases, const float* dout,
                             const float* out, const void* saved, const float* x, const float* const* weights,
                             const sg_multilink_plan* plan, int64_t in_dim, int64_t units_per_level, int order,
                             int accum, int act, float slope, void* workspace, size_t workspace_bytes,
                             void* stream);

/* ------------------------------------------------------------------------------------------------
 * (8b) aggregate -> contract in ONE kernel (csrc/agg_fused.hip): the aggregation of (8) without its R-expanded
 *     intermediate.  Reference: the same MultiLinkGCNAggregator.hybrid_forward (aggregators.py:141-160: R FullyConnected
 *     outputs written and re-read by R seg_weighted_pool ops).
 *        out (n_dst, ldo) = act( sum_r (A_r x) B_r + sum_r rowsum[:, r] b_r )        x (n_src, ldx), <= 256 wide; out <= 256 wide
 *     A workgroup owns 64 destination rows; per level the aggregate lives as two f16 planes in LDS and is multiplied by
 *     B_r on the matrix cores (three f16 MFMAs per product, error model of sg_gemm_backend 3 with one scale per
 *     (row, level) of the aggregate and per (level, 32 output columns) of B_r) while the next level is gathered.
 *     weights[r] (host array of R device pointers, leading dimension ldw): trans_w = 0 -> (256 out, 256 in), B_r = W_r^T
 *     (the forward); trans_w = 1 -> (256 in, 256 out), B_r = W_r (the data gradient over the transposed plan: out = dx,
 *     x = dpre).  biases (host array of R device pointers) / rowsum (n_dst, R) may be NULL: no bias term.
 *     zsave (n_dst, ldz), optional: receives the fp32 aggregates [r * 256 + k] (what the weight gradient contracts with).
 *     Plan: sg_agg_fused_plan_build_hip reorders the edges of every 64-row tile of a (row, level)-major CSR -- segment
 *     i * R + r at indptr[i * R + r], the c_* / t_* arrays of sg_multilink_plan -- level-major: f_ptr (tiles * R * 65
 *     absolute edge offsets, 65 per (launch slot, level)), f_idx / f_w (nnz; the edges stay inside their tile's range),
 *     optional f_pos (nnz: source position of every edge, for sg_agg_fused_refresh_hip after the weights were rewritten).
 *     tile_order (tiles; NULL = identity): the tile of every launch slot, e.g. by descending edge count; must be the
 *     same array at build and at launch.  nt_loads != 0: gathered rows are read non-temporally (sources far beyond the
 *     Infinity Cache: keeps B_r's planes in L2).  Deterministic; no atomics.
 * ---------------------------------------------------------------------------------------------- */
int64_t sg_agg_fused_tiles(int64_t n_dst);
int sg_agg_fused_supported(int64_t in_dim, int64_t out_dim, int32_t num_links);
int sg_agg_fused_plan_build_hip(int32_t* f_ptr, int32_t* f_idx, float* f_w, int32_t* f_pos, const int32_t* tile_order,
                                const int32_t* indptr, const int32_t* indices, const float* weights, int64_t n_dst,
                                int32_t num_links, int64_t nnz, void* stream);
int sg_agg_fused_refresh_hip(float* f_w, const int32_t* f_pos, const float* weights, int64_t nnz, void* stream);
size_t sg_agg_fused_workspace_bytes(int32_t num_links);
int sg_agg_fused_hip(float* out, int64_t ldo, float* zsave, int64_t ldz, const float* x, int64_t ldx,
                     const float* const* weights, int64_t ldw, int trans_w, const float* const* biases,
                     const float* rowsum, const int32_t* f_ptr, const int32_t* f_idx, const float* f_w,
                     const int32_t* tile_order, int64_t n_dst, int64_t n_src, int32_t num_links, int64_t nnz,
                     int64_t in_dim, int64_t out_dim, int act, float slope, int nt_loads, void* workspace,
                     size_t workspace_bytes, void* stream);      /* n_src: rows of x, or 0 = unknown (rows are then addressed
                                                                  * with 64-bit pointers instead of 32-bit buffer offsets) */
/* The general form (round 6): rows of 4 .. 256 floats (multiple of 4) x 1 .. 256 output columns per level -- the reference's own
 * widths (EMBED.UNITS 32 / 64 -> GCN.AGG.UNITS 250, experiments/cfg, the yml files) -- and both accumulations of
 * MultiLinkGCNAggregator (aggregators.py:79-81, 151-159):
 *     accum = SG_ACCUM_STACK: out (n_dst, ldo >= R * out_dim), column block r = act( (A_r x) B_r + rowsum[:, r] b_r )
 *     x_level_stride: level r gathers its rows from x + r * x_level_stride floats (multiple of 4): the data gradient of 'stack',
 *                     where level r reads column block r of the output gradient (accum = SG_ACCUM_SUM there: dx sums the levels)
 *     in_dim: floats per gathered row that exist in memory; k_valid <= in_dim: the contraction length present in the weights
 *             (rows padded to a multiple of 4 by the caller: the padding must be finite, the planes beyond k_valid are zero)
 *     zsave (n_dst, ldz): level r's fp32 aggregate at [r * in_dim, (r + 1) * in_dim).
 * The kernel is built for 256 x 256 tiles: narrower rows leave gather lanes idle and narrower outputs multiply zero-padded
 * planes -- supported and exact, not tuned; SG_ORDER_AUTO routes only 256 -> 256 'sum' here. */
int sg_agg_fused2_hip(float* out, int64_t ldo, float* zsave, int64_t ldz, const float* x, int64_t ldx, int64_t x_level_stride,
                      const float* const* weights, int64_t ldw, int trans_w, int64_t k_valid, const float* const* biases,
                      const float* rowsum, const int32_t* f_ptr, const int32_t* f_idx, const float* f_w,
                      const int32_t* tile_order, int64_t n_dst, int64_t n_src, int32_t num_links, int64_t nnz,
                      int64_t in_dim, int64_t out_dim, int accum, int act, float slope, int nt_loads, void* workspace,
                      size_t workspace_bytes, void* stream);
/* measurement aid for bench.py, as sg_gather_profile_*: HIP events around every fused launch on its own stream.  read():
 * (elapsed ms, edges, 1 when the launch also wrote the aggregates) per launch, in launch order; clears the records. */
int sg_agg_fused_profile_enable(int on);
int64_t sg_agg_fused_profile_read(float* ms, int64_t* nnz, int32_t* zsave, int64_t capacity);

/* ------------------------------------------------------------------------------------------------
 * (9) per-batch edge removal on the device.  Reference: HeterGraph.remove_edges_by_id in both
 *     directions (graph.py:952-974 -> graph_sampler.cpp:154-201), then fresh degrees + support (graph.py:401-429), a
 *     new plan (layers.py:260-337) and new uploads (layers.py:366-377) on EVERY training iteration.
 *     Here the full-graph plan stays resident and only its weights are rewritten: for every edge e of the graph
 *         w(e) = removed(e) ? 0 : support(d_row'(e), d_col'(e))        d' = degrees after the removal
 *     is stored at w_out[o][pos[o][e]] for each of the n_out weight arrays (c_w / t_w of the resident plans; pos[o] maps
 *     the edge id to its slot in that array).  support = sqrt(1/d_row/d_col) | 1/d_row for transposed[o] = 0 and sqrt(1/d_col/d_row) | 1/d_col for
 *     transposed[o] = 1 (symm | not): the expression sg_get_support_cpu evaluates for that matrix, bit-identical.  Edge ids: position in
 *     the (edge_row, edge_col) COO; ids outside [0, nnz) are ignored, duplicates count once.  n_rm = 0 restores the
 *     full graph.  w_out / pos / transposed are HOST arrays (of device pointers / ints).
 * ---------------------------------------------------------------------------------------------- */
#define SG_MAX_MASK_OUT 16
size_t sg_mask_edges_workspace_bytes(int64_t n_rows, int64_t n_cols, int64_t nnz);
int sg_mask_edges_hip(float* const* w_out, const int32_t* const* pos, const int32_t* transposed, int32_t n_out,
                      const int32_t* edge_row, const int32_t* edge_col, const int32_t* row_degrees,
                      const int32_t* col_degrees, const int32_t* rm_edges, int64_t n_rm, int64_t n_rows,
                      int64_t n_cols, int64_t nnz, int symm, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * (10) measurement aid for bench.py's `roofline`: while enabled, every seg_gather launch (direct, or issued inside
 *      sg_multilink_agg_*_hip) is bracketed with HIP events on the stream it is launched on.  enable(1) clears old
 *      records and returns the previous state; read() synchronises the recorded events, returns up to `capacity`
 *      (elapsed ms, edges visited, feature width) triples in launch order and clears the records.
 * ---------------------------------------------------------------------------------------------- */
int sg_gather_profile_enable(int on);
int64_t sg_gather_profile_read(float* ms, int64_t* nnz, int64_t* feat_dim, int64_t capacity);
/* same, plus the footprint of the gathered matrix of every launch (the caller's hint, 0 = unknown) */
int64_t sg_gather_profile_read2(float* ms, int64_t* nnz, int64_t* feat_dim, int64_t* src_bytes, int64_t capacity);
/* tuning aid (tests, profiling scripts): column-slice count of eligible gather launches (`slices`) and of EVERY launch
 * it divides (`slices_force`); 0 = library default, -1 = leave unchanged.  The environment variables SG_GATHER_SLICES /
 * SG_GATHER_SLICES_FORCE give the initial values and are read once, at the first launch.  Returns 0. */
int sg_gather_tuning(int slices, int slices_force);
/* tuning aid: GEMM backend of sg_gemm_f32_hip -- 0 exact-fp32 MFMA, 1 bf16x6, 2 x6v2 (wave-specialised bf16x6; both
 * bf16 backends drop terms <= 2^-24 |a b|: fp32-roundoff class), 3 f16x3 (the DEFAULT from K = 96 on; needs the workspace
 * sg_gemm_f32_workspace_bytes reports); -1 = SG_GEMM_BACKEND / build default.  Returns 0.
 * ERROR MODEL of backend 3 (block floating point on two f16 planes, three MFMAs per product): an operand is cut into blocks
 * of 32 rows of op(X) x 64 k (32 x 32 where a huge fp32 operand is split inside the kernel); a block is scaled by a power
 * of two that brings its largest FINITE magnitude into [2^14, 2^15).  A product term then carries <= 3 * 2^-22 = 7.2e-7
 * RELATIVE error as long as both factors lie within 2^18 of their block's maximum; an element further below keeps an
 * ABSOLUTE error of 2^-40 of that maximum instead (a row 1e6 times smaller than a neighbour in the same 32-row block still
 * comes out to ~1e-6 relative).  Over a dot product the errors are independent: ~7e-7 / sqrt(K) of sum |a||b|.  An inf or
 * NaN element makes every output it takes part in non-finite (an inf may come out as NaN: it meets both planes of the other
 * operand) and does not disturb the scale of its finite block-mates.  Operands with a wider dynamic range
 * inside a block, or callers that need <= 2^-24 per term, select the exact kernel: SG_GEMM_BACKEND=fp32 / sg_gemm_backend(0). */
int sg_gemm_backend(int backend);
/* measurement aid (bench.py `dense_roofline`): HIP events around every sg_gemm_f32_hip call (conversion passes and split-K
 * reduce included) on its own stream; read returns one record per call: ms, (M, N, K) at mnk[3 i ..], backend used */
int sg_gemm_profile_enable(int on);
int64_t sg_gemm_profile_read(float* ms, int64_t* mnk, int* backend, int64_t capacity);
/* tuning aid for backend 3 (f16x3): 0 automatic; 1-3, 6 and 7 force a plane-kernel geometry (6 = the default one, 7 = software-
 * pipelined fragment reads); 4 never use the in-kernel split of a huge fp32 operand ("hybrid"); 5 use it whenever the layout
 * allows, whatever the size (tests); 9 = 5 with one K tile of the fp32 operand in flight instead of two.  -1 = SG_X3_VARIANT / 0. */
int sg_gemm_x3_variant(int variant);
/* measurement aid: best-case streaming read with the gather's launch geometry: `workgroups` single-wave workgroups each
 * read `bursts` consecutive 1 KiB bursts (4 in flight) of a `bytes`-long buffer, wrapping around; bench.py uses it to
 * measure, in the same run, the Infinity-Cache and L2 ceilings that price cache-resident shapes */
int sg_stream_read_hip(const void* buf, int64_t bytes, int bursts, int64_t workgroups, float* sink, void* stream);
/* the same with a burst STRIDE per wave: wave w reads bursts w, w + stride, w + 2 stride, ... (mod bytes / 1024).  With
 * stride = workgroups the resident waves sweep one contiguous window through the buffer and never ask for the same burst, so
 * a buffer larger than the L2s is served by the Infinity Cache (or HBM) without sibling-wave L2 hits: the clean bandwidth of
 * that level (tools/mall_sweep.py -> profiles/r4_mall_sweep.txt; bench.py's bound for cache-resident launch classes) */
int sg_stream_read_strided_hip(const void* buf, int64_t bytes, int bursts, int64_t workgroups, int64_t stride, float* sink,
                               void* stream);

/* ------------------------------------------------------------------------------------------------
 * (11) DEVICE-side plan builders (csrc/plan_build.hip): hand-written wave64 exclusive scan + stable LSD radix sort.
 *      All pointers are device pointers, nothing is copied to the host, everything is enqueued on `stream`.
 *      Outputs are bit-identical to the host builders of (5) for in-range indices (stable order = original edge order).
 *
 *   sg_build_transpose_hip          = sg_build_transpose_cpu on the device.  Replaces the per-call iota + stable radix
 *                                     sort + GetSegId of the reference backward (seg_op.cu:882-926, :91-110).  Edges at
 *                                     positions >= indptr[seg_num] (padding) and out-of-range indices are dropped.
 *   sg_seg_weighted_pool_bwd_data_dev_hip
 *                                   = `_contrib__backward_seg_take_k_corr_embed2` with the reference operator's own
 *                                     inputs (ograd(K,nnz) = weights, embed1(K,N,C) = ograd rows, neighbor_ids,
 *                                     neighbor_indptr -- seg_op.cc:718-752): the transposed plan is built in the caller's
 *                                     workspace on every call (as the reference re-sorts on every call) and consumed by
 *                                     the gather.  Callers that keep a graph for more than one step should build the
 *                                     plan once (sg_build_transpose_hip) and call sg_seg_weighted_pool_bwd_data_hip.
 *   sg_multilink_fuse_hip           = sg_multilink_fuse_cpu on the device (per-level lists as HOST arrays of device
 *                                     pointers); nnz = sum_r indptr_l[r][n_dst].  d_indptr / s_indptr (optional): the
 *                                     un-split row pointers c_indptr[::R] / t_indptr[::R].
 *   sg_multilink_fuse_csr_hip       the same plan straight from a device-resident CSR (indptr over n_dst rows,
 *                                     end_points, per-edge level in [0, R), per-edge support): replaces
 *                                     sample_neighbors(num_neighbors = -1) + multi_link_split + fuse (graph.py:677-748,
 *                                     graph_sampler.cpp:277-376) for full-neighbourhood plans.  nnz must equal
 *                                     indptr[n_dst].  c_from / t_from (optional): CSR edge id held by every slot.
 *   sg_gen_row_indices_hip, sg_count_indices_hip, sg_get_support_hip, sg_level_index_hip
 *                                   device twins of gen_row_indices_by_indptr, the degree count, get_support
 *                                     (graph_sampler.cpp:393-420) and the level match of multi_link_split (:300-311).
 * ---------------------------------------------------------------------------------------------- */
size_t sg_build_transpose_workspace_bytes(int64_t seg_num, int64_t total_ind_num, int64_t nnz);
int sg_build_transpose_hip(int32_t* t_indptr, int32_t* t_pos, int32_t* t_seg, const int32_t* indices,
                           const int32_t* indptr, int64_t seg_num, int64_t total_ind_num, int64_t nnz, void* workspace,
                           size_t workspace_bytes, void* stream);
size_t sg_seg_weighted_pool_bwd_data_dev_workspace_bytes(int64_t batch, int64_t seg_num, int64_t total_ind_num,
                                                         int64_t nnz, int64_t feat_dim);
int sg_seg_weighted_pool_bwd_data_dev_hip(float* ddata, const float* weights, const float* ograd, const int32_t* indices,
                                          const int32_t* indptr, int64_t batch, int64_t seg_num, int64_t total_ind_num,
                                          int64_t nnz, int64_t feat_dim, int req, void* workspace, size_t workspace_bytes,
                                          void* stream);
size_t sg_multilink_fuse_workspace_bytes(int64_t num_links, int64_t n_dst, int64_t n_src, int64_t nnz);
int sg_multilink_fuse_hip(int32_t* c_indptr, int32_t* c_idx, int32_t* c_q, float* c_w, int32_t* t_indptr, int32_t* t_idx,
                          int32_t* t_q, float* t_w, int32_t* d_indptr, int32_t* s_indptr,
                          const int32_t* const* end_points_l, const int32_t* const* indptr_l,
                          const float* const* support_l, int64_t num_links, int64_t n_dst, int64_t n_src, int64_t nnz,
                          void* workspace, size_t workspace_bytes, void* stream);
int sg_multilink_fuse_csr_hip(int32_t* c_indptr, int32_t* c_idx, int32_t* c_q, float* c_w, int32_t* t_indptr,
                              int32_t* t_idx, int32_t* t_q, float* t_w, int32_t* d_indptr, int32_t* s_indptr,
                              int32_t* c_from, int32_t* t_from, const int32_t* indptr, const int32_t* end_points,
                              const int32_t* level, const float* support, int64_t num_links, int64_t n_dst,
                              int64_t n_src, int64_t nnz, void* workspace, size_t workspace_bytes, void* stream);
/* keys[j] = part(src_ids[j]) * n_seg + segment(j) for the source-partitioned gather plan (parts <= 64; bounds[1..parts)
 * = first source row of each part, device array); padding positions get parts * n_seg. */
int sg_part_keys_hip(int32_t* keys, const int32_t* src_ids, const int32_t* indptr, const int32_t* bounds, int64_t parts,
                     int64_t n_seg, int64_t n, void* stream);
int sg_gen_row_indices_hip(int32_t* edge_row, const int32_t* ind_ptr, int64_t row_num, int64_t nnz, void* stream);
int sg_count_indices_hip(int32_t* counts, const int32_t* idx, int64_t n, int64_t total, void* stream);
int sg_get_support_hip(float* support, const int32_t* row_degrees, const int32_t* col_degrees, const int32_t* end_points,
                       const int32_t* edge_row, int64_t nnz, int symm, void* stream);
int sg_level_index_hip(int32_t* level, const float* values, const float* multi_link, int64_t n, int64_t num_links,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * (12) per-iteration samplers and batch plans ON THE DEVICE.  Reference: DataIterator.rating_sampler /
 *      recon_nodes_sampler (iterators.py:264-370) draw with numpy's Mersenne Twister on the host and every batch's index
 *      arrays are uploaded.  The device samplers are COUNTER-BASED (a keyed 4-round Feistel bijection of the index range
 *      with cycle walking: element i of the sample depends on (seed, counter, i) only): the same distribution -- a
 *      uniform sample without replacement -- but deliberately not the reference's random stream.
 *   sg_sample_distinct_hip     k distinct uniform elements of [0, n)
 *   sg_recon_mask_hip          noise / reconstruction-node arrays of one node type
 *   sg_sort_i32_hip, sg_bounds_from_sorted_hip, sg_gather_i32_hip, sg_inverse_index_hip
 *                              the pieces a batch plan is made of: sorted batch edge ids -> (user, item) pairs grouped by
 *                              user (CSR) -> transpose (sg_build_transpose_hip); unique take -> inverse index
 * ---------------------------------------------------------------------------------------------- */
int sg_sample_distinct_hip(int32_t* out, int64_t n, int64_t k, uint64_t seed, uint64_t counter, void* stream);
int sg_recon_mask_hip(int32_t* noise, int32_t* recon, int64_t n, int64_t k, float p_zero, uint64_t seed, uint64_t counter,
                      void* stream);
/* the same draws with the counter taken as counter + *dev_counter (device pointer or NULL): a captured hipGraph freezes its
 * kernel arguments, so a replayed training iteration advances its draw through device memory (sg_counter_add_hip) */
int sg_sample_distinct_dev_hip(int32_t* out, int64_t n, int64_t k, uint64_t seed, uint64_t counter,
                               const uint64_t* dev_counter, void* stream);
int sg_recon_mask_dev_hip(int32_t* noise, int32_t* recon, int64_t n, int64_t k, float p_zero, uint64_t seed,
                          uint64_t counter, const uint64_t* dev_counter, void* stream);
/* inductive form (iterators.py:332-346, "nodes unseen in the training graph are masked as -1"): `cand` = the m distinct node
 * ids that occur in the training graph; noise = -1 for every other node, recon = k distinct CANDIDATES */
int sg_recon_mask_cand_dev_hip(int32_t* noise, int32_t* recon, int64_t n, const int32_t* cand, int64_t m, int64_t k,
                               float p_zero, uint64_t seed, uint64_t counter, const uint64_t* dev_counter, void* stream);
int sg_counter_add_hip(uint64_t* counter, uint64_t v, void* stream);
size_t sg_sort_i32_workspace_bytes(int64_t n);
int sg_sort_i32_hip(int32_t* keys_out, int32_t* vals_out, const int32_t* keys, const int32_t* vals, int64_t n,
                    int64_t max_key, void* workspace, size_t workspace_bytes, void* stream);
int sg_bounds_from_sorted_hip(int32_t* out_indptr, const int32_t* sorted_keys, int64_t n, int64_t total, void* stream);
int sg_gather_i32_hip(int32_t* dst, const int32_t* src, const int32_t* pos, int64_t n, void* stream);
int sg_inverse_index_hip(int32_t* inv, const int32_t* ids, int64_t n_ids, int64_t n_rows, void* stream);

/* Device twins of the last host-only plan primitives.
 *   sg_unique_inverse_hip       unique_inverse / unique_cnt, graph_sampler.h:441-534 (py_ext.cpp:612-627): unique ids in
 *                               FIRST-OCCURRENCE order for ids in [0, max_id], + inverse, + counts (may be NULL).  Same
 *                               outputs as sg_unique_inverse_cpu; the number of unique ids goes to DEVICE memory
 *                               (*n_uniq_dev), *bad_dev (may be NULL) becomes 1 when an id lies outside [0, max_id].
 *                               Integer atomicMin / atomicAdd only: the result does not depend on execution order.
 *   sg_sample_fix_neighbor_hip  random_sample_fix_neighbor, graph_sampler.cpp:742-779: bit-identical to
 *                               sg_sample_fix_neighbor_cpu for the same seed (row i's draw depends on (seed, i) only,
 *                               positions ascending).  sampled == NULL: only dst_ind_ptr (sel_num + 1, device). */
size_t sg_unique_inverse_workspace_bytes(int64_t n, int64_t max_id);
int sg_unique_inverse_hip(int32_t* uniq, int32_t* inverse, int32_t* counts, int32_t* n_uniq_dev, int32_t* bad_dev,
                          const int32_t* ids, int64_t n, int64_t max_id, void* workspace, size_t workspace_bytes,
                          void* stream);
size_t sg_sample_fix_neighbor_workspace_bytes(int64_t sel_num);
int sg_sample_fix_neighbor_hip(int32_t* sampled, int32_t* dst_ind_ptr, const int32_t* src_ind_ptr,
                               const int32_t* sel_indices, int64_t sel_num, int64_t neighbor_num, uint64_t seed,
                               void* workspace, size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STARGCN_H_ */
