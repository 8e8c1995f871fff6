// edge_mask.hip -- per-batch edge removal on the device.
//
// Reference training step (experiments/STAR-GCN.py:583-600): the batch's own ratings are removed from the training
// graph in BOTH directions (graph.py:952-974 -> graph_sampler.cpp:154-201, a fresh CSR), degrees and the support
// 1/sqrt(d_row d_col) are recomputed for the new matrices (graph.py:414-429), a new plan is generated on the host
// (layers.py:260-337) and every plan array is uploaded again (layers.py:366-377) -- O(E) host work per iteration.
//
// Here the full-graph plan stays resident in HBM and only its WEIGHTS change: one pass marks the removed edges and
// decrements the degrees of their end points, a second pass writes, for every edge e of the graph,
//     w(e) = removed(e) ? 0 : support(d_row'(e), d_col'(e))
// into each weight array of the resident plans (c_w / t_w of every MultiLinkPlan, through precomputed position maps).
// A zero-weight edge adds an exact 0 to its segment sum, so the aggregation equals the reference's on the reduced
// graph, with the same per-row summation order; the support expression is the one of sg_get_support_cpu, evaluated
// with correctly rounded fp32 divide / sqrt (hipcc default), i.e. bit-identical to the host path.
#include "common.hpp"

namespace sg {
namespace {

struct MaskTable {
  float* w[SG_MAX_MASK_OUT];
  const int32_t* pos[SG_MAX_MASK_OUT];
  int32_t transposed[SG_MAX_MASK_OUT];
};

__global__ void mask_init_kernel(int32_t* __restrict__ rd, const int32_t* __restrict__ rd0, long long n_rows,
                                 int32_t* __restrict__ cd, const int32_t* __restrict__ cd0, long long n_cols,
                                 uint32_t* __restrict__ flag_words, long long n_words) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n_rows) rd[i] = rd0[i];
  if (i < n_cols) cd[i] = cd0[i];
  if (i < n_words) flag_words[i] = 0u;
}

// one bit per edge; atomicOr returns the old word, so an edge listed twice is counted once (integer atomics only:
// the result does not depend on the execution order)
__global__ void mask_mark_kernel(uint32_t* __restrict__ flag_words, int32_t* __restrict__ rd, int32_t* __restrict__ cd,
                                 const int32_t* __restrict__ edge_row, const int32_t* __restrict__ edge_col,
                                 const int32_t* __restrict__ rm, long long n_rm, long long nnz) {
  const long long k = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (k >= n_rm) return;
  const long long e = rm[k];
  if (e < 0 || e >= nnz) return;   // ids that are not edges of this graph are ignored, like graph_sampler.cpp:154-201
  const uint32_t bit = 1u << (e & 31);
  const uint32_t old = atomicOr(&flag_words[e >> 5], bit);
  if (!(old & bit)) {
    atomicSub(&rd[edge_row[e]], 1);
    atomicSub(&cd[edge_col[e]], 1);
  }
}

__global__ void mask_write_kernel(MaskTable t, int n_out, const uint32_t* __restrict__ flag_words,
                                  const int32_t* __restrict__ rd, const int32_t* __restrict__ cd,
                                  const int32_t* __restrict__ edge_row, const int32_t* __restrict__ edge_col,
                                  long long nnz, int symm) {
  const long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  const bool removed = (flag_words[e >> 5] >> (e & 31)) & 1u;
  const int32_t dr = rd[edge_row[e]], dc = cd[edge_col[e]];
  // the symmetric support of the transposed matrix divides in the other order ((1/d_col)/d_row): same value up to
  // the last ulp, kept separate so that every array is bit-identical to what sg_get_support_cpu gives for ITS matrix
  float w_symm = 0.f, w_symm_t = 0.f, w_row = 0.f, w_col = 0.f;
  if (!removed) {
    if (symm) {
      if (dr != 0 && dc != 0) {
        w_symm = sqrtf(1.0f / static_cast<float>(dr) / static_cast<float>(dc));
        w_symm_t = sqrtf(1.0f / static_cast<float>(dc) / static_cast<float>(dr));
      }
    } else {
      w_row = dr == 0 ? 0.f : 1.0f / static_cast<float>(dr);
      w_col = dc == 0 ? 0.f : 1.0f / static_cast<float>(dc);
    }
  }
  for (int o = 0; o < n_out; ++o) {
    const float w = symm ? (t.transposed[o] ? w_symm_t : w_symm) : (t.transposed[o] ? w_col : w_row);
    t.w[o][t.pos[o][e]] = w;
  }
}

inline size_t al256(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }

}  // namespace
}  // namespace sg

using namespace sg;

SG_API size_t sg_mask_edges_workspace_bytes(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  if (n_rows < 0 || n_cols < 0 || nnz < 0) return 0;
  return al256(n_rows * sizeof(int32_t)) + al256(n_cols * sizeof(int32_t)) + al256(((nnz + 31) / 32) * sizeof(uint32_t)) + 256;
}

SG_API int sg_mask_edges_hip(float* const* w_out, const int32_t* const* pos, const int32_t* transposed, int32_t n_out,
                             const int32_t* edge_row, const int32_t* edge_col, const int32_t* row_degrees,
                             const int32_t* col_degrees, const int32_t* rm_edges, int64_t n_rm, int64_t n_rows,
                             int64_t n_cols, int64_t nnz, int symm, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (n_rows < 0 || n_cols < 0 || nnz < 0 || n_rm < 0) return fail(SG_ERR_INVALID, "negative dimension");
  if (n_out < 1 || n_out > SG_MAX_MASK_OUT) return fail(SG_ERR_INVALID, "n_out %d outside [1, %d]", n_out, SG_MAX_MASK_OUT);
  if (!w_out || !pos) return fail(SG_ERR_INVALID, "w_out / pos is null");
  if (nnz == 0) return SG_OK;
  if (!edge_row || !edge_col || !row_degrees || !col_degrees || (n_rm > 0 && !rm_edges))
    return fail(SG_ERR_INVALID, "null graph array");
  const size_t need = sg_mask_edges_workspace_bytes(n_rows, n_cols, nnz);
  if (!workspace || workspace_bytes < need)
    return fail(SG_ERR_WORKSPACE, "edge-mask workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  MaskTable t;
  for (int o = 0; o < SG_MAX_MASK_OUT; ++o) {
    t.w[o] = nullptr; t.pos[o] = nullptr; t.transposed[o] = 0;
  }
  for (int o = 0; o < n_out; ++o) {
    if (!w_out[o] || !pos[o]) return fail(SG_ERR_INVALID, "w_out[%d] / pos[%d] is null", o, o);
    t.w[o] = w_out[o]; t.pos[o] = pos[o]; t.transposed[o] = transposed ? transposed[o] : 0;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  int32_t* rd = reinterpret_cast<int32_t*>(base);
  int32_t* cd = reinterpret_cast<int32_t*>(base + al256(n_rows * sizeof(int32_t)));
  uint32_t* flags = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(cd) + al256(n_cols * sizeof(int32_t)));
  const long long n_words = (nnz + 31) / 32;
  const long long n_init = std::max<long long>(std::max<long long>(n_rows, n_cols), n_words);
  hipLaunchKernelGGL(mask_init_kernel, dim3(static_cast<unsigned>((n_init + 255) / 256)), dim3(256), 0, st, rd,
                     row_degrees, static_cast<long long>(n_rows), cd, col_degrees, static_cast<long long>(n_cols), flags,
                     n_words);
  if (n_rm > 0)
    hipLaunchKernelGGL(mask_mark_kernel, dim3(static_cast<unsigned>((n_rm + 255) / 256)), dim3(256), 0, st, flags, rd, cd,
                       edge_row, edge_col, rm_edges, static_cast<long long>(n_rm), static_cast<long long>(nnz));
  hipLaunchKernelGGL(mask_write_kernel, dim3(static_cast<unsigned>((nnz + 255) / 256)), dim3(256), 0, st, t, n_out, flags,
                     rd, cd, edge_row, edge_col, static_cast<long long>(nnz), symm);
  return check_launch("mask_write_kernel");
}
