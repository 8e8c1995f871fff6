// embed.hip -- masked embedding gather of the STAR-GCN reconstruction path (reference
// experiments/STAR-GCN.py:264-300 Net.get_embed):
//   id' = noise ? noise[ids[i]] : ids[i];   out[i,:] = (id' == -1) ? 0 : table[id', :]
// The reference composes take -> (!= -1) -> mul -> Embedding -> mul as five MXNet ops; here it is one
// coalesced row copy (float4 when dim % 4 == 0).  Its gradient is a segment sum over the plan built on the
// resolved ids (sg_resolve_ids_hip -> sg_build_transpose_cpu -> sg_seg_gather_sum_hip), atomic-free.
#include <algorithm>

#include "common.hpp"

namespace sg {

template <int VEC>
__global__ void masked_embed_kernel(float* __restrict__ out, const float* __restrict__ table,
                                    const int32_t* __restrict__ ids, const int32_t* __restrict__ noise,
                                    long long n_ids, int dim) {
  const int per_row = dim / VEC;
  const long long total = n_ids * per_row;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += stride) {
    const long long i = e / per_row;
    const int c = static_cast<int>(e - i * per_row) * VEC;
    int id = ids[i];
    if (noise) id = noise[id];
    float v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = 0.f;
    if (id >= 0) {
      const float* src = table + static_cast<long long>(id) * dim + c;
      if (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(src);
        v[0] = t.x; v[1 % VEC] = t.y; v[2 % VEC] = t.z; v[3 % VEC] = t.w;
      } else {
        v[0] = src[0];
      }
    }
    float* o = out + i * dim + c;
    if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]);
    else o[0] = v[0];
  }
}

__global__ void resolve_ids_kernel(int32_t* __restrict__ resolved, const int32_t* __restrict__ ids,
                                   const int32_t* __restrict__ noise, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) resolved[i] = noise ? noise[ids[i]] : ids[i];
}

// ---- rating loss: gluon L2Loss = 0.5 (x - y)^2 (reference STAR-GCN.py:550,612) with its gradient in the same pass ----
constexpr int kLossBlocks = 1024;
__global__ __launch_bounds__(256) void l2_loss_partial_kernel(float* __restrict__ partial, float* __restrict__ grad,
                                                              const float* __restrict__ pred,
                                                              const float* __restrict__ target, long long n, float scale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
       i += static_cast<long long>(gridDim.x) * blockDim.x) {
    const float d = pred[i] - target[i];
    acc = fmaf(0.5f * d, d, acc);
    if (grad) grad[i] = scale * d;
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void l2_loss_final_kernel(float* __restrict__ loss, const float* __restrict__ partial,
                                                            int blocks, float scale) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < blocks; i += 256) acc += partial[i];
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) *loss = scale * ((red[0] + red[1]) + (red[2] + red[3]));
}

}  // namespace sg

using namespace sg;

SG_API size_t sg_l2_loss_workspace_bytes(int64_t n) { (void)n; return kLossBlocks * sizeof(float); }

SG_API int sg_l2_loss_hip(float* loss, float* grad, const float* pred, const float* target, int64_t n, float scale,
                          void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0) return fail(SG_ERR_INVALID, "negative n");
  if (!loss || (n > 0 && (!pred || !target))) return fail(SG_ERR_INVALID, "null pointer argument");
  if (!workspace || workspace_bytes < sg_l2_loss_workspace_bytes(n)) return fail(SG_ERR_WORKSPACE, "l2_loss workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int blocks = static_cast<int>(std::min<int64_t>(kLossBlocks, std::max<int64_t>(1, (n + 255) / 256)));
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(l2_loss_partial_kernel, dim3(blocks), dim3(256), 0, st, partial, grad, pred, target,
                     static_cast<long long>(n), scale);
  hipLaunchKernelGGL(l2_loss_final_kernel, dim3(1), dim3(256), 0, st, loss, partial, blocks, scale);
  return check_launch("l2_loss");
}

SG_API int sg_masked_embed_hip(float* out, const float* table, const int32_t* ids, const int32_t* noise, int64_t n_ids,
                               int64_t n_rows, int64_t dim, void* stream) {
  (void)n_rows;
  if (n_ids < 0 || dim < 0 || dim >= (1ll << 31)) return fail(SG_ERR_INVALID, "bad masked_embed shape");
  if (n_ids == 0 || dim == 0) return SG_OK;
  if (!out || !table || !ids) return fail(SG_ERR_INVALID, "null pointer argument");
  const bool v4 = (dim % 4 == 0) && aligned(out, 16) && aligned(table, 16);
  const long long total = n_ids * (v4 ? dim / 4 : dim);
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (v4) hipLaunchKernelGGL(masked_embed_kernel<4>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, out, table, ids, noise, static_cast<long long>(n_ids), static_cast<int>(dim));
  else hipLaunchKernelGGL(masked_embed_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, st, out, table, ids, noise, static_cast<long long>(n_ids), static_cast<int>(dim));
  return check_launch("masked_embed");
}

SG_API int sg_resolve_ids_hip(int32_t* resolved, const int32_t* ids, const int32_t* noise, int64_t n_ids, void* stream) {
  if (n_ids < 0) return fail(SG_ERR_INVALID, "negative n_ids");
  if (n_ids == 0) return SG_OK;
  hipLaunchKernelGGL(resolve_ids_kernel, dim3(static_cast<unsigned>((n_ids + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), resolved, ids, noise, static_cast<long long>(n_ids));
  return check_launch("resolve_ids");
}
