// embed.hip -- masked embedding gather of the STAR-GCN reconstruction path (reference
// experiments/STAR-GCN.py:264-300 Net.get_embed):
//   id' = noise ? noise[ids[i]] : ids[i];   out[i,:] = (id' == -1) ? 0 : table[id', :]
// The reference composes take -> (!= -1) -> mul -> Embedding -> mul as five MXNet ops; here it is one
// coalesced row copy (float4 when dim % 4 == 0).  Its gradient is a segment sum over the plan built on the
// resolved ids (sg_resolve_ids_hip -> sg_build_transpose_cpu -> sg_seg_gather_sum_hip), atomic-free.
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

}  // namespace sg

using namespace sg;

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
