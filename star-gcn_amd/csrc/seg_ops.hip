// seg_ops.hip -- the rest of the reference's segment-operator surface on gfx950 (wave64):
//   seg_take_k_corr (reference seg_op.cc:150-178, seg_op.cu:573-664), seg_sum (:7-50), seg_broadcast_*
//   (:52-78), seg_softmax fwd/bwd (:80-148), seg_pool fwd/bwd (:242-332).
// None of these is executed by STAR-GCN training except through the gather kernel (seg_gather.hip);
// they exist for operator-API parity (reference test_seg_ops.py).  Simple wave-per-segment mappings with
// __shfl_xor reductions -- no 32-lane warp idioms (the reference's SumSharedMem, seg_op.cu:30-60, is
// warp-synchronous over 32 lanes and invalid on a 64-lane wavefront).
#include "common.hpp"

namespace sg {

int launch_gather(float* dst, int64_t dst_group, int64_t dst_ld, int64_t dst_bs, const float* src, int64_t src_group,
                  int64_t src_ld, int64_t src_bs, const float* w, int64_t w_bs, const int32_t* wpos,
                  const int32_t* idx, const int32_t* indptr, int64_t batch, int64_t seg_num, int64_t nnz, int64_t C,
                  int req, int mean, int act, float slope, void* workspace, size_t workspace_bytes, hipStream_t st,
                  int64_t src_bytes);
size_t gather_workspace_bytes(int64_t batch, int64_t nnz, int64_t C);

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

constexpr int kBlock = 256;             // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;

// ---------------------------------------------------------------------------------------------------
// seg_take_k_corr: edge-balanced like the gather kernel -- one wave per chunk of 256 consecutive edges.
//   1. the chunk's neighbour ids and the segment of every edge are staged in LDS once: segment starts inside the
//      chunk are marked (ds_max, so runs of empty segments resolve to the one that owns the edge) and a prefix-max
//      over the 256 slots turns the marks into a per-edge segment id;
//   2. each edge's dot product is taken by LPR lanes holding VEC consecutive channels each (float4 -> a 64-wide row
//      is 16 lanes, 4 edges per wave step), partial dots combined inside the lane group with __shfl_xor.  The edge
//      loop carries no dependence on the segment walk any more, so it is unrolled 4x: 8 independent row loads in
//      flight per lane instead of the id -> row -> reduce chain of round 1 (10 M pairs x 64: 336 -> see profiles/);
//   3. the 256 results leave through LDS as coalesced stores; edges past indptr[node_num] are zero-filled here
//      (kWriteTo) instead of by a separate launch.
// ---------------------------------------------------------------------------------------------------
constexpr int kCorrChunk = 256;

template <int VEC, int LPR>   // LPR > 0: lanes per edge, a row fits one pass of the lane group; 0: general (run-time lpr)
__global__ __launch_bounds__(kWave) void take_k_corr_kernel(float* __restrict__ dst, const float* __restrict__ e1,
                                                            const float* __restrict__ e2,
                                                            const int32_t* __restrict__ ids,
                                                            const int32_t* __restrict__ indptr, int node_num,
                                                            long long nbr_num, long long nnz, int C, int lpr_rt, int add) {
  constexpr bool ONE = LPR > 0;
  const int lpr = ONE ? LPR : lpr_rt;
  __shared__ int32_t s_id[kCorrChunk];
  __shared__ int32_t s_seg[kCorrChunk];
  __shared__ float s_out[kCorrChunk];
  const int lane = threadIdx.x;
  const int k = blockIdx.y;
  const long long E = indptr[node_num];
  const long long cb64 = static_cast<long long>(blockIdx.x) * kCorrChunk;
  float* out = dst + static_cast<long long>(k) * nnz;
  const int n_here = static_cast<int>(min(static_cast<long long>(kCorrChunk), nnz - cb64));   // slots of this chunk
  if (cb64 >= E) {                                   // nothing but uncovered positions
    if (!add)
      for (int i = lane; i < n_here; i += kWave) out[cb64 + i] = 0.f;
    return;
  }
  const int cb = static_cast<int>(cb64);
  const int ce = static_cast<int>(min(cb64 + kCorrChunk, E));
  const int n = ce - cb;
  // segments holding the first / last edge of the chunk: largest s with indptr[s] <= edge  (uniform binary searches)
  int lo = 0, hi = node_num;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (indptr[mid] <= cb) lo = mid; else hi = mid;
  }
  const int s0 = lo;
  hi = node_num;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (indptr[mid] <= ce - 1) lo = mid; else hi = mid;
  }
  const int s1 = lo;
  for (int i = lane; i < kCorrChunk; i += kWave) {
    s_seg[i] = i == 0 ? s0 : 0;
    s_id[i] = i < n ? ids[cb + i] : 0;
  }
  __syncthreads();
  for (int s = s0 + 1 + lane; s <= s1; s += kWave) atomicMax(&s_seg[indptr[s] - cb], s);   // 0 < indptr[s] - cb < n
  __syncthreads();
  {  // inclusive prefix-max over the 256 slots: 4 per lane, then across lanes
    int m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = s_seg[lane * 4 + q];
#pragma unroll
    for (int q = 1; q < 4; ++q) m[q] = max(m[q], m[q - 1]);
    int run = m[3];
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int up = __shfl_up(run, off);
      if (lane >= off) run = max(run, up);
    }
    const int before = __shfl_up(run, 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) s_seg[lane * 4 + q] = lane == 0 ? m[q] : max(m[q], before);
  }
  __syncthreads();

  const int epg = kWave / lpr;
  const int grp = lane / lpr, slot = lane % lpr;
  const float* base1 = e1 + static_cast<long long>(k) * node_num * C;
  const float* base2 = e2 + static_cast<long long>(k) * nbr_num * C;
  auto dot = [&](const float* row1, const float* row2, int c) -> float {
    if (VEC == 4) {
      const float4 a = *reinterpret_cast<const float4*>(row1 + c);
      const float4 b = *reinterpret_cast<const float4*>(row2 + c);
      return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
    }
    return row1[c] * row2[c];
  };
  if (ONE) {   // a row fits one pass of the lane group: nothing but independent loads inside the unrolled loop
    const bool has = slot * VEC < C;
    const int c = has ? slot * VEC : 0;
    constexpr int kBatch = 4;                       // edges per lane group in flight: 2 * kBatch independent row loads
    for (int i0 = 0; i0 < n; i0 += kBatch * epg) {  // 256 % (kBatch * epg) == 0, so every slot index stays < 256
      float acc[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int i = i0 + u * epg + grp;           // slots >= n hold id 0 / a valid segment: they read real rows
        acc[u] = dot(base1 + static_cast<long long>(s_seg[i]) * C, base2 + static_cast<long long>(s_id[i]) * C, c);
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        float v = has ? acc[u] : 0.f;
#pragma unroll
        for (int off = 1; off < (ONE ? LPR : 1); off <<= 1) v += __shfl_xor(v, off);
        if (slot == 0) s_out[i0 + u * epg + grp] = v;
      }
    }
  } else {
    for (int i0 = 0; i0 < n; i0 += epg) {
      const int i = i0 + grp;
      const bool live = i < n;
      const float* row1 = base1 + static_cast<long long>(live ? s_seg[i] : s0) * C;
      const float* row2 = base2 + static_cast<long long>(s_id[i]) * C;
      float acc = 0.f;
      for (int c = slot * VEC; c < C; c += lpr * VEC) acc += dot(row1, row2, c);
      for (int off = 1; off < lpr; off <<= 1) acc += __shfl_xor(acc, off);
      if (slot == 0) s_out[i] = acc;
    }
  }
  __syncthreads();
  for (int i = lane; i < n_here; i += kWave) {
    const float v = s_out[i];
    float* o = out + cb64 + i;
    if (i < n) *o = add ? (*o + v) : v;
    else if (!add) *o = 0.f;
  }
}

// zero dst[k, j] for j in [E, nnz) (uncovered positions) -- E read on device
__global__ void zero_uncovered_kernel(float* __restrict__ dst, const int32_t* __restrict__ indptr, int seg_num,
                                      long long nnz) {
  const long long E = indptr[seg_num];
  const long long j = E + static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j < nnz) dst[static_cast<long long>(blockIdx.y) * nnz + j] = 0.f;
}

// ---------------------------------------------------------------------------------------------------
// seg_sum / seg_softmax: G lanes per (segment, batch) with G in {4, 8, 16, 32, 64} chosen from the average segment
// length (a (user, level) segment of the rating graph holds 14 edges: a whole wave per segment leaves 3/4 of the lanes
// idle and needs 4x the waves); reductions stay inside the lane group (__shfl_xor with offsets < G).
// ---------------------------------------------------------------------------------------------------
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
static inline int lanes_per_segment(int64_t nnz, int64_t seg_num) {
  const int64_t avg = seg_num > 0 ? nnz / seg_num : 0;
  int g = 4;
  while (g < kWave && g < avg) g <<= 1;
  return g;
}
static inline dim3 group_grid(int64_t segs, int g, int64_t batch) {
  const int64_t per_block = kBlock / g;
  return dim3(static_cast<unsigned>((segs + per_block - 1) / per_block), static_cast<unsigned>(batch));
}

template <int G>
__global__ __launch_bounds__(kBlock) void seg_sum_kernel(float* __restrict__ dst, const float* __restrict__ data,
                                                         const int32_t* __restrict__ indptr, int seg_num, long long nnz,
                                                         int add) {
  const int slot = threadIdx.x & (G - 1);
  const long long seg = static_cast<long long>(blockIdx.x) * (kBlock / G) + threadIdx.x / G;
  const bool live = seg < seg_num;                       // dead groups run the shuffles with empty ranges
  const int k = blockIdx.y;
  const float* d = data + static_cast<long long>(k) * nnz;
  const int pb = live ? indptr[seg] : 0, pe = live ? indptr[seg + 1] : 0;
  float acc = 0.f;
  for (int j = pb + slot; j < pe; j += G) acc += d[j];
  acc = group_sum<G>(acc);
  if (live && slot == 0) {
    float* o = dst + static_cast<long long>(k) * seg_num + seg;
    *o = add ? (*o + acc) : acc;
  }
}

// ---------------------------------------------------------------------------------------------------
// seg_broadcast_{add,mul,to}: one wave per chunk of 256 edge positions; the segment of every position comes from the
// chunk-level staging seg_take_k_corr uses (segment starts marked with ds_max, prefix-max over the 256 slots) instead
// of a 17-step binary search per element.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kWave) void seg_broadcast_kernel(float* __restrict__ dst, const float* __restrict__ lhs,
                                                              const float* __restrict__ rhs,
                                                              const int32_t* __restrict__ indptr, int seg_num,
                                                              long long nnz, int op, int add) {
  __shared__ int32_t s_seg[kCorrChunk];
  const int lane = threadIdx.x;
  const int k = blockIdx.y;
  const long long E = indptr[seg_num];
  const long long cb64 = static_cast<long long>(blockIdx.x) * kCorrChunk;
  const int n_here = static_cast<int>(min(static_cast<long long>(kCorrChunk), nnz - cb64));
  float* out = dst + static_cast<long long>(k) * nnz + cb64;
  if (cb64 >= E) {
    if (!add)
      for (int i = lane; i < n_here; i += kWave) out[i] = 0.f;
    return;
  }
  const int cb = static_cast<int>(cb64);
  const int ce = static_cast<int>(min(cb64 + kCorrChunk, E));
  const int n = ce - cb;
  int lo = 0, hi = seg_num;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (indptr[mid] <= cb) lo = mid; else hi = mid;
  }
  const int s0 = lo;
  hi = seg_num;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (indptr[mid] <= ce - 1) lo = mid; else hi = mid;
  }
  const int s1 = lo;
  for (int i = lane; i < kCorrChunk; i += kWave) s_seg[i] = i == 0 ? s0 : 0;
  __syncthreads();
  for (int s = s0 + 1 + lane; s <= s1; s += kWave) atomicMax(&s_seg[indptr[s] - cb], s);
  __syncthreads();
  int m[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) m[q] = s_seg[lane * 4 + q];
#pragma unroll
  for (int q = 1; q < 4; ++q) m[q] = max(m[q], m[q - 1]);
  int run = m[3];
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int up = __shfl_up(run, off);
    if (lane >= off) run = max(run, up);
  }
  const int before = __shfl_up(run, 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) s_seg[lane * 4 + q] = lane == 0 ? m[q] : max(m[q], before);
  __syncthreads();
  const float* l = lhs ? lhs + static_cast<long long>(k) * nnz + cb64 : nullptr;
  const float* r = rhs + static_cast<long long>(k) * seg_num;
#pragma unroll
  for (int q = 0; q < 4; ++q) {          // position i = 64 q + lane: every wave instruction is one 256-byte burst
    const int i = q * kWave + lane;
    if (i < n) {
      const float rv = r[s_seg[i]];
      float v;
      if (op == 0) v = l[i] + rv;
      else if (op == 1) v = l[i] * rv;
      else v = rv;
      out[i] = add ? (out[i] + v) : v;
    } else if (i < n_here && !add) {
      out[i] = 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// seg_softmax forward / backward: G lanes per (segment, batch)
// ---------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(kBlock) void seg_softmax_kernel(float* __restrict__ dst, const float* __restrict__ data,
                                                             const int32_t* __restrict__ indptr, int seg_num,
                                                             long long nnz) {
  const int slot = threadIdx.x & (G - 1);
  const long long seg = static_cast<long long>(blockIdx.x) * (kBlock / G) + threadIdx.x / G;
  const bool live = seg < seg_num;
  const int k = blockIdx.y;
  const float* d = data + static_cast<long long>(k) * nnz;
  float* o = dst + static_cast<long long>(k) * nnz;
  const int pb = live ? indptr[seg] : 0, pe = live ? indptr[seg + 1] : 0;
  float m = -3.402823466e+38f;
  for (int j = pb + slot; j < pe; j += G) m = fmaxf(m, d[j]);
  m = group_max<G>(m);
  float s = 0.f;
  for (int j = pb + slot; j < pe; j += G) {
    const float e = expf(d[j] - m);
    o[j] = e;
    s += e;
  }
  s = group_sum<G>(s);
  for (int j = pb + slot; j < pe; j += G) o[j] = o[j] / s;
}

template <int G>
__global__ __launch_bounds__(kBlock) void seg_softmax_bwd_kernel(float* __restrict__ dst,
                                                                 const float* __restrict__ ograd,
                                                                 const float* __restrict__ val,
                                                                 const int32_t* __restrict__ indptr, int seg_num,
                                                                 long long nnz, int add) {
  const int slot = threadIdx.x & (G - 1);
  const long long seg = static_cast<long long>(blockIdx.x) * (kBlock / G) + threadIdx.x / G;
  const bool live = seg < seg_num;
  const long long off = static_cast<long long>(blockIdx.y) * nnz;
  const int pb = live ? indptr[seg] : 0, pe = live ? indptr[seg + 1] : 0;
  float s = 0.f;
  for (int j = pb + slot; j < pe; j += G) s = fmaf(ograd[off + j], val[off + j], s);
  s = group_sum<G>(s);
  for (int j = pb + slot; j < pe; j += G) {
    const float g = val[off + j] * (ograd[off + j] - s);
    dst[off + j] = add ? (dst[off + j] + g) : g;
  }
}

#define SG_BY_GROUP(G_, CALL)  \
  switch (G_) {                \
    case 4: { constexpr int G = 4; CALL; } break;    \
    case 8: { constexpr int G = 8; CALL; } break;    \
    case 16: { constexpr int G = 16; CALL; } break;  \
    case 32: { constexpr int G = 32; CALL; } break;  \
    default: { constexpr int G = 64; CALL; } break;  \
  }

// ---------------------------------------------------------------------------------------------------
// seg_pool max forward: one wave per (segment, batch); a lane owns VEC consecutive channels; the edges of the segment
// are taken four at a time (four independent id -> row loads in flight; the first version's one-edge loop was a
// dependent id -> row -> compare chain: 4.1 ms for 10 M edges x 256 channels against 0.55 ms for the sum).  Candidates
// are compared in CSR order with a strict '>' so the first maximum wins; empty segment -> 0 / -1 (seg_op.cc:264-283).
// ---------------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(kBlock) void seg_pool_max_kernel(float* __restrict__ dst, int32_t* __restrict__ arg,
                                                              const float* __restrict__ data,
                                                              const int32_t* __restrict__ indices,
                                                              const int32_t* __restrict__ indptr, int seg_num,
                                                              long long total, int C) {
  const int lane = threadIdx.x & (kWave - 1);
  const long long seg = static_cast<long long>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (seg >= seg_num) return;
  const int k = blockIdx.y;
  const int pb = indptr[seg], pe = indptr[seg + 1];
  const float* base = data + static_cast<long long>(k) * total * C;
  const long long orow = (static_cast<long long>(k) * seg_num + seg) * C;
  for (int c = lane * VEC; c < C; c += kWave * VEC) {
    float best[VEC];
    int bi[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { best[v] = (pe == pb) ? 0.f : -3.402823466e+38f; bi[v] = -1; }
    int j = pb;
    for (; j + 4 <= pe; j += 4) {
      float x[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* row = base + static_cast<long long>(indices[j + u]) * C + c;
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(row);
          x[u][0] = t.x; x[u][1 % VEC] = t.y; x[u][2 % VEC] = t.z; x[u][3 % VEC] = t.w;
        } else {
          x[u][0] = row[0];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (x[u][v] > best[v]) { best[v] = x[u][v]; bi[v] = j + u; }
    }
    for (; j < pe; ++j) {
      const float* row = base + static_cast<long long>(indices[j]) * C + c;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float xv = row[v];
        if (xv > best[v]) { best[v] = xv; bi[v] = j; }
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) { dst[orow + c + v] = best[v]; arg[orow + c + v] = bi[v]; }
  }
}

// seg_pool max backward over the transposed plan: ddata[n,c] (+)= sum_p ograd[t_seg[p],c] * (arg[t_seg[p],c]==t_pos[p]).
// A lane owns VEC consecutive channels; four transposed edges in flight (the terms are still added in plan order).
template <int VEC>
__global__ __launch_bounds__(kBlock) void seg_pool_max_bwd_kernel(float* __restrict__ ddata,
                                                                  const float* __restrict__ ograd,
                                                                  const int32_t* __restrict__ arg,
                                                                  const int32_t* __restrict__ t_indptr,
                                                                  const int32_t* __restrict__ t_pos,
                                                                  const int32_t* __restrict__ t_seg, int seg_num,
                                                                  long long total, int C, int add) {
  const int lane = threadIdx.x & (kWave - 1);
  const long long n = static_cast<long long>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
  if (n >= total) return;
  const int k = blockIdx.y;
  const int pb = t_indptr[n], pe = t_indptr[n + 1];
  const long long gbase = static_cast<long long>(k) * seg_num * C;
  float* o = ddata + (static_cast<long long>(k) * total + n) * C;
  for (int c = lane * VEC; c < C; c += kWave * VEC) {
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
    int p = pb;
    for (; p + 4 <= pe; p += 4) {
      float g[4][VEC];
      int a[4][VEC], pos[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long r = gbase + static_cast<long long>(t_seg[p + u]) * C + c;
        pos[u] = t_pos[p + u];
        if (VEC == 4) {
          const float4 tg = *reinterpret_cast<const float4*>(ograd + r);
          const int4 ta = *reinterpret_cast<const int4*>(arg + r);
          g[u][0] = tg.x; g[u][1 % VEC] = tg.y; g[u][2 % VEC] = tg.z; g[u][3 % VEC] = tg.w;
          a[u][0] = ta.x; a[u][1 % VEC] = ta.y; a[u][2 % VEC] = ta.z; a[u][3 % VEC] = ta.w;
        } else {
          g[u][0] = ograd[r];
          a[u][0] = arg[r];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += (a[u][v] == pos[u]) ? g[u][v] : 0.f;
    }
    for (; p < pe; ++p) {
      const long long r = gbase + static_cast<long long>(t_seg[p]) * C + c;
      const int pos = t_pos[p];
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += (arg[r + v] == pos) ? ograd[r + v] : 0.f;
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) o[c + v] = add ? (o[c + v] + acc[v]) : acc[v];
  }
}

// w_e[p] = 1 / len(segment of transposed edge p)   (avg-pool backward weights)
__global__ void inv_len_kernel(float* __restrict__ w, const int32_t* __restrict__ t_seg,
                               const int32_t* __restrict__ indptr, const int32_t* __restrict__ t_indptr,
                               long long total) {
  const long long p = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= t_indptr[total]) return;
  const int s = t_seg[p];
  w[p] = 1.f / static_cast<float>(indptr[s + 1] - indptr[s]);
}

static inline dim3 seg_grid(int64_t segs, int64_t batch) {
  return dim3(static_cast<unsigned>((segs + kWavesPerBlock - 1) / kWavesPerBlock), static_cast<unsigned>(batch));
}

static int common_checks(int req, int64_t batch, int64_t seg_num, int64_t nnz) {
  if (!valid_req(req)) return fail(SG_ERR_INVALID, "req must be 0, 1 or 3, got %d", req);
  if (batch < 0 || seg_num < 0 || nnz < 0) return fail(SG_ERR_INVALID, "negative dimension");
  if (batch > 65535) return fail(SG_ERR_INVALID, "batch > 65535 not supported");
  if (seg_num >= (1ll << 31) - 1 || nnz >= (1ll << 31) - 1) return fail(SG_ERR_INVALID, "int32 index overflow");
  return SG_OK;
}

}  // namespace sg

using namespace sg;

SG_API int sg_seg_take_k_corr_hip(float* dst, const float* embed1, const float* embed2, const int32_t* neighbor_ids,
                                  const int32_t* neighbor_indptr, int64_t K, int64_t node_num,
                                  int64_t neighbor_node_num, int64_t nnz, int64_t feat_dim, int req, void* stream) {
  if (int rc = common_checks(req, K, node_num, nnz)) return rc;
  if (req == SG_REQ_NULL || K == 0 || nnz == 0) return SG_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (req == SG_REQ_WRITE && (node_num == 0 || feat_dim == 0))     // otherwise the main kernel zero-fills
    hipLaunchKernelGGL(zero_uncovered_kernel, dim3(static_cast<unsigned>((nnz + 255) / 256), static_cast<unsigned>(K)),
                       dim3(256), 0, st, dst, neighbor_indptr, static_cast<int>(node_num), static_cast<long long>(nnz));
  if (node_num > 0 && feat_dim > 0) {
    const bool v4 = (feat_dim % 4 == 0) && aligned(embed1, 16) && aligned(embed2, 16);
    const int64_t per_row = v4 ? feat_dim / 4 : feat_dim;
    int lpr = 1;
    while (lpr < kWave && lpr < per_row) lpr <<= 1;
    dim3 grid(static_cast<unsigned>((nnz + kCorrChunk - 1) / kCorrChunk), static_cast<unsigned>(K));
#define SG_CORR(V, L)                                                                                                 \
  hipLaunchKernelGGL((take_k_corr_kernel<V, L>), grid, dim3(kWave), 0, st, dst, embed1, embed2, neighbor_ids,         \
                     neighbor_indptr, static_cast<int>(node_num), static_cast<long long>(neighbor_node_num),          \
                     static_cast<long long>(nnz), static_cast<int>(feat_dim), lpr, req == SG_REQ_ADD)
#define SG_CORR_V(V)                                                                                                  \
  do {                                                                                                                \
    if (per_row > kWave) SG_CORR(V, 0);                                                                               \
    else switch (lpr) {                                                                                               \
      case 1: SG_CORR(V, 1); break;                                                                                   \
      case 2: SG_CORR(V, 2); break;                                                                                   \
      case 4: SG_CORR(V, 4); break;                                                                                   \
      case 8: SG_CORR(V, 8); break;                                                                                   \
      case 16: SG_CORR(V, 16); break;                                                                                 \
      case 32: SG_CORR(V, 32); break;                                                                                 \
      default: SG_CORR(V, 64); break;                                                                                 \
    }                                                                                                                 \
  } while (0)
    if (v4) SG_CORR_V(4); else SG_CORR_V(1);
#undef SG_CORR_V
#undef SG_CORR
  }
  return check_launch("seg_take_k_corr");
}

SG_API int sg_seg_sum_hip(float* dst, const float* data, const int32_t* indptr, int64_t batch, int64_t seg_num,
                          int64_t nnz, int req, void* stream) {
  if (int rc = common_checks(req, batch, seg_num, nnz)) return rc;
  if (req == SG_REQ_NULL || batch == 0 || seg_num == 0) return SG_OK;
  const int lanes = lanes_per_segment(nnz, seg_num);
  SG_BY_GROUP(lanes, hipLaunchKernelGGL(seg_sum_kernel<G>, group_grid(seg_num, G, batch), dim3(kBlock), 0,
                                        static_cast<hipStream_t>(stream), dst, data, indptr, static_cast<int>(seg_num),
                                        static_cast<long long>(nnz), req == SG_REQ_ADD));
  return check_launch("seg_sum");
}

SG_API int sg_seg_broadcast_hip(float* dst, const float* lhs, const float* rhs, const int32_t* indptr, int64_t batch,
                                int64_t seg_num, int64_t nnz, int op, int req, void* stream) {
  if (int rc = common_checks(req, batch, seg_num, nnz)) return rc;
  if (op < 0 || op > 2) return fail(SG_ERR_INVALID, "op must be 0 (add), 1 (mul) or 2 (to)");
  if (op != 2 && lhs == nullptr && nnz > 0) return fail(SG_ERR_INVALID, "lhs is null");
  if (req == SG_REQ_NULL || batch == 0 || nnz == 0) return SG_OK;
  hipLaunchKernelGGL(seg_broadcast_kernel,
                     dim3(static_cast<unsigned>((nnz + kCorrChunk - 1) / kCorrChunk), static_cast<unsigned>(batch)),
                     dim3(kWave), 0, static_cast<hipStream_t>(stream), dst, lhs, rhs, indptr, static_cast<int>(seg_num),
                     static_cast<long long>(nnz), op, req == SG_REQ_ADD);
  return check_launch("seg_broadcast");
}

SG_API int sg_seg_softmax_hip(float* dst, const float* data, const int32_t* indptr, int64_t batch, int64_t seg_num,
                              int64_t nnz, int req, void* stream) {
  if (int rc = common_checks(req, batch, seg_num, nnz)) return rc;
  if (req == SG_REQ_ADD) return fail(SG_ERR_UNSUPPORTED, "AddTo for seg_softmax is not supported (as the reference)");
  if (req == SG_REQ_NULL || batch == 0 || nnz == 0) return SG_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(zero_uncovered_kernel, dim3(static_cast<unsigned>((nnz + 255) / 256), static_cast<unsigned>(batch)),
                     dim3(256), 0, st, dst, indptr, static_cast<int>(seg_num), static_cast<long long>(nnz));
  if (seg_num > 0) {
    const int lanes = lanes_per_segment(nnz, seg_num);
    SG_BY_GROUP(lanes, hipLaunchKernelGGL(seg_softmax_kernel<G>, group_grid(seg_num, G, batch), dim3(kBlock), 0, st, dst,
                                          data, indptr, static_cast<int>(seg_num), static_cast<long long>(nnz)));
  }
  return check_launch("seg_softmax");
}

SG_API int sg_seg_softmax_bwd_hip(float* dst, const float* ograd, const float* val, const int32_t* indptr,
                                  int64_t batch, int64_t seg_num, int64_t nnz, int req, void* stream) {
  if (int rc = common_checks(req, batch, seg_num, nnz)) return rc;
  if (req == SG_REQ_NULL || batch == 0 || nnz == 0) return SG_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (req == SG_REQ_WRITE)
    hipLaunchKernelGGL(zero_uncovered_kernel, dim3(static_cast<unsigned>((nnz + 255) / 256), static_cast<unsigned>(batch)),
                       dim3(256), 0, st, dst, indptr, static_cast<int>(seg_num), static_cast<long long>(nnz));
  if (seg_num > 0) {
    const int lanes = lanes_per_segment(nnz, seg_num);
    SG_BY_GROUP(lanes, hipLaunchKernelGGL(seg_softmax_bwd_kernel<G>, group_grid(seg_num, G, batch), dim3(kBlock), 0, st,
                                          dst, ograd, val, indptr, static_cast<int>(seg_num),
                                          static_cast<long long>(nnz), req == SG_REQ_ADD));
  }
  return check_launch("seg_softmax_bwd");
}

SG_API size_t sg_seg_pool_workspace_bytes(int64_t batch, int64_t seg_num, int64_t nnz, int64_t feat_dim) {
  (void)seg_num;
  return gather_workspace_bytes(batch, nnz, feat_dim);
}

SG_API int sg_seg_pool_hip(float* dst, int32_t* pool_indices, const float* data, const int32_t* indices,
                           const int32_t* indptr, int64_t batch, int64_t seg_num, int64_t total_ind_num, int64_t nnz,
                           int64_t feat_dim, int pool_type, int req, void* workspace, size_t workspace_bytes,
                           void* stream) {
  if (int rc = common_checks(req, batch, seg_num, nnz)) return rc;
  if (req == SG_REQ_ADD) return fail(SG_ERR_UNSUPPORTED, "AddTo for seg_pool forward is not supported (as the reference)");
  if (pool_type < SG_POOL_SUM || pool_type > SG_POOL_MAX) return fail(SG_ERR_INVALID, "bad pool_type %d", pool_type);
  if (req == SG_REQ_NULL || batch == 0 || seg_num == 0 || feat_dim == 0) return SG_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (pool_type == SG_POOL_MAX) {
    if (!pool_indices) return fail(SG_ERR_INVALID, "pool_indices is required for max pooling");
    if (feat_dim % 4 == 0 && aligned(data, 16))
      hipLaunchKernelGGL(seg_pool_max_kernel<4>, seg_grid(seg_num, batch), dim3(kBlock), 0, st, dst, pool_indices, data,
                         indices, indptr, static_cast<int>(seg_num), static_cast<long long>(total_ind_num),
                         static_cast<int>(feat_dim));
    else
      hipLaunchKernelGGL(seg_pool_max_kernel<1>, seg_grid(seg_num, batch), dim3(kBlock), 0, st, dst, pool_indices, data,
                         indices, indptr, static_cast<int>(seg_num), static_cast<long long>(total_ind_num),
                         static_cast<int>(feat_dim));
    return check_launch("seg_pool_max");
  }
  return launch_gather(dst, 1, feat_dim, seg_num * feat_dim, data, 1, feat_dim, total_ind_num * feat_dim, nullptr, 0,
                       nullptr, indices, indptr, batch, seg_num, nnz, feat_dim, req, pool_type == SG_POOL_AVG, SG_ACT_NONE, 0.f,
                       workspace, workspace_bytes, st, batch * total_ind_num * feat_dim * static_cast<int64_t>(sizeof(float)));
}

SG_API size_t sg_seg_pool_bwd_workspace_bytes(int64_t batch, int64_t total_ind_num, int64_t nnz, int64_t feat_dim) {
  (void)total_ind_num;
  return gather_workspace_bytes(batch, nnz, feat_dim) + static_cast<size_t>(nnz) * sizeof(float) + 16;
}

SG_API int sg_seg_pool_bwd_hip(float* ddata, const float* ograd, const int32_t* pool_indices, const int32_t* indptr,
                               const int32_t* t_indptr, const int32_t* t_pos, const int32_t* t_seg, int64_t batch,
                               int64_t seg_num, int64_t total_ind_num, int64_t nnz, int64_t feat_dim, int pool_type,
                               int req, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = common_checks(req, batch, seg_num, nnz)) return rc;
  if (pool_type < SG_POOL_SUM || pool_type > SG_POOL_MAX) return fail(SG_ERR_INVALID, "bad pool_type %d", pool_type);
  if (req == SG_REQ_NULL || batch == 0 || total_ind_num == 0 || feat_dim == 0) return SG_OK;
  if (!t_indptr || (nnz > 0 && (!t_pos || !t_seg))) return fail(SG_ERR_INVALID, "transposed plan is required");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (pool_type == SG_POOL_MAX) {
    if (!pool_indices) return fail(SG_ERR_INVALID, "pool_indices is required for max pooling");
    if (feat_dim % 4 == 0 && aligned(ograd, 16) && aligned(pool_indices, 16))
      hipLaunchKernelGGL(seg_pool_max_bwd_kernel<4>, seg_grid(total_ind_num, batch), dim3(kBlock), 0, st, ddata, ograd,
                         pool_indices, t_indptr, t_pos, t_seg, static_cast<int>(seg_num),
                         static_cast<long long>(total_ind_num), static_cast<int>(feat_dim), req == SG_REQ_ADD);
    else
      hipLaunchKernelGGL(seg_pool_max_bwd_kernel<1>, seg_grid(total_ind_num, batch), dim3(kBlock), 0, st, ddata, ograd,
                         pool_indices, t_indptr, t_pos, t_seg, static_cast<int>(seg_num),
                         static_cast<long long>(total_ind_num), static_cast<int>(feat_dim), req == SG_REQ_ADD);
    return check_launch("seg_pool_max_bwd");
  }
  const size_t gbytes = gather_workspace_bytes(batch, nnz, feat_dim);
  if (!workspace || workspace_bytes < gbytes + static_cast<size_t>(nnz) * sizeof(float) + 16)
    return fail(SG_ERR_WORKSPACE, "workspace too small for seg_pool backward");
  const float* w = nullptr;
  if (pool_type == SG_POOL_AVG && nnz > 0) {
    char* p = static_cast<char*>(workspace) + gbytes;
    p = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p) + 15) & ~static_cast<uintptr_t>(15));
    float* wl = reinterpret_cast<float*>(p);
    hipLaunchKernelGGL(inv_len_kernel, dim3(static_cast<unsigned>((nnz + 255) / 256)), dim3(256), 0, st, wl, t_seg,
                       indptr, t_indptr, static_cast<long long>(total_ind_num));
    w = wl;
  }
  // same weights for every batch element (w_bs = 0)
  return launch_gather(ddata, 1, feat_dim, total_ind_num * feat_dim, ograd, 1, feat_dim, seg_num * feat_dim, w, 0,
                       nullptr, t_seg, t_indptr, batch, total_ind_num, nnz, feat_dim, req, 0, SG_ACT_NONE, 0.f, workspace, gbytes,
                       st, batch * seg_num * feat_dim * static_cast<int64_t>(sizeof(float)));
}

// ---- a gather issued as source-range phases (include/stargcn.h: sg_gather_phases; built by sg_gather_phases_build_hip) ----
SG_API int sg_seg_gather_sum_phased_hip(float* dst, int64_t dst_group, int64_t dst_ld, const float* src, int64_t src_group,
                                        int64_t src_ld, const float* weights, const sg_gather_phases* ph, int64_t seg_num,
                                        int64_t feat_dim, int req, int act, float slope, void* workspace,
                                        size_t workspace_bytes, void* stream, int64_t src_bytes) {
  if (!ph || ph->num_phases != 2 || !ph->indptr) return sg::fail(SG_ERR_INVALID, "gather phases missing");
  if (req != SG_REQ_NULL && req != SG_REQ_WRITE && req != SG_REQ_ADD) return sg::fail(SG_ERR_INVALID, "req must be 0, 1 or 3, got %d", req);
  if (req == SG_REQ_NULL) return SG_OK;
  int64_t off = 0;
  for (int p = 0; p < 2; ++p) {
    // every launch visits every segment once: phase 0 writes (or adds to) the destination, phase 1 adds and applies the
    // activation to the finished sum
    const int rc = sg::launch_gather(dst, dst_group, dst_ld, 0, src, src_group, src_ld, 0, weights, 0, ph->wpos + off,
                                     ph->idx + off, ph->indptr + p * (seg_num + 1), 1, seg_num, ph->nnz_p[p], feat_dim,
                                     p == 0 ? req : SG_REQ_ADD, 0, p == 1 ? act : SG_ACT_NONE, slope, workspace,
                                     workspace_bytes, static_cast<hipStream_t>(stream), src_bytes);
    if (rc != SG_OK) return rc;
    off += ph->nnz_p[p];
  }
  return SG_OK;
}
