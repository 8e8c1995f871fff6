// plan_build.hip -- DEVICE-side builders of the integer plans of the hot path (gfx950, wave64).
//
// Reference behaviour being replaced:
//   * `_backward_seg_take_k_corr_embed2` receives (ograd, embed1, neighbor_ids, neighbor_indptr) as DEVICE tensors and
//     sorts the edges by neighbour id on every call: iota + cub::DeviceRadixSort::SortPairs (stable) + FillSegStartIndex
//     + inclusive max-scan (seg_op.cu:882-926, GetSegId :91-110).  Round 1 of this library only had a HOST builder
//     (sg_build_transpose_cpu), i.e. a D2H copy + host sort + H2D copy per new graph.
//   * the per-level neighbour lists of the multi-link aggregator come from host code (graph_sampler.cpp:277-376,
//     layers.py:260-337) and are uploaded on every call (layers.py:366-377).
//
// Here the same integer structures are built where the data already lives:
//   exclusive_scan   3-kernel decoupled block scan of int32 (wave64 __shfl_up scan + LDS across the 4 waves)
//   radix_sort_pairs stable LSD radix sort of (uint32 key, int32 payload), 8-bit digits.  One 64-lane wavefront owns a
//                    tile of 2048 consecutive elements; the rank of an element among the equal digits before it is
//                    (running per-digit tile counter in LDS) + (lanes below with the same digit, found with 8
//                    __ballot rounds).  No floating point, no order-dependent atomics: the result is the unique stable
//                    order, bit-identical to the host counting sorts (sg_build_transpose_cpu, sg_multilink_fuse_cpu).
// All kernels take their sizes from host arguments except the number of COVERED edges E = indptr[seg_num], which is read
// on the device (positions >= E are padding, as in the reference's empty_as_zero convention, graph.py:221-222).
#include "common.hpp"

namespace sg {
namespace {

constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;   // 2048

// ---- block-level exclusive scan helpers ---------------------------------------------------------------------------
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int o = __shfl_up(v, d);
    if (lane >= d) v += o;
  }
  return v;
}

// exclusive scan of one value per thread over a 256-thread block; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* s_wave /* [4] */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int incl = wave_incl_scan(v, lane);
  if (lane == 63) s_wave[w] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kScanThreads / kWave; ++i) {
    const int c = s_wave[i];
    if (i < w) base += c;
    tot += c;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(kScanThreads) void scan_tile_sums_kernel(int32_t* __restrict__ sums,
                                                                      const int32_t* __restrict__ in, long long n) {
  __shared__ int s_wave[4];
  const long long base = static_cast<long long>(blockIdx.x) * kScanTile + static_cast<long long>(threadIdx.x) * kScanItems;
  int v = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i)
    if (base + i < n) v += in[base + i];
  int tot;
  (void)block_excl_scan(v, &tot, s_wave);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// single workgroup: exclusive scan of the tile sums in place (m is small: n / 2048)
__global__ __launch_bounds__(kScanThreads) void scan_sums_kernel(int32_t* __restrict__ sums, long long m) {
  __shared__ int s_wave[4];
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (long long b = 0; b < m; b += kScanTile) {
    const long long base = b + static_cast<long long>(threadIdx.x) * kScanItems;
    int x[kScanItems];
    int v = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
      x[i] = (base + i < m) ? sums[base + i] : 0;
      v += x[i];
    }
    int tot;
    int pre = block_excl_scan(v, &tot, s_wave) + s_carry;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
      if (base + i < m) sums[base + i] = pre;
      pre += x[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) s_carry += tot;
    __syncthreads();
  }
}

// out[i] = offset(tile) + exclusive prefix inside the tile; out[n] = grand total when `with_total`
__global__ __launch_bounds__(kScanThreads) void scan_tiles_kernel(int32_t* __restrict__ out, const int32_t* __restrict__ in,
                                                                  const int32_t* __restrict__ sums, long long n,
                                                                  int with_total) {
  __shared__ int s_wave[4];
  const long long base = static_cast<long long>(blockIdx.x) * kScanTile + static_cast<long long>(threadIdx.x) * kScanItems;
  int x[kScanItems];
  int v = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    x[i] = (base + i < n) ? in[base + i] : 0;
    v += x[i];
  }
  int tot;
  int pre = block_excl_scan(v, &tot, s_wave) + sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    if (base + i < n) out[base + i] = pre;
    pre += x[i];
  }
  if (with_total && base <= n - 1 && n - 1 < base + kScanItems) out[n] = pre;   // the thread owning element n-1
}

inline size_t al256(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }
inline long long scan_tiles(long long n) { return (n + kScanTile - 1) / kScanTile; }
inline size_t scan_ws_bytes(long long n) { return al256((scan_tiles(n) + 1) * sizeof(int32_t)); }

// exclusive scan of n int32; `in` and `out` may alias; out has n (+1 with_total) entries
int exclusive_scan(int32_t* out, const int32_t* in, long long n, bool with_total, void* ws, hipStream_t st) {
  if (n <= 0) {
    if (with_total && hipMemsetAsync(out, 0, sizeof(int32_t), st) != hipSuccess) return fail(SG_ERR_HIP, "memset");
    return SG_OK;
  }
  int32_t* sums = static_cast<int32_t*>(ws);
  const long long m = scan_tiles(n);
  hipLaunchKernelGGL(scan_tile_sums_kernel, dim3(static_cast<unsigned>(m)), dim3(kScanThreads), 0, st, sums, in, n);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kScanThreads), 0, st, sums, m);
  hipLaunchKernelGGL(scan_tiles_kernel, dim3(static_cast<unsigned>(m)), dim3(kScanThreads), 0, st, out, in, sums, n,
                     with_total ? 1 : 0);
  return check_launch("exclusive_scan");
}

// ---- stable LSD radix sort of (key, payload) pairs ----------------------------------------------------------------
constexpr int kRadixBits = 8;
constexpr int kRadix = 1 << kRadixBits;
constexpr int kSortRounds = 32;
constexpr int kSortTile = kWave * kSortRounds;   // 2048 elements per wavefront

__global__ __launch_bounds__(kWave) void radix_hist_kernel(int32_t* __restrict__ hist, const uint32_t* __restrict__ keys,
                                                           long long n, long long n_tiles, int shift) {
  __shared__ int s_cnt[kRadix];
  const int lane = threadIdx.x;
  for (int d = lane; d < kRadix; d += kWave) s_cnt[d] = 0;
  __syncthreads();
  const long long base = static_cast<long long>(blockIdx.x) * kSortTile;
#pragma unroll 4
  for (int r = 0; r < kSortRounds; ++r) {
    const long long p = base + static_cast<long long>(r) * kWave + lane;
    if (p < n) atomicAdd(&s_cnt[(keys[p] >> shift) & (kRadix - 1)], 1);   // integer LDS atomics: order-independent
  }
  __syncthreads();
  for (int d = lane; d < kRadix; d += kWave) hist[static_cast<long long>(d) * n_tiles + blockIdx.x] = s_cnt[d];
}

__global__ __launch_bounds__(kWave) void radix_scatter_kernel(uint32_t* __restrict__ keys_out, int32_t* __restrict__ vals_out,
                                                              const uint32_t* __restrict__ keys,
                                                              const int32_t* __restrict__ vals,  // null: payload = position
                                                              const int32_t* __restrict__ offs, long long n,
                                                              long long n_tiles, int shift) {
  __shared__ int s_cnt[kRadix];     // elements of each digit already placed from this tile
  __shared__ int s_off[kRadix];     // global start of this tile's run of each digit
  const int lane = threadIdx.x;
  for (int d = lane; d < kRadix; d += kWave) {
    s_cnt[d] = 0;
    s_off[d] = offs[static_cast<long long>(d) * n_tiles + blockIdx.x];
  }
  __syncthreads();
  const long long base = static_cast<long long>(blockIdx.x) * kSortTile;
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < kSortRounds; ++r) {
    const long long p = base + static_cast<long long>(r) * kWave + lane;
    const bool valid = p < n;
    const uint32_t key = valid ? keys[p] : 0u;
    const int32_t val = valid ? (vals ? vals[p] : static_cast<int32_t>(p)) : 0;
    const int digit = static_cast<int>((key >> shift) & (kRadix - 1));
    unsigned long long same = __ballot(valid);       // lanes holding the same digit as this lane
#pragma unroll
    for (int b = 0; b < kRadixBits; ++b) {
      const bool bit = (digit >> b) & 1;
      const unsigned long long m = __ballot(valid && bit);
      same &= bit ? m : ~m;
    }
    const int rank = __popcll(same & below);
    const int prior = valid ? s_cnt[digit] : 0;
    __syncthreads();                                 // every lane has read the counter before a leader bumps it
    if (valid) {
      const long long dst = static_cast<long long>(s_off[digit]) + prior + rank;
      keys_out[dst] = key;
      vals_out[dst] = val;
      if (rank == 0) s_cnt[digit] = prior + __popcll(same);
    }
    __syncthreads();
  }
}

inline long long sort_tiles(long long n) { return (n + kSortTile - 1) / kSortTile; }
inline int key_passes(long long max_key) {   // digits needed for keys in [0, max_key]
  int bits = 1;
  while (bits < 32 && (1ll << bits) <= max_key) ++bits;
  return (bits + kRadixBits - 1) / kRadixBits;
}
struct SortLayout {
  size_t keys_b, vals_b, hist, scan, total;
};
SortLayout sort_layout(long long n) {
  SortLayout L{};
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += al256(b); return o; };
  const long long nt = sort_tiles(n > 0 ? n : 1);
  L.keys_b = take(static_cast<size_t>(n > 0 ? n : 1) * 4);
  L.vals_b = take(static_cast<size_t>(n > 0 ? n : 1) * 4);
  L.hist = take(static_cast<size_t>(kRadix) * nt * 4);
  L.scan = take(scan_ws_bytes(static_cast<long long>(kRadix) * nt));
  L.total = off;
  return L;
}

// Sorts (keys, vals) stably by key.  keys_a/vals_a hold the input and are clobbered; on return *keys_res / *vals_res
// point at whichever of the a / b buffers holds the result.  vals_a may be null on input (payload = position); then
// `vals_a_storage` is the buffer to use for the a side.
int radix_sort_pairs(uint32_t* keys_a, int32_t* vals_a_storage, bool vals_is_iota, long long n, long long max_key,
                     char* ws, uint32_t** keys_res, int32_t** vals_res, hipStream_t st) {
  const SortLayout L = sort_layout(n);
  uint32_t* kb = reinterpret_cast<uint32_t*>(ws + L.keys_b);
  int32_t* vb = reinterpret_cast<int32_t*>(ws + L.vals_b);
  int32_t* hist = reinterpret_cast<int32_t*>(ws + L.hist);
  void* scan_ws = ws + L.scan;
  uint32_t* kin = keys_a;
  int32_t* vin = vals_a_storage;
  uint32_t* kout = kb;
  int32_t* vout = vb;
  *keys_res = kin;
  *vals_res = vin;
  if (n <= 0) return SG_OK;
  const long long nt = sort_tiles(n);
  const int passes = key_passes(max_key);
  for (int p = 0; p < passes; ++p) {
    const int shift = p * kRadixBits;
    hipLaunchKernelGGL(radix_hist_kernel, dim3(static_cast<unsigned>(nt)), dim3(kWave), 0, st, hist, kin, n, nt, shift);
    int rc = exclusive_scan(hist, hist, static_cast<long long>(kRadix) * nt, false, scan_ws, st);
    if (rc != SG_OK) return rc;
    hipLaunchKernelGGL(radix_scatter_kernel, dim3(static_cast<unsigned>(nt)), dim3(kWave), 0, st, kout, vout, kin,
                       (p == 0 && vals_is_iota) ? static_cast<const int32_t*>(nullptr) : vin, hist, n, nt, shift);
    uint32_t* tk = kin; kin = kout; kout = tk;
    int32_t* tv = vin; vin = vout; vout = tv;
  }
  *keys_res = kin;
  *vals_res = vin;
  return check_launch("radix_sort_pairs");
}

// ---- small element-wise kernels ------------------------------------------------------------------------------------
// key[j] = (j < E && 0 <= idx[j] < T) ? idx[j] : T      (padding and out-of-range entries sort behind every real key)
__global__ void transpose_keys_kernel(uint32_t* __restrict__ keys, const int32_t* __restrict__ idx,
                                      const int32_t* __restrict__ indptr, long long seg_num, long long T, long long n) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const long long E = indptr[seg_num];
  const long long v = idx[j];
  keys[j] = static_cast<uint32_t>((j < E && v >= 0 && v < T) ? v : T);
}

// seg[j] = the segment s with indptr[s] <= j < indptr[s+1]  (for j < indptr[seg_num]; else seg_num - 1, never read)
__global__ void edge_seg_kernel(int32_t* __restrict__ seg, const int32_t* __restrict__ indptr, long long seg_num, long long n) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= n) return;
  long long lo = 0, hi = seg_num;                     // upper_bound(indptr[1..seg_num], j)
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (indptr[mid + 1] <= j) lo = mid + 1; else hi = mid;
  }
  seg[j] = static_cast<int32_t>(lo < seg_num ? lo : seg_num - 1);
}

// out_indptr[k] = first sorted position whose key is >= k, for k in [0, T]  (keys are sorted; keys >= T are padding)
__global__ void bounds_from_sorted_kernel(int32_t* __restrict__ out_indptr, const uint32_t* __restrict__ keys, long long n,
                                          long long T) {
  const long long p = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p > n) return;
  const long long prev = (p == 0) ? -1 : static_cast<long long>(keys[p - 1] < T ? keys[p - 1] : T);
  const long long cur = (p < n) ? static_cast<long long>(keys[p] < T ? keys[p] : T) : T;
  for (long long k = prev + 1; k <= cur; ++k) out_indptr[k] = static_cast<int32_t>(p);
}

__global__ void gather_i32_kernel(int32_t* __restrict__ dst, const int32_t* __restrict__ src,
                                  const int32_t* __restrict__ pos, long long n) {
  const long long p = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p < n) dst[p] = src[pos[p]];
}

inline unsigned blocks(long long n) { return static_cast<unsigned>((n + 255) / 256); }

struct TransposeLayout {
  size_t keys, vals, seg, sort, total;
};
TransposeLayout transpose_layout(long long nnz, bool need_seg) {
  TransposeLayout L{};
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += al256(b); return o; };
  const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 1);
  L.keys = take(n * 4);
  L.vals = take(n * 4);
  L.seg = need_seg ? take(n * 4) : 0;
  L.sort = take(sort_layout(nnz).total);
  L.total = off + 256;
  return L;
}

}  // namespace
}  // namespace sg

using namespace sg;

// ------------------------------------------------------------------------------------------------------------------
// sg_build_transpose_hip: device twin of sg_build_transpose_cpu (same outputs, bit-exact for in-range indices)
// ------------------------------------------------------------------------------------------------------------------
SG_API size_t sg_build_transpose_workspace_bytes(int64_t seg_num, int64_t total_ind_num, int64_t nnz) {
  (void)seg_num; (void)total_ind_num;
  if (nnz < 0) return 0;
  return transpose_layout(nnz, true).total;
}

SG_API int sg_build_transpose_hip(int32_t* t_indptr, int32_t* t_pos, int32_t* t_seg, const int32_t* indices,
                                  const int32_t* indptr, int64_t seg_num, int64_t total_ind_num, int64_t nnz,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  if (seg_num < 0 || total_ind_num < 0 || nnz < 0) return fail(SG_ERR_INVALID, "negative dimension");
  if (!t_indptr || !indptr) return fail(SG_ERR_INVALID, "null pointer argument");
  if (nnz > 0 && (!t_pos || !indices)) return fail(SG_ERR_INVALID, "null pointer argument");
  if (total_ind_num >= (1ll << 31) - 1 || nnz >= (1ll << 31) - 1) return fail(SG_ERR_INVALID, "size overflows int32");
  const TransposeLayout L = transpose_layout(nnz, t_seg != nullptr);
  if (!workspace || workspace_bytes < L.total)
    return fail(SG_ERR_WORKSPACE, "transpose workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  uint32_t* keys = reinterpret_cast<uint32_t*>(base + L.keys);
  int32_t* vals = reinterpret_cast<int32_t*>(base + L.vals);
  const long long T = total_ind_num;
  uint32_t* ks = keys;
  int32_t* vs = vals;
  if (nnz > 0) {
    hipLaunchKernelGGL(transpose_keys_kernel, dim3(blocks(nnz)), dim3(256), 0, st, keys, indices, indptr,
                       static_cast<long long>(seg_num), T, static_cast<long long>(nnz));
    const int rc = radix_sort_pairs(keys, vals, true, nnz, T, base + L.sort, &ks, &vs, st);
    if (rc != SG_OK) return rc;
    if (hipMemcpyAsync(t_pos, vs, static_cast<size_t>(nnz) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess)
      return fail(SG_ERR_HIP, "memcpy");
    if (t_seg) {
      int32_t* seg = reinterpret_cast<int32_t*>(base + L.seg);
      if (seg_num > 0) {
        hipLaunchKernelGGL(edge_seg_kernel, dim3(blocks(nnz)), dim3(256), 0, st, seg, indptr,
                           static_cast<long long>(seg_num), static_cast<long long>(nnz));
        hipLaunchKernelGGL(gather_i32_kernel, dim3(blocks(nnz)), dim3(256), 0, st, t_seg, seg, vs,
                           static_cast<long long>(nnz));
      }
    }
  }
  hipLaunchKernelGGL(bounds_from_sorted_kernel, dim3(blocks(nnz + 1)), dim3(256), 0, st, t_indptr, ks,
                     static_cast<long long>(nnz), T);
  return check_launch("sg_build_transpose_hip");
}

// ------------------------------------------------------------------------------------------------------------------
// Reference-shaped backward operator: _backward_seg_take_k_corr_embed2(ograd = weights, embed1 = ograd rows,
// neighbor_ids, neighbor_indptr) with DEVICE index tensors only (seg_op.cc:718-752, GPU path seg_op.cu:882-926): the
// transposed plan is built in the caller's workspace (like the reference's per-call sort) and consumed by the gather.
// ------------------------------------------------------------------------------------------------------------------
SG_API size_t sg_seg_weighted_pool_bwd_data_dev_workspace_bytes(int64_t batch, int64_t seg_num, int64_t total_ind_num,
                                                                int64_t nnz, int64_t feat_dim) {
  if (nnz < 0 || total_ind_num < 0) return 0;
  const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 1);
  return al256((static_cast<size_t>(total_ind_num) + 1) * 4) + 2 * al256(n * 4) +
         al256(sg_build_transpose_workspace_bytes(seg_num, total_ind_num, nnz)) +
         al256(sg_seg_weighted_pool_bwd_data_workspace_bytes(batch, total_ind_num, nnz, feat_dim)) + 256;
}

SG_API int sg_seg_weighted_pool_bwd_data_dev_hip(float* ddata, const float* weights, const float* ograd,
                                                 const int32_t* indices, const int32_t* indptr, int64_t batch,
                                                 int64_t seg_num, int64_t total_ind_num, int64_t nnz, int64_t feat_dim,
                                                 int req, void* workspace, size_t workspace_bytes, void* stream) {
  if (!valid_req(req)) return fail(SG_ERR_INVALID, "req must be 0 (null), 1 (write) or 3 (add), got %d", req);
  if (req == SG_REQ_NULL) return SG_OK;
  if (batch < 0 || seg_num < 0 || total_ind_num < 0 || nnz < 0 || feat_dim < 0) return fail(SG_ERR_INVALID, "negative dimension");
  const size_t need = sg_seg_weighted_pool_bwd_data_dev_workspace_bytes(batch, seg_num, total_ind_num, nnz, feat_dim);
  if (!workspace || workspace_bytes < need)
    return fail(SG_ERR_WORKSPACE, "bwd-data workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 1);
  size_t off = 0;
  int32_t* t_indptr = reinterpret_cast<int32_t*>(base + off); off += al256((static_cast<size_t>(total_ind_num) + 1) * 4);
  int32_t* t_pos = reinterpret_cast<int32_t*>(base + off); off += al256(n * 4);
  int32_t* t_seg = reinterpret_cast<int32_t*>(base + off); off += al256(n * 4);
  const size_t tb = al256(sg_build_transpose_workspace_bytes(seg_num, total_ind_num, nnz));
  void* tws = base + off; off += tb;
  const size_t gb = sg_seg_weighted_pool_bwd_data_workspace_bytes(batch, total_ind_num, nnz, feat_dim);
  void* gws = base + off;
  int rc = sg_build_transpose_hip(t_indptr, t_pos, t_seg, indices, indptr, seg_num, total_ind_num, nnz, tws, tb, stream);
  if (rc != SG_OK) return rc;
  return sg_seg_weighted_pool_bwd_data_hip(ddata, weights, ograd, t_indptr, t_pos, t_seg, batch, seg_num, total_ind_num,
                                           nnz, feat_dim, req, gws, gb, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Multi-link plan builders on the device
// ------------------------------------------------------------------------------------------------------------------
namespace sg {
namespace {

struct LevelTable {
  const int32_t* ep[SG_MAX_LINKS];
  const int32_t* ip[SG_MAX_LINKS];
  const float* sp[SG_MAX_LINKS];
};

// len[i*R + r] = indptr_r[i+1] - indptr_r[i]
__global__ void level_lens_kernel(int32_t* __restrict__ lens, LevelTable t, long long n_dst, int R) {
  const long long s = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s >= n_dst * R) return;
  const long long i = s / R;
  const int r = static_cast<int>(s - i * R);
  lens[s] = t.ip[r][i + 1] - t.ip[r][i];
}

// slot w of the fused CSR <- entry of level r = seg % R at row i = seg / R
__global__ void level_fill_kernel(int32_t* __restrict__ c_idx, float* __restrict__ c_w, const int32_t* __restrict__ c_seg,
                                  const int32_t* __restrict__ c_indptr, LevelTable t, long long nnz, int R) {
  const long long w = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (w >= nnz) return;
  const int s = c_seg[w];
  const int i = s / R, r = s - i * R;
  const long long src = static_cast<long long>(t.ip[r][i]) + (w - c_indptr[s]);
  c_idx[w] = t.ep[r][src];
  c_w[w] = t.sp[r][src];
}

// key of edge j of a plain CSR for the (row, level)-major fused order; levels outside [0, R) are dropped (key = n_dst*R)
__global__ void csr_level_keys_kernel(uint32_t* __restrict__ keys, const int32_t* __restrict__ seg,
                                      const int32_t* __restrict__ level, long long nnz, long long n_dst, int R) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= nnz) return;
  const int l = level[j];
  keys[j] = static_cast<uint32_t>((l >= 0 && l < R) ? static_cast<long long>(seg[j]) * R + l : n_dst * R);
}

__global__ void csr_fill_kernel(int32_t* __restrict__ c_idx, float* __restrict__ c_w, int32_t* __restrict__ c_from,
                                const int32_t* __restrict__ order, const int32_t* __restrict__ end_points,
                                const float* __restrict__ support, long long nnz) {
  const long long w = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (w >= nnz) return;
  const int j = order[w];
  c_idx[w] = end_points[j];
  c_w[w] = support ? support[j] : 1.f;
  if (c_from) c_from[w] = j;
}

// c_q[w] = c_idx[w]*R + (c_seg[w] % R); transposed sort key = c_q (or n_src*R for out-of-range sources)
__global__ void cq_keys_kernel(int32_t* __restrict__ c_q, uint32_t* __restrict__ keys, const int32_t* __restrict__ c_idx,
                               const int32_t* __restrict__ c_seg, long long nnz, long long n_src, int R) {
  const long long w = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (w >= nnz) return;
  const long long n = c_idx[w];
  const int r = c_seg[w] % R;
  const bool ok = n >= 0 && n < n_src;
  const long long q = n * R + r;
  if (c_q) c_q[w] = static_cast<int32_t>(ok ? q : 0);
  keys[w] = static_cast<uint32_t>(ok ? q : n_src * R);
}

// transposed arrays from the sorted order: t_q = fused segment (dst*R + r), t_idx = destination node, t_w = weight
__global__ void transposed_fill_kernel(int32_t* __restrict__ t_idx, int32_t* __restrict__ t_q, float* __restrict__ t_w,
                                       int32_t* __restrict__ t_from, const int32_t* __restrict__ order,
                                       const int32_t* __restrict__ c_seg, const float* __restrict__ c_w,
                                       const int32_t* __restrict__ c_from, long long nnz, int R) {
  const long long p = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (p >= nnz) return;
  const int w = order[p];
  const int s = c_seg[w];
  if (t_q) t_q[p] = s;
  t_idx[p] = s / R;
  t_w[p] = c_w[w];
  if (t_from) t_from[p] = c_from ? c_from[w] : w;
}

__global__ void strided_copy_kernel(int32_t* __restrict__ dst, const int32_t* __restrict__ src, long long n, int stride) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i * stride];
}

struct FuseLayout {
  size_t lens, c_seg, keys, vals, seg_in, sort, scan, total;
};
FuseLayout fuse_layout(long long n_dst, long long n_src, long long nnz, int R) {
  (void)n_src;
  FuseLayout L{};
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += al256(b); return o; };
  const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 1);
  L.lens = take((static_cast<size_t>(n_dst) * R + 1) * 4);
  L.c_seg = take(n * 4);
  L.keys = take(n * 4);
  L.vals = take(n * 4);
  L.seg_in = take(n * 4);
  L.sort = take(sort_layout(nnz).total);
  L.scan = take(scan_ws_bytes(n_dst * R + 1));
  L.total = off + 256;
  return L;
}

// shared second half: given the fused CSR (c_indptr, c_idx, c_w) and the segment of every slot (c_seg), produce c_q,
// the transposed arrays and the un-split row pointers
int finish_fuse(int32_t* c_indptr, int32_t* c_idx, int32_t* c_q, float* c_w, int32_t* t_indptr, int32_t* t_idx,
                int32_t* t_q, float* t_w, int32_t* d_indptr, int32_t* s_indptr, int32_t* t_from, const int32_t* c_from,
                const int32_t* c_seg, long long n_dst, long long n_src, long long nnz, int R, char* base,
                const FuseLayout& L, hipStream_t st) {
  uint32_t* keys = reinterpret_cast<uint32_t*>(base + L.keys);
  int32_t* vals = reinterpret_cast<int32_t*>(base + L.vals);
  uint32_t* ks = keys;
  int32_t* vs = vals;
  const long long T = n_src * R;
  if (nnz > 0) {
    hipLaunchKernelGGL(cq_keys_kernel, dim3(blocks(nnz)), dim3(256), 0, st, c_q, keys, c_idx, c_seg, nnz, n_src, R);
    const int rc = radix_sort_pairs(keys, vals, true, nnz, T, base + L.sort, &ks, &vs, st);
    if (rc != SG_OK) return rc;
    hipLaunchKernelGGL(transposed_fill_kernel, dim3(blocks(nnz)), dim3(256), 0, st, t_idx, t_q, t_w, t_from, vs, c_seg,
                       c_w, c_from, nnz, R);
  }
  hipLaunchKernelGGL(bounds_from_sorted_kernel, dim3(blocks(nnz + 1)), dim3(256), 0, st, t_indptr, ks, nnz, T);
  if (d_indptr)
    hipLaunchKernelGGL(strided_copy_kernel, dim3(blocks(n_dst + 1)), dim3(256), 0, st, d_indptr, c_indptr, n_dst + 1, R);
  if (s_indptr)
    hipLaunchKernelGGL(strided_copy_kernel, dim3(blocks(n_src + 1)), dim3(256), 0, st, s_indptr, t_indptr, n_src + 1, R);
  return check_launch("multilink fuse");
}

int check_fuse_dims(int64_t num_links, int64_t n_dst, int64_t n_src, int64_t nnz) {
  if (num_links < 1 || num_links > SG_MAX_LINKS) return fail(SG_ERR_INVALID, "num_links outside [1, %d]", SG_MAX_LINKS);
  if (n_dst < 0 || n_src < 0 || nnz < 0) return fail(SG_ERR_INVALID, "negative dimension");
  if (n_dst * num_links >= (1ll << 31) - 1 || n_src * num_links >= (1ll << 31) - 1 || nnz >= (1ll << 31) - 1)
    return fail(SG_ERR_INVALID, "n*R / nnz overflows int32");
  return SG_OK;
}

}  // namespace
}  // namespace sg

SG_API size_t sg_multilink_fuse_workspace_bytes(int64_t num_links, int64_t n_dst, int64_t n_src, int64_t nnz) {
  if (num_links < 1 || n_dst < 0 || n_src < 0 || nnz < 0) return 0;
  return fuse_layout(n_dst, n_src, nnz, static_cast<int>(num_links)).total;
}

// device twin of sg_multilink_fuse_cpu: inputs are the R per-level lists (device pointers in HOST arrays); `nnz` = total
// number of covered edges = sum_r indptr_l[r][n_dst] (the caller knows it: it sized the output arrays with it).
SG_API int sg_multilink_fuse_hip(int32_t* c_indptr, int32_t* c_idx, int32_t* c_q, float* c_w, int32_t* t_indptr,
                                 int32_t* t_idx, int32_t* t_q, float* t_w, int32_t* d_indptr, int32_t* s_indptr,
                                 const int32_t* const* end_points_l, const int32_t* const* indptr_l,
                                 const float* const* support_l, int64_t num_links, int64_t n_dst, int64_t n_src,
                                 int64_t nnz, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_fuse_dims(num_links, n_dst, n_src, nnz);
  if (rc != SG_OK) return rc;
  if (!c_indptr || !t_indptr || !end_points_l || !indptr_l || !support_l) return fail(SG_ERR_INVALID, "null pointer argument");
  if (nnz > 0 && (!c_idx || !c_w || !t_idx || !t_w)) return fail(SG_ERR_INVALID, "null output array");
  const int R = static_cast<int>(num_links);
  const FuseLayout L = fuse_layout(n_dst, n_src, nnz, R);
  if (!workspace || workspace_bytes < L.total)
    return fail(SG_ERR_WORKSPACE, "fuse workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
  LevelTable t{};
  for (int r = 0; r < R; ++r) {
    if (!indptr_l[r] || (nnz > 0 && (!end_points_l[r] || !support_l[r]))) return fail(SG_ERR_INVALID, "level %d: null array", r);
    t.ep[r] = end_points_l[r]; t.ip[r] = indptr_l[r]; t.sp[r] = support_l[r];
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  int32_t* lens = reinterpret_cast<int32_t*>(base + L.lens);
  int32_t* c_seg = reinterpret_cast<int32_t*>(base + L.c_seg);
  const long long S = n_dst * R;
  if (S > 0) hipLaunchKernelGGL(level_lens_kernel, dim3(blocks(S)), dim3(256), 0, st, lens, t, static_cast<long long>(n_dst), R);
  rc = exclusive_scan(c_indptr, lens, S, true, base + L.scan, st);
  if (rc != SG_OK) return rc;
  if (nnz > 0) {
    hipLaunchKernelGGL(edge_seg_kernel, dim3(blocks(nnz)), dim3(256), 0, st, c_seg, c_indptr, S, static_cast<long long>(nnz));
    hipLaunchKernelGGL(level_fill_kernel, dim3(blocks(nnz)), dim3(256), 0, st, c_idx, c_w, c_seg, c_indptr, t,
                       static_cast<long long>(nnz), R);
  }
  return finish_fuse(c_indptr, c_idx, c_q, c_w, t_indptr, t_idx, t_q, t_w, d_indptr, s_indptr, nullptr, nullptr, c_seg, n_dst,
                     n_src, nnz, R, base, L, st);
}

// Same plan straight from a device-resident CSR of the (destination x source) graph: `level[j]` in [0, R) is the
// rating level of edge j, `support[j]` its weight (null: 1).  Replaces sample_neighbors + multi_link_split + fuse
// (graph.py:677-748, graph_sampler.cpp:277-376) for full-neighbourhood plans without touching the host.  Optional
// c_from / t_from: the CSR edge id stored in every slot of the two edge orders (what per-batch edge masking needs).
SG_API int sg_multilink_fuse_csr_hip(int32_t* c_indptr, int32_t* c_idx, int32_t* c_q, float* c_w, int32_t* t_indptr,
                                     int32_t* t_idx, int32_t* t_q, float* t_w, int32_t* d_indptr, int32_t* s_indptr,
                                     int32_t* c_from, int32_t* t_from, const int32_t* indptr, const int32_t* end_points,
                                     const int32_t* level, const float* support, int64_t num_links, int64_t n_dst,
                                     int64_t n_src, int64_t nnz, void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_fuse_dims(num_links, n_dst, n_src, nnz);
  if (rc != SG_OK) return rc;
  if (!c_indptr || !t_indptr || !indptr) return fail(SG_ERR_INVALID, "null pointer argument");
  if (nnz > 0 && (!c_idx || !c_w || !t_idx || !t_w || !end_points || !level)) return fail(SG_ERR_INVALID, "null array");
  if (t_from && !c_from) return fail(SG_ERR_INVALID, "t_from needs c_from");
  const int R = static_cast<int>(num_links);
  const FuseLayout L = fuse_layout(n_dst, n_src, nnz, R);
  if (!workspace || workspace_bytes < L.total)
    return fail(SG_ERR_WORKSPACE, "fuse workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  int32_t* c_seg = reinterpret_cast<int32_t*>(base + L.c_seg);
  int32_t* seg_in = reinterpret_cast<int32_t*>(base + L.seg_in);
  uint32_t* keys = reinterpret_cast<uint32_t*>(base + L.keys);
  int32_t* vals = reinterpret_cast<int32_t*>(base + L.vals);
  const long long S = n_dst * R;
  uint32_t* ks = keys;
  int32_t* vs = vals;
  if (nnz > 0) {
    hipLaunchKernelGGL(edge_seg_kernel, dim3(blocks(nnz)), dim3(256), 0, st, seg_in, indptr, static_cast<long long>(n_dst),
                       static_cast<long long>(nnz));
    hipLaunchKernelGGL(csr_level_keys_kernel, dim3(blocks(nnz)), dim3(256), 0, st, keys, seg_in, level,
                       static_cast<long long>(nnz), static_cast<long long>(n_dst), R);
    rc = radix_sort_pairs(keys, vals, true, nnz, S, base + L.sort, &ks, &vs, st);
    if (rc != SG_OK) return rc;
    hipLaunchKernelGGL(csr_fill_kernel, dim3(blocks(nnz)), dim3(256), 0, st, c_idx, c_w, c_from, vs, end_points, support,
                       static_cast<long long>(nnz));
  }
  hipLaunchKernelGGL(bounds_from_sorted_kernel, dim3(blocks(nnz + 1)), dim3(256), 0, st, c_indptr, ks,
                     static_cast<long long>(nnz), S);
  if (nnz > 0)   // the sort buffers are reused by finish_fuse: take the slot -> segment map first
    hipLaunchKernelGGL(edge_seg_kernel, dim3(blocks(nnz)), dim3(256), 0, st, c_seg, c_indptr, S, static_cast<long long>(nnz));
  return finish_fuse(c_indptr, c_idx, c_q, c_w, t_indptr, t_idx, t_q, t_w, d_indptr, s_indptr, t_from, c_from, c_seg, n_dst,
                     n_src, nnz, R, base, L, st);
}

// ------------------------------------------------------------------------------------------------------------------
// Small graph primitives on the device (twins of the host helpers in graph_host.cpp)
// ------------------------------------------------------------------------------------------------------------------
namespace sg {
namespace {
__global__ void support_kernel(float* __restrict__ support, const int32_t* __restrict__ row_deg,
                               const int32_t* __restrict__ col_deg, const int32_t* __restrict__ end_points,
                               const int32_t* __restrict__ edge_row, long long nnz, int symm) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= nnz) return;
  const int dr = row_deg[edge_row[j]];
  float v;
  if (symm) {
    const int dc = col_deg[end_points[j]];
    v = (dr == 0 || dc == 0) ? 0.f : sqrtf(1.0f / static_cast<float>(dr) / static_cast<float>(dc));
  } else {
    v = dr == 0 ? 0.f : 1.0f / static_cast<float>(dr);
  }
  support[j] = v;
}
__global__ void count_kernel(int32_t* __restrict__ cnt, const int32_t* __restrict__ idx, long long n, long long T) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const long long v = idx[j];
  if (v >= 0 && v < T) atomicAdd(&cnt[v], 1);   // integer atomics: the result does not depend on the order
}
__global__ void level_index_kernel(int32_t* __restrict__ level, const float* __restrict__ values,
                                   const float* __restrict__ multi_link, long long n, int R) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const float v = values[j];
  int l = -1;
  for (int r = 0; r < R; ++r)
    if (multi_link[r] == v) { l = r; break; }   // exact float equality, as graph_sampler.cpp:300-311
  level[j] = l;
}
}  // namespace
}  // namespace sg

// edge_row[j] = row of edge j (device twin of sg_gen_row_indices_cpu, reference py_ext.cpp gen_row_indices_by_indptr)
SG_API int sg_gen_row_indices_hip(int32_t* edge_row, const int32_t* ind_ptr, int64_t row_num, int64_t nnz, void* stream) {
  if (row_num < 0 || nnz < 0) return fail(SG_ERR_INVALID, "negative dimension");
  if (nnz == 0 || row_num == 0) return SG_OK;
  hipLaunchKernelGGL(edge_seg_kernel, dim3(blocks(nnz)), dim3(256), 0, static_cast<hipStream_t>(stream), edge_row, ind_ptr,
                     static_cast<long long>(row_num), static_cast<long long>(nnz));
  return check_launch("sg_gen_row_indices_hip");
}

// counts[v] = #{j : idx[j] == v}, v in [0, T)  (column degrees of a CSR; `counts` is overwritten)
SG_API int sg_count_indices_hip(int32_t* counts, const int32_t* idx, int64_t n, int64_t total, void* stream) {
  if (n < 0 || total < 0) return fail(SG_ERR_INVALID, "negative dimension");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (total > 0 && hipMemsetAsync(counts, 0, static_cast<size_t>(total) * 4, st) != hipSuccess) return fail(SG_ERR_HIP, "memset");
  if (n > 0 && total > 0)
    hipLaunchKernelGGL(count_kernel, dim3(blocks(n)), dim3(256), 0, st, counts, idx, static_cast<long long>(n),
                       static_cast<long long>(total));
  return check_launch("sg_count_indices_hip");
}

// device twin of sg_get_support_cpu (graph_sampler.cpp:393-420); `edge_row` from sg_gen_row_indices_hip
SG_API int sg_get_support_hip(float* support, const int32_t* row_degrees, const int32_t* col_degrees,
                              const int32_t* end_points, const int32_t* edge_row, int64_t nnz, int symm, void* stream) {
  if (nnz < 0) return fail(SG_ERR_INVALID, "negative nnz");
  if (nnz == 0) return SG_OK;
  if (!support || !row_degrees || !edge_row || (symm && (!col_degrees || !end_points))) return fail(SG_ERR_INVALID, "null pointer argument");
  hipLaunchKernelGGL(support_kernel, dim3(blocks(nnz)), dim3(256), 0, static_cast<hipStream_t>(stream), support, row_degrees,
                     col_degrees, end_points, edge_row, static_cast<long long>(nnz), symm);
  return check_launch("sg_get_support_hip");
}

// level[j] = index of values[j] in multi_link (exact float equality), -1 when it matches no level
SG_API int sg_level_index_hip(int32_t* level, const float* values, const float* multi_link, int64_t n, int64_t num_links,
                              void* stream) {
  if (n < 0 || num_links < 1 || num_links > SG_MAX_LINKS) return fail(SG_ERR_INVALID, "bad size");
  if (n == 0) return SG_OK;
  hipLaunchKernelGGL(level_index_kernel, dim3(blocks(n)), dim3(256), 0, static_cast<hipStream_t>(stream), level, values,
                     multi_link, static_cast<long long>(n), static_cast<int>(num_links));
  return check_launch("sg_level_index_hip");
}

// ------------------------------------------------------------------------------------------------------------------
// Per-iteration samplers on the device (second half).  Reference: DataIterator.rating_sampler draws
// `rng.choice(n, batch, replace=False)` and recon_nodes_sampler permutes the node ids with numpy's Mersenne Twister on the
// host (iterators.py:264-370).  A stateful sequential generator cannot be reproduced in parallel, so the device samplers
// use a COUNTER-BASED construction instead (documented deviation: same distribution -- a uniform sample without
// replacement -- not the same stream): a keyed bijection pi of [0, 2^b) (4-round Feistel network over the two halves of
// the b-bit index, b = smallest even width with 2^b >= n; round function = a 32-bit integer finaliser keyed by
// (seed, counter, round)), restricted to [0, n) by cycle walking.  pi(0), pi(1), ..., pi(k-1) are k DISTINCT uniform
// elements of [0, n); every thread computes its own element from (seed, counter, i) alone.
// ------------------------------------------------------------------------------------------------------------------
namespace sg {
namespace {
__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // integer finaliser (avalanche), the Feistel round function
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint64_t feistel(uint64_t x, int half_bits, uint32_t k0, uint32_t k1) {
  const uint32_t mask = (half_bits >= 32) ? 0xffffffffU : ((1U << half_bits) - 1U);
  uint32_t l = static_cast<uint32_t>(x >> half_bits) & mask, r = static_cast<uint32_t>(x) & mask;
#pragma unroll
  for (int round = 0; round < 4; ++round) {
    const uint32_t f = mix32(r ^ mix32(k0 + 0x9e3779b9U * (round + 1)) ^ (k1 * (2 * round + 1))) & mask;
    const uint32_t nl = r;
    r = l ^ f;
    l = nl;
  }
  return (static_cast<uint64_t>(l) << half_bits) | r;
}
// keys of a draw from (seed, counter [+ *dev_counter]): the optional device-resident counter lets a captured hipGraph draw
// a fresh batch at every replay (kernel arguments are frozen at capture time, device memory is not)
__host__ __device__ __forceinline__ void sampler_keys(uint64_t seed, uint64_t counter, uint32_t* k0, uint32_t* k1) {
  const uint64_t a = seed * 0x9e3779b97f4a7c15ULL + counter;
  const uint64_t b = (a ^ (a >> 31)) * 0xbf58476d1ce4e5b9ULL + (counter << 1 | 1);
  *k0 = static_cast<uint32_t>(a ^ (a >> 32));
  *k1 = static_cast<uint32_t>(b ^ (b >> 29)) | 1U;
}
__global__ void sample_kernel(int32_t* __restrict__ out, long long n, long long k, int half_bits, uint64_t seed,
                              uint64_t counter, const uint64_t* __restrict__ dev_counter) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= k) return;
  uint32_t k0, k1;
  sampler_keys(seed, counter + (dev_counter ? *dev_counter : 0ull), &k0, &k1);
  uint64_t x = static_cast<uint64_t>(i);
  do { x = feistel(x, half_bits, k0, k1); } while (x >= static_cast<uint64_t>(n));   // cycle walking: expected < 4 rounds
  out[i] = static_cast<int32_t>(x);
}
__global__ void iota_kernel(int32_t* __restrict__ a, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) a[i] = static_cast<int32_t>(i);
}
__global__ void fill_i32_kernel(int32_t* __restrict__ a, long long n, int32_t v) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}
__global__ void inverse_index_kernel(int32_t* __restrict__ inv, const int32_t* __restrict__ ids, long long n_ids, long long n_rows) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= n_ids) return;
  const long long r = ids[j];
  if (r >= 0 && r < n_rows) inv[r] = static_cast<int32_t>(j);
}
// noise[node] of the nodes picked for reconstruction: -1 (zero-mask) with probability p_zero, the node itself otherwise
__global__ void recon_noise_kernel(int32_t* __restrict__ noise, const int32_t* __restrict__ recon, long long k, long long n,
                                   float p_zero, uint64_t seed, uint64_t counter, const uint64_t* __restrict__ dev_counter) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= k) return;
  uint32_t k0, k1;
  sampler_keys(seed, counter + (dev_counter ? *dev_counter : 0ull), &k0, &k1);
  const long long node = recon[j];
  if (node < 0 || node >= n) return;
  const float u = (mix32(static_cast<uint32_t>(j) ^ mix32(k0 ^ 0x5bd1e995U) ^ (k1 * 0x27d4eb2fU)) >> 8) * (1.0f / 16777216.0f);
  noise[node] = (u < p_zero) ? -1 : static_cast<int32_t>(node);
}
// noise[cand[i]] = cand[i]: the nodes of the training graph keep their own embedding (iterators.py:341-342)
__global__ void scatter_identity_kernel(int32_t* __restrict__ noise, const int32_t* __restrict__ cand, long long m, long long n) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const long long v = cand[i];
  if (v >= 0 && v < n) noise[v] = static_cast<int32_t>(v);
}
// recon[j] = cand[recon[j]] (in place: positions in the candidate list -> node ids)
__global__ void map_through_kernel(int32_t* __restrict__ recon, const int32_t* __restrict__ cand, long long k, long long m) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= k) return;
  const long long p = recon[j];
  recon[j] = (p >= 0 && p < m) ? cand[p] : -1;
}
__global__ void add_u64_kernel(uint64_t* __restrict__ c, uint64_t v) {
  if (blockIdx.x == 0 && threadIdx.x == 0) *c += v;
}
}  // namespace
}  // namespace sg

SG_API int sg_sample_distinct_dev_hip(int32_t* out, int64_t n, int64_t k, uint64_t seed, uint64_t counter,
                                      const uint64_t* dev_counter, void* stream);
// out[i] = pi_{seed,counter}(i) for i < k: k distinct uniform elements of [0, n)  (k <= n < 2^31)
SG_API int sg_sample_distinct_hip(int32_t* out, int64_t n, int64_t k, uint64_t seed, uint64_t counter, void* stream) {
  return sg_sample_distinct_dev_hip(out, n, k, seed, counter, nullptr, stream);
}
// same with counter + *dev_counter (dev_counter: device pointer or null), for draws inside a captured graph
SG_API int sg_sample_distinct_dev_hip(int32_t* out, int64_t n, int64_t k, uint64_t seed, uint64_t counter,
                                      const uint64_t* dev_counter, void* stream) {
  if (n < 0 || k < 0 || k > n || n >= (1ll << 31)) return fail(SG_ERR_INVALID, "need 0 <= k <= n < 2^31 (k=%lld n=%lld)", (long long)k, (long long)n);
  if (k == 0) return SG_OK;
  if (!out) return fail(SG_ERR_INVALID, "null pointer argument");
  int bits = 2;
  while ((1ll << bits) < n) bits += 2;       // even width: two equal halves
  hipLaunchKernelGGL(sample_kernel, dim3(blocks(k)), dim3(256), 0, static_cast<hipStream_t>(stream), out,
                     static_cast<long long>(n), static_cast<long long>(k), bits / 2, seed, counter, dev_counter);
  return check_launch("sg_sample_distinct_hip");
}

// *counter += v on the stream (advances the device-resident draw counter once per captured iteration)
SG_API int sg_counter_add_hip(uint64_t* counter, uint64_t v, void* stream) {
  if (!counter) return fail(SG_ERR_INVALID, "null pointer argument");
  hipLaunchKernelGGL(add_u64_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), counter, v);
  return check_launch("sg_counter_add_hip");
}

// stable ascending sort of non-negative int32 keys (<= max_key) with an int32 payload (vals == null: payload = position)
SG_API size_t sg_sort_i32_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  return 2 * al256(static_cast<size_t>(n > 0 ? n : 1) * 4) + al256(sort_layout(n).total) + 256;
}
SG_API int sg_sort_i32_hip(int32_t* keys_out, int32_t* vals_out, const int32_t* keys, const int32_t* vals, int64_t n,
                           int64_t max_key, void* workspace, size_t workspace_bytes, void* stream) {
  if (n < 0 || max_key < 0 || n >= (1ll << 31) - 1) return fail(SG_ERR_INVALID, "bad size");
  if (n == 0) return SG_OK;
  if (!keys || (!keys_out && !vals_out)) return fail(SG_ERR_INVALID, "null pointer argument");
  if (!workspace || workspace_bytes < sg_sort_i32_workspace_bytes(n)) return fail(SG_ERR_WORKSPACE, "sort workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  const size_t nb = al256(static_cast<size_t>(n) * 4);
  uint32_t* ka = reinterpret_cast<uint32_t*>(base);
  int32_t* va = reinterpret_cast<int32_t*>(base + nb);
  if (hipMemcpyAsync(ka, keys, static_cast<size_t>(n) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(SG_ERR_HIP, "memcpy");
  if (vals && hipMemcpyAsync(va, vals, static_cast<size_t>(n) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(SG_ERR_HIP, "memcpy");
  uint32_t* ks; int32_t* vs;
  const int rc = radix_sort_pairs(ka, va, vals == nullptr, n, max_key, base + 2 * nb, &ks, &vs, st);
  if (rc != SG_OK) return rc;
  if (keys_out && hipMemcpyAsync(keys_out, ks, static_cast<size_t>(n) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(SG_ERR_HIP, "memcpy");
  if (vals_out && hipMemcpyAsync(vals_out, vs, static_cast<size_t>(n) * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(SG_ERR_HIP, "memcpy");
  return SG_OK;
}

// out_indptr[k] = first position of the SORTED keys whose key is >= k, k in [0, total]  (CSR row pointer of sorted keys)
SG_API int sg_bounds_from_sorted_hip(int32_t* out_indptr, const int32_t* sorted_keys, int64_t n, int64_t total, void* stream) {
  if (n < 0 || total < 0 || !out_indptr) return fail(SG_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(bounds_from_sorted_kernel, dim3(blocks(n + 1)), dim3(256), 0, static_cast<hipStream_t>(stream), out_indptr,
                     reinterpret_cast<const uint32_t*>(sorted_keys), static_cast<long long>(n), static_cast<long long>(total));
  return check_launch("sg_bounds_from_sorted_hip");
}

// dst[p] = src[pos[p]]
SG_API int sg_gather_i32_hip(int32_t* dst, const int32_t* src, const int32_t* pos, int64_t n, void* stream) {
  if (n < 0) return fail(SG_ERR_INVALID, "negative size");
  if (n == 0) return SG_OK;
  hipLaunchKernelGGL(gather_i32_kernel, dim3(blocks(n)), dim3(256), 0, static_cast<hipStream_t>(stream), dst, src, pos,
                     static_cast<long long>(n));
  return check_launch("sg_gather_i32_hip");
}

// inv[r] = j where ids[j] == r (ids distinct; -1 entries ignored), -1 for rows that are not listed
SG_API int sg_inverse_index_hip(int32_t* inv, const int32_t* ids, int64_t n_ids, int64_t n_rows, void* stream) {
  if (n_ids < 0 || n_rows < 0) return fail(SG_ERR_INVALID, "negative size");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n_rows > 0) hipLaunchKernelGGL(fill_i32_kernel, dim3(blocks(n_rows)), dim3(256), 0, st, inv, static_cast<long long>(n_rows), -1);
  if (n_ids > 0) hipLaunchKernelGGL(inverse_index_kernel, dim3(blocks(n_ids)), dim3(256), 0, st, inv, ids, static_cast<long long>(n_ids),
                                    static_cast<long long>(n_rows));
  return check_launch("sg_inverse_index_hip");
}

// Masked-reconstruction sampler of one node type (reference iterators.py:309-370): recon[0..k) = k distinct uniform node
// indices, noise[i] = i for every other node, noise[recon[j]] = -1 with probability p_zero else recon[j].
SG_API int sg_recon_mask_dev_hip(int32_t* noise, int32_t* recon, int64_t n, int64_t k, float p_zero, uint64_t seed,
                                 uint64_t counter, const uint64_t* dev_counter, void* stream);
SG_API int sg_recon_mask_hip(int32_t* noise, int32_t* recon, int64_t n, int64_t k, float p_zero, uint64_t seed, uint64_t counter,
                             void* stream) {
  return sg_recon_mask_dev_hip(noise, recon, n, k, p_zero, seed, counter, nullptr, stream);
}
SG_API int sg_recon_mask_dev_hip(int32_t* noise, int32_t* recon, int64_t n, int64_t k, float p_zero, uint64_t seed,
                                 uint64_t counter, const uint64_t* dev_counter, void* stream) {
  if (n < 0 || k < 0 || k > n || n >= (1ll << 31)) return fail(SG_ERR_INVALID, "need 0 <= k <= n < 2^31");
  if (n == 0) return SG_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(iota_kernel, dim3(blocks(n)), dim3(256), 0, st, noise, static_cast<long long>(n));
  int rc = sg_sample_distinct_dev_hip(recon, n, k, seed, counter, dev_counter, stream);
  if (rc != SG_OK) return rc;
  if (k > 0)
    hipLaunchKernelGGL(recon_noise_kernel, dim3(blocks(k)), dim3(256), 0, st, noise, recon, static_cast<long long>(k),
                       static_cast<long long>(n), p_zero, seed ^ 0xa5a5a5a5ULL, counter, dev_counter);
  return check_launch("sg_recon_mask_hip");
}

// The INDUCTIVE form of the same sampler (reference iterators.py:332-346 with _recon_train_candidates a strict subset of the
// graph's nodes): `cand` lists the m nodes that occur in the training graph (distinct ids in [0, n)).  noise starts as -1
// for EVERY node ("nodes unseen in the training graph are masked as -1"), the candidates get their own id, k distinct
// candidates are drawn for reconstruction (recon = their node ids) and take -1 with probability p_zero.
SG_API int sg_recon_mask_cand_dev_hip(int32_t* noise, int32_t* recon, int64_t n, const int32_t* cand, int64_t m, int64_t k,
                                      float p_zero, uint64_t seed, uint64_t counter, const uint64_t* dev_counter, void* stream) {
  if (n < 0 || m < 0 || k < 0 || k > m || m > n || n >= (1ll << 31)) return fail(SG_ERR_INVALID, "need 0 <= k <= m <= n < 2^31");
  if (n == 0) return SG_OK;
  if (!noise || (m > 0 && !cand) || (k > 0 && !recon)) return fail(SG_ERR_INVALID, "null pointer argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(fill_i32_kernel, dim3(blocks(n)), dim3(256), 0, st, noise, static_cast<long long>(n), -1);
  if (m > 0)
    hipLaunchKernelGGL(scatter_identity_kernel, dim3(blocks(m)), dim3(256), 0, st, noise, cand, static_cast<long long>(m),
                       static_cast<long long>(n));
  if (k > 0) {
    int rc = sg_sample_distinct_dev_hip(recon, m, k, seed, counter, dev_counter, stream);
    if (rc != SG_OK) return rc;
    hipLaunchKernelGGL(map_through_kernel, dim3(blocks(k)), dim3(256), 0, st, recon, cand, static_cast<long long>(k),
                       static_cast<long long>(m));
    hipLaunchKernelGGL(recon_noise_kernel, dim3(blocks(k)), dim3(256), 0, st, noise, recon, static_cast<long long>(k),
                       static_cast<long long>(n), p_zero, seed ^ 0xa5a5a5a5ULL, counter, dev_counter);
  }
  return check_launch("sg_recon_mask_cand_dev_hip");
}

// ------------------------------------------------------------------------------------------------------------------
// Source-partitioned view of a gather plan (sg_seg_gather_sum_parts_hip): key[j] = part(src[j]) * n_seg + seg(j), with
// part(v) = #{p in [1, parts): bounds[p] <= v}; a stable sort by this key orders the edges [part 0: segment by segment |
// part 1: ... ] and sg_bounds_from_sorted_hip over parts * n_seg keys gives the row pointer of the sub-segments.
// Positions >= indptr[n_seg] (padding) get the key parts * n_seg and sort behind everything.
// ------------------------------------------------------------------------------------------------------------------
namespace sg {
namespace {
__global__ void part_keys_kernel(int32_t* __restrict__ keys, const int32_t* __restrict__ src, const int32_t* __restrict__ indptr,
                                 const int32_t* __restrict__ bounds, int parts, long long n_seg, long long n) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (j >= indptr[n_seg]) { keys[j] = static_cast<int32_t>(parts * n_seg); return; }
  long long lo = 0, hi = n_seg;                       // segment of edge j: upper_bound(indptr[1..n_seg], j)
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (indptr[mid + 1] <= j) lo = mid + 1; else hi = mid;
  }
  const int v = src[j];
  int p = 0;
  for (int q = 1; q < parts; ++q) p += bounds[q] <= v ? 1 : 0;
  keys[j] = static_cast<int32_t>(p * n_seg + lo);
}
}  // namespace
}  // namespace sg

SG_API int sg_part_keys_hip(int32_t* keys, const int32_t* src_ids, const int32_t* indptr, const int32_t* bounds,
                            int64_t parts, int64_t n_seg, int64_t n, void* stream) {
  if (parts < 1 || parts > 64 || n_seg < 0 || n < 0 || (parts + 1) * n_seg >= (1ll << 31) - 1) return fail(SG_ERR_INVALID, "bad size");
  if (n == 0) return SG_OK;
  if (!keys || !src_ids || !indptr || (parts > 1 && !bounds)) return fail(SG_ERR_INVALID, "null pointer argument");
  hipLaunchKernelGGL(part_keys_kernel, dim3(blocks(n)), dim3(256), 0, static_cast<hipStream_t>(stream), keys, src_ids, indptr,
                     bounds, static_cast<int>(parts), static_cast<long long>(n_seg), static_cast<long long>(n));
  return check_launch("sg_part_keys_hip");
}

// ---- device twins of the last host-only plan primitives --------------------------------------------
namespace sg {
namespace {
// unique_inverse, first-occurrence order (graph_sampler.h:465-534), for ids in [0, max_id]:
//   first[v] = smallest position holding v (atomicMin on integers: order-independent, so deterministic);
//   a position is a HEAD when it is the first occurrence of its value; heads numbered by an exclusive scan over the
//   positions = rank of the value in first-occurrence order; slot[v] = that rank; inverse[i] = slot[ids[i]].
__global__ void ui_first_kernel(int32_t* __restrict__ first, int32_t* __restrict__ bad, const int32_t* __restrict__ ids,
                                long long n, long long max_id) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t v = ids[i];
  if (v < 0 || v > max_id) { atomicMax(bad, 1); return; }
  atomicMin(first + v, static_cast<int32_t>(i));
}
__global__ void ui_heads_kernel(int32_t* __restrict__ head, const int32_t* __restrict__ first, const int32_t* __restrict__ ids,
                                long long n, long long max_id) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t v = ids[i];
  head[i] = (v >= 0 && v <= max_id && first[v] == static_cast<int32_t>(i)) ? 1 : 0;
}
__global__ void ui_uniq_kernel(int32_t* __restrict__ uniq, int32_t* __restrict__ slot, const int32_t* __restrict__ rank,
                               const int32_t* __restrict__ first, const int32_t* __restrict__ ids, long long n,
                               long long max_id) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t v = ids[i];
  if (v >= 0 && v <= max_id && first[v] == static_cast<int32_t>(i)) {
    uniq[rank[i]] = v;
    slot[v] = rank[i];
  }
}
__global__ void ui_inverse_kernel(int32_t* __restrict__ inverse, int32_t* __restrict__ counts, const int32_t* __restrict__ slot,
                                  const int32_t* __restrict__ ids, long long n, long long max_id) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t v = ids[i];
  if (v < 0 || v > max_id) return;
  const int32_t s = slot[v];
  if (inverse) inverse[i] = s;
  if (counts) atomicAdd(counts + s, 1);       // integer counts: order-independent
}

// random_sample_fix_neighbor (graph_sampler.cpp:742-779): the arithmetic of sg_sample_fix_neighbor_cpu, one thread per
// selected row -- row i's draw depends on (seed, i) only, positions ascending -- so host and device agree bit for bit
__device__ __forceinline__ uint64_t splitmix64_dev(uint64_t& s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void sfn_lens_kernel(int32_t* __restrict__ lens, const int32_t* __restrict__ src_ind_ptr,
                                const int32_t* __restrict__ sel, long long sel_num, long long k) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= sel_num) return;
  const int32_t len = src_ind_ptr[sel[i] + 1] - src_ind_ptr[sel[i]];
  lens[i] = (k < 0) ? len : static_cast<int32_t>(min(static_cast<long long>(len), k));
}
__global__ void sfn_fill_kernel(int32_t* __restrict__ sampled, const int32_t* __restrict__ dst_ind_ptr,
                                const int32_t* __restrict__ src_ind_ptr, const int32_t* __restrict__ sel, long long sel_num,
                                uint64_t seed) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= sel_num) return;
  const int32_t b = src_ind_ptr[sel[i]], e = src_ind_ptr[sel[i] + 1];
  const int32_t k = dst_ind_ptr[i + 1] - dst_ind_ptr[i], len = e - b;
  int32_t* out = sampled + dst_ind_ptr[i];
  if (k == len) {
    for (int32_t j = 0; j < len; ++j) out[j] = b + j;
    return;
  }
  uint64_t st = seed ^ (0xD1B54A32D192ED03ull * static_cast<uint64_t>(i + 1));
  int32_t n = 0;
  for (int32_t j = len - k; j < len; ++j) {      // Floyd: k distinct values of [0, len), kept sorted
    const int32_t t = static_cast<int32_t>(splitmix64_dev(st) % static_cast<uint64_t>(j + 1));
    int32_t v = b + t;
    int32_t lo = 0, hi = n;                      // lower_bound(out, out + n, v)
    while (lo < hi) {
      const int32_t mid = (lo + hi) >> 1;
      if (out[mid] < v) lo = mid + 1; else hi = mid;
    }
    if (lo != n && out[lo] == v) {               // already chosen -> take j itself (larger than everything chosen so far)
      v = b + j;
      lo = n;
    }
    for (int32_t q = n; q > lo; --q) out[q] = out[q - 1];
    out[lo] = v;
    ++n;
  }
}
}  // namespace
}  // namespace sg

SG_API size_t sg_unique_inverse_workspace_bytes(int64_t n, int64_t max_id) {
  if (n < 0 || max_id < -1) return 0;
  // first[max_id+1] | slot[max_id+1] | head / rank [n+1] | bad flag | scan
  return 2 * sg::al256(static_cast<size_t>(max_id + 1) * 4 + 4) + sg::al256(static_cast<size_t>(n + 1) * 4) + 256 +
         sg::scan_ws_bytes(n) + 256;
}
// uniq (n entries, *n_uniq_dev used), inverse (n) and counts (n, may be NULL) as sg_unique_inverse_cpu; the number of
// unique ids is written to DEVICE memory (no host synchronisation); *bad_dev (may be NULL) is set to 1 when an id lies
// outside [0, max_id] (such positions get no inverse)
SG_API int sg_unique_inverse_hip(int32_t* uniq, int32_t* inverse, int32_t* counts, int32_t* n_uniq_dev, int32_t* bad_dev,
                                 const int32_t* ids, int64_t n, int64_t max_id, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  using namespace sg;
  if (n < 0 || max_id < -1 || n >= (1ll << 31) - 1) return fail(SG_ERR_INVALID, "bad unique_inverse arguments");
  if (!n_uniq_dev || (n > 0 && (!uniq || !ids))) return fail(SG_ERR_INVALID, "null pointer argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (n == 0) {
    if (hipMemsetAsync(n_uniq_dev, 0, 4, st) != hipSuccess) return fail(SG_ERR_HIP, "memset");
    return SG_OK;
  }
  if (!workspace || workspace_bytes < sg_unique_inverse_workspace_bytes(n, max_id))
    return fail(SG_ERR_WORKSPACE, "unique_inverse workspace too small");
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  const size_t tb = al256(static_cast<size_t>(max_id + 1) * 4 + 4);
  int32_t* first = reinterpret_cast<int32_t*>(base);
  int32_t* slot = reinterpret_cast<int32_t*>(base + tb);
  int32_t* rank = reinterpret_cast<int32_t*>(base + 2 * tb);
  int32_t* bad = reinterpret_cast<int32_t*>(base + 2 * tb + al256(static_cast<size_t>(n + 1) * 4));
  void* scan_ws = reinterpret_cast<char*>(bad) + 256;
  if (hipMemsetAsync(first, 0x7f, tb, st) != hipSuccess || hipMemsetAsync(bad, 0, 4, st) != hipSuccess)
    return fail(SG_ERR_HIP, "memset");
  if (counts && hipMemsetAsync(counts, 0, static_cast<size_t>(n) * 4, st) != hipSuccess) return fail(SG_ERR_HIP, "memset");
  hipLaunchKernelGGL(ui_first_kernel, dim3(blocks(n)), dim3(256), 0, st, first, bad, ids, static_cast<long long>(n),
                     static_cast<long long>(max_id));
  hipLaunchKernelGGL(ui_heads_kernel, dim3(blocks(n)), dim3(256), 0, st, rank, first, ids, static_cast<long long>(n),
                     static_cast<long long>(max_id));
  int rc = exclusive_scan(rank, rank, n, true, scan_ws, st);
  if (rc != SG_OK) return rc;
  hipLaunchKernelGGL(ui_uniq_kernel, dim3(blocks(n)), dim3(256), 0, st, uniq, slot, rank, first, ids, static_cast<long long>(n),
                     static_cast<long long>(max_id));
  if (inverse || counts)
    hipLaunchKernelGGL(ui_inverse_kernel, dim3(blocks(n)), dim3(256), 0, st, inverse, counts, slot, ids,
                       static_cast<long long>(n), static_cast<long long>(max_id));
  if (hipMemcpyAsync(n_uniq_dev, rank + n, 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(SG_ERR_HIP, "memcpy");
  if (bad_dev && hipMemcpyAsync(bad_dev, bad, 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail(SG_ERR_HIP, "memcpy");
  return check_launch("sg_unique_inverse_hip");
}

SG_API size_t sg_sample_fix_neighbor_workspace_bytes(int64_t sel_num) {
  return sel_num < 0 ? 0 : sg::scan_ws_bytes(sel_num) + 256;
}
// device twin of sg_sample_fix_neighbor_cpu, bit-identical for the same seed.  sampled == NULL: only dst_ind_ptr
// (sel_num + 1, device) is filled -- read its last entry to size `sampled`, then call again with both.
SG_API int sg_sample_fix_neighbor_hip(int32_t* sampled, int32_t* dst_ind_ptr, const int32_t* src_ind_ptr,
                                      const int32_t* sel_indices, int64_t sel_num, int64_t neighbor_num, uint64_t seed,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  using namespace sg;
  if (sel_num < 0 || sel_num >= (1ll << 31) - 1) return fail(SG_ERR_INVALID, "bad dimension");
  if (!dst_ind_ptr || !src_ind_ptr || (sel_num > 0 && !sel_indices)) return fail(SG_ERR_INVALID, "null argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!sampled) {
    if (sel_num > 0 && (!workspace || workspace_bytes < sg_sample_fix_neighbor_workspace_bytes(sel_num)))
      return fail(SG_ERR_WORKSPACE, "sample_fix_neighbor workspace too small");
    if (sel_num > 0)
      hipLaunchKernelGGL(sfn_lens_kernel, dim3(blocks(sel_num)), dim3(256), 0, st, dst_ind_ptr, src_ind_ptr, sel_indices,
                         static_cast<long long>(sel_num), static_cast<long long>(neighbor_num));
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
    return exclusive_scan(dst_ind_ptr, dst_ind_ptr, sel_num, true, base, st);
  }
  if (sel_num > 0)
    hipLaunchKernelGGL(sfn_fill_kernel, dim3(blocks(sel_num)), dim3(256), 0, st, sampled, dst_ind_ptr, src_ind_ptr,
                       sel_indices, static_cast<long long>(sel_num), seed);
  return check_launch("sg_sample_fix_neighbor_hip");
}

// ---- source-range phases of a gather view --------------------------------------------------------------
// A gather whose source matrix is a few times the aggregate L2 runs faster as TWO launches that each touch one half of the
// source rows (the second accumulates): per XCD and column slice the working set halves and the L2 hit rate rises
// (10 M edges over 68 MB of user rows: 1.02 -> 0.80 ms, over 104 MB of grouped rows 0.83 -> 0.69 ms).  The plan of such a
// pair is a STABLE partition of the view's edges by phase(idx) = idx >= split: inside a phase the edges keep their
// segment-major order, so each phase is a CSR over the same segments.
namespace sg {
namespace {
__global__ void phase_flag_kernel(int32_t* __restrict__ flag, const int32_t* __restrict__ idx, long long n, int32_t split) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j < n) flag[j] = idx[j] < split ? 1 : 0;
}
// before[j] = edges of phase 0 among the first j (n + 1 entries).  Edge j goes to before[j] (phase 0) or
// before[n] + j - before[j] (phase 1); wpos = its position in the view's weight array
__global__ void phase_scatter_kernel(int32_t* __restrict__ idx_p, int32_t* __restrict__ wpos_p, const int32_t* __restrict__ idx,
                                     const int32_t* __restrict__ before, long long n, int32_t split) {
  const long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int32_t q = idx[j], b = before[j];
  const long long pos = q < split ? b : static_cast<long long>(before[n]) + (j - b);
  idx_p[pos] = q;
  wpos_p[pos] = static_cast<int32_t>(j);
}
__global__ void phase_indptr_kernel(int32_t* __restrict__ indptr_p, int32_t* __restrict__ nnz_p,
                                    const int32_t* __restrict__ indptr, const int32_t* __restrict__ before, long long seg_num,
                                    long long n) {
  const long long s = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (s <= seg_num) {
    const int32_t p = indptr[s], b = before[p];
    indptr_p[s] = b;
    indptr_p[seg_num + 1 + s] = p - b;
  }
  if (s == 0) { nnz_p[0] = before[n]; nnz_p[1] = static_cast<int32_t>(n - before[n]); }
}
}  // namespace
}  // namespace sg

SG_API size_t sg_gather_phases_workspace_bytes(int64_t nnz) {
  return nnz < 0 ? 0 : sg::al256((static_cast<size_t>(nnz) + 1) * sizeof(int32_t)) + sg::scan_ws_bytes(nnz + 1) + 256;
}
// two source-range phases of the view (indices, indptr) whose indices address n_rows source rows: phase 0 = rows
// [0, ceil(n_rows / 2)), phase 1 = the rest.  idx_p / wpos_p: (nnz) each, phase 0's edges first; indptr_p: (2, seg_num + 1);
// nnz_p: (2) on the device -- the caller reads it back once at plan time.
SG_API int sg_gather_phases_build_hip(int32_t* idx_p, int32_t* wpos_p, int32_t* indptr_p, int32_t* nnz_p,
                                      const int32_t* indices, const int32_t* indptr, int64_t seg_num, int64_t nnz,
                                      int64_t n_rows, void* workspace, size_t workspace_bytes, void* stream) {
  using namespace sg;
  if (seg_num < 0 || nnz < 0 || n_rows < 0 || nnz >= (1ll << 31) - 1 || seg_num >= (1ll << 31) - 1 || n_rows >= (1ll << 31))
    return fail(SG_ERR_INVALID, "bad dimension");
  if (!indptr_p || !nnz_p || !indptr || (nnz > 0 && (!idx_p || !wpos_p || !indices))) return fail(SG_ERR_INVALID, "null argument");
  if (!workspace || workspace_bytes < sg_gather_phases_workspace_bytes(nnz)) return fail(SG_ERR_WORKSPACE, "gather phases workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  int32_t* before = reinterpret_cast<int32_t*>(base);
  void* scan_ws = base + al256((static_cast<size_t>(nnz) + 1) * sizeof(int32_t));
  const int32_t split = static_cast<int32_t>((n_rows + 1) / 2);
  if (nnz > 0)
    hipLaunchKernelGGL(phase_flag_kernel, dim3(blocks(nnz)), dim3(256), 0, st, before, indices, static_cast<long long>(nnz), split);
  int rc = exclusive_scan(before, before, nnz, true, scan_ws, st);
  if (rc != SG_OK) return rc;
  if (nnz > 0)
    hipLaunchKernelGGL(phase_scatter_kernel, dim3(blocks(nnz)), dim3(256), 0, st, idx_p, wpos_p, indices, before,
                       static_cast<long long>(nnz), split);
  hipLaunchKernelGGL(phase_indptr_kernel, dim3(blocks(seg_num + 1)), dim3(256), 0, st, indptr_p, nnz_p, indptr, before,
                     static_cast<long long>(seg_num), static_cast<long long>(nnz));
  return check_launch("sg_gather_phases_build_hip");
}
