// seg_gather.hip -- the HBM-bound hot kernel of the STAR-GCN multi-link graph convolution on gfx950.
//
//   dst[row(s), :] (+)= sum_{j = indptr[s]}^{indptr[s+1]-1}  w[j] * src[row(idx[j]), :]
//
// This one kernel serves (a) `seg_weighted_pool` forward (reference seg_op.cc:180-207, seg_op.cu:682-722),
// (b) its gradient w.r.t. data through a cached transposed plan (reference seg_op.cc:209-240,
// seg_op.cu:747-790/882-926 re-sorts on every call), (c) `seg_pool` sum/avg, and (d) the fused multi-link
// aggregation where the R per-rating-level aggregates of a node are written side by side into one row of
// the matrix that feeds the MFMA contraction (reference aggregators.py:133-159 does R separate op calls).
//
// CDNA4 mapping (not the reference's 32-thread-block-per-(row,128 channels) design):
//   * work unit = a CHUNK of 256 consecutive edges handled by ONE 64-lane wavefront (one wave per
//     workgroup, so the only synchronisation is wave-local).  Edge-balanced: a 35k-edge hub row is split
//     over ~140 waves, a run of 14-edge rows shares one wave.  No atomics, deterministic.
//   * the chunk's (source-row offset, weight) pairs are computed/loaded once, coalesced, into LDS; the CSR
//     row-pointer tile of the segments that start in the chunk is staged in LDS 64 pointers at a time.
//   * each gathered neighbour row is read as ONE coalesced burst: lanes hold VEC consecutive floats
//     (float4 -> 1 KiB per wave instruction at C = 256).  For narrow rows (C*4 < 1 KiB) the wave splits
//     into 64/LPR edge groups that gather different edges concurrently and combine with __shfl_xor.
//   * segments that straddle a chunk boundary write fp32 partial rows to a workspace; a tiny second
//     kernel adds the partials of each such segment in chunk order.
//   * XCD column slicing (cache-resident sources): XCD i only touches the 256-byte column slice i % 4 of every source
//     row, so its private 4 MB L2 faces a 4x smaller working set (L2 hit 7 % -> 25 % on a 109 MB source).
#include <type_traits>
#include "common.hpp"

#include <atomic>
#include <mutex>
#include <vector>

namespace sg {

#ifndef SG_GATHER_DEFAULT_SLICES
#define SG_GATHER_DEFAULT_SLICES 4
#endif
#ifndef SG_GATHER_CHUNK
#define SG_GATHER_CHUNK 256
#endif
constexpr int kChunk = SG_GATHER_CHUNK;  // edges per wavefront
constexpr int kPtrTile = 64; // CSR row pointers staged in LDS per refill

struct GatherArgs {
  float* dst;
  const float* src;
  const float* w;        // may be null (all ones)
  const int32_t* wpos;   // may be null; else weight of edge j is w[wpos[j]]
  const int32_t* idx;
  const int32_t* indptr; // seg_num + 1
  float* ws_head;        // [batch][n_chunks][C] partial of the segment that started in an earlier chunk
  float* ws_tail;        // [batch][n_chunks][C] partial of the segment that continues into the next chunk
  int32_t* ws_tailseg;   // [batch][n_chunks]    that segment's id, or -1
  long long dst_ld, src_ld, dst_bs, src_bs, w_bs;
  int32_t dst_group, src_group;
  uint32_t src_magic;    // q / src_group == (uint64(q) * src_magic) >> (31 + src_shift)   for q < 2^31
  int32_t src_shift;
  int32_t seg_num, C, n_chunks, lpr;
  int32_t n_slices, Cs;  // column slicing across XCDs: slice i covers channels [i*Cs, (i+1)*Cs); 1 / C = off
  int32_t xcd_ranges;    // 1: XCD x takes the CONTIGUOUS chunk range [x * span, (x + 1) * span) instead of every 8th chunk --
                         // for source-partitioned plans (edges ordered by source-row range first): each private L2 then
                         // only ever sees one range of the gathered matrix
  int32_t req, mean;
  int32_t act;
  float slope;
  // DOT mode (rating head, sg_pair_l2_hip): the weight of edge j is NOT read but formed from the row it gathers:
  //   r_j = < src[idx[j]], dot_other[seg(j) % dot_mod] > - w[j]        (w holds the targets y in edge order)
  //   weight_j = dot_scale * (*dot_scale_dev) * r_j,   loss_part[chunk] = sum_j r_j^2
  const float* dot_other;
  const float* dot_scale_dev;   // may be null (= 1)
  float* loss_part;             // [batch][n_chunks], may be null
  float dot_scale;
  int32_t dot_mod;              // rows of dot_other (a source-partitioned plan has parts * dot_mod sub-segments)
};

__device__ __forceinline__ float gather_act(float v, int act, float slope) {
  switch (act) {
    case SG_ACT_LEAKY: return v > 0.f ? v : slope * v;
    case SG_ACT_RELU: return v > 0.f ? v : 0.f;
    case SG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case SG_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

template <int V> struct VecT;
template <> struct VecT<1> { using T = float; };
template <> struct VecT<2> { using T = float2; };
template <> struct VecT<4> { using T = float4; };

template <int V> __device__ __forceinline__ void ld_vec(float (&r)[V], const float* p) {
  typename VecT<V>::T t = *reinterpret_cast<const typename VecT<V>::T*>(p);
  const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
  for (int v = 0; v < V; ++v) r[v] = f[v];
}
// gathered source rows.  PMC (tools/pmc_gather.sh): the per-CU L1 waits for L2 returns 47-65 % of its busy cycles
// (TCP_PENDING_STALL) at 0.18-0.25 line requests per clock, request latency 119 (L2-resident) .. 414 cycles (bench shape),
// no TLB misses: the kernel is bound by the L1's outstanding-miss capacity over the L2 / Infinity-Cache latency.
// SG_GATHER_NT=1 issues the row loads non-temporal to keep them out of the L1 -- measured 1.6-1.7 ms instead of
// 0.86-0.92 ms per launch: the hint also keeps the rows out of L2, and the L2 hits are what the kernel lives on.  Off.
#ifndef SG_GATHER_NT
#define SG_GATHER_NT 0
#endif
template <int V> __device__ __forceinline__ void ld_row(float (&r)[V], const float* p) {
#if SG_GATHER_NT
  typedef float vec_t __attribute__((ext_vector_type(V)));
  const vec_t t = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(p));
#pragma unroll
  for (int v = 0; v < V; ++v) r[v] = t[v];
#else
  ld_vec<V>(r, p);
#endif
}
template <int V> __device__ __forceinline__ void st_vec(float* p, const float (&r)[V]) {
  typename VecT<V>::T t;
  float* f = reinterpret_cast<float*>(&t);
#pragma unroll
  for (int v = 0; v < V; ++v) f[v] = r[v];
  *reinterpret_cast<typename VecT<V>::T*>(p) = t;
}

// first s in [0, n] with a[s] >= t (a is non-decreasing, n+1 entries, a[n] >= t guaranteed by callers).
// 64-ary search: every step all lanes probe, __ballot narrows the range 64x (wave-uniform result).
__device__ __forceinline__ int wave_lower_bound(const int32_t* __restrict__ a, int n, int t, int lane) {
  int lo = 0, hi = n + 1;
  while (lo < hi) {
    const int len = hi - lo;
    const int step = (len + 63) >> 6;
    const int p = lo + lane * step;
    const bool pred = (p < hi) && (a[p] >= t);
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) {
      const int nprobe = (len + step - 1) / step;
      lo = lo + (nprobe - 1) * step + 1;
    } else {
      const int f = __ffsll(static_cast<long long>(m)) - 1;
      const int nhi = lo + f * step;
      lo = (f == 0) ? nhi : (lo + (f - 1) * step + 1);
      hi = nhi;
    }
  }
  return lo;
}

template <bool GROUPED>
__device__ __forceinline__ long long src_row_off(const GatherArgs& a, int q) {
  if (!GROUPED) return static_cast<long long>(q) * a.src_ld;
  const unsigned hi = static_cast<unsigned>((static_cast<unsigned long long>(static_cast<unsigned>(q)) * a.src_magic) >>
                                            (31 + a.src_shift));
  const unsigned lo = static_cast<unsigned>(q) - hi * static_cast<unsigned>(a.src_group);
  return static_cast<long long>(hi) * a.src_ld + static_cast<long long>(lo) * a.C;
}

// sum over aligned groups of G lanes (G a power of two), result on every lane of the group.  Up to 16 lanes -- one DPP
// row -- the butterfly runs on the vector ALU (xor 1 / xor 2 inside quads, then the mirrors inside 8 and 16 lanes) instead
// of one ds_bpermute round trip through the LDS pipe per level; wider groups finish with shuffles.
template <int G> __device__ __forceinline__ float group_sum(float d) {
  auto step = [&](auto ctrl) __attribute__((always_inline)) {
    d += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(d), decltype(ctrl)::value, 0xf, 0xf, false));
  };
  if (G >= 2) step(std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
  if (G >= 4) step(std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
  if (G >= 8) step(std::integral_constant<int, 0x141>{});     // row_half_mirror: the other quad of the 8
  if (G >= 16) step(std::integral_constant<int, 0x140>{});    // row_mirror: the other 8 of the 16
#pragma unroll
  for (int off = 16; off < G; off <<= 1) d += __shfl_xor(d, off);
  return d;
}

// wave-uniform 64-bit value -> SGPR pair (scalar address arithmetic for the row burst)
__device__ __forceinline__ long long uniform_ll(long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(static_cast<unsigned long long>(v) >> 32));
  return static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo);
}

// Accumulate edges [ea, eb) (chunk-relative LDS positions) for the channel tile starting at ct; the
// result (summed over edge groups) is returned in acc on every lane.
// `c` is the lane's channel offset CLAMPED into the tile by the caller (min(c, c_hi - VEC)): every lane loads, lanes past the
// row's end (`!chan_ok`) re-read its last vector and their accumulators are never stored.  With the loads under
// `if (chan_ok)` -- a lane-divergent condition -- every one of the U unrolled row loads sat in its own exec-masked block,
// and the compiler, unable to count outstanding loads across those blocks, put `s_waitcnt vmcnt(0)` in front of each: the
// U loads of a wave ran one after the other (ISA of rounds 1-3; U = 2 / 4 / 8 "measured equal" for that reason).
template <int VEC, bool GROUPED, bool UNI, int DLPR = 0>   // DLPR > 0: DOT mode with DLPR lanes per edge (compile time)
__device__ __forceinline__ void accumulate_piece(const GatherArgs& a, const float* __restrict__ src, const long long* s_off,
                                                 const float* s_w, int ea, int eb, int c, bool chan_ok, int grp,
                                                 int epg, float (&acc)[VEC], const float* dvec = nullptr,
                                                 float dscale = 0.f, float* lacc = nullptr) {
  constexpr bool DOT = DLPR > 0;
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  int e = ea + grp;
#ifndef SG_GATHER_U
#define SG_GATHER_U 4
#endif
  constexpr int U = SG_GATHER_U;   // rows in flight per edge group (2 / 4 / 8 measured equal on the unsliced path)
  for (; e + (U - 1) * epg < eb; e += U * epg) {
    float x[U][VEC];
    float wv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long long off = s_off[e + u * epg];
      wv[u] = s_w[e + u * epg];
      if (UNI) {
        off = uniform_ll(off);
        wv[u] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wv[u])));
      }
      ld_row<VEC>(x[u], src + off + c);      // unconditional (c is clamped by the caller): see the note above the function
    }
    if (DOT) {   // the row is in registers: its inner product with the segment's vector gives the weight
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float d = 0.f;
        if (chan_ok) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) d = fmaf(x[u][v], dvec[v], d);
        }
        d = group_sum<DOT ? DLPR : 1>(d);
        const float r = d - wv[u];
        *lacc += r * r;
        wv[u] = dscale * r;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = fmaf(wv[u], x[u][v], acc[v]);
  }
  for (; e < eb; e += epg) {
    long long off = s_off[e];
    float wv = s_w[e];
    if (UNI) {
      off = uniform_ll(off);
      wv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, wv)));
    }
    float x[VEC];
    ld_row<VEC>(x, src + off + c);
    if (DOT) {
      float d = 0.f;
      if (chan_ok) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) d = fmaf(x[v], dvec[v], d);
      }
      d = group_sum<DOT ? DLPR : 1>(d);
      const float r = d - wv;
      *lacc += r * r;
      wv = dscale * r;
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = fmaf(wv, x[v], acc[v]);
  }
  if (!UNI) {
    if (DOT) {
#pragma unroll
      for (int off = (DLPR > 0 ? DLPR : kWave); off < kWave; off <<= 1) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += __shfl_xor(acc[v], off);
      }
    } else {
      for (int off = a.lpr; off < kWave; off <<= 1) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] += __shfl_xor(acc[v], off);
      }
    }
  }
}

template <int VEC, bool GROUPED, bool UNI, int DLPR = 0>
__global__ __launch_bounds__(kWave) void seg_gather_kernel(const GatherArgs a) {
  constexpr bool DOT = DLPR > 0;
  __shared__ long long s_off[kChunk];   // element offset of each edge's source row (index -> row address done once)
  __shared__ float s_w[kChunk];
  __shared__ int32_t s_ptr[kPtrTile + 1];

  const int lane = threadIdx.x;
  // Column slicing (n_slices in {2,4,8}): workgroup x runs on XCD x % 8 (round-robin dispatch of the default SPX partition
  // mode; a locality assumption only -- results do not depend on it), and XCD i only ever
  // touches the 4*Cs-byte column slice (i % n_slices) of every source row, so its private 4 MB L2 faces a working set
  // n_slices times smaller than the source matrix.  The 8 / n_slices XCDs that share a slice split the chunks.
  int k = blockIdx.x, slice = 0;
  if (a.xcd_ranges) {
    const int span = (a.n_chunks + 7) >> 3;
    k = (blockIdx.x & 7) * span + (blockIdx.x >> 3);
    if (k >= a.n_chunks) return;
  } else if (a.n_slices > 1) {
    const int xcd = blockIdx.x & 7;
    slice = xcd % a.n_slices;
    k = (blockIdx.x >> 3) * (8 / a.n_slices) + xcd / a.n_slices;
    if (k >= a.n_chunks) return;
  }
  const int c_lo = slice * a.Cs, c_hi = min(a.C, c_lo + a.Cs);
  const int b = blockIdx.y;
  const int32_t* __restrict__ indptr = a.indptr;
  const int E = indptr[a.seg_num];  // covered edges; positions >= E are padding
  const long long cb64 = static_cast<long long>(k) * kChunk;
  int32_t* tailseg = a.ws_tailseg + static_cast<long long>(b) * a.n_chunks + k;
  if (cb64 >= E && !(E == 0 && k == 0)) {
    if (lane == 0 && slice == 0) {
      *tailseg = -1;
      if (DOT && a.loss_part) a.loss_part[static_cast<long long>(b) * a.n_chunks + k] = 0.f;
    }
    return;
  }
  const int cb = static_cast<int>(cb64);
  const int ce = min(cb + kChunk, E);
  const bool is_last = (ce == E);
  const float* __restrict__ src = a.src + static_cast<long long>(b) * a.src_bs;
  float* __restrict__ dst = a.dst + static_cast<long long>(b) * a.dst_bs;

  // ---- stage this chunk's edges in LDS (coalesced) -------------------------------------------------
  for (int q = lane; q < ce - cb; q += kWave) {
    const int j = cb + q;
    s_off[q] = src_row_off<GROUPED>(a, a.idx[j]);
    float wv = 1.f;
    if (a.w) wv = a.w[static_cast<long long>(b) * a.w_bs + (a.wpos ? a.wpos[j] : j)];
    s_w[q] = wv;
  }
  const int s_lo = wave_lower_bound(indptr, a.seg_num, cb, lane);
  __syncthreads();  // single-wave workgroup: only orders the LDS writes above before the reads below

  const int lpr = DOT ? DLPR : a.lpr;
  const int epg = kWave / lpr;
  const int grp = UNI ? 0 : lane / lpr;
  const int slot = UNI ? lane : lane % lpr;
  const int ctile = lpr * VEC;
  const long long wsrow = (static_cast<long long>(b) * a.n_chunks + k) * a.C;
  const bool add = (a.req == SG_REQ_ADD);

  // ---- head piece: the segment that started before this chunk --------------------------------------
  const int p_lo = indptr[s_lo];  // s_lo <= seg_num
  // DOT mode: one channel pass (C <= lpr * VEC, checked by the launcher); the segment's own vector is loaded per piece
  float lacc = 0.f;
  float dscale = 0.f;
  if (DOT) dscale = a.dot_scale * (a.dot_scale_dev ? *a.dot_scale_dev : 1.f);
  auto load_dvec = [&](float (&dv)[VEC], int seg, int c, bool ok) {
    const int row = a.dot_mod > 0 ? seg % a.dot_mod : seg;
#pragma unroll
    for (int v = 0; v < VEC; ++v) dv[v] = 0.f;
    if (ok) ld_vec<VEC>(dv, a.dot_other + static_cast<long long>(row) * a.C + c);
  };
  if (cb < ce && p_lo > cb) {
    const int hb = min(ce, p_lo);
    for (int ct = c_lo; ct < c_hi; ct += ctile) {
      const int c = ct + slot * VEC;
      const bool chan_ok = c < c_hi;
      const int cl = min(c, c_hi - VEC);      // clamped: the loads of accumulate_piece are unconditional
      float acc[VEC];
      if (DOT) {
        float dv[VEC];
        load_dvec(dv, s_lo - 1, c, chan_ok);       // the segment that holds edge cb
        accumulate_piece<VEC, GROUPED, UNI, DLPR>(a, src, s_off, s_w, 0, hb - cb, cl, chan_ok, grp, epg, acc, dv, dscale, &lacc);
      } else {
        accumulate_piece<VEC, GROUPED, UNI>(a, src, s_off, s_w, 0, hb - cb, cl, chan_ok, grp, epg, acc);
      }
      if (grp == 0 && chan_ok) st_vec<VEC>(a.ws_head + wsrow + c, acc);
    }
  }

  // ---- segments that start in this chunk ------------------------------------------------------------
  int s = s_lo;
  int tail_s = -1;
  bool done = false;
  while (!done && s < a.seg_num) {
    const int cnt = min(kPtrTile, a.seg_num - s);
    __syncthreads();
    if (lane <= cnt) s_ptr[lane] = indptr[s + lane];
    if (lane == 0 && cnt == kPtrTile) s_ptr[kPtrTile] = indptr[s + kPtrTile];
    __syncthreads();
    for (int t = 0; t < cnt; ++t, ++s) {
      const int pb = __builtin_amdgcn_readfirstlane(s_ptr[t]);
      const int pe = __builtin_amdgcn_readfirstlane(s_ptr[t + 1]);
      if (!is_last && pb >= ce) { done = true; break; }
      const bool whole = (pe <= ce);
      const int eb = whole ? pe : ce;
      float* out;
      if (whole) {
        const int dg = s / a.dst_group;
        out = dst + static_cast<long long>(dg) * a.dst_ld + static_cast<long long>(s - dg * a.dst_group) * a.C;
      } else {
        out = a.ws_tail + wsrow;
      }
      const float scale = (a.mean && pe > pb) ? 1.f / static_cast<float>(pe - pb) : 1.f;
      for (int ct = c_lo; ct < c_hi; ct += ctile) {
        const int c = ct + slot * VEC;
        const bool chan_ok = c < c_hi;
        const int cl = min(c, c_hi - VEC);    // clamped: the loads of accumulate_piece are unconditional
        float acc[VEC];
        if (DOT) {
          float dv[VEC];
          load_dvec(dv, s, c, chan_ok);
          accumulate_piece<VEC, GROUPED, UNI, DLPR>(a, src, s_off, s_w, pb - cb, eb - cb, cl, chan_ok, grp, epg, acc, dv, dscale,
                                                    &lacc);
        } else {
          accumulate_piece<VEC, GROUPED, UNI>(a, src, s_off, s_w, pb - cb, eb - cb, cl, chan_ok, grp, epg, acc);
        }
        if (grp == 0 && chan_ok) {
          if (whole) {
            if (a.mean) {
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc[v] *= scale;
            }
            if (add) {
              float old[VEC];
              ld_vec<VEC>(old, out + c);
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc[v] += old[v];
            }
            if (a.act) {
#pragma unroll
              for (int v = 0; v < VEC; ++v) acc[v] = gather_act(acc[v], a.act, a.slope);
            }
          }
          st_vec<VEC>(out + c, acc);
        }
      }
      if (!whole) { tail_s = s; done = true; break; }
    }
  }
  if (lane == 0 && slice == 0) *tailseg = tail_s;
  if (DOT && a.loss_part) {      // every lane of an edge's group holds the same residual: the wave sum counts it lpr times
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lacc += __shfl_xor(lacc, off);
    if (lane == 0) a.loss_part[static_cast<long long>(b) * a.n_chunks + k] = lacc / static_cast<float>(a.lpr);
  }
}

// Adds, in chunk order, the partial rows of every segment that straddles chunk boundaries.
template <int VEC>
__global__ __launch_bounds__(kWave) void seg_gather_fixup_kernel(const GatherArgs a) {
  const int k = blockIdx.x;
  const int b = blockIdx.y;
  const int s = a.ws_tailseg[static_cast<long long>(b) * a.n_chunks + k];
  if (s < 0) return;
  const int lane = threadIdx.x;
  const int pb = a.indptr[s];
  const int pe = a.indptr[s + 1];
  const long long base = static_cast<long long>(b) * a.n_chunks;
  const int dg = s / a.dst_group;
  float* out = a.dst + static_cast<long long>(b) * a.dst_bs + static_cast<long long>(dg) * a.dst_ld +
               static_cast<long long>(s - dg * a.dst_group) * a.C;
  const float scale = (a.mean && pe > pb) ? 1.f / static_cast<float>(pe - pb) : 1.f;
  for (int c = lane * VEC; c < a.C; c += kWave * VEC) {
    float acc[VEC];
    ld_vec<VEC>(acc, a.ws_tail + (base + k) * a.C + c);
    for (int kk = k + 1; kk < a.n_chunks && static_cast<long long>(kk) * kChunk < pe; ++kk) {
      float h[VEC];
      ld_vec<VEC>(h, a.ws_head + (base + kk) * a.C + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += h[v];
    }
    if (a.mean) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] *= scale;
    }
    if (a.req == SG_REQ_ADD) {
      float old[VEC];
      ld_vec<VEC>(old, out + c);
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] += old[v];
    }
    if (a.act) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] = gather_act(acc[v], a.act, a.slope);
    }
    st_vec<VEC>(out + c, acc);
  }
}

static inline int64_t n_chunks_for(int64_t nnz) { return nnz <= 0 ? 1 : (nnz + kChunk - 1) / kChunk; }

size_t gather_workspace_bytes(int64_t batch, int64_t nnz, int64_t C) {
  const int64_t nc = n_chunks_for(nnz);
  // head + tail partial rows, then the tail-segment ids (kept 16-byte aligned)
  size_t rows = static_cast<size_t>(batch) * nc * C * sizeof(float);
  rows = (rows + 15) & ~static_cast<size_t>(15);
  return 2 * rows + static_cast<size_t>(batch) * nc * sizeof(int32_t) + 16;
}

template <int VEC>
static void launch_variants(const GatherArgs& a, dim3 fix_grid, hipStream_t st, bool grouped, bool uni, bool dot = false) {
  dim3 grid = fix_grid;   // sliced: 8 workgroups (one per XCD) per group of 8 / n_slices chunks
  if (a.xcd_ranges) {
    grid.x = ((static_cast<unsigned>(a.n_chunks) + 7u) / 8u) * 8u;
  } else if (a.n_slices > 1) {
    const unsigned per = 8u / static_cast<unsigned>(a.n_slices);
    grid.x = (static_cast<unsigned>(a.n_chunks) + per - 1) / per * 8u;
  }
  if (dot) {      // VEC = 4, plain source rows, several edges per wave step (checked by the launcher)
    if (VEC == 4) {
      switch (a.lpr) {
        case 1: hipLaunchKernelGGL((seg_gather_kernel<4, false, false, 1>), grid, dim3(kWave), 0, st, a); break;
        case 2: hipLaunchKernelGGL((seg_gather_kernel<4, false, false, 2>), grid, dim3(kWave), 0, st, a); break;
        case 4: hipLaunchKernelGGL((seg_gather_kernel<4, false, false, 4>), grid, dim3(kWave), 0, st, a); break;
        case 8: hipLaunchKernelGGL((seg_gather_kernel<4, false, false, 8>), grid, dim3(kWave), 0, st, a); break;
        case 16: hipLaunchKernelGGL((seg_gather_kernel<4, false, false, 16>), grid, dim3(kWave), 0, st, a); break;
        default: hipLaunchKernelGGL((seg_gather_kernel<4, false, false, 32>), grid, dim3(kWave), 0, st, a); break;
      }
    }
  } else if (grouped) {
    if (uni) hipLaunchKernelGGL((seg_gather_kernel<VEC, true, true>), grid, dim3(kWave), 0, st, a);
    else hipLaunchKernelGGL((seg_gather_kernel<VEC, true, false>), grid, dim3(kWave), 0, st, a);
  } else {
    if (uni) hipLaunchKernelGGL((seg_gather_kernel<VEC, false, true>), grid, dim3(kWave), 0, st, a);
    else hipLaunchKernelGGL((seg_gather_kernel<VEC, false, false>), grid, dim3(kWave), 0, st, a);
  }
  hipLaunchKernelGGL((seg_gather_fixup_kernel<VEC>), fix_grid, dim3(kWave), 0, st, a);
}

// Generic launcher shared by every public entry point built on the gather kernel.
// ---- measurement aid (bench.py `roofline`): when enabled, every gather launch -- called directly or from inside the
// fused aggregator entry -- is bracketed with HIP events ON ITS OWN STREAM.  Off by default; the compute entry
// points stay stateless otherwise.
struct ProfRecord {
  hipEvent_t a, b;
  int64_t nnz, C, src_bytes;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRecord> g_prof;

// returns the index of the new record (-1: profiling off); prof_end records the stop event into THAT record, so
// concurrent launches from several host threads / streams cannot cross their events
static long prof_begin(hipStream_t st, int64_t nnz, int64_t C, int64_t src_bytes) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on) return -1;
  ProfRecord r{};
  r.nnz = nnz; r.C = C; r.src_bytes = src_bytes;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
  (void)hipEventRecord(r.a, st);
  g_prof.push_back(r);
  return static_cast<long>(g_prof.size()) - 1;
}
static void prof_end(long rec, hipStream_t st) {
  if (rec < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (rec < static_cast<long>(g_prof.size())) (void)hipEventRecord(g_prof[rec].b, st);
}

// tuning environment variables, read ONCE (getenv is off the launch path)
static int env_int_once(const char* name) {
  const char* e = getenv(name);
  return e ? atoi(e) : 0;
}
static std::atomic<int> g_slices{-1}, g_slices_force{-1};   // -1: not initialised from the environment yet
static int env_slices() {
  int v = g_slices.load(std::memory_order_relaxed);
  if (v < 0) { v = env_int_once("SG_GATHER_SLICES"); if (v < 0) v = 0; g_slices.store(v, std::memory_order_relaxed); }
  return v;
}
static int env_slices_force() {
  int v = g_slices_force.load(std::memory_order_relaxed);
  if (v < 0) { v = env_int_once("SG_GATHER_SLICES_FORCE"); if (v < 0) v = 0; g_slices_force.store(v, std::memory_order_relaxed); }
  return v;
}

struct DotArgs {   // DOT mode of the gather (see GatherArgs)
  const float* other;
  const float* scale_dev;
  float* loss_part;
  float scale;
  int32_t mod;
};
int launch_gather_ex(float* dst, int64_t dst_group, int64_t dst_ld, int64_t dst_bs, const float* src, int64_t src_group,
                     int64_t src_ld, int64_t src_bs, const float* w, int64_t w_bs, const int32_t* wpos,
                     const int32_t* idx, const int32_t* indptr, int64_t batch, int64_t seg_num, int64_t nnz, int64_t C,
                     int req, int mean, int act, float slope, void* workspace, size_t workspace_bytes, hipStream_t st,
                     int64_t src_bytes, int xcd_ranges, const DotArgs* dot = nullptr);

int launch_gather(float* dst, int64_t dst_group, int64_t dst_ld, int64_t dst_bs, const float* src, int64_t src_group,
                  int64_t src_ld, int64_t src_bs, const float* w, int64_t w_bs, const int32_t* wpos,
                  const int32_t* idx, const int32_t* indptr, int64_t batch, int64_t seg_num, int64_t nnz, int64_t C,
                  int req, int mean, int act, float slope, void* workspace, size_t workspace_bytes, hipStream_t st,
                  int64_t src_bytes) {
  return launch_gather_ex(dst, dst_group, dst_ld, dst_bs, src, src_group, src_ld, src_bs, w, w_bs, wpos, idx, indptr, batch,
                          seg_num, nnz, C, req, mean, act, slope, workspace, workspace_bytes, st, src_bytes, 0);
}

int launch_gather_ex(float* dst, int64_t dst_group, int64_t dst_ld, int64_t dst_bs, const float* src, int64_t src_group,
                     int64_t src_ld, int64_t src_bs, const float* w, int64_t w_bs, const int32_t* wpos,
                     const int32_t* idx, const int32_t* indptr, int64_t batch, int64_t seg_num, int64_t nnz, int64_t C,
                     int req, int mean, int act, float slope, void* workspace, size_t workspace_bytes, hipStream_t st,
                     int64_t src_bytes, int xcd_ranges, const DotArgs* dot) {
  if (!valid_req(req)) return fail(SG_ERR_INVALID, "req must be 0 (null), 1 (write) or 3 (add), got %d", req);
  if (req == SG_REQ_NULL) return SG_OK;
  if (batch < 0 || seg_num < 0 || nnz < 0 || C < 0) return fail(SG_ERR_INVALID, "negative dimension");
  if (batch == 0 || seg_num == 0 || C == 0) return SG_OK;
  if (seg_num >= (1ll << 31) - 1 || nnz >= (1ll << 31) - kChunk || C >= (1 << 24))
    return fail(SG_ERR_INVALID, "seg_num/nnz must fit int32 indices (seg_num=%lld nnz=%lld C=%lld)",
                (long long)seg_num, (long long)nnz, (long long)C);
  if (dst_group < 1 || src_group < 1 || src_group >= (1 << 30)) return fail(SG_ERR_INVALID, "bad group size");
  if (act < SG_ACT_NONE || act > SG_ACT_TANH) return fail(SG_ERR_INVALID, "bad activation %d", act);
  if (!dst || !src || !indptr || (nnz > 0 && !idx)) return fail(SG_ERR_INVALID, "null pointer argument");
  const size_t need = gather_workspace_bytes(batch, nnz, C);
  if (workspace_bytes < need || !workspace)
    return fail(SG_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);

  GatherArgs a{};
  a.dst = dst; a.src = src; a.w = w; a.wpos = wpos; a.idx = idx; a.indptr = indptr;
  a.n_chunks = static_cast<int32_t>(n_chunks_for(nnz));
  char* wsp = static_cast<char*>(workspace);
  wsp = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(wsp) + 15) & ~static_cast<uintptr_t>(15));
  size_t rows = static_cast<size_t>(batch) * a.n_chunks * C * sizeof(float);
  rows = (rows + 15) & ~static_cast<size_t>(15);
  a.ws_head = reinterpret_cast<float*>(wsp);
  a.ws_tail = reinterpret_cast<float*>(wsp + rows);
  a.ws_tailseg = reinterpret_cast<int32_t*>(wsp + 2 * rows);
  a.dst_ld = dst_ld; a.src_ld = src_ld; a.dst_bs = dst_bs; a.src_bs = src_bs; a.w_bs = w_bs;
  a.dst_group = static_cast<int32_t>(dst_group);
  a.src_group = static_cast<int32_t>(src_group);
  a.seg_num = static_cast<int32_t>(seg_num);
  a.C = static_cast<int32_t>(C);
  a.req = req; a.mean = mean; a.act = act; a.slope = slope;
  if (src_group > 1) {
    int l = 0;
    while ((1ll << l) < src_group) ++l;  // ceil(log2 d), >= 1
    const unsigned long long num = 1ull << (31 + l);
    a.src_magic = static_cast<uint32_t>((num + static_cast<unsigned long long>(src_group) - 1) /
                                        static_cast<unsigned long long>(src_group));
    a.src_shift = l;
  }
  // vector width by alignment of every row start
  int vec = 1;
  auto ok = [&](int v) {
    return C % v == 0 && dst_ld % v == 0 && src_ld % v == 0 && dst_bs % v == 0 && src_bs % v == 0 &&
           aligned(dst, 4 * v) && aligned(src, 4 * v);
  };
  if (ok(4)) vec = 4; else if (ok(2)) vec = 2;
  // Column slicing across XCDs (see the kernel).  Measured (bench.py, MovieLens-10M shape, 68-105 MB sources that sit
  // in the 256 MB Infinity Cache): 1.278 ms -> 1.187 ms (2 slices) -> 1.113 ms (4) -> 1.25 (8) per 10 M-edge launch;
  // with sources far beyond the Infinity Cache (hbm-stress, 0.5-8 GB) the 1 KiB row bursts are worth more than the
  // L2 hits: 10.2 ms -> 11.0 (2) -> 12.7 (4).  So: slice only when the caller tells us the source footprint
  // (src_bytes, 0 = unknown) and it fits the Infinity Cache next to the output, but not a single L2.
  int slices = 1;
  auto valid = [&](int v) { return (v == 1 || v == 2 || v == 4 || v == 8) && C % (v * vec) == 0; };
  if (vec == 4 && C >= 256 && nnz >= (1 << 20) && src_bytes >= (8ll << 20) && src_bytes <= (160ll << 20)) {
    slices = SG_GATHER_DEFAULT_SLICES;
    if (env_slices() > 0) slices = env_slices();     // tuning aid: applies to eligible launches only
    if (!valid(slices)) slices = 1;
  }
  if (env_slices_force() > 0 && valid(env_slices_force())) slices = env_slices_force();   // tests: slice every launch
  if (xcd_ranges || dot) slices = 1;     // the source-row partition replaces the column slices; DOT needs whole rows
  a.xcd_ranges = xcd_ranges ? 1 : 0;
  a.n_slices = slices;
  a.Cs = static_cast<int32_t>(C / slices);
  int lpr = 1;
  while (lpr < kWave && static_cast<int64_t>(lpr) * vec < a.Cs) lpr <<= 1;
  a.lpr = lpr;
  const bool uni = (lpr == kWave);
  const bool grouped = (src_group > 1);
  if (dot) {
    if (vec != 4 || grouped || uni || batch != 1 || mean || act != SG_ACT_NONE || !w || !dot->other)
      return fail(SG_ERR_UNSUPPORTED, "DOT-mode gather needs 16-byte aligned rows of 4..128 floats, plain source rows, batch 1");
    a.dot_other = dot->other; a.dot_scale_dev = dot->scale_dev; a.loss_part = dot->loss_part;
    a.dot_scale = dot->scale; a.dot_mod = dot->mod;
  }
  dim3 grid(static_cast<unsigned>(a.n_chunks), static_cast<unsigned>(batch));
  if (batch > 65535) return fail(SG_ERR_INVALID, "batch > 65535 not supported");
  const long rec = prof_begin(st, nnz * batch, C, src_bytes);
  if (vec == 4) launch_variants<4>(a, grid, st, grouped, uni, dot != nullptr);
  else if (vec == 2) launch_variants<2>(a, grid, st, grouped, uni);
  else launch_variants<1>(a, grid, st, grouped, uni);
  prof_end(rec, st);
  return check_launch("seg_gather");
}

// dst[s, :] (+)= act( sum_p part[p * seg_num + s, :] ), p ascending: joins the partial results of a source-partitioned
// gather.  One float4 (or float) per thread.
template <int VEC>
__global__ void sum_parts_kernel(float* __restrict__ dst, long long dst_group, long long dst_ld,
                                 const float* __restrict__ part, long long seg_num, int parts, int C, int req, int act,
                                 float slope) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int cv = C / VEC;
  if (i >= seg_num * cv) return;
  const long long s = i / cv;
  const int c = static_cast<int>(i - s * cv) * VEC;
  float acc[VEC];
  ld_vec<VEC>(acc, part + s * C + c);
  for (int p = 1; p < parts; ++p) {
    float t[VEC];
    ld_vec<VEC>(t, part + (static_cast<long long>(p) * seg_num + s) * C + c);
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += t[v];
  }
  const long long g = s / dst_group;
  float* o = dst + g * dst_ld + (s - g * dst_group) * C + c;
  if (req == SG_REQ_ADD) {
    float old[VEC];
    ld_vec<VEC>(old, o);
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] += old[v];
  }
  if (act) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = gather_act(acc[v], act, slope);
  }
  st_vec<VEC>(o, acc);
}

}  // namespace sg

// ------------------------------------------------------------------------------------------------------
// C ABI (see include/stargcn.h for the reference citations of every entry point)
// ------------------------------------------------------------------------------------------------------
SG_API size_t sg_seg_weighted_pool_workspace_bytes(int64_t batch, int64_t seg_num, int64_t nnz, int64_t feat_dim) {
  (void)seg_num;
  return sg::gather_workspace_bytes(batch, nnz, feat_dim);
}

SG_API int sg_seg_weighted_pool_hip(float* dst, const float* data, const float* weights, const int32_t* indices,
                                    const int32_t* indptr, int64_t batch, int64_t seg_num, int64_t total_ind_num,
                                    int64_t nnz, int64_t feat_dim, int req, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  if (weights == nullptr && nnz > 0 && req != SG_REQ_NULL) return sg::fail(SG_ERR_INVALID, "weights is null");
  return sg::launch_gather(dst, 1, feat_dim, seg_num * feat_dim, data, 1, feat_dim, total_ind_num * feat_dim, weights,
                           nnz, nullptr, indices, indptr, batch, seg_num, nnz, feat_dim, req, 0, SG_ACT_NONE, 0.f,
                           workspace, workspace_bytes, static_cast<hipStream_t>(stream),
                           batch * total_ind_num * feat_dim * static_cast<int64_t>(sizeof(float)));
}

SG_API int sg_seg_gather_sum_hinted_hip(float* dst, int64_t dst_group, int64_t dst_ld, const float* src,
                                        int64_t src_group, int64_t src_ld, const float* weights, const int32_t* indices,
                                        const int32_t* indptr, int64_t seg_num, int64_t nnz, int64_t feat_dim, int req,
                                        int act, float slope, void* workspace, size_t workspace_bytes, void* stream,
                                        int64_t src_bytes) {
  return sg::launch_gather(dst, dst_group, dst_ld, 0, src, src_group, src_ld, 0, weights, 0, nullptr, indices, indptr,
                           1, seg_num, nnz, feat_dim, req, 0, act, slope, workspace, workspace_bytes,
                           static_cast<hipStream_t>(stream), src_bytes);
}

SG_API size_t sg_seg_gather_sum_parts_workspace_bytes(int64_t seg_num, int64_t parts, int64_t nnz, int64_t feat_dim) {
  if (seg_num < 0 || parts < 1 || nnz < 0 || feat_dim < 0) return 0;
  return ((static_cast<size_t>(parts) * seg_num * feat_dim * sizeof(float) + 255) & ~static_cast<size_t>(255)) + 256 +
         sg::gather_workspace_bytes(1, nnz, feat_dim);
}

SG_API int sg_seg_gather_sum_parts_hip(float* dst, int64_t dst_group, int64_t dst_ld, const float* src,
                                       int64_t src_group, int64_t src_ld, const float* weights, const int32_t* wpos,
                                       const int32_t* indices_p, const int32_t* indptr_p, int64_t seg_num, int64_t parts,
                                       int64_t nnz, int64_t feat_dim, int req, int act, float slope, void* workspace,
                                       size_t workspace_bytes, void* stream, int64_t src_bytes) {
  if (!sg::valid_req(req)) return sg::fail(SG_ERR_INVALID, "req must be 0, 1 or 3, got %d", req);
  if (req == SG_REQ_NULL || seg_num == 0 || feat_dim == 0) return SG_OK;
  if (seg_num < 0 || nnz < 0 || feat_dim < 0 || parts < 1 || parts * seg_num >= (1ll << 31) - 1)
    return sg::fail(SG_ERR_INVALID, "bad size");
  if (dst_group < 1 || !dst) return sg::fail(SG_ERR_INVALID, "bad destination");
  if (!workspace || workspace_bytes < sg_seg_gather_sum_parts_workspace_bytes(seg_num, parts, nnz, feat_dim))
    return sg::fail(SG_ERR_WORKSPACE, "partitioned gather workspace too small");
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  const size_t pbytes = (static_cast<size_t>(parts) * seg_num * feat_dim * sizeof(float) + 255) & ~static_cast<size_t>(255);
  float* part = reinterpret_cast<float*>(base);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const size_t used = static_cast<size_t>(base - static_cast<char*>(workspace)) + pbytes;
  int rc = sg::launch_gather_ex(part, 1, feat_dim, 0, src, src_group, src_ld, 0, weights, 0, wpos, indices_p, indptr_p, 1,
                                parts * seg_num, nnz, feat_dim, SG_REQ_WRITE, 0, SG_ACT_NONE, 0.f, base + pbytes,
                                workspace_bytes - used, st, src_bytes, parts == 8 ? 1 : 0);
  if (rc != SG_OK) return rc;
  const bool v4 = feat_dim % 4 == 0 && dst_ld % 4 == 0 && sg::aligned(dst, 16);
  const long long n = seg_num * (v4 ? feat_dim / 4 : feat_dim);
  const dim3 grid(static_cast<unsigned>((n + 255) / 256));
  if (v4)
    hipLaunchKernelGGL(sg::sum_parts_kernel<4>, grid, dim3(256), 0, st, dst, static_cast<long long>(dst_group),
                       static_cast<long long>(dst_ld), part, static_cast<long long>(seg_num), static_cast<int>(parts),
                       static_cast<int>(feat_dim), req, act, slope);
  else
    hipLaunchKernelGGL(sg::sum_parts_kernel<1>, grid, dim3(256), 0, st, dst, static_cast<long long>(dst_group),
                       static_cast<long long>(dst_ld), part, static_cast<long long>(seg_num), static_cast<int>(parts),
                       static_cast<int>(feat_dim), req, act, slope);
  return sg::check_launch("sg_seg_gather_sum_parts_hip");
}

namespace sg {
// loss[0] (+)= loss_scale * sum of the per-chunk partial sums, in chunk order (one wave; deterministic)
__global__ __launch_bounds__(1024) void pair_l2_loss_kernel(float* __restrict__ loss, const float* __restrict__ part,
                                                             long long n, float loss_scale, int add) {
  __shared__ float red[16];
  float acc = 0.f;
  for (long long i = threadIdx.x; i < n; i += 1024) acc += part[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k];
    loss[0] = (add ? loss[0] : 0.f) + loss_scale * s;
  }
}
}  // namespace sg

SG_API size_t sg_pair_l2_workspace_bytes(int64_t seg_num, int64_t parts, int64_t nnz, int64_t feat_dim) {
  if (seg_num < 0 || parts < 1 || nnz < 0 || feat_dim < 0) return 0;
  const size_t nchunks = static_cast<size_t>(nnz <= 0 ? 1 : (nnz + sg::kChunk - 1) / sg::kChunk);
  return sg_seg_gather_sum_parts_workspace_bytes(seg_num, parts, nnz, feat_dim) + ((nchunks * sizeof(float) + 255) & ~static_cast<size_t>(255)) + 256;
}

SG_API int sg_pair_l2_hip(float* rows, float* loss, const float* src, const float* other, const float* y,
                          const int32_t* indices, const int32_t* indptr, int64_t seg_num, int64_t parts, int64_t nnz,
                          int64_t feat_dim, float scale, const float* scale_dev, float loss_scale, int req,
                          void* workspace, size_t workspace_bytes, void* stream, int64_t src_bytes) {
  if (!sg::valid_req(req)) return sg::fail(SG_ERR_INVALID, "req must be 0, 1 or 3, got %d", req);
  if (req == SG_REQ_NULL || seg_num == 0 || feat_dim == 0) return SG_OK;
  if (seg_num < 0 || nnz < 0 || parts < 1 || parts * seg_num >= (1ll << 31) - 1) return sg::fail(SG_ERR_INVALID, "bad size");
  if (feat_dim % 4 || feat_dim > 128) return sg::fail(SG_ERR_UNSUPPORTED, "pair_l2 handles rows of 4..128 floats (multiple of 4)");
  if (!rows || !src || !other || !y || !indptr || (nnz > 0 && !indices)) return sg::fail(SG_ERR_INVALID, "null pointer argument");
  if (!workspace || workspace_bytes < sg_pair_l2_workspace_bytes(seg_num, parts, nnz, feat_dim))
    return sg::fail(SG_ERR_WORKSPACE, "pair_l2 workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  const size_t nchunks = static_cast<size_t>(nnz <= 0 ? 1 : (nnz + sg::kChunk - 1) / sg::kChunk);
  const size_t lbytes = (nchunks * sizeof(float) + 255) & ~static_cast<size_t>(255);
  float* loss_part = reinterpret_cast<float*>(base);
  base += lbytes;
  const size_t left = workspace_bytes - static_cast<size_t>(base - static_cast<char*>(workspace));
  sg::DotArgs d{other, scale_dev, loss ? loss_part : nullptr, scale, static_cast<int32_t>(parts > 1 ? seg_num : 0)};
  int rc;
  if (parts == 1) {
    rc = sg::launch_gather_ex(rows, 1, feat_dim, 0, src, 1, feat_dim, 0, y, 0, nullptr, indices, indptr, 1, seg_num, nnz,
                              feat_dim, req, 0, SG_ACT_NONE, 0.f, base, left, st, src_bytes, 0, &d);
  } else {
    const size_t pbytes = (static_cast<size_t>(parts) * seg_num * feat_dim * sizeof(float) + 255) & ~static_cast<size_t>(255);
    float* part = reinterpret_cast<float*>(base);
    rc = sg::launch_gather_ex(part, 1, feat_dim, 0, src, 1, feat_dim, 0, y, 0, nullptr, indices, indptr, 1, parts * seg_num, nnz,
                              feat_dim, SG_REQ_WRITE, 0, SG_ACT_NONE, 0.f, base + pbytes, left - pbytes, st, src_bytes,
                              parts == 8 ? 1 : 0, &d);
    if (rc == SG_OK) {
      const long long n = seg_num * (feat_dim / 4);
      hipLaunchKernelGGL(sg::sum_parts_kernel<4>, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, rows, 1ll,
                         static_cast<long long>(feat_dim), part, static_cast<long long>(seg_num), static_cast<int>(parts),
                         static_cast<int>(feat_dim), req, SG_ACT_NONE, 0.f);
    }
  }
  if (rc != SG_OK) return rc;
  if (loss)
    hipLaunchKernelGGL(sg::pair_l2_loss_kernel, dim3(1), dim3(1024), 0, st, loss, loss_part,
                       static_cast<long long>(nchunks), loss_scale, 0);
  return sg::check_launch("sg_pair_l2_hip");
}

SG_API int sg_seg_gather_sum_hip(float* dst, int64_t dst_group, int64_t dst_ld, const float* src, int64_t src_group,
                                 int64_t src_ld, const float* weights, const int32_t* indices, const int32_t* indptr,
                                 int64_t seg_num, int64_t nnz, int64_t feat_dim, int req, int act, float slope,
                                 void* workspace, size_t workspace_bytes, void* stream) {
  return sg_seg_gather_sum_hinted_hip(dst, dst_group, dst_ld, src, src_group, src_ld, weights, indices, indptr, seg_num,
                                      nnz, feat_dim, req, act, slope, workspace, workspace_bytes, stream, 0);
}

SG_API size_t sg_seg_weighted_pool_bwd_data_workspace_bytes(int64_t batch, int64_t total_ind_num, int64_t nnz,
                                                            int64_t feat_dim) {
  (void)total_ind_num;
  return sg::gather_workspace_bytes(batch, nnz, feat_dim);
}

SG_API int sg_seg_weighted_pool_bwd_data_hip(float* ddata, const float* weights, const float* ograd,
                                             const int32_t* t_indptr, const int32_t* t_pos, const int32_t* t_seg,
                                             int64_t batch, int64_t seg_num, int64_t total_ind_num, int64_t nnz,
                                             int64_t feat_dim, int req, void* workspace, size_t workspace_bytes,
                                             void* stream) {
  if (req != SG_REQ_NULL && (!t_indptr || (nnz > 0 && (!t_pos || !t_seg || !weights))))
    return sg::fail(SG_ERR_INVALID, "transposed plan (t_indptr, t_pos, t_seg) and weights are required; build the "
                                    "plan once with sg_build_transpose_cpu");
  // gather over the transposed plan: segment n collects ograd rows t_seg[p] with weight weights[t_pos[p]]
  return sg::launch_gather(ddata, 1, feat_dim, total_ind_num * feat_dim, ograd, 1, feat_dim, seg_num * feat_dim,
                           weights, nnz, t_pos, t_seg, t_indptr, batch, total_ind_num, nnz, feat_dim, req, 0, SG_ACT_NONE,
                           0.f, workspace, workspace_bytes, static_cast<hipStream_t>(stream),
                           batch * seg_num * feat_dim * static_cast<int64_t>(sizeof(float)));
}

SG_API int sg_gather_tuning(int slices, int slices_force) {
  if (slices >= 0) sg::g_slices.store(slices, std::memory_order_relaxed);
  if (slices_force >= 0) sg::g_slices_force.store(slices_force, std::memory_order_relaxed);
  return SG_OK;
}

SG_API int sg_gather_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(sg::g_prof_mu);
  const int was = sg::g_prof_on ? 1 : 0;
  if (on && !was) {
    for (auto& r : sg::g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    sg::g_prof.clear();
  }
  sg::g_prof_on = on != 0;
  return was;
}

SG_API int64_t sg_gather_profile_read2(float* ms, int64_t* nnz, int64_t* feat_dim, int64_t* src_bytes, int64_t capacity);
SG_API int64_t sg_gather_profile_read(float* ms, int64_t* nnz, int64_t* feat_dim, int64_t capacity) {
  return sg_gather_profile_read2(ms, nnz, feat_dim, nullptr, capacity);
}
SG_API int64_t sg_gather_profile_read2(float* ms, int64_t* nnz, int64_t* feat_dim, int64_t* src_bytes, int64_t capacity) {
  std::lock_guard<std::mutex> lk(sg::g_prof_mu);
  int64_t n = 0;
  for (auto& r : sg::g_prof) {
    if (n < capacity) {
      float t = 0.f;
      (void)hipEventSynchronize(r.b);
      if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) t = -1.f;
      if (ms) ms[n] = t;
      if (nnz) nnz[n] = r.nnz;
      if (feat_dim) feat_dim[n] = r.C;
      if (src_bytes) src_bytes[n] = r.src_bytes;
      ++n;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  sg::g_prof.clear();
  return n;
}
