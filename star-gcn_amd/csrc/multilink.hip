// multilink.hip -- the fused multi-link aggregator behind ONE forward and ONE backward C-ABI call.
//
// Reference: MultiLinkGCNAggregator.hybrid_forward (mxgraph/layers/aggregators.py:111-163) issues, per layer and
// direction, R FullyConnected ops, R seg_weighted_pool ops, an add_n/concat and an activation, and MXNet's autograd
// replays the mirror image.  Here the whole thing is two entry points that sequence the library's own kernels on the
// caller's stream (no host synchronisation, no allocation: every intermediate lives in the caller's workspace):
//
//   transform first   H = x Wcat^T + bcat  (n_src, R*U')      -> ONE gather with R-grouped source rows (+ activation)
//   aggregate first   Zext = [A_0 x | .. | A_{R-1} x | A_r 1]  -> ONE MFMA contraction with [W_0 | .. | b | 0] (+ act)
//
// The per-level parameters keep the reference layout (R separate (U', D) weights and (U') biases); a pack kernel lays
// them out as Wcat / Wext in the workspace and an unpack kernel scatters the packed gradient back, so a caller that
// owns reference-shaped parameters (MXNet NDArrays, torch Parameters) needs no glue of its own.
#include <cstddef>
#include "common.hpp"

namespace sg {
namespace {

struct PtrTable {
  const float* p[SG_MAX_LINKS];
};
struct MutPtrTable {
  float* p[SG_MAX_LINKS];
};

// ---- parameter packing --------------------------------------------------------------------------------------------
// Wcat[(r*U + u), d] = W_r[u, d];  bcat[r*U + u] = b_r[u]
__global__ void pack_cat_kernel(float* __restrict__ wcat, float* __restrict__ bcat, PtrTable w, PtrTable b, int R,
                                int U, int D) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long per = static_cast<long long>(U) * D;
  if (i < per * R) {
    const int r = static_cast<int>(i / per);
    wcat[i] = w.p[r][i - r * per];
  }
  if (bcat && i < static_cast<long long>(R) * U) {
    const int r = static_cast<int>(i / U);
    bcat[i] = b.p[r] ? b.p[r][i - static_cast<long long>(r) * U] : 0.f;
  }
}
// (Up: rows per level in dwcat -- U, or U rounded up to a multiple of 4 where the fused order padded its R-expanded gradient)
__global__ void unpack_cat_kernel(MutPtrTable dw, MutPtrTable db, const float* __restrict__ dwcat,
                                  const float* __restrict__ dbcat, int R, int U, int D, int Up) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long per = static_cast<long long>(U) * D;
  if (dwcat && i < per * R) {
    const int r = static_cast<int>(i / per);
    if (dw.p[r]) dw.p[r][i - r * per] = dwcat[i + static_cast<long long>(r) * (Up - U) * D];
  }
  if (dbcat && i < static_cast<long long>(R) * U) {
    const int r = static_cast<int>(i / U);
    if (db.p[r]) db.p[r][i - static_cast<long long>(r) * U] = dbcat[i];
  }
}
// Wext (rows, ld): 'sum'   rows = U    row u        : [W_0[u,:] | .. | W_{R-1}[u,:] | b_0[u] .. b_{R-1}[u] | 0]
//                  'stack' rows = R*U  row r*U + u  : W_r[u,:] in column block r, b_r[u] in column R*D + r, else 0
__global__ void pack_ext_kernel(float* __restrict__ wext, PtrTable w, PtrTable b, int R, int U, int D, int ld,
                                int stack) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const int rows = stack ? R * U : U;
  if (i >= static_cast<long long>(rows) * ld) return;
  const int row = static_cast<int>(i / ld), col = static_cast<int>(i - static_cast<long long>(row) * ld);
  const int rr = stack ? row / U : -1, u = stack ? row - rr * U : row;
  float v = 0.f;
  if (col < R * D) {
    const int rc = col / D, d = col - rc * D;
    if (!stack || rc == rr) v = w.p[rc][static_cast<long long>(u) * D + d];
  } else if (col < R * D + R) {
    const int rc = col - R * D;
    if ((!stack || rc == rr) && b.p[rc]) v = b.p[rc][u];
  }
  wext[i] = v;
}
__global__ void unpack_ext_kernel(MutPtrTable dw, MutPtrTable db, const float* __restrict__ dwext, int R, int U, int D,
                                  int ld, int stack) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long per = static_cast<long long>(U) * D;
  if (i < per * R) {
    const int r = static_cast<int>(i / per);
    const long long e = i - r * per;
    const int u = static_cast<int>(e / D), d = static_cast<int>(e - static_cast<long long>(u) * D);
    const long long row = stack ? static_cast<long long>(r) * U + u : u;
    if (dw.p[r]) dw.p[r][e] = dwext[row * ld + static_cast<long long>(r) * D + d];
  }
  if (i < static_cast<long long>(R) * U) {
    const int r = static_cast<int>(i / U), u = static_cast<int>(i - static_cast<long long>(r) * U);
    const long long row = stack ? static_cast<long long>(r) * U + u : u;
    if (db.p[r]) db.p[r][u] = dwext[row * ld + static_cast<long long>(R) * D + r];
  }
}
// Zext[i, R*D + c] = c < R ? rowsum[i, c] : 0     (the bias rides on the support row sums; tail = alignment pad)
__global__ void fill_rowsum_kernel(float* __restrict__ zext, const float* __restrict__ rowsum, long long n_dst, int R,
                                   int D, int ld) {
  const int extra = ld - R * D;
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_dst * extra) return;
  const long long row = i / extra;
  const int c = static_cast<int>(i - row * extra);
  zext[row * ld + static_cast<long long>(R) * D + c] = c < R ? rowsum[row * R + c] : 0.f;
}

// db[r, u] = sum_i rowsum[i, r] * dpre[i, u]  -- the bias gradient of the fused order (= the column sums of dH_r without reading
// the 16-20 GB of dH).  Pass 1: workgroup b (1024 threads = 4 row groups x 256 columns) owns a contiguous block of rows; 64 rows
// of rowsum at a time go through LDS (read back as broadcasts), every thread keeps R running sums for its column over its
// group's rows, the four groups are added through LDS and part[b][r][u] is written.  Pass 2 adds the P partials in order.
// Bandwidth-bound on dpre (1.3 GB at the config-5 shard); deterministic.
constexpr int kDbParts = 512;
// dpre: row pitch ld; accum 'stack' (lstride > 0): level r reads its own column block, dpre[i, r * lstride + u].
template <int RMAX>
__global__ __launch_bounds__(1024) void bias_grad_partial_kernel(float* __restrict__ part, const float* __restrict__ rowsum,
                                                                  const float* __restrict__ dpre, long long n, int R, int U,
                                                                  long long ld, int lstride) {
  __shared__ float s_rs[64 * RMAX];
  __shared__ float s_red[3][256];
  const int t = threadIdx.x, grp = t >> 8, col = t & 255;
  const int u = blockIdx.y * 256 + col;
  const int b = blockIdx.x, P = gridDim.x;
  const long long chunk = (n + P - 1) / P;
  const long long r_lo = b * chunk, r_hi = (r_lo + chunk < n) ? r_lo + chunk : n;
  float acc[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) acc[r] = 0.f;
  for (long long base = r_lo; base < r_hi; base += 64) {
    const int rows = static_cast<int>((r_hi - base < 64) ? r_hi - base : 64);
    __syncthreads();
    for (int e = t; e < rows * R; e += 1024) s_rs[(e / R) * RMAX + (e % R)] = rowsum[base * R + e];
    __syncthreads();
    if (u < U) {
      if (lstride > 0) {                              // 'stack': one value per (row, level)
        for (int j = grp; j < rows; j += 4) {
          const float* rs = s_rs + j * RMAX;
          const float* gp = dpre + (base + j) * ld + u;
#pragma unroll
          for (int r = 0; r < RMAX; ++r)
            if (r < R) acc[r] = fmaf(rs[r], gp[static_cast<long long>(r) * lstride], acc[r]);
        }
      } else
      for (int j = grp; j < rows; j += 16) {          // rows j, j + 4, j + 8, j + 12 of the batch in flight
        float g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = (j + 4 * q < rows) ? dpre[(base + j + 4 * q) * ld + u] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float* rs = s_rs + ((j + 4 * q < rows) ? (j + 4 * q) : 0) * RMAX;
#pragma unroll
          for (int r = 0; r < RMAX; ++r) acc[r] = fmaf(rs[r], g[q], acc[r]);
        }
      }
    }
  }
  // the four row groups, in order
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    if (r < R) {          // (uniform)
      __syncthreads();
      if (grp > 0) s_red[grp - 1][col] = acc[r];
      __syncthreads();
      if (grp == 0 && u < U) {
        float v = acc[r];
        v += s_red[0][col]; v += s_red[1][col]; v += s_red[2][col];
        part[(static_cast<long long>(b) * R + r) * U + u] = v;
      }
    }
  }
}
__global__ void bias_grad_final_kernel(float* __restrict__ db, const float* __restrict__ part, int P, int RU) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= RU) return;
  float s = 0.f;
  for (int b = 0; b < P; ++b) s += part[static_cast<long long>(b) * RU + i];
  db[i] = s;
}

// dst[row, (col / U) * Up + col % U] = dout[row, col] * act'(out[row, col]) for col < W (= U or R U), zero in the padding columns
// (the fused order rounds every level's width up to a multiple of 4 floats: its gather lanes hold one float4 each)
__global__ void act_bwd_pitched_kernel(float* __restrict__ dst, long long dst_ld, const float* __restrict__ dout,
                                       const float* __restrict__ out, long long n_rows, int W, int U, int Up, int act, float slope) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n_rows * dst_ld) return;
  const long long row = i / dst_ld;
  const int pc = static_cast<int>(i - row * dst_ld), lvl = pc / Up, u = pc - lvl * Up;
  float v = 0.f;
  if (u < U) {
    const long long src = row * W + static_cast<long long>(lvl) * U + u;
    v = dout[src];
    if (act != SG_ACT_NONE) {
      const float y = out[src];
      float g = 1.f;
      switch (act) {
        case SG_ACT_LEAKY: g = y > 0.f ? 1.f : slope; break;
        case SG_ACT_RELU: g = y > 0.f ? 1.f : 0.f; break;
        case SG_ACT_SIGMOID: g = y * (1.f - y); break;
        case SG_ACT_TANH: g = 1.f - y * y; break;
        default: break;
      }
      v *= g;
    }
  }
  dst[i] = v;
}

inline unsigned blocks_for(long long n) { return static_cast<unsigned>((n + 255) / 256); }
inline size_t al(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }
inline size_t max2(size_t a, size_t b) { return a > b ? a : b; }

struct Dims {
  int64_t n_dst, n_src, nnz, D, U, ld, outw, RU;
  int64_t Up, outw_p;  // fused order: U rounded up to a multiple of 4 (one float4 per gather lane) and the pitch of its dpre rows
  int R, order, stack;
  bool saves_z;       // fused order: the forward writes the aggregates (see fused_saves_z)
};

int resolve_order(const sg_multilink_plan* p, int order) {
  if (order == SG_ORDER_AUTO) return p->n_src <= p->n_dst ? SG_ORDER_TRANSFORM_FIRST : SG_ORDER_AGGREGATE_FIRST;
  return order;
}
// SG_FUSED: 0 never, 1 whenever the fused kernel handles the widths, unset / other: the size rule
int fused_mode() {
  static const int m = [] { const char* e = getenv("SG_FUSED"); return e ? atoi(e) : -1; }();
  return m;
}
int fused_nt() {       // SG_FUSED_NT=1: gathered rows read non-temporally (measured slower at the config-5 shard: off)
  static const int m = [] { const char* e = getenv("SG_FUSED_NT"); return e ? atoi(e) : 0; }();
  return m;
}
bool has_fused(const sg_multilink_plan* p, int which) {
  return p->struct_bytes >= static_cast<int32_t>(offsetof(sg_multilink_plan, fused) + sizeof(p->fused)) &&
         p->fused[which].f_ptr && p->fused[which].f_idx && p->fused[which].f_w;
}
// Fused order: which R-expanded matrix feeds the weight gradient.  dW_r = dH_r^T x (dH = [A_r^T dpre]_r, n_src rows, written by
// the data-gradient launch) = dpre^T Z_r (Z = [A_r x]_r, n_dst rows, written by the FORWARD launch): the one on the smaller node
// side is 20 % smaller at the config-5 shard (16.4 against 20.5 GB) and so is its GEMM (K = 1 M against 1.25 M).
bool fused_saves_z(const sg_multilink_plan* p) {
  static const int mode = [] { const char* e = getenv("SG_FUSED_SAVEZ"); return e ? atoi(e) : -1; }();    // 0 / 1: never / always (A-B)
  return mode < 0 ? p->n_dst < p->n_src : mode != 0;
}
// the order for these widths; AUTO prefers the fused kernel where the R-expanded matrix would cost HBM time
int resolve_order2(const sg_multilink_plan* p, int order, int64_t in_dim, int64_t upl, int accum) {
  if (order != SG_ORDER_AUTO) return order;
  // (the fused kernel is built and measured for 256 -> 256 'sum'; the other widths / 'stack' it handles are for callers who ask)
  if (accum == SG_ACCUM_SUM && in_dim == 256 && upl == 256 && p->nnz > 0 && p->n_dst > 0 && p->n_src > 0 &&
      sg_agg_fused_supported(in_dim, upl, p->num_links)) {
    const int mode = fused_mode();
    const int64_t small_side = p->n_src < p->n_dst ? p->n_src : p->n_dst;
    const int64_t big_side = p->n_src < p->n_dst ? p->n_dst : p->n_src;
    // Measured per direction, forward + backward, on bench.py's synthetic graphs from 17 k x 16.5 k nodes / 1.5 M edges to the
    // config-5 shard, 2 .. 32 levels (tools/exp_r5_fused.py dir-ab, tools/mid_size_fused_ab.sh; profiles/r5_fused_kernel.md
    // section 7), fused time / time of the better unfused order:
    //   * node sides within a factor of two and the smaller side's R-expanded matrix beyond ~200 MB (it does not stay in the
    //     L2s / the Infinity Cache next to its sources): 0.65 .. 0.97 -- 30 k x 25 k, 10 levels 0.68; 60 k x 50 k, 16 levels 0.65;
    //     120 k x 100 k 0.67; 300 k x 250 k with TWO levels 0.95; the shard 0.78 / 0.83; down to 258 tiles (17 k x 16.5 k) 0.86 .. 0.95;
    //   * lopsided graphs: the unfused orders put the R-expanded matrix and its GEMM on the SMALL side, the fused kernel contracts
    //     on whichever side is the destination: ties at 2.5 .. 4 : 1 (100 k x 40 k 0.97, 400 k x 100 k 0.97 / 1.02, 1 M x 300 k 0.98),
    //     losses beyond (200 k x 33 k 1.02, 156 k x 1 M -- one rank's block of eight -- 1.30, 500 k x 20 k 1.5, the MovieLens-10M
    //     shape 70 k x 10.7 k 1.5: 167 item tiles on 256 CUs, sources that live in the L2s);
    //   * smaller expanded matrices (70 k x 35 k, 5 levels: 171 MB) tie.
    const bool big = small_side >= (1ll << 14) && p->num_links >= 2 && big_side <= 2 * small_side &&
                     small_side * p->num_links * in_dim * 4 > (192ll << 20);
    if (mode == 1 || (mode != 0 && big)) return SG_ORDER_FUSED;
  }
  return resolve_order(p, order);
}

int make_dims(Dims* d, const sg_multilink_plan* p, int64_t in_dim, int64_t upl, int order, int accum) {
  if (!p) return fail(SG_ERR_INVALID, "plan is null");
  if (p->num_links < 1 || p->num_links > SG_MAX_LINKS)
    return fail(SG_ERR_INVALID, "num_links %d outside [1, %d]", p->num_links, SG_MAX_LINKS);
  if (p->n_dst < 0 || p->n_src < 0 || p->nnz < 0 || in_dim < 1 || upl < 1) return fail(SG_ERR_INVALID, "negative / empty size");
  if (order < SG_ORDER_AUTO || order > SG_ORDER_FUSED) return fail(SG_ERR_INVALID, "order %d", order);
  if (accum != SG_ACCUM_SUM && accum != SG_ACCUM_STACK) return fail(SG_ERR_INVALID, "accum %d", accum);
  d->n_dst = p->n_dst; d->n_src = p->n_src; d->nnz = p->nnz; d->D = in_dim; d->U = upl; d->R = p->num_links;
  d->order = resolve_order2(p, order, in_dim, upl, accum);
  if (d->order == SG_ORDER_FUSED) {
    const bool can = p->nnz > 0 && sg_agg_fused_supported(in_dim, upl, p->num_links) && has_fused(p, 0) && has_fused(p, 1);
    if (!can && order == SG_ORDER_AUTO) d->order = resolve_order(p, SG_ORDER_AUTO);     // a caller that attached no f-plans
    else if (!can)
      return fail(SG_ERR_UNSUPPORTED, "SG_ORDER_FUSED needs in_dim in 4 .. 256 (multiple of 4), units_per_level in 1 .. 256, at least "
                                      "one edge and plan->fused[0..1] (sg_agg_fused_plan_build_hip)");
  }
  d->stack = accum == SG_ACCUM_STACK;
  d->RU = d->R * upl;
  d->saves_z = d->order == SG_ORDER_FUSED && fused_saves_z(p);
  d->outw = d->stack ? d->RU : upl;
  d->Up = (upl + 3) / 4 * 4;
  d->outw_p = d->stack ? d->R * d->Up : d->Up;
  const int64_t used = d->R * in_dim + d->R;
  // row pitch of the R-expanded matrices (Zext / dZ): every level block of a row is gathered / scattered as one burst of
  // 4*D bytes, so when that burst is a multiple of 256 B the pitch is rounded to 256 B as well -- otherwise (pitch =
  // R*D + R rounded to 16 B, round 1) every row starts 64 B off a cache-line boundary: 9 instead of 8 lines per 1 KiB
  // burst and 3 instead of 2 per 256-B column slice (measured: the dZ gather 1.27 ms vs 0.84 ms for the aligned H gather)
  const int64_t align = (in_dim % 64 == 0) ? 64 : 4;
  d->ld = (used + align - 1) / align * align;
  return SG_OK;
}

// workspace layout, shared by the size query and the launchers
struct Layout {
  size_t wpack, bpack, a, b, c, scratch, scratch_bytes, total;
};
Layout make_layout(const Dims& d, bool backward) {
  Layout L{};
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
  const size_t f = sizeof(float);
  size_t sc = 0;
  if (d.order == SG_ORDER_FUSED) {
    sc = sg_agg_fused_workspace_bytes(d.R);
    if (backward) {
      L.a = take(d.n_dst * d.outw_p * f);                                            // dpre (levels padded to Up floats)
      if (!d.saves_z) {
        L.b = take(d.n_src * d.R * d.Up * f);                                        // dH (written by the fused data gradient)
        L.wpack = take(d.n_src * d.D * f);                                           // stand-in dx when the caller wants only dW
      }
      L.c = take((d.R * d.Up * d.D + d.RU) * f);                                     // dWcat (R Up, D) | dbcat   (or dWext (U, R D) | dbcat)
      const size_t gw = d.saves_z ? (d.stack ? sg_gemm_f32_workspace_bytes(d.U, d.D, d.n_dst, 1)
                                             : sg_gemm_f32_workspace_bytes(d.U, d.R * d.D, d.n_dst, 1))
                                  : sg_gemm_f32_workspace_bytes(d.R * d.Up, d.D, d.n_src, 1);
      sc = max2(sc, max2(gw, static_cast<size_t>(kDbParts) * d.RU * f));
    }
  } else if (d.order == SG_ORDER_TRANSFORM_FIRST) {
    L.wpack = take(d.RU * d.D * f);
    L.bpack = take(d.RU * f);
    if (!backward) {
      L.a = take(d.n_src * d.RU * f);                                              // H
      sc = max2(sg_gemm_f32_workspace_bytes(d.n_src, d.RU, d.D, 0),
                sg_seg_weighted_pool_workspace_bytes(1, d.stack ? d.n_dst * d.R : d.n_dst, d.nnz, d.U));
    } else {
      L.a = take(d.n_dst * d.outw * f);                                            // dpre
      L.b = take(d.n_src * d.RU * f);                                              // dH
      L.c = take((d.RU * d.D + d.RU) * f);                                         // dWcat | dbcat
      sc = max2(sg_seg_weighted_pool_workspace_bytes(1, d.n_src * d.R, d.nnz, d.U),
                max2(sg_gemm_f32_workspace_bytes(d.n_src, d.D, d.RU, 0),
                     max2(sg_gemm_f32_workspace_bytes(d.RU, d.D, d.n_src, 1), sg_colsum_workspace_bytes(d.n_src, d.RU))));
    }
  } else {
    L.wpack = take(d.outw * d.ld * f);                                             // Wext
    if (!backward) {
      sc = max2(sg_seg_weighted_pool_workspace_bytes(1, d.n_dst * d.R, d.nnz, d.D),
                sg_gemm_f32_workspace_bytes(d.n_dst, d.outw, d.ld, 0));
    } else {
      L.a = take(d.n_dst * d.outw * f);                                            // dpre
      L.b = take(d.n_dst * d.ld * f);                                              // dZ
      L.c = take(d.outw * d.ld * f);                                               // dWext
      sc = max2(sg_seg_weighted_pool_workspace_bytes(1, d.n_src, d.nnz, d.D),
                max2(sg_gemm_f32_workspace_bytes(d.n_dst, d.ld, d.outw, 0), sg_gemm_f32_workspace_bytes(d.outw, d.ld, d.n_dst, 1)));
    }
  }
  L.scratch = take(sc + 64);
  L.scratch_bytes = sc + 64;
  L.total = off + 256;   // slack for aligning the caller's base pointer
  return L;
}

int fill_table(PtrTable* t, const float* const* host, int R, bool required, const char* what) {
  for (int r = 0; r < SG_MAX_LINKS; ++r) t->p[r] = nullptr;
  if (!host) return required ? fail(SG_ERR_INVALID, "%s is null", what) : SG_OK;
  for (int r = 0; r < R; ++r) {
    if (required && !host[r]) return fail(SG_ERR_INVALID, "%s[%d] is null", what, r);
    t->p[r] = host[r];
  }
  return SG_OK;
}

// degenerate shapes (no destination / source nodes): the parameter gradients are exact zeros
int zero_param_grads(const MutPtrTable& dw, const MutPtrTable& db, const Dims& d, hipStream_t st) {
  for (int r = 0; r < d.R; ++r) {
    if (dw.p[r] && hipMemsetAsync(dw.p[r], 0, d.U * d.D * sizeof(float), st) != hipSuccess) return fail(SG_ERR_HIP, "memset");
    if (db.p[r] && hipMemsetAsync(db.p[r], 0, d.U * sizeof(float), st) != hipSuccess) return fail(SG_ERR_HIP, "memset");
  }
  return SG_OK;
}

// one gather of the plan: as source-range phases when the plan carries them for this view and the source matrix is
// cache-resident but larger than the L2s; SG_GATHER_PHASES=0 keeps the single launch
// the ONE eligibility rule of source-range phases (also exported: sg_multilink_agg_phased_view): feature width of at
// least one 256-byte column slice and a gathered matrix that is cache-resident but larger than the L2s
bool phases_on() {
  static const int on = [] { const char* e = getenv("SG_GATHER_PHASES"); return e ? atoi(e) : 1; }();
  return on != 0;
}
bool phase_shape_ok(int64_t C, int64_t src_bytes) { return C >= 64 && src_bytes >= (24ll << 20) && src_bytes <= (256ll << 20); }

int gather_view(const sg_multilink_plan* plan, int view, float* dst, int64_t dst_group, int64_t dst_ld, const float* src,
                int64_t src_group, int64_t src_ld, const float* w, const int32_t* idx, const int32_t* indptr,
                int64_t seg_num, int64_t nnz, int64_t C, int act, float slope, void* scratch, size_t scratch_bytes,
                void* stream, int64_t src_bytes) {
  const bool phases_on = sg::phases_on();
  // `phases` exists only in callers built against the header that has it (struct_bytes says so; 0 / older = absent)
  const bool has_phases = plan->struct_bytes >= static_cast<int32_t>(offsetof(sg_multilink_plan, phases) + sizeof(plan->phases));
  const sg_gather_phases* ph = &plan->phases[view];
  if (phases_on && has_phases && ph->num_phases == 2 && ph->idx && phase_shape_ok(C, src_bytes))
    return sg_seg_gather_sum_phased_hip(dst, dst_group, dst_ld, src, src_group, src_ld, w, ph, seg_num, C, SG_REQ_WRITE,
                                        act, slope, scratch, scratch_bytes, stream, src_bytes);
  return sg_seg_gather_sum_hinted_hip(dst, dst_group, dst_ld, src, src_group, src_ld, w, idx, indptr, seg_num, nnz, C,
                                      SG_REQ_WRITE, act, slope, scratch, scratch_bytes, stream, src_bytes);
}

#define SG_TRY(expr)          \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != SG_OK) return rc_; \
  } while (0)

}  // namespace
}  // namespace sg

using namespace sg;

// Which gather view (SG_VIEW_*) sg_multilink_agg_{fwd,bwd}_hip would issue as source-range phases for these sizes, or -1
// when it would not (width / footprint outside the rule, SG_GATHER_PHASES=0): callers build phases for that view only.
SG_API int sg_multilink_agg_phased_view(const sg_multilink_plan* plan, int64_t in_dim, int64_t units_per_level, int order,
                                        int accum, int backward) {
  Dims d;
  const int rc = make_dims(&d, plan, in_dim, units_per_level, order, accum);
  if (rc != SG_OK) return rc;
  if (!sg::phases_on() || d.n_dst == 0 || d.n_src == 0 || d.order == SG_ORDER_FUSED) return -1;
  int view;
  int64_t C, src_bytes;
  if (d.order == SG_ORDER_TRANSFORM_FIRST) {
    C = d.U;
    if (!backward) { view = d.stack ? SG_VIEW_C_Q_C : SG_VIEW_C_Q_D; src_bytes = d.n_src * d.RU * 4; }
    else { view = d.stack ? SG_VIEW_T_Q_T : SG_VIEW_T_IDX_T; src_bytes = d.n_dst * d.outw * 4; }
  } else {
    C = d.D;
    if (!backward) { view = SG_VIEW_C_IDX_C; src_bytes = d.n_src * d.D * 4; }
    else { view = SG_VIEW_T_Q_S; src_bytes = d.n_dst * d.ld * 4; }
  }
  return phase_shape_ok(C, src_bytes) ? view : -1;
}

SG_API int sg_multilink_agg_resolve_order(const sg_multilink_plan* plan, int order) {
  if (!plan) return fail(SG_ERR_INVALID, "plan is null");
  return resolve_order(plan, order);
}

SG_API int sg_multilink_agg_resolve_order2(const sg_multilink_plan* plan, int order, int64_t in_dim, int64_t units_per_level,
                                           int accum) {
  if (!plan) return fail(SG_ERR_INVALID, "plan is null");
  if (order < SG_ORDER_AUTO || order > SG_ORDER_FUSED) return fail(SG_ERR_INVALID, "order %d", order);
  return resolve_order2(plan, order, in_dim, units_per_level, accum);
}

SG_API size_t sg_multilink_agg_saved_bytes(const sg_multilink_plan* plan, int64_t in_dim, int64_t units_per_level,
                                           int order, int accum) {
  Dims d;
  if (make_dims(&d, plan, in_dim, units_per_level, order, accum) != SG_OK) return 0;
  if (d.saves_z) return static_cast<size_t>(d.n_dst) * d.R * d.D * sizeof(float);       // fused, destination side smaller: Z
  return d.order == SG_ORDER_AGGREGATE_FIRST ? static_cast<size_t>(d.n_dst) * d.ld * sizeof(float) : 0;
}

SG_API size_t sg_multilink_agg_workspace_bytes(const sg_multilink_plan* plan, int64_t in_dim, int64_t units_per_level,
                                               int order, int accum, int backward) {
  Dims d;
  if (make_dims(&d, plan, in_dim, units_per_level, order, accum) != SG_OK) return 0;
  return make_layout(d, backward != 0).total;
}

SG_API int sg_multilink_agg_fwd_hip(float* out, void* saved, const float* x, const float* const* weights,
                                    const float* const* biases, const sg_multilink_plan* plan, int64_t in_dim,
                                    int64_t units_per_level, int order, int accum, int act, float slope,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  Dims d;
  SG_TRY(make_dims(&d, plan, in_dim, units_per_level, order, accum));
  if (d.n_dst == 0) return SG_OK;
  if (!out || (!x && d.n_src > 0)) return fail(SG_ERR_INVALID, "out / x is null");
  const Layout L = make_layout(d, false);
  if (!workspace || workspace_bytes < L.total)
    return fail(SG_ERR_WORKSPACE, "multilink fwd workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
  PtrTable w, b;
  SG_TRY(fill_table(&w, weights, d.R, true, "weights"));
  SG_TRY(fill_table(&b, biases, d.R, false, "biases"));
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  void* scratch = base + L.scratch;

  if (d.order == SG_ORDER_FUSED) {
    if (d.n_src == 0) return fail(SG_ERR_INVALID, "fused aggregation with edges but no source rows");
    if (biases && !plan->rowsum) return fail(SG_ERR_INVALID, "the fused order needs plan->rowsum for the bias term");
    const sg_fused_plan& fp = plan->fused[0];
    if (d.saves_z && !saved) return fail(SG_ERR_INVALID, "the fused order needs the `saved` buffer here (sg_multilink_agg_saved_bytes)");
    return sg_agg_fused2_hip(out, d.outw, d.saves_z ? static_cast<float*>(saved) : nullptr, d.R * d.D, x, d.D, 0, weights, d.D, 0, d.D,
                             biases, plan->rowsum, fp.f_ptr, fp.f_idx, fp.f_w, fp.tile_order, d.n_dst, d.n_src, d.R, d.nnz, d.D, d.U,
                             accum, act, slope, fused_nt(), scratch, L.scratch_bytes, stream);
  }

  if (d.order == SG_ORDER_TRANSFORM_FIRST) {
    float* wcat = reinterpret_cast<float*>(base + L.wpack);
    float* bcat = reinterpret_cast<float*>(base + L.bpack);
    float* h = reinterpret_cast<float*>(base + L.a);
    hipLaunchKernelGGL(pack_cat_kernel, dim3(blocks_for(d.RU * d.D)), dim3(256), 0, st, wcat, bcat, w, b, d.R,
                       static_cast<int>(d.U), static_cast<int>(d.D));
    SG_TRY(check_launch("pack_cat_kernel"));
    if (d.n_src > 0)
      SG_TRY(sg_gemm_f32_hip(h, d.RU, x, d.D, 0, wcat, d.D, 1, d.n_src, d.RU, d.D, bcat, SG_ACT_NONE, 0.f, 0, scratch,
                             L.scratch_bytes, stream));
    if (!d.stack)
      return gather_view(plan, SG_VIEW_C_Q_D, out, 1, d.U, h, d.R, d.RU, plan->c_w, plan->c_q, plan->d_indptr, d.n_dst, d.nnz,
                         d.U, act, slope, scratch, L.scratch_bytes, stream, d.n_src * d.RU * 4);
    return gather_view(plan, SG_VIEW_C_Q_C, out, d.R, d.RU, h, d.R, d.RU, plan->c_w, plan->c_q, plan->c_indptr, d.n_dst * d.R,
                       d.nnz, d.U, act, slope, scratch, L.scratch_bytes, stream, d.n_src * d.RU * 4);
  }

  if (!saved) return fail(SG_ERR_INVALID, "aggregate-first needs the `saved` buffer (sg_multilink_agg_saved_bytes)");
  if (!plan->rowsum) return fail(SG_ERR_INVALID, "aggregate-first needs plan->rowsum");
  float* wext = reinterpret_cast<float*>(base + L.wpack);
  float* zext = static_cast<float*>(saved);
  hipLaunchKernelGGL(pack_ext_kernel, dim3(blocks_for(d.outw * d.ld)), dim3(256), 0, st, wext, w, b, d.R,
                     static_cast<int>(d.U), static_cast<int>(d.D), static_cast<int>(d.ld), d.stack);
  SG_TRY(check_launch("pack_ext_kernel"));
  SG_TRY(gather_view(plan, SG_VIEW_C_IDX_C, zext, d.R, d.ld, x, 1, d.D, plan->c_w, plan->c_idx, plan->c_indptr, d.n_dst * d.R,
                     d.nnz, d.D, SG_ACT_NONE, 0.f, scratch, L.scratch_bytes, stream, d.n_src * d.D * 4));
  hipLaunchKernelGGL(fill_rowsum_kernel, dim3(blocks_for(d.n_dst * (d.ld - d.R * d.D))), dim3(256), 0, st, zext,
                     plan->rowsum, static_cast<long long>(d.n_dst), d.R, static_cast<int>(d.D), static_cast<int>(d.ld));
  SG_TRY(check_launch("fill_rowsum_kernel"));
  return sg_gemm_f32_hip(out, d.outw, zext, d.ld, 0, wext, d.ld, 1, d.n_dst, d.outw, d.ld, nullptr, act, slope, 0,
                         scratch, L.scratch_bytes, stream);
}

SG_API int sg_multilink_agg_bwd_hip(float* dx, float* const* dweights, float* const* dbiases, const float* dout,
                                    const float* out, const void* saved, const float* x, const float* const* weights,
                                    const sg_multilink_plan* plan, int64_t in_dim, int64_t units_per_level, int order,
                                    int accum, int act, float slope, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  Dims d;
  SG_TRY(make_dims(&d, plan, in_dim, units_per_level, order, accum));
  const Layout L = make_layout(d, true);
  if (!workspace || workspace_bytes < L.total)
    return fail(SG_ERR_WORKSPACE, "multilink bwd workspace too small: need %zu bytes, got %zu", L.total, workspace_bytes);
  if (d.n_dst > 0 && (!dout || (act != SG_ACT_NONE && !out))) return fail(SG_ERR_INVALID, "dout / out is null");
  PtrTable w;
  SG_TRY(fill_table(&w, weights, d.R, true, "weights"));
  MutPtrTable dw, db;
  bool want_w = false, want_b = false;
  for (int r = 0; r < SG_MAX_LINKS; ++r) {
    dw.p[r] = (dweights && r < d.R) ? dweights[r] : nullptr;
    db.p[r] = (dbiases && r < d.R) ? dbiases[r] : nullptr;
    want_w = want_w || dw.p[r];
    want_b = want_b || db.p[r];
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  void* scratch = base + L.scratch;
  float* dpre_buf = reinterpret_cast<float*>(base + L.a);
  const float* dpre = dout;
  const bool repitch = d.order == SG_ORDER_FUSED && d.Up != d.U;      // the fused branch applies act' while it pads the levels
  if (act != SG_ACT_NONE && d.n_dst > 0 && !repitch) {
    SG_TRY(sg_act_bwd_hip(dpre_buf, dout, out, d.n_dst * d.outw, act, slope, stream));
    dpre = dpre_buf;
  }
  PtrTable nob;
  for (int r = 0; r < SG_MAX_LINKS; ++r) nob.p[r] = nullptr;

  if (d.order == SG_ORDER_FUSED) {
    // dx[n] = sum_r (A_r^T dpre_r)[n] W_r in one kernel over the transposed plan (dpre_r = dpre for 'sum', column block r of it for
    // 'stack'), which also leaves dH = [A_r^T dpre_r]_r for the weight gradient dW_r = dH_r^T x; the bias gradient is
    // db_r = sum_i rowsum[i, r] dpre_r[i] (= the column sums of dH_r).  Every level of dpre / dH is Up = U rounded up to a multiple
    // of 4 floats wide (one float4 per gather lane); the padding columns are zero.
    float* dh = reinterpret_cast<float*>(base + L.b);
    float* dwcat = reinterpret_cast<float*>(base + L.c);
    float* dbcat = dwcat + d.R * d.Up * d.D;
    if (d.n_src == 0 || d.n_dst == 0) return fail(SG_ERR_INVALID, "fused aggregation with edges but no rows");
    if (want_b && !plan->rowsum) return fail(SG_ERR_INVALID, "the fused order needs plan->rowsum for the bias gradient");
    if (d.Up != d.U) {         // re-pitch (and apply act' in the same pass): the generic pass above wrote nothing usable
      hipLaunchKernelGGL(act_bwd_pitched_kernel, dim3(blocks_for(d.n_dst * d.outw_p)), dim3(256), 0, st, dpre_buf, static_cast<long long>(d.outw_p),
                         dout, out, static_cast<long long>(d.n_dst), static_cast<int>(d.outw), static_cast<int>(d.U), static_cast<int>(d.Up),
                         act, slope);
      SG_TRY(check_launch("act_bwd_pitched_kernel"));
      dpre = dpre_buf;
    }
    const int64_t lstride = d.stack ? d.Up : 0;
    auto bias_grad = [&]() -> int {
      float* part = reinterpret_cast<float*>(scratch);
      const int P = static_cast<int>(d.n_dst < kDbParts ? d.n_dst : kDbParts);
      const dim3 grid(static_cast<unsigned>(P), static_cast<unsigned>((d.U + 255) / 256));
      if (d.R <= 16)
        hipLaunchKernelGGL(bias_grad_partial_kernel<16>, grid, dim3(1024), 0, st, part, plan->rowsum, dpre,
                           static_cast<long long>(d.n_dst), d.R, static_cast<int>(d.U), static_cast<long long>(d.outw_p), static_cast<int>(lstride));
      else
        hipLaunchKernelGGL(bias_grad_partial_kernel<SG_MAX_LINKS>, grid, dim3(1024), 0, st, part, plan->rowsum, dpre,
                           static_cast<long long>(d.n_dst), d.R, static_cast<int>(d.U), static_cast<long long>(d.outw_p), static_cast<int>(lstride));
      hipLaunchKernelGGL(bias_grad_final_kernel, dim3(static_cast<unsigned>((d.RU + 63) / 64)), dim3(64), 0, st, dbcat, part, P, static_cast<int>(d.RU));
      return check_launch("bias_grad kernels");
    };
    MutPtrTable none;
    for (int r = 0; r < SG_MAX_LINKS; ++r) none.p[r] = nullptr;
    const sg_fused_plan& fp = plan->fused[1];
    if (d.saves_z) {      // the forward kept Z = [A_r x]_r: dW_r = dpre_r^T Z_r; the data gradient writes nothing but dx
      if (want_w && !saved) return fail(SG_ERR_INVALID, "fused backward needs the `saved` buffer of the forward");
      if (dx)
        SG_TRY(sg_agg_fused2_hip(dx, d.D, nullptr, 0, dpre, d.outw_p, lstride, weights, d.D, 1, d.U, nullptr, nullptr, fp.f_ptr, fp.f_idx,
                                 fp.f_w, fp.tile_order, d.n_src, d.n_dst, d.R, d.nnz, d.Up, d.D, SG_ACCUM_SUM, SG_ACT_NONE, 0.f, fused_nt(),
                                 scratch, L.scratch_bytes, stream));
      const float* z = static_cast<const float*>(saved);
      if (want_w && !d.stack) {        // 'sum': ONE product dWext (U, R D) = dpre^T Z
        SG_TRY(sg_gemm_f32_hip(dwcat, d.R * d.D, dpre, d.outw_p, 1, z, d.R * d.D, 0, d.U, d.R * d.D, d.n_dst, nullptr, SG_ACT_NONE, 0.f, 0,
                               scratch, L.scratch_bytes, stream));
      } else if (want_w) {             // 'stack': level r contracts its own column blocks, dW_r (U, D) = dpre_r^T Z_r
        for (int r = 0; r < d.R; ++r)
          SG_TRY(sg_gemm_f32_hip(dwcat + static_cast<int64_t>(r) * d.U * d.D, d.D, dpre + r * d.Up, d.outw_p, 1, z + r * d.D, d.R * d.D, 0,
                                 d.U, d.D, d.n_dst, nullptr, SG_ACT_NONE, 0.f, 0, scratch, L.scratch_bytes, stream));
      }
      if (want_b) SG_TRY(bias_grad());
      if (want_w && !d.stack) {
        hipLaunchKernelGGL(unpack_ext_kernel, dim3(blocks_for(d.RU * d.D)), dim3(256), 0, st, dw, none, dwcat, d.R,
                           static_cast<int>(d.U), static_cast<int>(d.D), static_cast<int>(d.R * d.D), 0);
        SG_TRY(check_launch("unpack_ext_kernel"));
      }
      if ((want_w && d.stack) || want_b) {
        hipLaunchKernelGGL(unpack_cat_kernel, dim3(blocks_for(d.RU * d.D)), dim3(256), 0, st, (want_w && d.stack) ? dw : none,
                           want_b ? db : none, (want_w && d.stack) ? dwcat : static_cast<const float*>(nullptr),
                           want_b ? dbcat : static_cast<const float*>(nullptr), d.R, static_cast<int>(d.U), static_cast<int>(d.D),
                           static_cast<int>(d.U));
        SG_TRY(check_launch("unpack_cat_kernel"));
      }
      return SG_OK;
    }
    if (dx || want_w) {      // (a caller that wants only dW still runs the data-gradient kernel: it is what writes dH)
      float* dxo = dx ? dx : reinterpret_cast<float*>(base + L.wpack);
      SG_TRY(sg_agg_fused2_hip(dxo, d.D, want_w ? dh : nullptr, d.R * d.Up, dpre, d.outw_p, lstride, weights, d.D, 1, d.U, nullptr, nullptr,
                               fp.f_ptr, fp.f_idx, fp.f_w, fp.tile_order, d.n_src, d.n_dst, d.R, d.nnz, d.Up, d.D, SG_ACCUM_SUM, SG_ACT_NONE,
                               0.f, fused_nt(), scratch, L.scratch_bytes, stream));
    }
    if (want_w) {
      if (!x) return fail(SG_ERR_INVALID, "x is null");
      SG_TRY(sg_gemm_f32_hip(dwcat, d.D, dh, d.R * d.Up, 1, x, d.D, 0, d.R * d.Up, d.D, d.n_src, nullptr, SG_ACT_NONE, 0.f, 0,
                             scratch, L.scratch_bytes, stream));
    }
    if (want_b) SG_TRY(bias_grad());
    if (want_w || want_b) {
      hipLaunchKernelGGL(unpack_cat_kernel, dim3(blocks_for(d.RU * d.D)), dim3(256), 0, st, dw, db,
                         want_w ? dwcat : static_cast<const float*>(nullptr),
                         want_b ? dbcat : static_cast<const float*>(nullptr), d.R, static_cast<int>(d.U),
                         static_cast<int>(d.D), static_cast<int>(d.Up));
      SG_TRY(check_launch("unpack_cat_kernel"));
    }
    return SG_OK;
  }

  if (d.order == SG_ORDER_TRANSFORM_FIRST) {
    float* wcat = reinterpret_cast<float*>(base + L.wpack);
    float* dh = reinterpret_cast<float*>(base + L.b);
    float* dwcat = reinterpret_cast<float*>(base + L.c);
    float* dbcat = dwcat + d.RU * d.D;
    if (d.n_src == 0 || d.n_dst == 0) {
      if (dx && d.n_src > 0 && hipMemsetAsync(dx, 0, d.n_src * d.D * sizeof(float), st) != hipSuccess)
        return fail(SG_ERR_HIP, "memset");
      return zero_param_grads(dw, db, d, st);
    }
    // dH[(n, r), :] = sum over the transposed plan of t_w * dpre[dest (, level r block)]
    if (!d.stack)
      SG_TRY(gather_view(plan, SG_VIEW_T_IDX_T, dh, d.R, d.RU, dpre, 1, d.U, plan->t_w, plan->t_idx, plan->t_indptr,
                         d.n_src * d.R, d.nnz, d.U, SG_ACT_NONE, 0.f, scratch, L.scratch_bytes, stream,
                         d.n_dst * d.outw * 4));
    else
      SG_TRY(gather_view(plan, SG_VIEW_T_Q_T, dh, d.R, d.RU, dpre, d.R, d.RU, plan->t_w, plan->t_q, plan->t_indptr,
                         d.n_src * d.R, d.nnz, d.U, SG_ACT_NONE, 0.f, scratch, L.scratch_bytes, stream,
                         d.n_dst * d.outw * 4));
    if (dx) {
      hipLaunchKernelGGL(pack_cat_kernel, dim3(blocks_for(d.RU * d.D)), dim3(256), 0, st, wcat,
                         static_cast<float*>(nullptr), w, nob, d.R, static_cast<int>(d.U), static_cast<int>(d.D));
      SG_TRY(check_launch("pack_cat_kernel"));
      SG_TRY(sg_gemm_f32_hip(dx, d.D, dh, d.RU, 0, wcat, d.D, 0, d.n_src, d.D, d.RU, nullptr, SG_ACT_NONE, 0.f, 0,
                             scratch, L.scratch_bytes, stream));
    }
    if (want_w) {
      if (!x) return fail(SG_ERR_INVALID, "x is null");
      SG_TRY(sg_gemm_f32_hip(dwcat, d.D, dh, d.RU, 1, x, d.D, 0, d.RU, d.D, d.n_src, nullptr, SG_ACT_NONE, 0.f, 0,
                             scratch, L.scratch_bytes, stream));
    }
    if (want_b)
      SG_TRY(sg_colsum_hip(dbcat, dh, d.RU, d.n_src, d.RU, SG_REQ_WRITE, scratch, L.scratch_bytes, stream));
    if (want_w || want_b) {
      hipLaunchKernelGGL(unpack_cat_kernel, dim3(blocks_for(d.RU * d.D)), dim3(256), 0, st, dw, db,
                         want_w ? dwcat : static_cast<const float*>(nullptr),
                         want_b ? dbcat : static_cast<const float*>(nullptr), d.R, static_cast<int>(d.U),
                         static_cast<int>(d.D), static_cast<int>(d.U));
      SG_TRY(check_launch("unpack_cat_kernel"));
    }
    return SG_OK;
  }

  if (!saved && d.n_dst > 0) return fail(SG_ERR_INVALID, "aggregate-first backward needs the `saved` buffer of the forward");
  float* wext = reinterpret_cast<float*>(base + L.wpack);
  float* dz = reinterpret_cast<float*>(base + L.b);
  float* dwext = reinterpret_cast<float*>(base + L.c);
  const float* zext = static_cast<const float*>(saved);
  if (dx && d.n_src > 0) {
    hipLaunchKernelGGL(pack_ext_kernel, dim3(blocks_for(d.outw * d.ld)), dim3(256), 0, st, wext, w, nob, d.R,
                       static_cast<int>(d.U), static_cast<int>(d.D), static_cast<int>(d.ld), d.stack);
    SG_TRY(check_launch("pack_ext_kernel"));
    if (d.n_dst > 0)
      SG_TRY(sg_gemm_f32_hip(dz, d.ld, dpre, d.outw, 0, wext, d.ld, 0, d.n_dst, d.ld, d.outw, nullptr, SG_ACT_NONE, 0.f,
                             0, scratch, L.scratch_bytes, stream));
    SG_TRY(gather_view(plan, SG_VIEW_T_Q_S, dx, 1, d.D, dz, d.R, d.ld, plan->t_w, plan->t_q, plan->s_indptr, d.n_src, d.nnz,
                       d.D, SG_ACT_NONE, 0.f, scratch, L.scratch_bytes, stream, d.n_dst * d.ld * 4));
  }
  if ((want_w || want_b) && d.n_dst == 0) return zero_param_grads(dw, db, d, st);
  if (want_w || want_b) {
    SG_TRY(sg_gemm_f32_hip(dwext, d.ld, dpre, d.outw, 1, zext, d.ld, 0, d.outw, d.ld, d.n_dst, nullptr, SG_ACT_NONE, 0.f,
                           0, scratch, L.scratch_bytes, stream));
    hipLaunchKernelGGL(unpack_ext_kernel, dim3(blocks_for(d.RU * d.D)), dim3(256), 0, st, dw, db, dwext, d.R,
                       static_cast<int>(d.U), static_cast<int>(d.D), static_cast<int>(d.ld), d.stack);
    SG_TRY(check_launch("unpack_ext_kernel"));
  }
  return SG_OK;
}
