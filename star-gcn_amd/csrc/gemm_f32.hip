// gemm_f32.hip -- fp32 MFMA GEMM with fused bias / activation epilogue for gfx950.
//
// The per-rating-level dense mix of the multi-link graph conv (reference aggregators.py:141-145 runs R
// separate MXNet FullyConnected calls BEFORE aggregation) is done here as ONE contraction AFTER aggregation:
//   pre[N_dst, U] = Zext[N_dst, R*D + pad] * Wext[U, R*D + pad]^T
// and the same kernel serves the output Dense (layers.py:183), the decoder embed_maps
// (STAR-GCN.py:241-245) and the rating projections (:257), plus all their gradients (NN / TN forms).
//
//   C[M,N] = act( opA(A)[M,K] * opB(B)[K,N] + bias[N] (+ C) ),   row-major.
//
// CDNA4 mapping: v_mfma_f32_32x32x2_f32 (exact fp32 products and fp32 accumulation -- gfx950 has no
// TF32/xf32 path, and the 1e-5 parity budget rules out bf16), 128x128x32 block tile, 256 threads = 4 waves
// in a 2x2 arrangement, each wave a 64x64 sub-tile = 2x2 MFMA tiles (64 accumulator registers).  Both
// operands are staged in LDS K-MAJOR ([k][m] / [k][n]) so that an MFMA operand read is 32 consecutive
// floats per half-wave (bank-conflict free) whatever the global layout; K-contiguous global operands are
// transposed on the way in (LDS leading dim 129 makes those b32 stores conflict-free), M/N-contiguous ones
// are copied with b128 stores (leading dim 132).  Two register staging sets keep the global loads of K tiles t+1
// and t+2 in flight while tile t is multiplied; a staged tile is written to the other LDS buffer after the MFMAs of
// the tile before it: one barrier per K tile.  A 64-row tile variant (32x64 per wave) fills the chip better when M*N is small.  Workgroups are remapped so
// that the tiles sharing an A row-panel run on the same XCD (private 4 MiB L2 each).  Small-tile-count /
// large-K problems (weight gradients: K = #nodes) use deterministic split-K through a workspace.
#include <atomic>
#include <cmath>
#include <cstdlib>

#include "common.hpp"

#include <mutex>
#include <vector>

namespace sg {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int kThreads = 256;
constexpr int LD_T = BM + 1;  // leading dim when the operand is transposed on the way into LDS
constexpr int LD_D = BM + 4;  // leading dim for direct (b128) stores
constexpr int LD_MAX = LD_D;

using f32x16 = __attribute__((ext_vector_type(16))) float;

#ifndef SG_GEMM_DEFAULT_BF16X6
#define SG_GEMM_DEFAULT_BF16X6 0
#endif

struct GemmArgs {
  float* C;
  const float* A;
  const float* B;
  const float* bias;
  float* ws;  // split-K partials [splits][M][N]
  long long lda, ldb, ldc;
  int M, N, K;
  int act;
  float slope;
  int accumulate;
  int splits, tiles_per_split;
  int tiles_m, tiles_n;
  int vecA, vecB;  // 16-byte vector loads allowed
};

static std::atomic<int> g_backend_override{-1};
static inline int g_backend_override_load() { return g_backend_override.load(std::memory_order_relaxed); }
void launch_gemm_bf16x6(const GemmArgs& g, bool transA, bool transB, hipStream_t st);   // gemm_bf16x6.hip
void launch_gemm_x6v2(const GemmArgs& g, bool transA, bool transB, int n_cus, hipStream_t st);   // gemm_x6v2.hip
bool x6v2_supported(const GemmArgs& g, bool transA, bool transB);
size_t f16x3_plane_bytes(long long M, long long N, long long K);                                  // gemm_f16x3.hip
int launch_gemm_f16x3(const GemmArgs& g, bool transA, bool transB, char* scratch, hipStream_t st, bool* reduced,
                      int* splits_used);

#ifndef SG_GEMM_DEFAULT_BACKEND
#define SG_GEMM_DEFAULT_BACKEND 3      // 0 exact-fp32 MFMA (this file), 1 bf16x6, 2 x6v2 (wave-specialised bf16x6),
#endif                                 // 3 f16x3 (pre-split f16 planes, three MFMAs per product)
static int gemm_backend() {            // SG_GEMM_BACKEND = fp32 | bf16x6 | x6v2 | f16x3, read once
  static const int v = [] {
    const char* e = getenv("SG_GEMM_BACKEND");
    if (!e) return SG_GEMM_DEFAULT_BACKEND;
    if (e[0] == 'f' && e[1] == '1') return 3;
    if (e[0] == 'x') return 2;
    if (e[0] == 'b') return 1;
    return 0;
  }();
  return v;
}
// Backend for a product, from what BOTH the workspace query and the launch know (sizes, layout, the override): the f16x3
// planes need workspace, so the two must agree.  f16x3 pays one conversion pass per operand: it takes over from K = 96 on
// (below that the bf16x6 kernel's in-loop split, or the exact kernel, has less to amortise).
static int backend_for(int64_t M, int64_t N, int64_t K, int transA) {
  int backend = g_backend_override_load();
  const bool forced = backend >= 0;
  if (!forced) backend = gemm_backend();
  if (K < 1) return 0;
  // ... and needs enough output tiles to fill the chip with its non-persistent workgroups: a 256 x 256 weight gradient over
  // a few thousand rows (4 tiles, everything in split-K slices) stays on the persistent x6v2 kernel
  // (a 64 MB operand against a <= 256-wide one is taken by the in-kernel-split form whatever the tile count: 95 vs 89)
  if (backend == 3 && !forced && (K < 96 || (transA && M <= 64) ||
                                  (((M + 127) / 128) * ((N + 127) / 128) < 16 && (M > N ? M : N) * K < (16ll << 20))))
    backend = 2;
  if (backend == 2 && !forced && (K <= 64 || (transA && M <= 64))) backend = 0;
  return backend;
}
static int cu_count() {
  static const int n = [] {
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0)
      c = 256;
    return c;
  }();
  return n;
}

// ---- measurement aid (bench.py `dense_roofline`): when enabled, every sg_gemm_f32_hip call -- direct or from inside the
// fused aggregator entries -- is bracketed with HIP events on its own stream (conversion passes and split-K reduce
// included).  Off by default.
struct GemmProfRecord {
  hipEvent_t a, b;
  int64_t M, N, K;
  int backend;
};
static std::mutex g_gprof_mu;
static bool g_gprof_on = false;
static std::vector<GemmProfRecord> g_gprof;
static long gprof_begin(hipStream_t st, int64_t M, int64_t N, int64_t K, int backend) {
  std::lock_guard<std::mutex> lk(g_gprof_mu);
  if (!g_gprof_on) return -1;
  GemmProfRecord r{};
  r.M = M; r.N = N; r.K = K; r.backend = backend;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
  (void)hipEventRecord(r.a, st);
  g_gprof.push_back(r);
  return static_cast<long>(g_gprof.size()) - 1;
}
static void gprof_end(long rec, hipStream_t st) {
  if (rec < 0) return;
  std::lock_guard<std::mutex> lk(g_gprof_mu);
  if (rec < static_cast<long>(g_gprof.size())) (void)hipEventRecord(g_gprof[rec].b, st);
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
  switch (act) {
    case SG_ACT_LEAKY: return v > 0.f ? v : slope * v;
    case SG_ACT_RELU: return v > 0.f ? v : 0.f;
    case SG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case SG_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// Loads the (EXT x BK) slab of a K-CONTIGUOUS operand (element (r,k) at p[r*ld + k]) for this thread:
// EXT/32 float4 = rows (t/8 + 32*i), k = 4*(t%8)..+3.
template <int EXT>
__device__ __forceinline__ void gload_kcontig(float (&r)[4][4], const float* __restrict__ p, long long ld, int row0,
                                              int k0, int R, int K, int vec, int t) {
  const int kc = (t & 7) * 4;
  const int rr = t >> 3;
#pragma unroll
  for (int i = 0; i < EXT / 32; ++i) {
    const int row = row0 + rr + 32 * i;
    const int k = k0 + kc;
    if (row < R && vec && k + 3 < K) {
      const float4 v = *reinterpret_cast<const float4*>(p + static_cast<long long>(row) * ld + k);
      r[i][0] = v.x; r[i][1] = v.y; r[i][2] = v.z; r[i][3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        r[i][j] = (row < R && k + j < K) ? p[static_cast<long long>(row) * ld + k + j] : 0.f;
    }
  }
}
template <int EXT>
__device__ __forceinline__ void sstore_kcontig(float* __restrict__ s, const float (&r)[4][4], int t) {
  const int kc = (t & 7) * 4;
  const int rr = t >> 3;
#pragma unroll
  for (int i = 0; i < EXT / 32; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) s[(kc + j) * (EXT + 1) + rr + 32 * i] = r[i][j];
}

// Loads the (BK x EXT) slab of an M/N-CONTIGUOUS operand (element (k,c) at p[k*ld + c]):
// EXT/32 float4 = k rows (t/(EXT/4) + (1024/EXT)*i), c = 4*(t%(EXT/4))..+3.
template <int EXT>
__device__ __forceinline__ void gload_mcontig(float (&r)[4][4], const float* __restrict__ p, long long ld, int col0,
                                              int k0, int Ccols, int K, int vec, int t) {
  constexpr int TPR = EXT / 4;          // threads per k row
  constexpr int KSTEP = kThreads / TPR; // k rows per pass
  const int cc = (t % TPR) * 4;
  const int kr = t / TPR;
#pragma unroll
  for (int i = 0; i < EXT / 32; ++i) {
    const int k = k0 + kr + KSTEP * i;
    const int c = col0 + cc;
    if (k < K && vec && c + 3 < Ccols) {
      const float4 v = *reinterpret_cast<const float4*>(p + static_cast<long long>(k) * ld + c);
      r[i][0] = v.x; r[i][1] = v.y; r[i][2] = v.z; r[i][3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        r[i][j] = (k < K && c + j < Ccols) ? p[static_cast<long long>(k) * ld + c + j] : 0.f;
    }
  }
}
template <int EXT>
__device__ __forceinline__ void sstore_mcontig(float* __restrict__ s, const float (&r)[4][4], int t) {
  constexpr int TPR = EXT / 4;
  constexpr int KSTEP = kThreads / TPR;
  const int cc = (t % TPR) * 4;
  const int kr = t / TPR;
#pragma unroll
  for (int i = 0; i < EXT / 32; ++i) {
    float4 v = make_float4(r[i][0], r[i][1], r[i][2], r[i][3]);
    *reinterpret_cast<float4*>(s + (kr + KSTEP * i) * (EXT + 4) + cc) = v;
  }
}

// TA: A is stored (K,M) (M-contiguous).  TB: B is stored (N,K) (K-contiguous, Linear weight layout).
// TM: tile rows (128 -> each wave a 64x64 sub-tile, 2 resident workgroups per CU; 64 -> 32x64 sub-tiles, 3 per CU).
template <bool TA, bool TB, int TM>
__global__ __launch_bounds__(kThreads, TM == 128 ? 2 : 3) void gemm_f32_kernel(const GemmArgs g) {
  constexpr int MT = TM / 64;               // MFMA tiles per wave along M
  constexpr int LDA_S = TA ? TM + 4 : TM + 1;   // A K-contiguous (TA = false) -> transposing store
  constexpr int LDB_S = TB ? LD_T : LD_D;       // B K-contiguous (TB = true)  -> transposing store
  __shared__ __attribute__((aligned(16))) float sA[2][BK * (TM + 4)];
  __shared__ __attribute__((aligned(16))) float sB[2][BK * LD_MAX];

  const int t = threadIdx.x;
  const int lane = t & 63;
  const int wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware bijective remap: hardware places block b on XCD b % 8; give each XCD a contiguous run of
  // logical tiles (n fastest) so tiles sharing an A row-panel hit the same L2.
  const int nwg = g.tiles_m * g.tiles_n;
  const int bid = blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7;
  const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  const int tm = wg / g.tiles_n, tn = wg - tm * g.tiles_n;
  const int m0 = tm * TM, n0 = tn * BN;

  const int z = blockIdx.y;
  const int ktiles = (g.K + BK - 1) / BK;
  const int kt_begin = z * g.tiles_per_split;
  const int kt_end = min(ktiles, kt_begin + g.tiles_per_split);

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // two register staging sets: the global loads of K tile t+2 are in flight while tile t is multiplied and tile t+1
  // waits in the other set (timing ablations showed 16 % of the K loop exposed to global-load latency with one set)
  float ra0[4][4], rb0[4][4], ra1[4][4], rb1[4][4];
  // interior tiles (the vast majority) take unguarded 16-byte loads: no per-load branches in the K loop
  const bool full_mn = (m0 + TM <= g.M) && (n0 + BN <= g.N) && g.vecA && g.vecB;
  auto gload = [&](int kt, float (&ra)[4][4], float (&rb)[4][4]) {
    const int k0 = kt * BK;
    if (full_mn && k0 + BK <= g.K) {
      if (TA) {
        constexpr int TPR = TM / 4, KSTEP = kThreads / TPR;
        const float* p = g.A + static_cast<long long>(k0 + t / TPR) * g.lda + m0 + (t % TPR) * 4;
#pragma unroll
        for (int i = 0; i < TM / 32; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(p + static_cast<long long>(KSTEP * i) * g.lda);
          ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
        }
      } else {
        const float* p = g.A + static_cast<long long>(m0 + (t >> 3)) * g.lda + k0 + (t & 7) * 4;
#pragma unroll
        for (int i = 0; i < TM / 32; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(p + static_cast<long long>(32 * i) * g.lda);
          ra[i][0] = v.x; ra[i][1] = v.y; ra[i][2] = v.z; ra[i][3] = v.w;
        }
      }
      if (TB) {
        const float* p = g.B + static_cast<long long>(n0 + (t >> 3)) * g.ldb + k0 + (t & 7) * 4;
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(p + static_cast<long long>(32 * i) * g.ldb);
          rb[i][0] = v.x; rb[i][1] = v.y; rb[i][2] = v.z; rb[i][3] = v.w;
        }
      } else {
        constexpr int TPR = BN / 4, KSTEP = kThreads / TPR;
        const float* p = g.B + static_cast<long long>(k0 + t / TPR) * g.ldb + n0 + (t % TPR) * 4;
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(p + static_cast<long long>(KSTEP * i) * g.ldb);
          rb[i][0] = v.x; rb[i][1] = v.y; rb[i][2] = v.z; rb[i][3] = v.w;
        }
      }
      return;
    }
    if (TA) gload_mcontig<TM>(ra, g.A, g.lda, m0, k0, g.M, g.K, g.vecA, t);
    else gload_kcontig<TM>(ra, g.A, g.lda, m0, k0, g.M, g.K, g.vecA, t);
    if (TB) gload_kcontig<BN>(rb, g.B, g.ldb, n0, k0, g.N, g.K, g.vecB, t);
    else gload_mcontig<BN>(rb, g.B, g.ldb, n0, k0, g.N, g.K, g.vecB, t);
  };
  auto sstore = [&](int buf, const float (&ra)[4][4], const float (&rb)[4][4]) {
    if (TA) sstore_mcontig<TM>(sA[buf], ra, t); else sstore_kcontig<TM>(sA[buf], ra, t);
    if (TB) sstore_kcontig<BN>(sB[buf], rb, t); else sstore_mcontig<BN>(sB[buf], rb, t);
  };

  const int nt = kt_end - kt_begin;
  if (nt > 0) {
    gload(kt_begin, ra0, rb0);
    sstore(0, ra0, rb0);
  }
  if (nt > 1) gload(kt_begin + 1, ra0, rb0);
  if (nt > 2) gload(kt_begin + 2, ra1, rb1);
  __syncthreads();
  const int kh = lane >> 5;   // which k of the pair this lane supplies
  const int l31 = lane & 31;
  auto compute = [&](int buf) {
    const float* pa = sA[buf] + wm * (TM / 2) + l31;
    const float* pb = sB[buf] + wn * 64 + l31;
    // operand reads run one k-pair ahead of the MFMAs (register double buffer) so LDS latency hides under them
    float av[MT], b0, b1;
#pragma unroll
    for (int i = 0; i < MT; ++i) av[i] = pa[kh * LDA_S + 32 * i];
    b0 = pb[kh * LDB_S];
    b1 = pb[kh * LDB_S + 32];
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float an[MT], bn0 = 0.f, bn1 = 0.f;
      if (kk + 2 < BK) {
#pragma unroll
        for (int i = 0; i < MT; ++i) an[i] = pa[(kk + 2 + kh) * LDA_S + 32 * i];
        bn0 = pb[(kk + 2 + kh) * LDB_S];
        bn1 = pb[(kk + 2 + kh) * LDB_S + 32];
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b0, acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], b1, acc[i][1], 0, 0, 0);
      }
      if (kk + 2 < BK) {
#pragma unroll
        for (int i = 0; i < MT; ++i) av[i] = an[i];
        b0 = bn0;
        b1 = bn1;
      }
    }
  };
  for (int it = 0; it < nt; it += 2) {
    compute(0);                                         // tile it     (LDS buffer 0)
    if (it + 1 < nt) sstore(1, ra0, rb0);               // tile it + 1 -> buffer 1
    if (it + 3 < nt) gload(kt_begin + it + 3, ra0, rb0);
    __syncthreads();
    if (it + 1 >= nt) break;
    compute(1);                                         // tile it + 1 (LDS buffer 1)
    if (it + 2 < nt) sstore(0, ra1, rb1);               // tile it + 2 -> buffer 0
    if (it + 4 < nt) gload(kt_begin + it + 4, ra1, rb1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) -------
  const bool partial = (g.splits > 1);
  float* out = partial ? g.ws + static_cast<long long>(z) * g.M * g.N : g.C;
  const long long ldo = partial ? g.N : g.ldc;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = (!partial && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * (TM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (row >= g.M) continue;
        float v = acc[i][j][e];
        float* o = out + static_cast<long long>(row) * ldo + col;
        if (!partial) {
          v += bv;
          if (g.accumulate) v += *o;
          v = apply_act(v, g.act, g.slope);
        }
        *o = v;
      }
    }
  }
}

// out = act( sum_z ws[z] + bias (+ C) )
__global__ void splitk_reduce_kernel(const GemmArgs g) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long total = static_cast<long long>(g.M) * g.N;
  if (i >= total) return;
  const int row = static_cast<int>(i / g.N), col = static_cast<int>(i - static_cast<long long>(row) * g.N);
  // The slices are summed in increasing z (the order is part of the result), but read 16 at a time: with 100-300
  // slices of a small C (weight gradients over 70 k rows) one dependent load per slice made this kernel a 45-65 us
  // latency chain on a near-empty chip.
  float v = 0.f;
  const float* p = g.ws + i;
  int z = 0;
  for (; z + 16 <= g.splits; z += 16) {
    float t[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = p[static_cast<long long>(z + u) * total];
#pragma unroll
    for (int u = 0; u < 16; ++u) v += t[u];
  }
  for (; z < g.splits; ++z) v += p[static_cast<long long>(z) * total];
  if (g.bias) v += g.bias[col];
  float* o = g.C + static_cast<long long>(row) * g.ldc + col;
  if (g.accumulate) v += *o;
  *o = apply_act(v, g.act, g.slope);
}

// 128-row tiles have the better MFMA/LDS ratio; 64-row tiles quantise less at the tail (more, smaller tiles and 3
// resident workgroups per CU).  Pick by a simple "rounds of resident workgroups x work per tile" model.
static int choose_tm(int64_t M, int64_t N, int splits) {
  const int64_t tn = (N + BN - 1) / BN;
  const double r128 = std::ceil(static_cast<double>(((M + 127) / 128) * tn * splits) / (256.0 * 2)) * 1.00;
  const double r64 = std::ceil(static_cast<double>(((M + 63) / 64) * tn * splits) / (256.0 * 3)) * 0.56;
  return r64 < r128 ? 64 : 128;
}

static void plan_split(int64_t M, int64_t N, int64_t K, int* splits, int* tiles_per_split) {
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  const int64_t ktiles = (K + BK - 1) / BK;
  int64_t s = 1;
  if (tiles < 192 && ktiles >= 16) {
    s = (768 + tiles - 1) / tiles;            // aim for ~3 workgroups per CU
    const int64_t max_s = ktiles / 8;          // keep >= 8 K tiles (256 k) per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
  }
  if (const char* e = getenv("SG_GEMM_SPLITS")) {   // tuning aid
    const int64_t v = atoll(e);
    if (v >= 1 && v <= ktiles) s = v;
  }
  int64_t per = (ktiles + s - 1) / s;
  if (per < 1) per = 1;
  s = (ktiles + per - 1) / per;
  if (s < 1) s = 1;
  *splits = static_cast<int>(s);
  *tiles_per_split = static_cast<int>(per);
}

// ---- elementwise helpers used by the dense backward -------------------------------------------------
__device__ __forceinline__ float act_slope_from_output(float y, int act, float slope) {
  switch (act) {
    case SG_ACT_LEAKY: return y > 0.f ? 1.f : slope;   // leaky preserves sign for slope > 0
    case SG_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case SG_ACT_SIGMOID: return y * (1.f - y);
    case SG_ACT_TANH: return 1.f - y * y;
    default: return 1.f;
  }
}
__global__ void act_bwd_kernel(float* __restrict__ dpre, const float* __restrict__ dout, const float* __restrict__ out,
                               long long n, int act, float slope) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long p = i; p < n; p += stride) dpre[p] = dout[p] * act_slope_from_output(out[p], act, slope);
}

// rows per workgroup in pass 1 of colsum: 64 keeps >= 160 workgroups in flight for the 10 k-row matrices of the step,
// 256 keeps the number of partial rows (pass 2 reads them all) small for the 70 k-row ones
static inline int colsum_rows(long long M) { return M > 32768 ? 256 : 64; }
// pass 1: partial[chunk][col] = sum over the chunk's rows.  256 threads = 4 waves; a wave reads whole row
// segments of 64*VEC consecutive floats (1 KiB with float4) so each load instruction is one coalesced burst;
// the 4 waves take rows r, r+4, ... and are combined through LDS.
// FUSED: X = dout, and the kernel first forms dpre = dout * act'(out) (written to `dpre`, dense M x N like `out`), then
// sums THAT: the Dense layer's bias gradient without a second pass over dpre.
template <int VEC, bool FUSED>
__global__ __launch_bounds__(256) void colsum_partial_kernel(float* __restrict__ partial, const float* __restrict__ X,
                                                             long long ldx, long long M, int N, int kColRows,
                                                             float* __restrict__ dpre, const float* __restrict__ out,
                                                             int act, float slope) {
#pragma clang fp contract(off)   // the product dout * act' is rounded (it IS dpre) before it enters the column sum
  __shared__ float red[4][64 * VEC];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.y * 64 * VEC + lane * VEC;
  const long long r0 = static_cast<long long>(blockIdx.x) * kColRows;
  const long long r1 = min(M, r0 + kColRows);
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  if (col < N) {
    for (long long r = r0 + w; r < r1; r += 4) {
      if (VEC == 4) {
        float4 t = *reinterpret_cast<const float4*>(X + r * ldx + col);
        if (FUSED) {
          const float4 y = *reinterpret_cast<const float4*>(out + r * N + col);
          t.x *= act_slope_from_output(y.x, act, slope); t.y *= act_slope_from_output(y.y, act, slope);
          t.z *= act_slope_from_output(y.z, act, slope); t.w *= act_slope_from_output(y.w, act, slope);
          *reinterpret_cast<float4*>(dpre + r * N + col) = t;
        }
        acc[0] += t.x; acc[1 % VEC] += t.y; acc[2 % VEC] += t.z; acc[3 % VEC] += t.w;
      } else {
        float t = X[r * ldx + col];
        if (FUSED) {
          t *= act_slope_from_output(out[r * N + col], act, slope);
          dpre[r * N + col] = t;
        }
        acc[0] += t;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) red[w][lane * VEC + v] = acc[v];
  __syncthreads();
  if (w == 0 && col < N) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int i = lane * VEC + v;
      partial[static_cast<long long>(blockIdx.x) * N + col + v] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
  }
}
// pass 2: 1024 threads = 64 columns x 16 row groups; group g sums partial rows g, g+16, ... (one coalesced 256-byte
// row segment per wave and iteration), the 16 group sums are combined through LDS in a fixed order (deterministic).
// The first version walked all `chunks` partial rows serially with one thread per column: 118 us average, 0.94 ms
// per step at ML-10M (rocprofv3, profiles/r1_bench_ml10m_kernel_stats_v4.csv).
__global__ __launch_bounds__(1024) void colsum_final_kernel(float* __restrict__ dst, const float* __restrict__ partial,
                                                            int chunks, int N, int add) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float acc = 0.f;
  if (col < N)
    for (int c = g; c < chunks; c += 16) acc += partial[static_cast<long long>(c) * N + col];
  red[g][lane] = acc;
  __syncthreads();
  if (g == 0 && col < N) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[k][lane];
    dst[col] = add ? dst[col] + s : s;
  }
}

}  // namespace sg

using namespace sg;

static inline size_t ws_align(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

SG_API size_t sg_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K, int transA) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  int s, per;
  plan_split(M, N, K, &s, &per);
  size_t need = s > 1 ? static_cast<size_t>(s) * M * N * sizeof(float) : 0;
  if (backend_for(M, N, K, transA) == 3) need = ws_align(need) + f16x3_plane_bytes(M, N, K);
  return need;
}

SG_API int sg_gemm_f32_hip(float* C, int64_t ldc, const float* A, int64_t lda, int transA, const float* B, int64_t ldb,
                           int transB, int64_t M, int64_t N, int64_t K, const float* bias, int act, float slope,
                           int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  if (M < 0 || N < 0 || K < 0) return fail(SG_ERR_INVALID, "negative GEMM dimension");
  if (M == 0 || N == 0) return SG_OK;
  if (M >= (1ll << 31) || N >= (1ll << 31) || K >= (1ll << 31)) return fail(SG_ERR_INVALID, "GEMM dimension overflow");
  if (act < SG_ACT_NONE || act > SG_ACT_TANH) return fail(SG_ERR_INVALID, "bad activation %d", act);
  if (!C || (K > 0 && (!A || !B))) return fail(SG_ERR_INVALID, "null pointer argument");
  if (ldc < N || lda < (transA ? M : K) || ldb < (transB ? K : N)) return fail(SG_ERR_INVALID, "leading dimension too small");
  GemmArgs g{};
  g.C = C; g.A = A; g.B = B; g.bias = bias;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = static_cast<int>(M); g.N = static_cast<int>(N); g.K = static_cast<int>(K);
  g.act = act; g.slope = slope; g.accumulate = accumulate ? 1 : 0;
  plan_split(M, N, K, &g.splits, &g.tiles_per_split);
  if (K == 0) { g.splits = 1; g.tiles_per_split = 1; }
  int tm = choose_tm(M, N, g.splits);
  if (const char* force = getenv("SG_GEMM_TM")) tm = atoi(force) == 64 ? 64 : 128;   // tuning aid
  // backend: exact-fp32 MFMA (this file) or the fp32-accurate bf16x6 split on the bf16 matrix cores (gemm_bf16x6.hip)
  // measured (profiles/): bf16x6 wins for row-major A (forward / data-gradient GEMMs), fp32 MFMA for the transposed-A
  // weight gradients; tiny K has nothing to amortise the split
  int backend = backend_for(M, N, K, transA);
  g.vecA = (lda % 4 == 0) && aligned(A, 16);
  g.vecB = (ldb % 4 == 0) && aligned(B, 16);
  if (backend == 2 && !x6v2_supported(g, transA != 0, transB != 0)) backend = 0;   // odd K / unaligned operands: exact-fp32 kernel
  const bool use_bx6 = backend == 1, use_v2 = backend == 2, use_x3 = backend == 3;
  if (use_bx6 || use_v2 || use_x3) tm = 128;
  if (use_x3 && g.splits > 1 && (g.tiles_per_split & 1)) {   // f16x3 scales 64-k blocks = two K tiles: keep slices block-aligned
    const int ktiles = static_cast<int>((K + BK - 1) / BK);
    g.tiles_per_split += 1;
    g.splits = (ktiles + g.tiles_per_split - 1) / g.tiles_per_split;     // never more slices than the workspace query assumed
  }
  g.tiles_m = static_cast<int>((M + tm - 1) / tm);
  g.tiles_n = static_cast<int>((N + BN - 1) / BN);
  if (static_cast<int64_t>(g.tiles_m) * g.tiles_n >= (1ll << 31)) return fail(SG_ERR_INVALID, "too many tiles");
  size_t need = g.splits > 1 ? static_cast<size_t>(g.splits) * M * N * sizeof(float) : 0;
  char* planes = nullptr;
  if (use_x3) {
    planes = static_cast<char*>(workspace) + ws_align(need);
    need = ws_align(need) + f16x3_plane_bytes(M, N, K);
  }
  if (need) {
    if (!workspace || workspace_bytes < need)
      return fail(SG_ERR_WORKSPACE, "GEMM workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    g.ws = static_cast<float*>(workspace);
  }
  g.vecA = (lda % 4 == 0) && aligned(A, 16);
  g.vecB = (ldb % 4 == 0) && aligned(B, 16);
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 grid(static_cast<unsigned>(g.tiles_m * g.tiles_n), static_cast<unsigned>(g.splits));
#define SG_TRY_RC(expr_) do { const int rc_ = (expr_); if (rc_ != SG_OK) return rc_; } while (0)
#define SG_LAUNCH_GEMM(TA_, TB_)                                                                         \
  do {                                                                                                  \
    if (tm == 128) hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, 128>), grid, dim3(kThreads), 0, st, g); \
    else hipLaunchKernelGGL((gemm_f32_kernel<TA_, TB_, 64>), grid, dim3(kThreads), 0, st, g);            \
  } while (0)
  const long prof = gprof_begin(st, M, N, K, backend);
  bool reduced = false;
  if (use_x3) {
    int used = g.splits;
    SG_TRY_RC(launch_gemm_f16x3(g, transA != 0, transB != 0, planes, st, &reduced, &used));
    g.splits = used;              // the launcher may have taken fewer, fuller slices
  } else if (use_v2) {
    launch_gemm_x6v2(g, transA != 0, transB != 0, cu_count(), st);
  } else if (use_bx6) {
    launch_gemm_bf16x6(g, transA != 0, transB != 0, st);
  } else if (transA) {
    if (transB) SG_LAUNCH_GEMM(true, true); else SG_LAUNCH_GEMM(true, false);
  } else {
    if (transB) SG_LAUNCH_GEMM(false, true); else SG_LAUNCH_GEMM(false, false);
  }
#undef SG_LAUNCH_GEMM
  if (g.splits > 1 && !reduced) {
    const long long total = static_cast<long long>(M) * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, g);
  }
  gprof_end(prof, st);
  return check_launch("gemm_f32");
}

SG_API int sg_gemm_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(sg::g_gprof_mu);
  const int was = sg::g_gprof_on ? 1 : 0;
  if (on && !was) {
    for (auto& r : sg::g_gprof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    sg::g_gprof.clear();
  }
  sg::g_gprof_on = on != 0;
  return was;
}
// one record per sg_gemm_f32_hip call since the profile was enabled: milliseconds, (M, N, K) as mnk[3 i .. 3 i + 2], backend
SG_API int64_t sg_gemm_profile_read(float* ms, int64_t* mnk, int* backend, int64_t capacity) {
  std::lock_guard<std::mutex> lk(sg::g_gprof_mu);
  int64_t n = 0;
  for (auto& r : sg::g_gprof) {
    if (n < capacity) {
      float t = 0.f;
      (void)hipEventSynchronize(r.b);
      if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) t = -1.f;
      if (ms) ms[n] = t;
      if (mnk) { mnk[3 * n] = r.M; mnk[3 * n + 1] = r.N; mnk[3 * n + 2] = r.K; }
      if (backend) backend[n] = r.backend;
      ++n;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  sg::g_gprof.clear();
  return n;
}

// tuning aid (tests, benchmarks): GEMM backend 0 exact-fp32 MFMA, 1 bf16x6, 2 x6v2, 3 f16x3; -1 = environment / build default
SG_API int sg_gemm_backend(int backend) {
  g_backend_override.store(backend < 0 || backend > 3 ? -1 : backend, std::memory_order_relaxed);
  return SG_OK;
}

namespace sg {
// out[i] = act(in[i]); 16 bytes per lane when both pointers allow it.  `out` may alias `in`.
__global__ void act_fwd_kernel(float* __restrict__ out, const float* __restrict__ in, long long n, int act, float slope, int vec) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
  if (vec) {
    const long long n4 = n >> 2;
    for (long long p = i; p < n4; p += stride) {
      float4 v = reinterpret_cast<const float4*>(in)[p];
      v.x = apply_act(v.x, act, slope); v.y = apply_act(v.y, act, slope);
      v.z = apply_act(v.z, act, slope); v.w = apply_act(v.w, act, slope);
      reinterpret_cast<float4*>(out)[p] = v;
    }
    for (long long p = (n4 << 2) + i; p < n; p += stride) out[p] = apply_act(in[p], act, slope);
  } else {
    for (long long p = i; p < n; p += stride) out[p] = apply_act(in[p], act, slope);
  }
}
}  // namespace sg

// Elementwise activation (reference common.py:32-57) for the places where it cannot ride on a GEMM / gather epilogue: after
// the all-reduce of a node-partitioned aggregate (the activation follows the SUM over ranks).  out may alias in.
SG_API int sg_act_hip(float* out, const float* in, int64_t n, int act, float slope, void* stream) {
  if (n < 0) return fail(SG_ERR_INVALID, "negative size");
  if (act < SG_ACT_NONE || act > SG_ACT_TANH) return fail(SG_ERR_INVALID, "bad activation %d", act);
  if (n == 0) return SG_OK;
  if (!out || !in) return fail(SG_ERR_INVALID, "null pointer argument");
  const int vec = aligned(out, 16) && aligned(in, 16);
  int64_t blocks = ((vec ? (n + 3) / 4 : n) + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(sg::act_fwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream), out,
                     in, static_cast<long long>(n), act, slope, vec);
  return check_launch("act_fwd");
}

SG_API int sg_act_bwd_hip(float* dpre, const float* dout, const float* out, int64_t n, int act, float slope,
                          void* stream) {
  if (n < 0) return fail(SG_ERR_INVALID, "negative size");
  if (act < SG_ACT_NONE || act > SG_ACT_TANH) return fail(SG_ERR_INVALID, "bad activation %d", act);
  if (n == 0) return SG_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dpre, dout, out, static_cast<long long>(n), act, slope);
  return check_launch("act_bwd");
}

SG_API size_t sg_colsum_workspace_bytes(int64_t M, int64_t N) {
  if (M <= 0 || N <= 0) return 0;
  const int64_t kColRows = colsum_rows(M);
  return static_cast<size_t>((M + kColRows - 1) / kColRows) * N * sizeof(float);
}

static int colsum_launch(float* dst, const float* X, int64_t ldx, int64_t M, int64_t N, int req, void* workspace,
                         size_t workspace_bytes, hipStream_t st, float* dpre, const float* out, int act, float slope) {
  const bool fused = dpre != nullptr;
  const int kColRows = colsum_rows(M);
  const int64_t chunks = (M + kColRows - 1) / kColRows;
  if (chunks >= (1ll << 31) || (N + 63) / 64 > 65535) return fail(SG_ERR_INVALID, "colsum: shape too large");
  if (chunks > 0) {
    if (!workspace || workspace_bytes < sg_colsum_workspace_bytes(M, N))
      return fail(SG_ERR_WORKSPACE, "colsum workspace too small");
    const bool v4 = N % 4 == 0 && ldx % 4 == 0 && aligned(X, 16) && (!fused || (aligned(dpre, 16) && aligned(out, 16)));
    const dim3 grid(static_cast<unsigned>(chunks), static_cast<unsigned>(v4 ? (N + 255) / 256 : (N + 63) / 64));
#define SG_COLSUM(V, F)                                                                                         \
  hipLaunchKernelGGL((colsum_partial_kernel<V, F>), grid, dim3(256), 0, st, static_cast<float*>(workspace), X,   \
                     static_cast<long long>(ldx), static_cast<long long>(M), static_cast<int>(N), kColRows, dpre, \
                     out, act, slope)
    if (v4) { if (fused) SG_COLSUM(4, true); else SG_COLSUM(4, false); }
    else { if (fused) SG_COLSUM(1, true); else SG_COLSUM(1, false); }
#undef SG_COLSUM
  }
  hipLaunchKernelGGL(colsum_final_kernel, dim3(static_cast<unsigned>((N + 63) / 64)), dim3(1024), 0, st, dst,
                     static_cast<const float*>(workspace), static_cast<int>(chunks), static_cast<int>(N),
                     req == SG_REQ_ADD);
  return check_launch("colsum");
}

SG_API int sg_colsum_hip(float* dst, const float* X, int64_t ldx, int64_t M, int64_t N, int req, void* workspace,
                         size_t workspace_bytes, void* stream) {
  if (!valid_req(req)) return fail(SG_ERR_INVALID, "bad req %d", req);
  if (req == SG_REQ_NULL || N <= 0) return SG_OK;
  if (M < 0 || ldx < N || N >= (1ll << 31)) return fail(SG_ERR_INVALID, "bad colsum shape");
  return colsum_launch(dst, X, ldx, M, N, req, workspace, workspace_bytes, static_cast<hipStream_t>(stream), nullptr,
                       nullptr, SG_ACT_NONE, 0.f);
}

SG_API int sg_act_bwd_colsum_hip(float* dpre, float* dbias, const float* dout, const float* out, int64_t M, int64_t N,
                                 int act, float slope, int req, void* workspace, size_t workspace_bytes, void* stream) {
  if (!valid_req(req)) return fail(SG_ERR_INVALID, "bad req %d", req);
  if (act < SG_ACT_NONE || act > SG_ACT_TANH) return fail(SG_ERR_INVALID, "bad activation %d", act);
  if (M < 0 || N < 0 || N >= (1ll << 31)) return fail(SG_ERR_INVALID, "bad shape");
  if (N == 0) return SG_OK;
  if (!dpre || !dbias || (M > 0 && (!dout || !out))) return fail(SG_ERR_INVALID, "null pointer argument");
  if (req == SG_REQ_NULL) return sg_act_bwd_hip(dpre, dout, out, M * N, act, slope, stream);
  return colsum_launch(dbias, dout, N, M, N, req, workspace, workspace_bytes, static_cast<hipStream_t>(stream), dpre, out,
                       act, slope);
}
