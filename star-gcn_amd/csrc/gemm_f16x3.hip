// gemm_f16x3.hip -- fp32-accurate GEMM on the f16 matrix cores of gfx950 with THREE matrix instructions per product
// (the bf16 backends need six), operands pre-split ONCE per call into a fragment-major plane format.
//
// Arithmetic (block floating point).  Each operand is cut into blocks of 32 rows of op(X) x 64 k.  A block gets ONE
// power-of-two scale s that brings its largest magnitude into [2^14, 2^15) -- exact, nothing is lost by it -- and every
// scaled value is split into two f16 planes
//     x s = h1 + h2 + e,   h1 = rne_f16(x s),  h2 = rne_f16(x s - h1),   |e| <= 2^-22 |x s|
// (11 significant bits per plane; the residual is an exact fp32 subtraction).  A product is formed from the three plane
// pairs a1 b1 + a1 b2 + a2 b1 on v_mfma_f32_32x32x16_f16 with fp32 accumulation; the dropped a2 b2 is <= 2^-22 |a b|,
// so a term carries a relative error <= 3 * 2^-22 = 7.2e-7 (fp32 product rounding: 6e-8), independent from term to
// term: over a dot product of length K it adds ~ 7e-7 / sqrt(K) of sum |a||b| -- below the rounding of the fp32
// accumulation itself for the K >= 96 this backend is used for, and inside the fp64-referenced bound of the exact-fp32
// kernel (tests/test_gpu_dense_multilink.py).  The three products of a 64-k block accumulate in a block-local register
// set P (four k steps: short sums, so the 2^-11 corrections are not rounded against a long running sum); at the end of
// the block P is scaled by 1 / (s_A s_B) with v_ldexp_f32 (exact) and added to the running result -- one fp32 rounding
// per 64 k, like a plain fp32-accumulating MFMA chain.  Dynamic range: an element more than 2^18 below the largest
// magnitude of ITS block (32 rows x 64 k) falls into the f16 subnormal range of h2 and keeps an absolute error of 2^-40
// of the block maximum instead of 2^-22 relative: a row 10^6 times smaller than a neighbour in the same 32-row block
// still comes out to 1e-6 relative.  Because the scale is per block there is no pass over the operand to find maxima:
// the split is ONE streaming pass (4 B read + 4 B written per element).
//
// Why pre-split.  In the wave-specialised bf16x6 kernel (gemm_x6v2.hip) every 128 x 128 tile re-splits its operand
// panels: the A panel of the 1 M x 4096 x 256 forward GEMM is converted 32 times, once per column tile, and that VALU +
// LDS-store work shares issue slots and the LDS pipe with the matrix instructions (each role gains 30 % when
// another is removed).  Here a streaming pass converts each operand once into "fragment-major" planes: for every
// 32-row block and 16-k step the 64 x 16 bytes that the 64 lanes of a wave feed to ONE MFMA operand are stored
// contiguously in lane order.  The GEMM kernel then is nothing but matrix work:
//   * global -> LDS by global_load_lds_dwordx4 (LDS-DMA: no VGPRs, no VALU, no ds_write); a tile's bytes are contiguous
//     4 KiB runs in memory, the LDS image is a plain copy and every ds_read_b128 of a fragment reads 1 KiB contiguous --
//     conflict-free without any swizzle arithmetic;
//   * 128 x 128 x 32 tiles, 4 waves (64 x 64 each), two 32 KiB LDS stages = 64 KiB: TWO workgroups per CU, so one
//     workgroup's barrier / epilogue overlaps the other's matrix work; the DMA of K tile t+1 is issued before the
//     fragments of K tile t are read, one barrier per K tile;
//   * epilogue through a private LDS block per wave (aliasing the stages) -> 16-byte stores, 256 B per row segment.
// Conversion costs one read + one write of the operand (two f16 planes = 4 B per element, the bytes of the fp32 original);
// it is charged to this backend in every measurement.
//
// Epilogue / split-K contract identical to gemm_f32.hip (GemmArgs); selected by sg_gemm_f32_hip (backend 3).
#include "gemm_x3_shared.hpp"

#include <atomic>
#include <cmath>
#include <cstdlib>

namespace sg {

namespace f16x3 {

// ---------------------------------------------------------------------------------------------------------------------
// split: fp32 operand -> fragment-major f16 planes.  16-byte unit u(rb, ks, plane, lane) = ((rb*KS + ks)*2 + plane)*64 + lane
// holds op(X)[row = 32 rb + (lane & 31)][k = 16 ks + 8 (lane >> 5) + 0..7]  (the operand layout of v_mfma_f32_32x32x16_f16).
// One workgroup converts 32 rows x 64 k: coalesced 128-byte row segments in, a [row][k] tile in LDS, 1 KiB units out.
// ---------------------------------------------------------------------------------------------------------------------
// (rb: 32-row block of the operand, kb: 64-k block; the kernels below map their grids onto them)
template <bool RC>
__device__ __forceinline__ void split_block(char* __restrict__ planes, int* __restrict__ expo, const float* __restrict__ p,
                                            long long ld, int R, int K, int KS, int vec, int rb, int kb) {
  __shared__ float tile[32][65];
  __shared__ float wmax[4];
  const int t = threadIdx.x;
  const int k0 = kb * 64;
  const int r0 = rb * 32;
  float m = 0.f;
  if (!RC) {            // element (r, k) at p[r * ld + k]: thread = (row t/8, 4 consecutive k), two passes of 32 k
    const int r = r0 + (t >> 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + h * 32 + (t & 7) * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (r < R) {
        if (vec && k + 3 < K) {
          const float4 x = *reinterpret_cast<const float4*>(p + static_cast<long long>(r) * ld + k);
          v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (k + j < K) v[j] = p[static_cast<long long>(r) * ld + k + j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tile[t >> 3][h * 32 + (t & 7) * 4 + j] = v[j];
        m = fmaxf(m, fabsf(v[j]));
      }
    }
  } else {              // element (k, c) at p[k * ld + c], op row = c: thread = (k row t/8 + 32 h, 4 consecutive c)
    const int c = r0 + (t & 7) * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int k = k0 + h * 32 + (t >> 3);
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (k < K) {
        if (vec && c + 3 < R) {
          const float4 x = *reinterpret_cast<const float4*>(p + static_cast<long long>(k) * ld + c);
          v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (c + j < R) v[j] = p[static_cast<long long>(k) * ld + c + j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        tile[(t & 7) * 4 + j][h * 32 + (t >> 3)] = v[j];
        m = fmaxf(m, fabsf(v[j]));
      }
    }
  }
  // block maximum -> exponent e with max * 2^e in [2^14, 2^15) (fmaxf drops NaNs; a block holding an inf takes the slow
  // path below)
  m = wave_max_nonneg(m);
  if ((t & 63) == 0) wmax[t >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  bool nonfinite = false;
  if (__builtin_expect(!(m <= 3.402823466e38f), 0)) {
    // the block holds an inf (block-uniform, rare): scale by its largest FINITE magnitude instead, so that finite
    // block-mates above 65504 do not overflow the f16 planes next to it; the inf itself stays inf in the leading plane and
    // gets a zero residual (inf - inf would turn the product's inf into NaN)
    nonfinite = true;
    __syncthreads();                                   // everyone has read wmax
    float mf = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = fabsf(tile[t >> 3][(t & 7) * 8 + j]);
      mf = fmaxf(mf, a <= 3.402823466e38f ? a : 0.f);  // drops inf and NaN
    }
    mf = wave_max_nonneg(mf);
    if ((t & 63) == 0) wmax[t >> 6] = mf;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  }
  int e = 0;
  {
    const unsigned b = __float_as_uint(m);
    const int ex = static_cast<int>((b >> 23) & 0xffu);
    if (b != 0u && ex != 0xff) e = 14 - (max(ex, 1) - 127);
    e = min(max(e, -126), 126);
  }
  const float s = __uint_as_float(static_cast<unsigned>(127 + e) << 23);
  if (t == 0) expo[static_cast<long long>(rb) * (KS >> 2) + kb] = -e;
  // thread -> unit: k step ks = t / 64 of this block's four, lane = t % 64
  const int lane = t & 63, ks = t >> 6;
  const int row = lane & 31, kk = ks * 16 + (lane >> 5) * 8;
  unsigned h1[4], h2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t0 = tile[row][kk + 2 * j], t1 = tile[row][kk + 2 * j + 1];
    const f32x2 x = {t0 * s, t1 * s};
    const f16x2 a = __builtin_convertvector(x, f16x2);
    f32x2 res = {__builtin_fmaf(t0, s, -static_cast<float>(a[0])), __builtin_fmaf(t1, s, -static_cast<float>(a[1]))};   // v_fma_mix_f32
    if (nonfinite) {      // block-uniform
      if (fabsf(x[0]) == __builtin_inff()) res[0] = 0.f;
      if (fabsf(x[1]) == __builtin_inff()) res[1] = 0.f;
    }
    const f16x2 b = __builtin_convertvector(res, f16x2);
    h1[j] = __builtin_bit_cast(unsigned, a);
    h2[j] = __builtin_bit_cast(unsigned, b);
  }
  const long long u = (static_cast<long long>(rb) * KS + (k0 >> 4) + ks) * 2;
  uint4* out = reinterpret_cast<uint4*>(planes) + u * 64 + lane;
  out[0] = make_uint4(h1[0], h1[1], h1[2], h1[3]);
  out[64] = make_uint4(h2[0], h2[1], h2[2], h2[3]);
}

template <bool RC>
__global__ __launch_bounds__(256) void split_kernel(char* __restrict__ planes, int* __restrict__ expo,
                                                    const float* __restrict__ p, long long ld, int R, int K, int KS, int vec,
                                                    int* __restrict__ clear_flag) {
  if (clear_flag && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) *clear_flag = 0;      // the exactness flag of gemm_x3w.hip
  split_block<RC>(planes, expo, p, ld, R, K, KS, vec, blockIdx.x, blockIdx.y);
}

// BOTH operands of a plane-kernel product in one launch (they share K): blocks [0, rbA) of the grid's x convert A, the rest B.
// One launch instead of two: the ML-10M step has nine such products of 5 us conversions each (profiles/r6_ml10m_step_timeline.md).
struct SplitOp {
  char* planes; int* expo; const float* p; long long ld; int R; int vec;
};
template <bool RCA, bool RCB>
__global__ __launch_bounds__(256) void split2_kernel(SplitOp a, SplitOp b, int rbA, int K, int KS) {
  if (static_cast<int>(blockIdx.x) < rbA) split_block<RCA>(a.planes, a.expo, a.p, a.ld, a.R, K, KS, a.vec, blockIdx.x, blockIdx.y);
  else split_block<RCB>(b.planes, b.expo, b.p, b.ld, b.R, K, KS, b.vec, blockIdx.x - rbA, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------------
// the GEMM on planes
// ---------------------------------------------------------------------------------------------------------------------

#ifndef SG_X3_ABLATE
#define SG_X3_ABLATE 0      // development (timing only): 1 no MFMAs, 2 no fragment reads, 3 no DMA
#endif
#ifndef SG_X3_ISSUE_POS
#define SG_X3_ISSUE_POS 3   // where a wave issues the DMA of tile t+NST-1 inside iteration t: 0 before its fragment reads,
#endif                      // 1 after them (the reads then queue behind the DMA: -30 %), 2 after its matrix instructions,
                            // 3 in two halves between the three groups of matrix instructions (+3..6 % over 0, measured)
#ifndef SG_X3H_ISSUE_POS
#define SG_X3H_ISSUE_POS 3  // the same choice for the B planes of the hybrid kernel (0 or 3; 3 measured +1 %)
#endif
#ifndef SG_X3H_ABLATE
#define SG_X3H_ABLATE 0     // hybrid kernel, timing only: 1 A always read from the same 16 KB, 2 no conversion VALU (planes = raw bits)
#endif
#ifndef SG_X3_NOFOLD
#define SG_X3_NOFOLD 0      // development (timing only, wrong results): no block-local accumulator and no fold; variant 2 then
#endif                      // runs <2,1,2,4> (four workgroups per CU)
#ifndef SG_X3_PRIO
#define SG_X3_PRIO 0        // s_setprio 1 over a K tile's matrix instructions: 1 including the DMA issue between them, 2 not
#endif
#ifndef SG_X3_TIMING
#define SG_X3_TIMING 0      // development: per-phase cycle counters of wave 0 of every workgroup (sg_x3_timing_read)
#endif
#if SG_X3_TIMING
// [0] vmcnt wait  [1] barrier  [2] DMA issue  [3] fragment reads (issue -> data)  [4] MFMAs (issue -> last result)  [5] K tiles
__device__ unsigned long long g_x3_timing[6];
#define X3_T(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define X3_T(var)
#endif
// ---- epilogue shared by the kernels: one 64 x 64 wave tile (2 x 2 MFMA tiles).  TRANSPOSED: the product was formed with
// the operands swapped (rows of this tile are COLUMNS of C); it is written as C^T into `out` (leading dimension ldo) and a
// small pass transposes it afterwards -- used for weight gradients, whose C is a few MB.
__device__ __forceinline__ void store_tile(const GemmArgs& g, f32x16 (&acc)[2][2], char* smem, int wave, int lane, int m0,
                                           int n0, int z) {
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kh = lane >> 5;
  // ---- epilogue; MFMA layout (col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 kh) turned
  // around in a private LDS block per wave so that every lane stores 16 bytes and a row segment is 256 contiguous bytes
  const bool partial = (g.splits > 1);
  float* out = partial ? g.ws + static_cast<long long>(z) * g.M * g.N : g.C;
  const long long ldo = partial ? g.N : g.ldc;
  const bool vec_c = ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && ((ldo & 3) == 0);
  float* cst = reinterpret_cast<float*>(smem + wave * CSTAGE);
  const int c4 = (lane & 15) * 4, r4 = lane >> 4;
  const int col = n0 + wn * 64 + c4;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (!partial && g.bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = (col + q < g.N) ? g.bias[col + q] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        cst[((e & 3) + 8 * (e >> 2) + 4 * kh) * CPITCH + j * 32 + l31] = acc[i][j][e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int r = r4 + 4 * it;
      const int row = m0 + wm * 64 + i * 32 + r;
      const float4 t4 = *reinterpret_cast<const float4*>(cst + r * CPITCH + c4);
      float v[4] = {t4.x, t4.y, t4.z, t4.w};
      if (row < g.M && col < g.N) {
        float* o = out + static_cast<long long>(row) * ldo + col;
        const bool full = vec_c && (col + 3 < g.N);
        if (!partial) {
          if (g.accumulate) {
            if (full) {
              const float4 old = *reinterpret_cast<const float4*>(o);
              v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
            } else {
#pragma unroll
              for (int q = 0; q < 4; ++q) if (col + q < g.N) v[q] += o[q];
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] = act_fn(v[q] + bv[q], g.act, g.slope);
        }
        if (full) {
          typedef float f4 __attribute__((ext_vector_type(4)));
          const f4 tt = {v[0], v[1], v[2], v[3]};
          // a finished C tile is not re-read by this kernel: streamed past the caches; split-K partials are re-read at
          // once by the reduce kernel and stay cacheable
          if (!partial) __builtin_nontemporal_store(tt, reinterpret_cast<f4*>(o));
          else *reinterpret_cast<f4*>(o) = tt;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (col + q < g.N) o[q] = v[q];
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// WM: waves along M (tile = 64 WM x 128, 2 WM waves of 64 x 64); BKS: 16-k steps per K tile; NST: LDS stages.
//   <2, 1, 3, 3>  128 x 128 x 16, 48 KiB, <= 168 VGPRs: THREE workgroups per CU, one K tile in flight each -- the default: the
//              third workgroup's matrix work fills the barrier / DMA-wait gaps of the other two (4-14 % over the rest)
//   <2, 2, 2>  128 x 128 x 32, 64 KiB: two workgroups per CU, one K tile in flight each
//   <2, 1, 5>  128 x 128 x 16, 80 KiB: two workgroups per CU, three K tiles in flight each
//   <4, 2, 3>  256 x 128 x 32, 144 KiB: one workgroup per CU, 25 % fewer operand bytes per flop, one to two K tiles in flight
// What bounds these kernels is bytes in flight: at the matrix-core rate a CU consumes 31-43 B / clk of planes, the loaded
// L2 / Infinity-Cache latency is ~2 us, and LDS (160 KiB) is the only place in-flight DMA data can land.
template <int WM, int BKS, int NST, int MINW = 2, bool PIPE = false>
__global__ __launch_bounds__(128 * WM, MINW) void gemm_f16x3_kernel(const GemmArgs g, const PlaneArgs pl) {
  constexpr int RBA = 2 * WM, RBB = 4;                 // 32-row blocks of A / B per tile
  constexpr int UPB = 2 * BKS;                         // units per row block and K tile (k step x plane)
  constexpr int UNITS = (RBA + RBB) * UPB, NW = 2 * WM, UPW = UNITS / NW;
  constexpr int STAGE_B = UNITS * UNIT;
  static_assert(UNITS % NW == 0, "units must divide over the waves");
  constexpr int SMEM_B = NST * STAGE_B > NW * CSTAGE ? NST * STAGE_B : NW * CSTAGE;   // the epilogue's staging aliases the stages
  __shared__ __attribute__((aligned(16))) char smem[SMEM_B];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // work item -> (tile, K slice); XCD-aware bijective remap of the tile index (workgroup b runs on XCD b % 8): the tiles of
  // one XCD are consecutive, consecutive tiles share their A panel
  const int nt = g.tiles_m * g.tiles_n;
  const int item = blockIdx.x;
  const int z = item / nt, lin = item - z * nt;
  const int q8 = nt >> 3, r8 = nt & 7, x8 = lin & 7;
  const int tile = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (lin >> 3);
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  // K range of this slice in tiles of BKS k-steps; g.tiles_per_split counts 32-k tiles and is even when g.splits > 1, so a
  // slice starts on a 64-k scale block
  const int ksteps = (g.K + 15) / 16;
  const int s0 = z * g.tiles_per_split * 2, s1 = min((ksteps + 3) & ~3, s0 + g.tiles_per_split * 2);
  const int T = (s1 - s0 + BKS - 1) / BKS;             // K tiles of this slice (planes are zero-padded to 64 k)

  // acc: running result (true scale); P: products of the current 64-k block (block scale)
#if SG_X3_NOFOLD
  f32x16 acc[2][2];
  f32x16 (&P)[2][2] = acc;
#else
  f32x16 acc[2][2], P[2][2];
#endif
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#if SG_X3_TIMING
  unsigned long long tq[6] = {0, 0, 0, 0, 0, 0};
#endif

  // DMA of one K tile: UNITS units of 1 KiB, UPW per wave.  unit u: row block u / UPB (A blocks first), part u % UPB of the
  // block's contiguous UPB KiB (k step x plane)
  const long long rb_stride = static_cast<long long>(pl.KS) * 2 * UNIT;
#ifdef SG_X3_EXP_SAME_TILE      // experiment: every workgroup loads panel 0 (operands L2-hot): is the kernel memory-latency bound?
  const char* a_base = pl.pa + static_cast<long long>(s0) * 2 * UNIT + lane * 16;
  const char* b_base = pl.pb + static_cast<long long>(s0) * 2 * UNIT + lane * 16;
#else
  const char* a_base = pl.pa + static_cast<long long>(tm) * RBA * rb_stride + static_cast<long long>(s0) * 2 * UNIT + lane * 16;
  const char* b_base = pl.pb + static_cast<long long>(tn) * RBB * rb_stride + static_cast<long long>(s0) * 2 * UNIT + lane * 16;
#endif
  auto issue_unit = [&](int kt, int i) {
    char* dst = smem + (kt % NST) * STAGE_B;
    const int u = wave * UPW + i;
    const int q = u / UPB, part = u - q * UPB;
    const char* src = (q < RBA ? a_base + q * rb_stride : b_base + (q - RBA) * rb_stride) +
                      static_cast<long long>(kt) * (UPB * UNIT) + part * UNIT;
    __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(dst + u * UNIT), 16, 0, 0);
  };
  auto issue = [&](int kt) {
#pragma unroll
    for (int i = 0; i < UPW; ++i) issue_unit(kt, i);
  };
  // block exponents of this wave's two A row blocks and two B row blocks (wave-uniform: scalar loads)
  const int kbs = pl.KS >> 2;
  cst_int* ea_p = (cst_int*)(pl.exp_a + static_cast<long long>(tm * RBA + wm * 2) * kbs + (s0 >> 2));
  cst_int* eb_p = (cst_int*)(pl.exp_b + static_cast<long long>(tn * RBB + wn * 2) * kbs + (s0 >> 2));

  // pipeline: tiles t+1 .. t+NST-2 stay in flight while tile t is multiplied; ONE barrier per K tile.  The barrier of
  // iteration t also says that every wave has finished tile t-1, whose stage then receives tile t+NST-1.
  // The loop runs over 64-k scale blocks (TPB K tiles each, compile-time positions inside a block): the first product of a
  // block starts from a zero accumulator operand (an inline constant: no register clearing), the last tile folds the block
  // into the running result with one FMA per element.
  constexpr int TPB = 4 / BKS;
#pragma unroll
  for (int i = 0; i < NST - 1; ++i)
    if (i < T) issue(i);
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if constexpr (PIPE) {
    // Software-pipelined form (BKS = 1, NST = 4): the fragments of tile t+1 are READ into the other register set while
    // tile t multiplies, so the LDS latency and the matrix instructions of a wave overlap instead of alternating
    // (ablations, DESIGN 3.3: each phase costs ~0.35 us per K tile and they ran back to back).  Invariants at the top of
    // iteration t: F[t & 1] holds tile t; tile t+1 has landed and is visible (the barrier at the end of iteration t-1
    // followed the wait for it); tiles t+2 (and t+3 after the issue below) are in flight.  The stage that receives tile
    // t+3 held tile t-1, whose fragments were read during iteration t-2 -- two barriers ago.
    static_assert(!PIPE || (BKS == 1 && NST == 4), "pipelined form: 16-k tiles, four stages");
    f16x8 Fa[2][2][2], Fb[2][2][2];      // [set][row block i][plane]
    auto read_frags = [&](int set, int kt) {
      const char* st = smem + (kt % NST) * STAGE_B;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          Fa[set][i][p] = *reinterpret_cast<const f16x8*>(st + ((wm * 2 + i) * UPB + p) * UNIT + lane * 16);
          Fb[set][i][p] = *reinterpret_cast<const f16x8*>(st + ((RBA + wn * 2 + i) * UPB + p) * UNIT + lane * 16);
        }
    };
    if (T > 0) {
      if (T > 2) wait_vm<2 * UPW>(); else if (T > 1) wait_vm<UPW>(); else wait_vm<0>();      // tile 0 landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      read_frags(0, 0);
      if (T > 1) {
        if (T > 2) wait_vm<UPW>(); else wait_vm<0>();                                        // tile 1 landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    for (int kb = 0; kb * TPB < T; ++kb) {
#pragma unroll
      for (int tt = 0; tt < TPB; ++tt) {
        const int kt = kb * TPB + tt;
        if (kt < T) {
          constexpr int dummy = 0; (void)dummy;
          const int set = tt & 1;                           // TPB is even: the parity of kt
          if (kt + 3 < T) issue(kt + 3);
          if (kt + 1 < T) read_frags(set ^ 1, kt + 1);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              P[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fa[set][i][1], Fb[set][j][0], tt == 0 ? zero : P[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) P[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fa[set][i][0], Fb[set][j][1], P[i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) P[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fa[set][i][0], Fb[set][j][0], P[i][j], 0, 0, 0);
          if (kt + 2 < T) {                                 // tile t+2 must be visible to the reads of the next iteration
            if (kt + 3 < T) wait_vm<UPW>(); else wait_vm<0>();
          }
          if (kt + 1 < T) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ex = ea_p[i * kbs + kb] + eb_p[j * kbs + kb];
          if (ex >= -126 && ex <= 127) {
            const float sc = __uint_as_float(static_cast<unsigned>(127 + ex) << 23);
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(P[i][j][e], sc, acc[i][j][e]);
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] += ldexpf(P[i][j][e], ex);
          }
        }
    }
  } else
  for (int kb = 0; kb * TPB < T; ++kb) {
    // the block's four exponents are requested now (volatile: the compiler would sink the scalar loads to the fold, where
    // their round trip is exposed)
    const int ea_s[2] = {*(volatile cst_int*)(ea_p + kb), *(volatile cst_int*)(ea_p + kbs + kb)};
    const int eb_s[2] = {*(volatile cst_int*)(eb_p + kb), *(volatile cst_int*)(eb_p + kbs + kb)};
#pragma unroll
    for (int tt = 0; tt < TPB; ++tt) {
      const int kt = kb * TPB + tt;
      if (kt < T) {                                         // wave-uniform; only the last block of a slice can be short
        X3_T(t0);
        if (kt + NST - 2 < T) wait_vm<UPW * (NST - 2)>();   // in order: everything up to tile kt has landed
        else wait_vm<0>();                                  // tail: fewer tiles in flight than the count assumes
        X3_T(t1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        X3_T(t2);
#if SG_X3_ABLATE != 3 && SG_X3_ISSUE_POS == 0
        if (kt + NST - 1 < T) issue(kt + NST - 1);
#endif
        X3_T(t3);
        const char* st = smem + (kt % NST) * STAGE_B;
#pragma unroll
        for (int ks = 0; ks < BKS; ++ks) {
          f16x8 a[2][2], b[2][2];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#if SG_X3_ABLATE == 2
              a[i][p] = __builtin_bit_cast(f16x8, make_uint4(kt, ks, i, p));
              b[i][p] = __builtin_bit_cast(f16x8, make_uint4(p, i, ks, kt));
#else
              a[i][p] = *reinterpret_cast<const f16x8*>(st + ((wm * 2 + i) * UPB + ks * 2 + p) * UNIT + lane * 16);
              b[i][p] = *reinterpret_cast<const f16x8*>(st + ((RBA + wn * 2 + i) * UPB + ks * 2 + p) * UNIT + lane * 16);
#endif
            }
#if SG_X3_ISSUE_POS == 1
          asm volatile("" ::: "memory");
          if (ks == BKS - 1 && kt + NST - 1 < T) issue(kt + NST - 1);
#endif
#if SG_X3_TIMING
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
          X3_T(t4);
#if SG_X3_ABLATE == 1
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int p = 0; p < 2; ++p)
                P[i][j][p] = ((tt == 0 && ks == 0) ? 0.f : P[i][j][p]) + static_cast<float>(a[i][p][0]) + static_cast<float>(b[j][p][1]);
#else
          // corrections first, leading product last; the four tiles interleave so consecutive MFMAs never depend on each other
#if SG_X3_PRIO
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              P[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], (!SG_X3_NOFOLD && tt == 0 && ks == 0) ? zero : P[i][j], 0, 0, 0);
#if SG_X3_ISSUE_POS == 3
          asm volatile("" ::: "memory");
          if (ks == BKS - 1 && kt + NST - 1 < T) {
#if SG_X3_PRIO == 2
            __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
            for (int u = 0; u < UPW / 2; ++u) issue_unit(kt + NST - 1, u);
#if SG_X3_PRIO == 2
            __builtin_amdgcn_s_setprio(1);
#endif
          }
          asm volatile("" ::: "memory");
#endif
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) P[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], P[i][j], 0, 0, 0);
#if SG_X3_ISSUE_POS == 3
          asm volatile("" ::: "memory");
          if (ks == BKS - 1 && kt + NST - 1 < T) {
#if SG_X3_PRIO == 2
            __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
            for (int u = UPW / 2; u < UPW; ++u) issue_unit(kt + NST - 1, u);
#if SG_X3_PRIO == 2
            __builtin_amdgcn_s_setprio(1);
#endif
          }
          asm volatile("" ::: "memory");
#endif
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) P[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], P[i][j], 0, 0, 0);
#if SG_X3_PRIO
          __builtin_amdgcn_s_setprio(0);
#endif
#if SG_X3_ISSUE_POS == 2
          asm volatile("" ::: "memory");
          if (ks == BKS - 1 && kt + NST - 1 < T) issue(kt + NST - 1);
#endif
#endif
#if SG_X3_TIMING
          asm volatile("s_nop 0" :: "v"(P[1][1][0]));       // the last MFMA result: the section's time is its execution time
          { X3_T(t5); tq[0] += t1 - t0; tq[1] += t2 - t1; tq[2] += t3 - t2; tq[3] += t4 - t3; tq[4] += t5 - t4; tq[5] += 1; }
#endif
        }
      }
    }
    // end of the 64-k block: fold it into the running result at its true scale 2^ex (exact)
#if !SG_X3_NOFOLD
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ex = ea_s[i] + eb_s[j];
        if (ex >= -126 && ex <= 127) {                      // wave-uniform
          const float sc = __uint_as_float(static_cast<unsigned>(127 + ex) << 23);
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(P[i][j][e], sc, acc[i][j][e]);
        } else {                                            // beyond a normal fp32 scale: two exact steps
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += ldexpf(P[i][j][e], ex);
        }
      }
#endif
  }
#if SG_X3_TIMING
  if (lane == 0 && wave == 0) {
#pragma unroll
    for (int q = 0; q < 6; ++q) atomicAdd(&g_x3_timing[q], tq[q]);
  }
#endif
  __syncthreads();       // every wave is done with the stages: they become the epilogue's staging blocks

  store_tile(g, acc, smem, wave, lane, tm * (64 * WM), tn * BN, z);
}

// ---------------------------------------------------------------------------------------------------------------------
// Hybrid: op(A) stays fp32 in memory and is split INSIDE the kernel; op(B) comes as planes by LDS-DMA.  For the products
// whose A operand is huge and is used by only one or two column tiles (N <= 256: the 16-21 GB R-expanded matrices of the
// config-5 step against a 256-wide weight or feature matrix) the streaming split pass would cost as much as the product
// (read 4 B + write 4 B per element, then the planes are read again); here A is read ONCE, as fp32.  Each of the four
// waves converts one 32-row block of the next K tile (32 x 32 values, 16 per lane) while the matrix pipe works on the
// current one: block maximum by a wave reduction -> exponent -> scale, two cvt_pk + one subtraction per pair of values,
// eight (K-contiguous A) or four (row-contiguous A) LDS stores into the fragment-major image the consumers read.  The
// scale block of A is 32 rows x 32 k here (one K tile), so the block-local products P are folded into the running result
// after every K tile; B keeps its 64-k blocks.  128 x 128 x 32 tiles, two stages, two workgroups per CU.
// ARC = false: A element (m, k) at A[m * lda + k] (K % 4 == 0, 16-byte aligned rows); true: at A[k * lda + m].
// ---------------------------------------------------------------------------------------------------------------------
// AV4 (row-contiguous A only): M % 4 == 0 and 16-byte aligned k rows -> float4 loads along m (four per K tile and lane, like
// the K-contiguous form) instead of sixteen dword loads.
// APF: K tiles of A in flight per wave.  1: tile t+1 is loaded while tile t multiplies (16 registers).  2: a second register
// set -- tile t+2 is requested at the top of iteration t, right after the DMA of B's tile t+1, and the barrier at the end of
// the iteration waits with a COUNTED vmcnt for everything but those youngest loads: A's HBM latency gets two tile times.
// (a third set -- 252 VGPRs -- measured no faster than two.)
template <bool ARC, bool AV4 = false, int APF = 1>
__global__ __launch_bounds__(256, 2) void gemm_f16x3h_kernel(const GemmArgs g, const PlaneArgs pl) {
  static_assert(APF == 1 || APF == 2, "one or two K tiles of A in flight");
  // launched as the FALLBACK of the 256-wide kernel (gemm_x3w.hip): nothing to do unless that kernel raised the flag
  if (pl.flag != nullptr && *reinterpret_cast<const volatile int*>(pl.flag) == 0) return;
  constexpr int UPB = 4, RBA = 4, STAGE_B = 32 * UNIT;      // per stage: A units 0..15, B units 16..31
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_B + 64];
  int* exp_lds = reinterpret_cast<int*>(smem + 2 * STAGE_B);               // [stage][row block]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = g.tiles_m * g.tiles_n;
  const int item = blockIdx.x;
  const int z = item / nt, lin = item - z * nt;
  const int q8 = nt >> 3, r8 = nt & 7, x8 = lin & 7;
  const int tile = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (lin >> 3);
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int ktiles = (g.K + 31) / 32;
  const int kt0 = z * g.tiles_per_split, kt1 = min(ktiles, kt0 + g.tiles_per_split);
  const int T = kt1 - kt0;

  f32x16 acc[2][2], P[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- B planes by DMA: 16 units per K tile, 4 per wave ----
  const long long rb_stride = static_cast<long long>(pl.KS) * 2 * UNIT;
  const char* b_base = pl.pb + static_cast<long long>(tn) * 4 * rb_stride + static_cast<long long>(kt0) * (UPB * UNIT) + lane * 16;
  auto issue_b_unit = [&](int kt, int i) {
    char* dst = smem + (kt & 1) * STAGE_B + 16 * UNIT;
    const int u = wave * 4 + i;
    const int q = u >> 2, part = u & 3;
    const char* src = b_base + q * rb_stride + static_cast<long long>(kt) * (UPB * UNIT) + part * UNIT;
    if (APF >= 2) {
      // assembly, so that the compiler keeps no record of a pending LDS write: before the plane stores of the conversion it
      // would otherwise wait with vmcnt(0) -- for this DMA, but also for A's younger loads, which it cannot see
      const unsigned lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_void*)(dst + u * UNIT)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is reserved: nothing else in this kernel uses it
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds) : "memory", "m0");
#pragma clang diagnostic pop
    } else {
      __builtin_amdgcn_global_load_lds((glb_void*)(src), (lds_void*)(dst + u * UNIT), 16, 0, 0);
    }
  };
  auto issue_b = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_b_unit(kt, i);
  };
  const int kbs = pl.KS >> 2;
  cst_int* eb_p = (cst_int*)(pl.exp_b + static_cast<long long>(tn * 4 + wn * 2) * kbs);

  // ---- A: this wave's 32-row block of a K tile, 16 fp32 per lane, loaded unconditionally from clamped coordinates ----
#if SG_X3H_ABLATE == 1
  const int m_blk = wave * 32;
#else
  const int m_blk = tm * BM + wave * 32;
#endif
  constexpr int NA = (ARC && !AV4) ? 16 : 4;             // vector-memory instructions of one load_a
  // APF == 2 (vector forms): the loads are inline assembly -- the compiler's own vmcnt before the conversion would be
  // vmcnt(0) (it does not carry exact counts around the loop), draining the younger loads of tile kt+2 too.  The registers
  // are pinned to the explicit counted wait (`wait_a`) as in/out operands, so no use can move above it.
  constexpr bool ASM_A = APF >= 2 && NA == 4;
  f32x4 va0[4], va1[4];
  auto ld4 = [&](f32x4& dst, const float* ptr) __attribute__((always_inline)) {
    if (ASM_A) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
    else dst = *reinterpret_cast<const f32x4*>(ptr);
  };
  auto wait_a = [&](f32x4 (&va)[4], auto n) __attribute__((always_inline)) {      // n: younger vector-memory operations
    constexpr int N = decltype(n)::value;
    static_assert(N == 0 || N == 8 || N == 16, "add the immediate");
    if (!ASM_A) return;
    if (N == 8) asm volatile("s_waitcnt vmcnt(8)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]) : : "memory");
    else if (N == 16) asm volatile("s_waitcnt vmcnt(16)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]) : : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(va[0]), "+v"(va[1]), "+v"(va[2]), "+v"(va[3]) : : "memory");
  };
  auto load_a = [&](int kt_in, f32x4 (&va)[4]) __attribute__((always_inline)) {
#if SG_X3H_ABLATE == 1          // development (timing only): every load of A hits the same 16 KB (L2-hot): is A's latency what binds?
    const int kt = (kt_in & 1) - kt0;
#else
    const int kt = kt_in;
#endif
    const int k0 = (kt0 + kt) * 32;
    if (!ARC) {           // lane = (row lane / 8 + 8 i, k = 4 (lane % 8) .. + 3)
      const int k = min(k0 + (lane & 7) * 4, g.K - 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = min(m_blk + (lane >> 3) + 8 * i, g.M - 1);
        ld4(va[i], g.A + static_cast<long long>(row) * g.lda + k);
      }
    } else if (AV4) {     // lane = (4 op rows m = 4 (lane % 8) .., k = 4 (lane / 8) + i): 128-byte k-row segments, va[4 i + j] = (k i, m j)
      const int m = min(m_blk + (lane & 7) * 4, g.M - 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = min(k0 + (lane >> 3) * 4 + i, g.K - 1);
        ld4(va[i], g.A + static_cast<long long>(k) * g.lda + m);
      }
    } else {              // lane = (op row m = lane % 32, k = 16 (lane / 32) + i): coalesced dword loads along m
      const int m = min(m_blk + (lane & 31), g.M - 1);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int k = min(k0 + (lane >> 5) * 16 + i, g.K - 1);
        va[i >> 2][i & 3] = g.A[static_cast<long long>(k) * g.lda + m];
      }
    }
  };
  auto store_a = [&](int kt, f32x4 (&va)[4]) __attribute__((always_inline)) {      // convert va (tile kt), write its planes + exponent into stage kt & 1
    const int k0 = (kt0 + kt) * 32;
    float x[16];
    float mx = 0.f;
    if (!ARC) {
      const bool kdead = k0 + (lane & 7) * 4 >= g.K;       // K % 4 == 0: a float4 is in or out as a whole
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool dead = kdead || (m_blk + (lane >> 3) + 8 * i >= g.M);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[4 * i + j] = dead ? 0.f : va[i][j]; mx = fmaxf(mx, fabsf(x[4 * i + j])); }
      }
    } else if (AV4) {     // transpose in registers: x[4 j + i] = (m j, k i) -> four consecutive k per op row, as in the K-contiguous form
      const bool mdead = m_blk + (lane & 7) * 4 >= g.M;       // M % 4 == 0: four rows are in or out together
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool dead = mdead || (k0 + (lane >> 3) * 4 + i >= g.K);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[4 * j + i] = dead ? 0.f : va[i][j]; mx = fmaxf(mx, fabsf(x[4 * j + i])); }
      }
    } else {
      const bool mdead = m_blk + (lane & 31) >= g.M;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        x[i] = (mdead || k0 + (lane >> 5) * 16 + i >= g.K) ? 0.f : va[i >> 2][i & 3];
        mx = fmaxf(mx, fabsf(x[i]));
      }
    }
    mx = wave_max_nonneg(mx);
    bool nonfinite = false;
    if (__builtin_expect(!(mx <= 3.402823466e38f), 0)) {      // wave-uniform, rare: an inf in the block -- see split_kernel
      nonfinite = true;
      float mf = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float a = fabsf(x[i]); mf = fmaxf(mf, a <= 3.402823466e38f ? a : 0.f); }
      mx = wave_max_nonneg(mf);
    }
    int e = 0;
    {
      const unsigned bits = __float_as_uint(mx);
      const int ex = static_cast<int>((bits >> 23) & 0xffu);
      if (bits != 0u && ex != 0xff) e = 14 - (max(ex, 1) - 127);
      e = min(max(e, -126), 126);
    }
    const float sc = __uint_as_float(static_cast<unsigned>(127 + e) << 23);
    unsigned h1[8], h2[8];
#if SG_X3H_ABLATE == 2
#pragma unroll
    for (int j = 0; j < 8; ++j) { h1[j] = __float_as_uint(x[2 * j]); h2[j] = __float_as_uint(x[2 * j + 1]); }
    if (false)
#endif
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x2 v = {x[2 * j] * sc, x[2 * j + 1] * sc};
      const f16x2 a = __builtin_convertvector(v, f16x2);
      // residual x * sc - h1 in one mixed-precision FMA per element (v_fma_mix_f32 reads the f16 half directly): the same
      // value as (x * sc) - float(h1) -- both are exact -- without the two conversions back to fp32
      f32x2 res = {__builtin_fmaf(x[2 * j], sc, -static_cast<float>(a[0])), __builtin_fmaf(x[2 * j + 1], sc, -static_cast<float>(a[1]))};
      if (nonfinite) {    // wave-uniform: an inf keeps a zero residual (inf - inf = NaN would poison the product's inf)
        if (fabsf(v[0]) == __builtin_inff()) res[0] = 0.f;
        if (fabsf(v[1]) == __builtin_inff()) res[1] = 0.f;
      }
      const f16x2 b = __builtin_convertvector(res, f16x2);
      h1[j] = __builtin_bit_cast(unsigned, a);
      h2[j] = __builtin_bit_cast(unsigned, b);
    }
    char* st = smem + (kt & 1) * STAGE_B + wave * (UPB * UNIT);
    if (!ARC || AV4) {    // 4 consecutive k of row r: half a 16-byte slot of unit (ks, plane), lane slot kg * 32 + r
      const int kq = AV4 ? (lane >> 3) : (lane & 7);
      const int off = ((kq >> 2) * 2) * UNIT + (((kq >> 1) & 1) * 32) * 16 + (kq & 1) * 8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = AV4 ? (lane & 7) * 4 + i : (lane >> 3) + 8 * i;
        *reinterpret_cast<uint2*>(st + off + r * 16) = make_uint2(h1[2 * i], h1[2 * i + 1]);
        *reinterpret_cast<uint2*>(st + off + UNIT + r * 16) = make_uint2(h2[2 * i], h2[2 * i + 1]);
      }
    } else {              // 16 consecutive k of op row m: k step ks = lane / 32, both k groups
      const int ks = lane >> 5, m = lane & 31;
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        *reinterpret_cast<uint4*>(st + (ks * 2) * UNIT + (kg * 32 + m) * 16) = make_uint4(h1[4 * kg], h1[4 * kg + 1], h1[4 * kg + 2], h1[4 * kg + 3]);
        *reinterpret_cast<uint4*>(st + (ks * 2 + 1) * UNIT + (kg * 32 + m) * 16) = make_uint4(h2[4 * kg], h2[4 * kg + 1], h2[4 * kg + 2], h2[4 * kg + 3]);
      }
    }
    if (lane == 0) exp_lds[(kt & 1) * 4 + wave] = -e;
  };

  if (T > 0) {
    issue_b(0);
    load_a(0, va0);
    wait_a(va0, std::integral_constant<int, 0>{});
    store_a(0, va0);
  }
  if (APF >= 2 && T > 1) load_a(1, va1);
  __syncthreads();
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // one K tile.  APF == 1: `vnext` receives tile kt+1 now and is converted at the end.  APF == 2: `vnext` already holds
  // tile kt+1 (requested one iteration ago), `vnew` receives tile kt+2.
  // `steady` (compile time): tiles kt+1 and kt+2 exist -- no branch around a vector-memory instruction, so the compiler's
  // own vmcnt before the conversion of `vnext` is exact (behind a branch it falls back to vmcnt(0), which would drain the
  // loads of tile kt+2 as well).
  auto body = [&](int kt, f32x4 (&vnew)[4], f32x4 (&vnext)[4], auto steady) __attribute__((always_inline)) {
    constexpr bool ST = decltype(steady)::value;
    const bool more = ST || kt + 1 < T;                    // wave-uniform
    const bool more2 = ST || (APF >= 2 && kt + APF < T);    // tile kt+APF exists: it is requested now
    if (APF >= 2) {
      if (more) issue_b(kt + 1);                           // program order: B's DMA first, A's loads after it (counted wait)
      asm volatile("" ::: "memory");
      if (more2) load_a(kt + APF, vnew);
    } else if (more) {
#if SG_X3H_ISSUE_POS == 0
      issue_b(kt + 1);                                     // the other stage was released by the barrier just passed
#endif
      load_a(kt + 1, vnext);
    }
    const char* st = smem + (kt & 1) * STAGE_B;
    // the tile's four block exponents are requested NOW (A's from LDS, written before the last barrier; B's from memory): at
    // the fold they used to cost a ds_read and an s_load round trip each, back to back, per K tile
    const int kb = (kt0 + kt) >> 1;
    const int ea_v0 = exp_lds[(kt & 1) * 4 + wm * 2], ea_v1 = exp_lds[(kt & 1) * 4 + wm * 2 + 1];
    const int eb_s[2] = {*(volatile cst_int*)(eb_p + kb), *(volatile cst_int*)(eb_p + kbs + kb)};   // volatile: not sunk to the fold
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      f16x8 a[2][2], b[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          a[i][p] = *reinterpret_cast<const f16x8*>(st + ((wm * 2 + i) * UPB + ks * 2 + p) * UNIT + lane * 16);
          b[i][p] = *reinterpret_cast<const f16x8*>(st + ((RBA + wn * 2 + i) * UPB + ks * 2 + p) * UNIT + lane * 16);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          P[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[j][0], ks == 0 ? zero : P[i][j], 0, 0, 0);
#if SG_X3H_ISSUE_POS == 3
      if (APF == 1) {
        asm volatile("" ::: "memory");
        if (more) issue_b_unit(kt + 1, ks * 2);
        asm volatile("" ::: "memory");
      }
#endif
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) P[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][1], P[i][j], 0, 0, 0);
#if SG_X3H_ISSUE_POS == 3
      if (APF == 1) {
        asm volatile("" ::: "memory");
        if (more) issue_b_unit(kt + 1, ks * 2 + 1);
        asm volatile("" ::: "memory");
      }
#endif
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) P[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[j][0], P[i][j], 0, 0, 0);
    }
    // fold the K tile at its true scale
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ea = __builtin_amdgcn_readfirstlane(i == 0 ? ea_v0 : ea_v1);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ex = ea + eb_s[j];
        if (ex >= -126 && ex <= 127) {
          const float sc = __uint_as_float(static_cast<unsigned>(127 + ex) << 23);
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(P[i][j][e], sc, acc[i][j][e]);
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] += ldexpf(P[i][j][e], ex);
        }
      }
    }
    if (more) {                             // the loads had the whole multiplication (APF 2: two of them) to land
      if (APF >= 2) {                       // younger than vnext's loads: per iteration since, B's DMA (4) and A's loads (4)
        if (ST) wait_a(vnext, std::integral_constant<int, 8 * (APF - 1)>{}); else wait_a(vnext, std::integral_constant<int, 0>{});
      }
      store_a(kt + 1, vnext);
    }
    if (APF >= 2) {
      // planes + exponent of tile kt+1 written (lgkmcnt), B's DMA of tile kt+1 landed: everything but the NA youngest
      // vector-memory operations -- the loads of tile kt+2 stay in flight across the barrier
      if (more2) wait_vm<NA>(); else wait_vm<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      __syncthreads();               // planes + exponent of tile kt+1 visible, B DMA landed (vmcnt(0) precedes the barrier)
    }
  };
  if (APF == 2) {
    int kt = 0;
    for (; kt + 3 < T; kt += 2) {                          // steady state: both iterations see tiles kt+1, kt+2 (and kt+3)
      body(kt, va0, va1, std::true_type{});
      body(kt + 1, va1, va0, std::true_type{});
    }
    for (; kt < T; kt += 2) {                              // the last two to three tiles
      body(kt, va0, va1, std::false_type{});
      if (kt + 1 < T) body(kt + 1, va1, va0, std::false_type{});
    }
  } else {
    for (int kt = 0; kt < T; ++kt) body(kt, va1, va0, std::false_type{});
  }
  store_tile(g, acc, smem, wave, lane, tm * BM, tn * BN, z);
}

// C^T (N x M, leading dimension M) -> C (M x N): the swapped-operand products of weight gradients
__global__ void transpose_out_kernel(float* __restrict__ C, long long ldc, const float* __restrict__ Ct, int M, int N) {
  __shared__ float tile[32][33];
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 256 threads: 8 rows per pass
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = (n0 + r < N && m0 + tx < M) ? Ct[static_cast<long long>(n0 + r) * M + m0 + tx] : 0.f;
  __syncthreads();
  for (int r = ty; r < 32; r += 8)
    if (m0 + r < M && n0 + tx < N) C[static_cast<long long>(m0 + r) * ldc + n0 + tx] = tile[tx][r];
}

static inline size_t align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

}  // namespace f16x3

// bytes of plane / exponent storage the backend needs for an M x N x K product (on top of the split-K partials)
size_t f16x3_plane_bytes(long long M, long long N, long long K) {
  using namespace f16x3;
  const long long Mp = (M + 255) / 256 * 256, Np = (N + 255) / 256 * 256, Kp = (K + 63) / 64 * 64;
  return align256(static_cast<size_t>(Mp) * Kp * 4) + align256(static_cast<size_t>(Np) * Kp * 4) +
         align256(static_cast<size_t>(Mp / 32) * (Kp / 64) * 4) + align256(static_cast<size_t>(Np / 32) * (Kp / 64) * 4) +
         (M <= 256 ? align256(static_cast<size_t>(M) * N * 4) : 0) +     // C^T of a swapped-operand product
         256;                                                             // the exactness flag of the 256-wide kernel
}

void launch_x3w_hybrid(const GemmArgs& g, const f16x3::PlaneArgs& pl, bool arc, int ctas, hipStream_t st);      // gemm_x3w.hip

constexpr int kWideCtas = 256;      // workgroups of the persistent 256-wide kernel: one per CU of an MI355X

// Does the 256-wide persistent kernel pay?  The chip is power-bound on these kernels: a launch that fills only half the CUs
// runs them at a higher clock and finishes the same work in about the same time (4096 x 256 x 1 M as 128 items on 256 CUs:
// 7.7 ms = 1.97 us per 32-k tile and workgroup, against 3.6 us with every CU busy).  Calibrated on the config-5 and the
// ML-10M products (tools/x3w_harness.cpp, profiles/r5_gemm_wide.md): a 256 x 256 x 32 tile takes 3.6 us with all CUs busy,
// 1.9 us unloaded; a 128 x 128 x 32 tile 2.3 us in each of a CU's two slots, 1.3 us unloaded.
static bool wide_pays(long long items_w, int tiles_per_item_w, long long items_o, int tiles_per_item_o) {
  // time of a launch = its full rounds at the loaded tile time + the last, partly filled round at a tile time interpolated
  // towards the unloaded one, plus per-round and per-launch costs that do not overlap the matrix work
  auto launch_us = [](long long items, int slots, int tiles, double t_full, double t_min, double per_round, double fixed) {
    const long long full = items / slots, rest = items % slots;
    const double part = rest ? t_min + (t_full - t_min) * static_cast<double>(rest) / slots : 0.0;
    return (static_cast<double>(full) * t_full + part) * tiles + static_cast<double>(full + (rest ? 1 : 0)) * per_round + fixed;
  };
  // 256 wide: 10 us per item for the unoverlapped epilogue, 15 us per launch (pipeline fill of the stream, fallback launch);
  // 128 wide: 6 us per round of items for the part of prologue + epilogue its three workgroups per CU do not hide
  const double tw = launch_us(items_w, kWideCtas, tiles_per_item_w, 3.6, 1.9, 10.0, 15.0);
  const double to = launch_us(items_o, 2 * kWideCtas, tiles_per_item_o, 2.3, 1.3, 6.0, 0.0);
  return tw < to;
}

// the persistent kernel's stream looks two K tiles ahead: every K slice must hold at least two 32-k tiles
static bool wide_slices_ok(const GemmArgs& g) {
  const int ktiles = (g.K + 31) / 32;
  if (g.splits <= 1) return ktiles >= 2;
  return g.tiles_per_split >= 2 && ktiles - (g.splits - 1) * g.tiles_per_split >= 2;
}

static std::atomic<int> g_x3_variant_override{-1};
static int x3_variant() {      // tuning aid: SG_X3_VARIANT / sg_gemm_x3_variant = 0 auto, 1 <2,2,2>, 2 <2,1,5>, 3 <4,2,3>, 8 the 256-wide hybrid at any size,
                               // 4 never hybrid, 5 hybrid at any size, 9 hybrid at any size with ONE K tile of A in flight (round-3 first form)
  const int o = g_x3_variant_override.load(std::memory_order_relaxed);
  if (o >= 0) return o;
  static const int v = [] { const char* e = getenv("SG_X3_VARIANT"); return e ? atoi(e) : 0; }();
  return v;
}

namespace f16x3 {
// C[m][n] = sum_z ws[z][n][m]: split-K partials of a swapped-operand product, summed in slice order and transposed
__global__ void reduce_t_kernel(float* __restrict__ C, long long ldc, const float* __restrict__ ws, int M, int N, int splits) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  const long long total = static_cast<long long>(M) * N;
  if (i >= total) return;
  const int n = static_cast<int>(i / M), m = static_cast<int>(i - static_cast<long long>(n) * M);      // coalesced reads
  float v = 0.f;
  for (int z = 0; z < splits; ++z) v += ws[static_cast<long long>(z) * total + i];
  C[static_cast<long long>(m) * ldc + n] = v;
}
}  // namespace f16x3

// the in-kernel-split kernel for A's layout (arc: row-contiguous in m; av4: float4 loads along m) and prefetch depth
static void launch_hybrid(const GemmArgs& g, const f16x3::PlaneArgs& pl, long long items, bool arc, bool av4, int deep,
                          hipStream_t st) {
  using namespace f16x3;
  const dim3 grid(static_cast<unsigned>(items)), block(256);
  if (deep == 2) {
    if (av4) hipLaunchKernelGGL((gemm_f16x3h_kernel<true, true, 2>), grid, block, 0, st, g, pl);
    else if (arc) hipLaunchKernelGGL((gemm_f16x3h_kernel<true, false, 2>), grid, block, 0, st, g, pl);
    else hipLaunchKernelGGL((gemm_f16x3h_kernel<false, false, 2>), grid, block, 0, st, g, pl);
  } else {
    if (av4) hipLaunchKernelGGL((gemm_f16x3h_kernel<true, true>), grid, block, 0, st, g, pl);
    else if (arc) hipLaunchKernelGGL((gemm_f16x3h_kernel<true>), grid, block, 0, st, g, pl);
    else hipLaunchKernelGGL((gemm_f16x3h_kernel<false>), grid, block, 0, st, g, pl);
  }
}

// called by sg_gemm_f32_hip when backend 3 is selected; g.tiles_n is for 128-wide tiles (g.tiles_m is recomputed here for
// the tile height chosen), g.tiles_per_split is EVEN when g.splits > 1 (a 64-k scale block = two K tiles must not straddle
// two slices), g.ws holds the split-K partials, `scratch` the plane storage (f16x3_plane_bytes).  *reduced is set when
// the split-K slices have already been combined here (swapped-operand products); *splits_used is the slice count written
// to g.ws (never more than the caller planned).
int launch_gemm_f16x3(const GemmArgs& g_in, bool transA, bool transB, char* scratch, hipStream_t st, bool* reduced,
                      int* splits_used) {
  using namespace f16x3;
  GemmArgs g = g_in;
  *reduced = false;
  *splits_used = g.splits;
  const long long Mp = (static_cast<long long>(g.M) + 255) / 256 * 256, Np = (static_cast<long long>(g.N) + 255) / 256 * 256;
  const long long Kp = (static_cast<long long>(g.K) + 63) / 64 * 64;
  const int KS = static_cast<int>(Kp / 16);
  if (g.splits > 1 && (g.tiles_per_split & 1)) return fail(SG_ERR_INVALID, "f16x3: odd K-tile count per split-K slice");
  int* flag = reinterpret_cast<int*>(scratch);        // exactness flag of the 256-wide kernel (cleared by the split pass)
  char* pa = scratch + 256;
  char* pb = pa + align256(static_cast<size_t>(Mp) * Kp * 4);
  int* ea = reinterpret_cast<int*>(pb + align256(static_cast<size_t>(Np) * Kp * 4));
  int* eb = reinterpret_cast<int*>(reinterpret_cast<char*>(ea) + align256(static_cast<size_t>(Mp / 32) * (Kp / 64) * 4));
  float* ct = reinterpret_cast<float*>(reinterpret_cast<char*>(eb) + align256(static_cast<size_t>(Np / 32) * (Kp / 64) * 4));
  if (Kp / 64 > 65535) return fail(SG_ERR_INVALID, "f16x3: K too large for the split grid");
  auto split = [&](char* planes, int* expo, const float* p, long long ld, bool rc, int R, long long Rp, int vec, int* clear) {
    const dim3 grid(static_cast<unsigned>(Rp / 32), static_cast<unsigned>(Kp / 64));
    if (rc) hipLaunchKernelGGL((split_kernel<true>), grid, dim3(256), 0, st, planes, expo, p, ld, R, g.K, KS, vec, clear);
    else hipLaunchKernelGGL((split_kernel<false>), grid, dim3(256), 0, st, planes, expo, p, ld, R, g.K, KS, vec, clear);
  };
  const int variant = x3_variant();
  const int tm128 = (g.M + 127) / 128, tn128 = (g.N + 127) / 128;
  // Split-K slice count against the number of workgroups the chip holds at once (`slots`): the caller's plan aims at ~3 per
  // CU for the 256-thread fp32 kernel; here 64 tiles x 12 slices = 768 workgroups on 512 slots run as 1.5 rounds (75 %
  // efficient).  Fewer slices are always allowed (the workspace was sized for the plan's count): take the count in
  // [splits / 2, splits] with the fullest rounds.
  auto fit_splits = [&](GemmArgs& a, long long tiles, int slots) {
    if (a.splits <= 1) return;
    const int ktiles = (a.K + 31) / 32;
    int best = a.splits;
    double best_eff = 0.0;
    for (int sp = a.splits; sp >= (a.splits + 1) / 2; --sp) {
      int per = (ktiles + sp - 1) / sp;
      per += per & 1;
      const int real = (ktiles + per - 1) / per;
      const long long items = tiles * real;
      const double eff = static_cast<double>(items) / (static_cast<double>((items + slots - 1) / slots) * slots);
      if (eff > best_eff + 1e-9) { best_eff = eff; best = real; }
    }
    int per = (ktiles + best - 1) / best;
    per += per & 1;
    a.tiles_per_split = per;
    a.splits = (ktiles + per - 1) / per;
  };
  // The in-kernel-split ("hybrid") forms, 128 wide (this file) or 256 wide and persistent (gemm_x3w.hip).  `h` is the
  // product as the kernel sees it (the swapped form exchanges the operands).  The wide kernel takes products that fill the
  // chip with 256 x 256 items; it is exact unless a scale block drops 2^60 below the slice's running scale -- then it raises
  // the flag and the 128-wide kernel, launched behind it with the same slices, redoes the product (it returns at once
  // otherwise).  variant 5 / 9: always 128 wide; 8: 256 wide at any size (tests).
  auto run_hybrid = [&](GemmArgs h, const PlaneArgs& pl_in, bool arc, bool av4, int t128m, int t128n) {
    PlaneArgs pl = pl_in;
    const int t256m = (h.M + 255) / 256, t256n = (h.N + 255) / 256;
    bool wide = variant != 5 && variant != 9 && (!arc || av4);
    if (wide) {
      GemmArgs w = h;
      w.tiles_m = t256m; w.tiles_n = t256n;
      fit_splits(w, static_cast<long long>(t256m) * t256n, kWideCtas);
      GemmArgs o = h;
      fit_splits(o, static_cast<long long>(t128m) * t128n, 512);
      const int ktiles = (h.K + 31) / 32;
      // accumulate with one K slice: the wide kernel adds into C in place, so a fallback run behind it would add the product a
      // second time -- those calls stay on the 128-wide kernel (with slices both kernels write partials and C is touched once)
      wide = wide_slices_ok(w) && !(h.accumulate && w.splits <= 1) &&
             (variant == 8 || wide_pays(static_cast<long long>(t256m) * t256n * w.splits, w.splits > 1 ? w.tiles_per_split : ktiles,
                                        static_cast<long long>(t128m) * t128n * o.splits, o.splits > 1 ? o.tiles_per_split : ktiles));
      if (wide) {
        h.splits = w.splits; h.tiles_per_split = w.tiles_per_split;      // the fallback uses the same slices
        pl.flag = flag;
        launch_x3w_hybrid(w, pl, arc, kWideCtas, st);
      }
    }
    h.tiles_m = t128m; h.tiles_n = t128n;
    if (!wide) fit_splits(h, static_cast<long long>(t128m) * t128n, 512);
    const long long items = static_cast<long long>(h.tiles_m) * h.tiles_n * h.splits;
    static const bool no_fallback = [] { const char* e = getenv("SG_X3_NOFALLBACK"); return e && atoi(e) != 0; }();      // development
    if (!(wide && no_fallback)) launch_hybrid(h, pl, items, arc, av4, variant == 9 ? 1 : 2, st);
    return h.splits;
  };
  const long long big = (variant == 5 || variant == 8 || variant == 9) ? 0 : 16ll << 20;  // elements: a 64 MB operand is worth keeping out of a split pass
  // A may stay fp32 when its layout allows the in-kernel loads (K-contiguous rows need 16-byte vectors)
  const bool a_fly_ok = transA || (g.vecA && g.K % 4 == 0 && g.K >= 4);
  const bool b_fly_ok = !transB || (g.vecB && g.K % 4 == 0 && g.K >= 4);
  const bool plain_epi = !g.bias && g.act == SG_ACT_NONE && !g.accumulate;
  if (variant != 4 && tn128 <= 2 && static_cast<long long>(g.M) * g.K >= big && a_fly_ok) {
    // ---- hybrid: A fp32 in the kernel, B planes ----
    split(pb, eb, g.B, g.ldb, !transB, g.N, Np, g.vecB, flag);
    PlaneArgs pl{nullptr, pb, nullptr, eb, KS, nullptr};
    *splits_used = run_hybrid(g, pl, transA, transA && g.vecA && g.M % 4 == 0, tm128, tn128);
    return SG_OK;
  }
  if (variant != 4 && tm128 <= 2 && g.M <= 256 && static_cast<long long>(g.N) * g.K >= big && b_fly_ok && plain_epi) {
    // ---- swapped hybrid: C^T = op(B)^T op(A)^T with op(B)^T (the huge operand) fp32 in the kernel, op(A)^T as planes ----
    split(pa, ea, g.A, g.lda, transA, g.M, Mp, g.vecA, flag);              // planes of op(A): rows m, K-contiguous units
    GemmArgs h = g;
    h.M = g.N; h.N = g.M;
    h.A = g.B; h.lda = g.ldb; h.vecA = g.vecB;
    h.C = ct; h.ldc = g.M;
    h.bias = nullptr;
    PlaneArgs pl{nullptr, pa, nullptr, ea, KS, nullptr};
    // op(B)^T element (n, k): B stored (K x N) when !transB -> row-contiguous in n (ARC); (N x K) when transB -> K-contiguous
    g.splits = run_hybrid(h, pl, !transB, !transB && h.vecA && h.M % 4 == 0, tn128, tm128);
    const long long total = static_cast<long long>(g.M) * g.N;
    if (g.splits > 1) {
      hipLaunchKernelGGL(reduce_t_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st, g.C, g.ldc, g.ws,
                         g.M, g.N, g.splits);
    } else {
      hipLaunchKernelGGL(transpose_out_kernel, dim3(static_cast<unsigned>((g.N + 31) / 32), static_cast<unsigned>((g.M + 31) / 32)),
                         dim3(256), 0, st, g.C, g.ldc, ct, g.M, g.N);
    }
    *splits_used = g.splits;
    *reduced = true;
    return SG_OK;
  }
  {     // both operands in ONE conversion launch
    const SplitOp sa{pa, ea, g.A, g.lda, g.M, g.vecA}, sb{pb, eb, g.B, g.ldb, g.N, g.vecB};
    const int rbA = static_cast<int>(Mp / 32);
    const dim3 grid(static_cast<unsigned>(rbA + Np / 32), static_cast<unsigned>(Kp / 64));
    const bool rca = transA, rcb = !transB;
    if (rca && rcb) hipLaunchKernelGGL((split2_kernel<true, true>), grid, dim3(256), 0, st, sa, sb, rbA, g.K, KS);
    else if (rca) hipLaunchKernelGGL((split2_kernel<true, false>), grid, dim3(256), 0, st, sa, sb, rbA, g.K, KS);
    else if (rcb) hipLaunchKernelGGL((split2_kernel<false, true>), grid, dim3(256), 0, st, sa, sb, rbA, g.K, KS);
    else hipLaunchKernelGGL((split2_kernel<false, false>), grid, dim3(256), 0, st, sa, sb, rbA, g.K, KS);
  }
  PlaneArgs pl{pa, pb, ea, eb, KS, nullptr};
  // variant: short K slices are dominated by the epilogue (two workgroups per CU overlap it); long ones by DMA latency
  const int ktiles32 = (g.K + 31) / 32;
  const int slice = g.splits > 1 ? g.tiles_per_split : ktiles32;
  // measured (tools/exp_x3_variants.py): three workgroups per CU (128 x 128 x 16, three stages, <= 168 VGPRs) beat the two-
  // workgroup and the 256-row geometries on every plane-path shape by 4-14 %: the third workgroup's matrix work fills the
  // barrier / DMA-wait gaps of the other two
  (void)slice;
  int v = ((variant >= 1 && variant <= 3) || variant == 6 || variant == 7) ? variant : 6;
  if (v == 3) {
    g.tiles_m = static_cast<int>((g.M + 255) / 256);
    const long long items = static_cast<long long>(g.tiles_m) * g.tiles_n * g.splits;
    hipLaunchKernelGGL((gemm_f16x3_kernel<4, 2, 3>), dim3(static_cast<unsigned>(items)), dim3(512), 0, st, g, pl);
  } else {
    g.tiles_m = tm128;
    const long long items = static_cast<long long>(g.tiles_m) * g.tiles_n * g.splits;
#if SG_X3_NOFOLD
    if (v == 2) hipLaunchKernelGGL((gemm_f16x3_kernel<2, 1, 2, 4>), dim3(static_cast<unsigned>(items)), dim3(256), 0, st, g, pl);
#else
    if (v == 2) hipLaunchKernelGGL((gemm_f16x3_kernel<2, 1, 5>), dim3(static_cast<unsigned>(items)), dim3(256), 0, st, g, pl);
#endif
    else if (v == 6) hipLaunchKernelGGL((gemm_f16x3_kernel<2, 1, 3, 3>), dim3(static_cast<unsigned>(items)), dim3(256), 0, st, g, pl);
    else if (v == 7) hipLaunchKernelGGL((gemm_f16x3_kernel<2, 1, 4, 2, true>), dim3(static_cast<unsigned>(items)), dim3(256), 0, st, g, pl);
    else hipLaunchKernelGGL((gemm_f16x3_kernel<2, 2, 2>), dim3(static_cast<unsigned>(items)), dim3(256), 0, st, g, pl);
  }
  return SG_OK;
}

}  // namespace sg

SG_API int sg_gemm_x3_variant(int variant) {
  sg::g_x3_variant_override.store(variant < 0 || variant > 9 ? -1 : variant, std::memory_order_relaxed);
  return SG_OK;
}

#if SG_X3_TIMING
// development build only (not declared in include/stargcn.h): read and reset the phase counters
extern "C" __attribute__((visibility("default"))) int sg_x3_timing_read(unsigned long long* out6) {
  unsigned long long z[6] = {0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out6, HIP_SYMBOL(sg::f16x3::g_x3_timing), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(sg::f16x3::g_x3_timing), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
