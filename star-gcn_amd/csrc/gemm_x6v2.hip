// gemm_x6v2.hip -- fp32-accurate GEMM on the bf16 matrix cores of gfx950, wave-specialised and persistent.
//
// Same arithmetic as gemm_bf16x6.hip (every fp32 operand split into three bf16 planes, a product formed from the six
// plane pairs with i + j <= 4, v_mfma_f32_32x32x16_bf16 with fp32 accumulation: dropped terms <= 2^-24 |a b|), but
// organised for the matrix pipe instead of around it.  What limited the first version (profiles/) was not
// the arithmetic: one LDS buffer with two barriers per K tile, the split VALU work and the matrix work of a workgroup
// running in lock-step (VALU idle while the MFMAs run and vice versa), and a prologue / epilogue bubble per 128x128
// tile that the K = 256 shapes of the step cannot amortise.  Here:
//
//   * 512 threads = 8 waves per workgroup, ONE workgroup per CU (2 x 60 KB LDS stages).  Waves 0-3 are CONSUMERS: each
//     owns a 64x64 sub-tile (2x2 MFMA tiles, 64 accumulators), reads bf16 fragments with ds_read_b128 and issues
//     nothing but LDS reads and MFMAs.  Waves 4-7 are PRODUCERS: global loads (two register staging sets, loads run two
//     K tiles ahead), the fp32 -> 3 x bf16 split (pure VALU) and the LDS stores of the NEXT stage.  One producer and
//     one consumer wave share each SIMD, so the VALU and the matrix pipe work at the same time by construction.
//   * double-buffered LDS, ONE barrier per K tile.
//   * persistent workgroups walk a flat list of work items (tile x split-K slice): the producers run ahead across item
//     boundaries, so the first K tiles of the next item are already in LDS while the consumers store the finished tile.
//   * row-contiguous operands (A^T of the weight gradients, B of the data gradients) are loaded with coalesced dword
//     loads that leave every thread with 16 consecutive k of ONE tile row: the bf16 planes are then written with
//     16-byte LDS stores at the conflict-free 80-byte row pitch (the first version's transposing 8-byte stores hit
//     two banks with all 32 lanes).
//
// Epilogue / split-K contract identical to gemm_f32.hip (GemmArgs); selected by sg_gemm_f32_hip.
#include "common.hpp"

namespace sg {

struct GemmArgs {   // must match gemm_f32.hip
  float* C;
  const float* A;
  const float* B;
  const float* bias;
  float* ws;
  long long lda, ldb, ldc;
  int M, N, K;
  int act;
  float slope;
  int accumulate;
  int splits, tiles_per_split;
  int tiles_m, tiles_n;
  int vecA, vecB;
};

#ifndef SG_X6V2_NT_STORE
#define SG_X6V2_NT_STORE 1    // C tiles are written once and never re-read by this kernel: stream them past the caches
#endif
#ifndef SG_X6V2_ABLATE
#define SG_X6V2_ABLATE 0      // development: 1 producers skip split + LDS stores, 2 consumers skip MFMAs, 3 consumers skip LDS reads,
                              // 4 two planes and three products per K step (cost model of a 3-product split scheme)
#endif

#ifndef SG_X6V2_TIMING
#define SG_X6V2_TIMING 0      // development: per-phase cycle counters of consumer wave 0 / producer wave 4 (sg_x6v2_timing_read)
#endif

namespace x6v2 {

#if SG_X6V2_TIMING
// [0] consumer barrier wait  [1] consumer LDS reads + MFMA  [2] consumer epilogue  [3] consumer steps
// [4] producer split + LDS stores  [5] producer global-load issue  [6] producer barrier wait  [7] producer steps
__device__ unsigned long long g_timing[8];
#define SG_T(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define SG_T(var)
#endif

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int kThreads = 512;
constexpr int kRole = 256;               // threads per role
constexpr int ROWB = 80;                 // bytes per LDS row: 32 bf16 (64 B) + 16 B pad -> conflict-free b128 accesses
constexpr int PLANE = BM * ROWB;         // one bf16 plane of one operand tile
constexpr int OPER = 3 * PLANE;
constexpr int STAGE = 2 * OPER;          // A planes then B planes
constexpr int CPITCH = 68;               // floats per row of a consumer wave's private C staging block (32 x 64 + pad)
constexpr int CSTAGE = 32 * CPITCH * 4;  // bytes per consumer wave

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// the six plane pairs (i + j <= 4) of a product, smallest terms first
__device__ constexpr int kPA[6] = {0, 2, 1, 0, 1, 0};
__device__ constexpr int kPB[6] = {2, 0, 1, 1, 0, 0};

__device__ __forceinline__ float act_fn(float v, int act, float slope) {
  switch (act) {
    case SG_ACT_LEAKY: return v > 0.f ? v : slope * v;
    case SG_ACT_RELU: return v > 0.f ? v : 0.f;
    case SG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case SG_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// two consecutive-k fp32 values -> one packed bf16 pair per plane: x = p1 + p2 + p3 with v_cvt_pk_bf16_f32 (round to
// nearest even) at every level; the residuals are exact fp32 subtractions
__device__ __forceinline__ void split2(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
  f32x2 v = {x0, x1};
  p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  f32x2 r = {v.x - __builtin_bit_cast(float, p1 << 16), v.y - __builtin_bit_cast(float, p1 & 0xffff0000u)};
  p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  f32x2 q = {r.x - __builtin_bit_cast(float, p2 << 16), r.y - __builtin_bit_cast(float, p2 & 0xffff0000u)};
  p3 = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2));
}

// ---- producer side: one operand tile (128 rows x 32 k) per K tile, 16 fp32 values per thread -----------------------
// Every load is issued UNCONDITIONALLY with clamped coordinates (rows / columns beyond the matrix only feed C entries that
// are never stored; k beyond K is zeroed at store time when K % 32 != 0): a fixed number of load instructions per step is
// what lets the compiler keep the loads of the next two K tiles in flight (a data-dependent count degrades every wait
// to vmcnt(0), i.e. exposes the full global-load latency once per K tile -- measured: 83 instead of > 160 TFLOP/s).
// K-contiguous operand (element (r,k) at p[r*ld + k], K % 4 == 0): thread pt holds rows (pt/8 + 32 i), k = 4 (pt%8) .. +3
__device__ __forceinline__ void gload_kc(float (&v)[16], const float* __restrict__ p, long long ld, int row0, int k0, int R,
                                         int K, int pt) {
  const int k = min(k0 + (pt & 7) * 4, K - 4), rr = row0 + (pt >> 3);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = min(rr + 32 * i, R - 1);
    const float4 x = *reinterpret_cast<const float4*>(p + static_cast<long long>(row) * ld + k);
    v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w;
  }
}
template <bool KTAIL>
__device__ __forceinline__ void sstore_kc(char* __restrict__ s, float (&v)[16], int k0, int K, int pt) {
  const int kc = (pt & 7) * 4, rr = pt >> 3;
  const bool dead = KTAIL && (k0 + kc >= K);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned a1, a2, a3, b1, b2, b3;
    split2(dead ? 0.f : v[4 * i], dead ? 0.f : v[4 * i + 1], a1, a2, a3);
    split2(dead ? 0.f : v[4 * i + 2], dead ? 0.f : v[4 * i + 3], b1, b2, b3);
    char* d = s + (rr + 32 * i) * ROWB + kc * 2;
    *reinterpret_cast<uint2*>(d) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(a2, b2);
#if SG_X6V2_ABLATE != 4
    *reinterpret_cast<uint2*>(d + 2 * PLANE) = make_uint2(a3, b3);
#endif
  }
}
// row-contiguous operand (element (k,c) at p[k*ld + c]): thread pt holds tile row c = pt % 128, k = 16 (pt/128) + i -> v[i]
// every load instruction of a wave reads 64 consecutive floats of one k row (coalesced dword loads)
__device__ __forceinline__ void gload_rc(float (&v)[16], const float* __restrict__ p, long long ld, int col0, int k0, int Ccols,
                                         int K, int pt) {
  const int c = min(col0 + (pt & 127), Ccols - 1), kb = k0 + (pt >> 7) * 16;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = p[static_cast<long long>(min(kb + i, K - 1)) * ld + c];
}
template <bool KTAIL>
__device__ __forceinline__ void sstore_rc(char* __restrict__ s, float (&v)[16], int k0, int K, int pt) {
  char* d = s + (pt & 127) * ROWB + (pt >> 7) * 32;
  const int kb = k0 + (pt >> 7) * 16;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    unsigned p1[4], p2[4], p3[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = 8 * h + 2 * j;
      const float x0 = (KTAIL && kb + i >= K) ? 0.f : v[i], x1 = (KTAIL && kb + i + 1 >= K) ? 0.f : v[i + 1];
      split2(x0, x1, p1[j], p2[j], p3[j]);
    }
    *reinterpret_cast<uint4*>(d + 16 * h) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
    *reinterpret_cast<uint4*>(d + 16 * h + PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
#if SG_X6V2_ABLATE != 4
    *reinterpret_cast<uint4*>(d + 16 * h + 2 * PLANE) = make_uint4(p3[0], p3[1], p3[2], p3[3]);
#endif
  }
}

// position of a workgroup in its flat list of work items; item = blockIdx.x + j * gridDim.x; item -> (tile, split slice)
struct Cursor {
  int item, kt, kt_end, m0, n0;
  bool valid;
};
__device__ __forceinline__ void cursor_set(Cursor& c, const GemmArgs& g, int item, int n_items, int ktiles) {
  c.item = item;
  c.valid = item < n_items;
  if (!c.valid) return;
  const int nt = g.tiles_m * g.tiles_n;
  const int z = item / nt, lin = item - z * nt;
  // XCD-aware bijective remap of the tile index (hardware places workgroup b on XCD b % 8; a persistent workgroup keeps
  // its XCD): tiles that share an A row panel are handled by workgroups of the same XCD
  const int q = nt >> 3, r = nt & 7, x = lin & 7;
  const int tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (lin >> 3);
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  c.m0 = tm * BM;
  c.n0 = tn * BN;
  c.kt = z * g.tiles_per_split;
  c.kt_end = min(ktiles, c.kt + g.tiles_per_split);
  (void)z;
}
__device__ __forceinline__ void cursor_next(Cursor& c, const GemmArgs& g, int n_items, int ktiles, int stride) {
  if (!c.valid) return;
  if (++c.kt >= c.kt_end) cursor_set(c, g, c.item + stride, n_items, ktiles);
}

// NSETS: register staging sets of the producers = how many K tiles of global loads are in flight (measured L2 / HBM load
// latency under this kernel's own traffic is about 2 us = three K-tile periods).  Bounded by the 6-bit vmcnt counter:
// a set is 8 (both operands K-contiguous), 20 or 32 load instructions.
template <bool TA, bool TB, bool KTAIL, int NSETS>
__global__ __launch_bounds__(kThreads) void gemm_x6v2_kernel(const GemmArgs g, int n_items) {
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + 4 * CSTAGE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool consumer = wave < 4;
  const int pt = t & (kRole - 1);
  const int ktiles = (g.K + BK - 1) / BK;
  const int stride = gridDim.x;

  // number of (item, K tile) steps of this workgroup: every thread needs it to run the same number of barriers
  long long total = 0;
  {
    const int per = g.tiles_per_split;
    for (int item = blockIdx.x; item < n_items; item += stride) {
      const int z = item / (g.tiles_m * g.tiles_n);
      const int b = z * per;
      total += max(0, min(ktiles, b + per) - b);
    }
  }
  if (total == 0) return;
  constexpr int kUnroll = (NSETS % 2 == 0) ? NSETS : 2 * NSETS;   // steps per producer-loop iteration (stage parity repeats)
  const long long total2 = (total + kUnroll - 1) / kUnroll * kUnroll;   // whole iterations only: no exit in mid-iteration

  if (consumer) {
    // ---------------------------------------------------------------------------------------------- consumers ----
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, kh = lane >> 5;
    // The consumer's MFMAs and its partner producer's VALU stream are arbitrated on one SIMD issue port by priority,
    // then age.  At equal priority the producer's ~270 independent VALU instructions per K tile win every slot and the
    // MFMAs only start once the burst is over (PMC: matrix pipe busy 44 %, VALU + MFMA issue time adds up to the step
    // time).  With the consumer at a higher static priority an MFMA takes its slot as soon as the pipe frees up and the
    // VALU work fills the 32-cycle shadows in between.
    __builtin_amdgcn_s_setprio(3);
    // Two accumulator sets: `acc` takes the leading product a1 b1 of every K step, `accs` the five correction terms
    // (2^-8 .. 2^-16 of it).  Adding the corrections straight into the large running sum would round each of them at the
    // ulp of the LARGE value -- six roundings per K step instead of one, which showed as 3x the exact-fp32 kernel's error
    // on a cancelling bias gradient; in their own accumulator their roundings are 2^-8 smaller and the leading sum is
    // rounded once per step, like a plain fp32-accumulating MFMA.  The two are added once, in the epilogue.
    f32x16 acc[2][2], accs[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accs[i][j][e] = 0.f; }
    Cursor c;
    cursor_set(c, g, blockIdx.x, n_items, ktiles);
    while (c.valid && c.kt >= c.kt_end) cursor_set(c, g, c.item + stride, n_items, ktiles);
    const int a_off = (wm * 64 + l31) * ROWB + kh * 16;
    const int b_off = OPER + (wn * 64 + l31) * ROWB + kh * 16;
#if SG_X6V2_TIMING
    unsigned long long tc_wait = 0, tc_work = 0, tc_epi = 0, tc_steps = 0;
#endif
    for (long long s = 0; s < total2; ++s) {
      SG_T(t_a);
      __syncthreads();               // stage (s & 1) holds step s; the producers go on to fill the other stage
      if (s >= total) break;         // odd number of steps: the pairing barrier only
      SG_T(t_b);
      const char* st = smem + (s & 1) * STAGE;
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < (SG_X6V2_ABLATE == 4 ? 2 : 3); ++p) {
#if SG_X6V2_ABLATE != 3
            a[i][p] = *reinterpret_cast<const bf16x8*>(st + a_off + i * 32 * ROWB + p * PLANE + ks * 32);
            b[i][p] = *reinterpret_cast<const bf16x8*>(st + b_off + i * 32 * ROWB + p * PLANE + ks * 32);
#else
            a[i][p] = __builtin_bit_cast(bf16x8, make_uint4(s, ks, i, p));
            b[i][p] = __builtin_bit_cast(bf16x8, make_uint4(p, i, ks, s));
#endif
          }
        // plane pairs ordered smallest terms first; the four accumulators interleave so that consecutive MFMAs never
        // depend on each other
#if SG_X6V2_ABLATE == 4
#pragma unroll
        for (int term = 3; term < 5; ++term) {     // a1 b2, a2 b1
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kPA[term]], b[j][kPB[term]], accs[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
#elif SG_X6V2_ABLATE != 2
#pragma unroll
        for (int term = 0; term < 5; ++term) {     // corrections, smallest first
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kPA[term]], b[j][kPB[term]], accs[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
#else
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int p = 0; p < 3; ++p) accs[i][0][p] += static_cast<float>(a[i][p][0]) + static_cast<float>(b[i][p][0]);
#endif
      }
#if SG_X6V2_TIMING
      // the last MFMA result is needed before the clock is read: make the MFMA section's time its execution time
      asm volatile("s_nop 0" :: "v"(accs[1][1][0]), "v"(acc[1][1][0]));
#endif
      SG_T(t_c);
      if (c.kt + 1 >= c.kt_end) {
        // ---- epilogue of this item.  The MFMA result layout (col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 kh) gives a
        // lane ONE column: stored directly, every wave instruction writes 128-byte pieces of 64 different rows, and at
        // small K the GEMM is bound by exactly those writes (measured: C written at 1.3 TB/s).  Each wave therefore
        // turns its 32 x 64 blocks around in a private LDS block and writes 16 bytes per lane, 256 contiguous bytes per row.
        const bool partial = (g.splits > 1);
        const int z = c.item / (g.tiles_m * g.tiles_n);
        float* out = partial ? g.ws + static_cast<long long>(z) * g.M * g.N : g.C;
        const long long ldo = partial ? g.N : g.ldc;
        const bool vec_c = ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && ((ldo & 3) == 0);
        float* cst = reinterpret_cast<float*>(smem + 2 * STAGE + wave * CSTAGE);
        const int c4 = (lane & 15) * 4, r4 = lane >> 4;
        const int col = c.n0 + wn * 64 + c4;
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
            for (int e = 0; e < 16; ++e) {
              cst[((e & 3) + 8 * (e >> 2) + 4 * kh) * CPITCH + j * 32 + l31] = acc[i][j][e] + accs[i][j][e];
              acc[i][j][e] = 0.f;
              accs[i][j][e] = 0.f;
            }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int it = 0; it < 8; ++it) {
            const int r = r4 + 4 * it;
            const int row = c.m0 + wm * 64 + i * 32 + r;
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
                const f4 t = {v[0], v[1], v[2], v[3]};
                // a finished C tile is not re-read by this kernel: streamed past the caches (K = 256, 262 144 x 4 096:
                // 106 -> 122 TFLOP/s); split-K partials ARE re-read at once by the reduce kernel and stay cacheable
                if (SG_X6V2_NT_STORE && !partial) __builtin_nontemporal_store(t, reinterpret_cast<f4*>(o));
                else *reinterpret_cast<f4*>(o) = t;
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
      cursor_next(c, g, n_items, ktiles, stride);
#if SG_X6V2_TIMING
      { SG_T(t_d); tc_wait += t_b - t_a; tc_work += t_c - t_b; tc_epi += t_d - t_c; ++tc_steps; }
#endif
    }
#if SG_X6V2_TIMING
    if (t == 0) {
      atomicAdd(&g_timing[0], tc_wait); atomicAdd(&g_timing[1], tc_work); atomicAdd(&g_timing[2], tc_epi);
      atomicAdd(&g_timing[3], tc_steps);
    }
#endif
  } else {
    // ---------------------------------------------------------------------------------------------- producers ----
    float va[NSETS][16], vb[NSETS][16];           // staging sets: the tile of step s lives in set (s % NSETS)
    int k0s[NSETS];                               // first k of the tile held by each set (K-tail masking)
    Cursor ld;      // next step to LOAD
    cursor_set(ld, g, blockIdx.x, n_items, ktiles);
    while (ld.valid && ld.kt >= ld.kt_end) cursor_set(ld, g, ld.item + stride, n_items, ktiles);
    int lm0 = ld.m0, ln0 = ld.n0, lk0 = ld.kt * BK;     // coordinates of the next load (held at the last valid step)
    auto gload = [&](float (&va)[16], float (&vb)[16], int& k0_set) {
      if (TA) gload_rc(va, g.A, g.lda, lm0, lk0, g.M, g.K, pt); else gload_kc(va, g.A, g.lda, lm0, lk0, g.M, g.K, pt);
      if (TB) gload_kc(vb, g.B, g.ldb, ln0, lk0, g.N, g.K, pt); else gload_rc(vb, g.B, g.ldb, ln0, lk0, g.N, g.K, pt);
      k0_set = lk0;
      cursor_next(ld, g, n_items, ktiles, stride);
      if (ld.valid) { lm0 = ld.m0; ln0 = ld.n0; lk0 = ld.kt * BK; }     // scalar bookkeeping only: no load is conditional
      // opaque to the optimiser: past the last step the coordinates stop changing, and the compiler would otherwise
      // prove the next loads redundant, make them conditional and lose the load count again
      asm volatile("" : "+s"(lm0), "+s"(ln0), "+s"(lk0));
    };
    auto sstore = [&](int stage, float (&va)[16], float (&vb)[16], int k0_set) {
      char* st = smem + stage * STAGE;
#if SG_X6V2_ABLATE == 1
      if (va[0] + vb[0] + va[15] + vb[15] + va[7] + vb[9] == 1.2345e-30f) *reinterpret_cast<float*>(st) = 1.f;   // keep the loads
      return;
#endif
      if (TA) sstore_rc<KTAIL>(st, va, k0_set, g.K, pt); else sstore_kc<KTAIL>(st, va, k0_set, g.K, pt);
      if (TB) sstore_kc<KTAIL>(st + OPER, vb, k0_set, g.K, pt); else sstore_rc<KTAIL>(st + OPER, vb, k0_set, g.K, pt);
    };
    // Steady state, identical on the entry edge and on the back edge of the loop (so the compiler's wait counts are
    // exact): the loads of NSETS steps are in flight, the oldest set is converted and stored, then refilled.
    //   producers:  store(s) -> stage0 | B | store(s+1) -> stage1 | B | store(s+2) -> stage0 | B ...
    //   consumers:                       B | multiply stage0        B | multiply stage1         B ...
    // A stage is rewritten only after the barrier that follows its multiplication.  Past the last step the producers
    // keep converting / loading clamped data that nobody reads: cheaper than a data-dependent load count.
    // (a mid-loop exit gives the loop a second path to its header on which fewer sets are in flight, and the compiler
    //  then waits for ALL sets at the top of every iteration -- total2 keeps the body branch-free)
#pragma unroll
    for (int u = 0; u < NSETS; ++u) gload(va[u], vb[u], k0s[u]);      // steps 0 .. NSETS-1
#if SG_X6V2_TIMING
    unsigned long long tp_store = 0, tp_load = 0, tp_wait = 0, tp_steps = 0;
#endif
    for (long long s = 0; s < total2; s += kUnroll) {
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        SG_T(p_a);
        sstore(u & 1, va[u % NSETS], vb[u % NSETS], k0s[u % NSETS]);  // step s + u
        SG_T(p_b);
        gload(va[u % NSETS], vb[u % NSETS], k0s[u % NSETS]);          // step s + u + NSETS
        SG_T(p_c);
        __syncthreads();
#if SG_X6V2_TIMING
        { SG_T(p_d); tp_store += p_b - p_a; tp_load += p_c - p_b; tp_wait += p_d - p_c; ++tp_steps; }
#endif
      }
    }
#if SG_X6V2_TIMING
    if (t == kRole) {
      atomicAdd(&g_timing[4], tp_store); atomicAdd(&g_timing[5], tp_load); atomicAdd(&g_timing[6], tp_wait);
      atomicAdd(&g_timing[7], tp_steps);
    }
#endif
  }
}

}  // namespace x6v2

int x6v2_items(const GemmArgs& g) { return g.tiles_m * g.tiles_n * g.splits; }

#if SG_X6V2_TIMING
// development build only (not declared in include/stargcn.h): read and reset the phase counters
extern "C" __attribute__((visibility("default"))) int sg_x6v2_timing_read(unsigned long long* out8) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(x6v2::g_timing), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(x6v2::g_timing), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif

// x6v2 handles 16-byte aligned operands whose K-contiguous dimension is a multiple of 4 (everything the step produces);
// sg_gemm_f32_hip falls back to the exact-fp32 kernel otherwise
bool x6v2_supported(const GemmArgs& g, bool transA, bool transB) {
  if (!g.vecA || !g.vecB || g.K < 4 || g.M < 1 || g.N < 1) return false;
  if ((!transA || transB) && (g.K % 4 != 0)) return false;     // a K-contiguous operand is loaded as float4 along k
  return true;
}

// launched by sg_gemm_f32_hip (gemm_f32.hip) when the x6v2 backend is selected; tiles are 128 x 128, g.tiles_m / tiles_n
// must have been computed for them
void launch_gemm_x6v2(const GemmArgs& g, bool transA, bool transB, int n_cus, hipStream_t st) {
  const int n_items = x6v2_items(g);
  const int grid_x = n_items < n_cus ? n_items : n_cus;      // persistent: at most one workgroup per CU
  dim3 grid(static_cast<unsigned>(grid_x));
  const bool ktail = (g.K % x6v2::BK) != 0;
#define SG_X6V2(TA_, TB_, NS_)                                                                                              \
  do {                                                                                                                     \
    if (ktail) hipLaunchKernelGGL((x6v2::gemm_x6v2_kernel<TA_, TB_, true, NS_>), grid, dim3(x6v2::kThreads), 0, st, g, n_items); \
    else hipLaunchKernelGGL((x6v2::gemm_x6v2_kernel<TA_, TB_, false, NS_>), grid, dim3(x6v2::kThreads), 0, st, g, n_items);      \
  } while (0)
  // loads per set: NT 8, NN / TT 20, TN 32; at most 63 may be outstanding per wave
  if (transA) {
    if (transB) SG_X6V2(true, true, 3); else SG_X6V2(true, false, 2);
  } else {
    if (transB) SG_X6V2(false, true, 4); else SG_X6V2(false, false, 3);
  }
#undef SG_X6V2
}

}  // namespace sg
