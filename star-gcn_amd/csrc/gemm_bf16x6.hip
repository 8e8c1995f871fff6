// gemm_bf16x6.hip -- fp32-accurate GEMM on the bf16 matrix cores of gfx950 ("bf16x6" split).
//
// gfx950 has no TF32 path and its fp32-input MFMA runs at the vector rate (157 TFLOP/s, 1/16 of bf16).  For the
// dense mix of the multi-link aggregation at many rating levels (R = 10..16) that contraction, not HBM, bounds the
// step.  Here every fp32 operand x is split, on its way from global memory into LDS, into three bf16 planes
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (3 x 8 = 24 significand bits)
// and a product a*b is formed from the six plane pairs with i + j <= 4:
//     a*b ~= a1 b3 + a3 b1 + a2 b2 + a1 b2 + a2 b1 + a1 b1      (dropped terms <= 2^-24 |a b|, fp32-roundoff class)
// each pair being one v_mfma_f32_32x32x16_bf16 (exact bf16 products, fp32 accumulation).  Six bf16 MFMAs of K = 16
// (6 x 32 cycles) replace eight fp32 MFMAs of K = 2 (8 x 64 cycles): 2.7x fewer matrix-pipe cycles at fp32-class
// accuracy (the parity tests hold it to the same fp64-referenced tolerance as the fp32-MFMA kernel).
//
// Tile 128x128x32, 256 threads = 2x2 waves of 64x64 (2x2 MFMA tiles), LDS image [row][k] in bf16 with an 80-byte row
// stride (conflict-free ds_read_b128 operand fetches, 16 B = 8 consecutive k per lane), one LDS buffer with
// register-staged prefetch of the next K tile (60 KB -> 2 workgroups per CU).  Same epilogue / split-K contract as
// gemm_f32.hip (GemmArgs), selected by sg_gemm_f32_hip.
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

namespace bx6 {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int kThreads = 256;
constexpr int ROWB = 80;                 // bytes per LDS row: 32 bf16 (64 B) + 16 B pad
constexpr int PLANE = BM * ROWB;         // bytes per plane of one operand tile

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ float act_fn(float v, int act, float slope) {
  switch (act) {
    case SG_ACT_LEAKY: return v > 0.f ? v : slope * v;
    case SG_ACT_RELU: return v > 0.f ? v : 0.f;
    case SG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case SG_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// round-to-nearest-even fp32 -> bf16 (bit pattern in the low 16 bits) and back
__device__ __forceinline__ unsigned bf16_bits(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float bf16_val(unsigned b) { return __builtin_bit_cast(float, b << 16); }

// split 4 consecutive-k fp32 values into three planes of 4 packed bf16 (8 bytes each)
__device__ __forceinline__ void split4(const float (&x)[4], uint2& p1, uint2& p2, uint2& p3) {
  unsigned h[4], m[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    h[j] = bf16_bits(x[j]);
    const float r1 = x[j] - bf16_val(h[j]);
    m[j] = bf16_bits(r1);
    const float r2 = r1 - bf16_val(m[j]);
    l[j] = bf16_bits(r2);
  }
  p1 = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
  p2 = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
  p3 = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}

// ---- global -> registers.  Both layouts deliver, per thread, 4 groups of 4 CONSECUTIVE-k values of one tile row. ----
// K-contiguous operand (element (r,k) at p[r*ld + k]): group i = row (t/8 + 32 i), k = 4 (t%8) .. +3
__device__ __forceinline__ void gload_kc(float (&r)[4][4], const float* __restrict__ p, long long ld, int row0, int k0,
                                         int R, int K, bool fast, int t) {
  const int kc = (t & 7) * 4, rr = t >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + rr + 32 * i, k = k0 + kc;
    if (fast) {
      const float4 v = *reinterpret_cast<const float4*>(p + static_cast<long long>(row) * ld + k);
      r[i][0] = v.x; r[i][1] = v.y; r[i][2] = v.z; r[i][3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) r[i][j] = (row < R && k + j < K) ? p[static_cast<long long>(row) * ld + k + j] : 0.f;
    }
  }
}
// row-contiguous operand (element (k,c) at p[k*ld + c]): rows c = 4 (t%32) + j, k = 4 (t/32) + i  -> r[j][i]
__device__ __forceinline__ void gload_mc(float (&r)[4][4], const float* __restrict__ p, long long ld, int col0, int k0,
                                         int Ccols, int K, bool fast, int t) {
  const int cc = (t & 31) * 4, kg = (t >> 5) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = k0 + kg + i, c = col0 + cc;
    if (fast) {
      const float4 v = *reinterpret_cast<const float4*>(p + static_cast<long long>(k) * ld + c);
      r[0][i] = v.x; r[1][i] = v.y; r[2][i] = v.z; r[3][i] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) r[j][i] = (k < K && c + j < Ccols) ? p[static_cast<long long>(k) * ld + c + j] : 0.f;
    }
  }
}
// registers -> LDS planes: one 8-byte store per plane per group
__device__ __forceinline__ void sstore_kc(char* __restrict__ s, const float (&r)[4][4], int t) {
  const int kc = (t & 7) * 4, rr = t >> 3;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint2 p1, p2, p3;
    split4(r[i], p1, p2, p3);
    char* d = s + (rr + 32 * i) * ROWB + kc * 2;
    *reinterpret_cast<uint2*>(d) = p1;
    *reinterpret_cast<uint2*>(d + PLANE) = p2;
    *reinterpret_cast<uint2*>(d + 2 * PLANE) = p3;
  }
}
__device__ __forceinline__ void sstore_mc(char* __restrict__ s, const float (&r)[4][4], int t) {
  const int cc = (t & 31) * 4, kg = (t >> 5) * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint2 p1, p2, p3;
    split4(r[j], p1, p2, p3);
    char* d = s + (cc + j) * ROWB + kg * 2;
    *reinterpret_cast<uint2*>(d) = p1;
    *reinterpret_cast<uint2*>(d + PLANE) = p2;
    *reinterpret_cast<uint2*>(d + 2 * PLANE) = p3;
  }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(kThreads, 2) void gemm_bf16x6_kernel(const GemmArgs g) {
  __shared__ __attribute__((aligned(16))) char sA[3 * PLANE];
  __shared__ __attribute__((aligned(16))) char sB[3 * PLANE];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nwg = g.tiles_m * g.tiles_n, bid = blockIdx.x;
  const int q = nwg >> 3, rm = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + (bid >> 3);   // XCD-aware bijective remap
  const int tm = wg / g.tiles_n, tn = wg - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y;
  const int ktiles = (g.K + BK - 1) / BK;
  const int kt_begin = z * g.tiles_per_split, kt_end = min(ktiles, kt_begin + g.tiles_per_split);
  const bool full_mn = (m0 + BM <= g.M) && (n0 + BN <= g.N) && g.vecA && g.vecB;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float ra[4][4], rb[4][4];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    const bool fast = full_mn && (k0 + BK <= g.K);
    if (TA) gload_mc(ra, g.A, g.lda, m0, k0, g.M, g.K, fast, t); else gload_kc(ra, g.A, g.lda, m0, k0, g.M, g.K, fast, t);
    if (TB) gload_kc(rb, g.B, g.ldb, n0, k0, g.N, g.K, fast, t); else gload_mc(rb, g.B, g.ldb, n0, k0, g.N, g.K, fast, t);
  };
  auto sstore = [&]() {
    if (TA) sstore_mc(sA, ra, t); else sstore_kc(sA, ra, t);
    if (TB) sstore_kc(sB, rb, t); else sstore_mc(sB, rb, t);
  };

  const int l31 = lane & 31, kh = lane >> 5;
  if (kt_begin < kt_end) gload(kt_begin);
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    __syncthreads();            // previous tile's fragment reads are done
    sstore();
    __syncthreads();
    if (kt + 1 < kt_end) gload(kt + 1);   // in flight during the MFMAs below
    const char* pa = sA + (wm * 64 + l31) * ROWB + kh * 16;
    const char* pb = sB + (wn * 64 + l31) * ROWB + kh * 16;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[i][p] = *reinterpret_cast<const bf16x8*>(pa + i * 32 * ROWB + p * PLANE + ks * 32);
          b[i][p] = *reinterpret_cast<const bf16x8*>(pb + i * 32 * ROWB + p * PLANE + ks * 32);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];   // smallest terms first
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
          acc[i][j] = c;
        }
    }
  }

  const bool partial = (g.splits > 1);
  float* out = partial ? g.ws + static_cast<long long>(z) * g.M * g.N : g.C;
  const long long ldo = partial ? g.N : g.ldc;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + l31;
      if (col >= g.N) continue;
      const float bv = (!partial && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
        if (row >= g.M) continue;
        float v = acc[i][j][e];
        float* o = out + static_cast<long long>(row) * ldo + col;
        if (!partial) {
          v += bv;
          if (g.accumulate) v += *o;
          v = act_fn(v, g.act, g.slope);
        }
        *o = v;
      }
    }
}

}  // namespace bx6

// launched by sg_gemm_f32_hip (gemm_f32.hip) when the bf16x6 backend is selected; tiles are 128x128
void launch_gemm_bf16x6(const GemmArgs& g, bool transA, bool transB, hipStream_t st) {
  dim3 grid(static_cast<unsigned>(g.tiles_m * g.tiles_n), static_cast<unsigned>(g.splits));
  if (transA) {
    if (transB) hipLaunchKernelGGL((bx6::gemm_bf16x6_kernel<true, true>), grid, dim3(bx6::kThreads), 0, st, g);
    else hipLaunchKernelGGL((bx6::gemm_bf16x6_kernel<true, false>), grid, dim3(bx6::kThreads), 0, st, g);
  } else {
    if (transB) hipLaunchKernelGGL((bx6::gemm_bf16x6_kernel<false, true>), grid, dim3(bx6::kThreads), 0, st, g);
    else hipLaunchKernelGGL((bx6::gemm_bf16x6_kernel<false, false>), grid, dim3(bx6::kThreads), 0, st, g);
  }
}

}  // namespace sg
