// gemm_x3_shared.hpp -- definitions shared by the f16x3 GEMM kernels (gemm_f16x3.hip: 128 x 128 tiles and the in-kernel-split
// "hybrid" forms; gemm_x3w.hip: 256 x 256 tiles, eight waves): argument blocks, the plane format constants, vector types,
// the wave maximum, the activation and the counted vector-memory wait.  See gemm_f16x3.hip for the arithmetic.
#pragma once
#include <type_traits>
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

namespace f16x3 {

constexpr int BM = 128, BN = 128;
constexpr int UNIT = 1024;                 // bytes one wave feeds to one MFMA operand: 64 lanes x 8 halves
constexpr int CPITCH = 68;                 // floats per row of a wave's private C staging block (32 x 64 + pad)
constexpr int CSTAGE = 32 * CPITCH * 4;    // 8704 B per wave

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

// maximum of a NON-NEGATIVE value over the 64 lanes of a wave, returned wave-uniform.  Six DPP steps on the vector ALU
// (xor 1, xor 2 inside quads, mirrors inside 8 and 16 lanes, then lane 15 / 31 broadcasts into the following rows: lane 63
// ends up with the maximum) instead of six dependent ds_bpermute round trips through the LDS pipe (~100+ cycles each
// under load, and each `s_waitcnt lgkmcnt(0)` also waits for every other LDS operation of the wave).  fmaxf drops NaNs,
// as the shuffle form did.
__device__ __forceinline__ float wave_max_nonneg(float f) {
  // on the BIT PATTERNS: non-negative floats order like unsigned integers (+inf above every finite value; the callers' fmaxf
  // chains have dropped NaNs), and v_max_u32 takes the DPP operand directly -- one instruction per step, where fmaxf on a DPP
  // move costs the move, the max and a canonicalising max
  unsigned v = __float_as_uint(f);
  auto step = [&](auto ctrl, auto row_mask) __attribute__((always_inline)) {
    const unsigned y = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), decltype(ctrl)::value,
                                                                         decltype(row_mask)::value, 0xf, true));
    v = v > y ? v : y;
  };
  using std::integral_constant;
  step(integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});     // quad_perm [1,0,3,2]
  step(integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});     // quad_perm [2,3,0,1]
  step(integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});    // row_half_mirror
  step(integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});    // row_mirror: every lane holds its row's maximum
  step(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});    // row_bcast:15 into rows 1 and 3
  step(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});    // row_bcast:31 into rows 2 and 3
  return __uint_as_float(static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63)));
}


__device__ __forceinline__ float act_fn(float v, int act, float slope) {
  switch (act) {
    case SG_ACT_LEAKY: return v > 0.f ? v : slope * v;
    case SG_ACT_RELU: return v > 0.f ? v : 0.f;
    case SG_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
    case SG_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// plane format: 16-byte unit u(rb, ks, plane, lane) = ((rb*KS + ks)*2 + plane)*64 + lane holds
// op(X)[row = 32 rb + (lane & 31)][k = 16 ks + 8 (lane >> 5) + 0..7] -- the operand layout of v_mfma_f32_32x32x16_f16
struct PlaneArgs {
  const char* pa;
  const char* pb;
  const int* exp_a;     // -log2(scale) of block (32-row block rb, 64-k block kb) at [rb * (KS / 4) + kb]
  const int* exp_b;
  int KS;               // 16-k steps per row block = Kp / 16 (a multiple of 4)
  int* flag;            // gemm_x3w.hip, direct-accumulation kernels: raised when a scale block lies too far below the running
                        // scale for the rescale form to be exact -- the launcher's fallback kernel then redoes the product
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef const __attribute__((address_space(4))) int cst_int;      // constant address space: scalar (s_load) access

template <int N>
__device__ __forceinline__ void wait_vm() {      // counted wait on this wave's own LDS-DMA / global loads
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else static_assert(N == 0, "add the immediate");
}


}  // namespace f16x3
}  // namespace sg
