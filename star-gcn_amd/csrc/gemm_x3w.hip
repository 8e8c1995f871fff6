// gemm_x3w.hip -- the f16x3 product (gemm_f16x3.hip: fp32-accurate GEMM on the f16 matrix cores, three matrix
// instructions per product on block-scaled f16 plane pairs) on 256 x 256 tiles with EIGHT waves, one workgroup per CU.
//
// Why.  The 128 x 128 kernels of gemm_f16x3.hip stop at ~0.9 PFLOP/s of f16 matrix-instruction issue: per
// 16-k step a wave issues 12 matrix instructions (384 pipe cycles) against 8 fragment reads, 2-3 LDS-DMA instructions and
// one workgroup barrier; at the matrix-core rate a CU would have to take in 43 B / clk of planes through the 64 B / clk
// vector-memory path and read 85 B / clk from the 128 B / clk LDS.  On a 256 x 256 tile a wave owns 128 x 64 of the
// result: 48 matrix instructions per 32-k tile against 24 fragment reads, 8 DMA instructions and ONE barrier -- half
// the global bytes (21 B / clk) and 3/4 of the LDS bytes (64 B / clk) per matrix instruction, a quarter of the barriers.
//
// Registers decide the shape of the loop.  The running result of a 128 x 64 wave tile is 128 VGPRs; a block-local product
// set of the same size (the P of gemm_f16x3.hip) would take the wave to 256 before a single fragment is loaded.  So the
// K tile is the scale block: the kernel folds after every 32 k (the planes keep their 64-k block exponents -- two
// consecutive folds use the same pair), and a K tile is walked in four PHASES, one per 32-row block i of the wave's A
// rows: the wave's B fragments of the whole K tile stay in registers (32 VGPRs), phase i reads the A fragments of row
// block i (16 VGPRs, double-buffered: phase i + 1's are requested before phase i's matrix instructions), forms the two
// 32 x 32 products of the row block -- 2 x 6 matrix instructions into a 32-register P, two interleaved dependent chains
// -- and folds P into acc[i] with one FMA per element.  128 + 32 + 32 + 32 = 224 VGPRs, two waves per SIMD.
//
// LDS: 160 KiB = three 32 KiB stages of A (256 rows x 32 k x 2 planes) + two of B.  ONE barrier per K tile, at the start
// of phase 3 (by then every wave has READ tile t completely: the fragments of phase 3 were requested in phase 2): it
// publishes tile t+1 and releases A's stage of tile t for tile t+3 and B's for tile t+2; the eight LDS-DMA instructions
// of a wave per K tile are issued two per phase, behind the phase's first matrix instructions, and the wait before the
// barrier is counted (`vmcnt(4)`: the four youngest DMA stay in flight across it).  The B fragments roll: phase 3 reads
// tile t+1's first k step into the registers its own first six matrix instructions have just released.
//
// Epilogue / split-K contract: GemmArgs, as in gemm_f16x3.hip (the launcher there splits the operands and calls in here).
#include "gemm_x3_shared.hpp"

namespace sg {
namespace f16x3 {

constexpr int WSTG = 32 * UNIT;                      // one stage of one operand: 8 row blocks x (2 k steps x 2 planes) units
constexpr int W_ASTAGES = 3, W_BSTAGES = 2;
constexpr int W_SMEM = (W_ASTAGES + W_BSTAGES) * WSTG;   // 160 KiB

// ABL (development, timing only -- wrong results): 1 no matrix instructions, 2 no fragment reads, 3 no DMA, 4 no fold.
// OPT (scheduling experiments): bit 0 s_setprio 1 over a phase's matrix instructions; bit 1 static s_setprio 1 for waves
// 4-7 (the younger wave of every SIMD) over the whole K loop; bit 2 the phase's two DMA instructions after its fold instead
// of between its k steps; bit 3 the fold as 32 plain v_fma_f32 (the compiler packs pairs into v_pk_fma_f32).

// OPT bit 4 (development): per-phase cycle counters (s_memtime) of waves 0 and 4 of every workgroup, summed into g_x3w_timing:
// [w][0..3] phase i, [w][4] of which waiting at the K tile's barrier (lgkmcnt + vmcnt + s_barrier), [w][5] K tiles
__device__ unsigned long long g_x3w_timing[2][8];

// LDS-DMA of 64 x 16 bytes: global (per-lane address) -> LDS (wave-uniform base in m0, lane-linear).  Assembly, so that
// the compiler keeps no record of a pending LDS write (it would wait with vmcnt(0) before the next LDS access that may
// alias it); the counted waits below are the only ordering.
__device__ __forceinline__ void dma_unit(const char* src, unsigned lds) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is reserved: nothing else in these kernels uses it
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds) : "memory", "m0");
#pragma clang diagnostic pop
}

// ---- epilogue of one wave: its 128 x 64 part of the tile (4 x 2 MFMA tiles) through a private LDS block -> 16-byte
// stores, 256 B per row segment.  MFMA layout: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
__device__ __forceinline__ void store_wave_tile(const GemmArgs& g, f32x16 (&acc)[4][2], char* smem, int wave, int lane,
                                                int row0, int col0, int z) {
  const int l31 = lane & 31, kh = lane >> 5;
  const bool partial = (g.splits > 1);
  float* out = partial ? g.ws + static_cast<long long>(z) * g.M * g.N : g.C;
  const long long ldo = partial ? g.N : g.ldc;
  const bool vec_c = ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && ((ldo & 3) == 0);
  float* cst = reinterpret_cast<float*>(smem + wave * CSTAGE);
  const int c4 = (lane & 15) * 4, r4 = lane >> 4;
  const int col = col0 + c4;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (!partial && g.bias) {
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = (col + q < g.N) ? g.bias[col + q] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
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
      const int row = row0 + i * 32 + r;
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
          const f32x4 tt = {v[0], v[1], v[2], v[3]};
          // a finished C tile is not re-read by this kernel: streamed past the caches; split-K partials are re-read at
          // once by the reduce kernel and stay cacheable
          if (!partial) __builtin_nontemporal_store(tt, reinterpret_cast<f32x4*>(o));
          else *reinterpret_cast<f32x4*>(o) = tt;
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

// one phase's matrix work: the products of A row block (fragments a[ks][plane]) with the wave's two B row blocks
// (b[ks][j][plane]) over the K tile's two k steps, into P (block scale).  Corrections first, leading product last; the two
// chains (j = 0, 1) alternate.  `mid` runs between the two k steps (the rolling B fragments are re-read there).
template <int ABL, int OPT, typename Mid>
__device__ __forceinline__ void phase_mfma(f32x16 (&P)[2], const f16x8 (&a)[2][2], const f16x8 (&b)[2][2][2], Mid&& mid) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    if constexpr (ABL == 1) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          P[j][p] = (ks == 0 ? 0.f : P[j][p]) + static_cast<float>(a[ks][p][0]) + static_cast<float>(b[ks][j][p][1]);
    } else {
      if constexpr (OPT & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 2; ++j) P[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][1], b[ks][j][0], ks == 0 ? zero : P[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) P[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][0], b[ks][j][1], P[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) P[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][0], b[ks][j][0], P[j], 0, 0, 0);
      if constexpr (OPT & 1) __builtin_amdgcn_s_setprio(0);
    }
    if (ks == 0) mid();
  }
}

// acc += P * 2^rel, rel <= 0: the block's scale relative to the largest block scale S of the (tile, K slice) -- the running
// result is kept in units of 2^S and brought to its true scale once, in the epilogue.  Branch-free: a block more than
// 2^126 below the largest one contributes less than 2^-90 of that block's product bound and is dropped (sc = 0).
template <int ABL, int OPT>
__device__ __forceinline__ void fold(f32x16& acc, const f32x16& P, int rel) {
  if constexpr (ABL == 4) {
    acc[0] += P[0] + static_cast<float>(rel);
  } else if constexpr (OPT & 8) {
    const float sc = rel >= -126 ? __uint_as_float(static_cast<unsigned>(127 + rel) << 23) : 0.f;      // scalar
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float a = acc[e];
      asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(P[e]), "s"(sc));
      acc[e] = a;
    }
  } else {
    const float sc = rel >= -126 ? __uint_as_float(static_cast<unsigned>(127 + rel) << 23) : 0.f;      // scalar
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = fmaf(P[e], sc, acc[e]);
  }
}

// wave maximum of an int (DPP, see wave_max_nonneg), returned wave-uniform
__device__ __forceinline__ int wave_max_int(int v) {
  auto step = [&](auto ctrl, auto row_mask) __attribute__((always_inline)) {
    const int y = __builtin_amdgcn_update_dpp(v, v, decltype(ctrl)::value, decltype(row_mask)::value, 0xf, false);
    v = max(v, y);
  };
  using std::integral_constant;
  step(integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});
  step(integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});
  step(integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});
  step(integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});
  step(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});
  step(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});
  return __builtin_amdgcn_readlane(v, 63);
}

// ---------------------------------------------------------------------------------------------------------------------
// planes x planes
// ---------------------------------------------------------------------------------------------------------------------
template <int ABL, int OPT>
__global__ __launch_bounds__(512, 1) void gemm_x3w_kernel(const GemmArgs g, const PlaneArgs pl) {
  __shared__ __attribute__((aligned(1024))) char smem[W_SMEM];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // work item -> (tile, K slice); XCD-aware bijective remap of the tile index (workgroup b runs on XCD b % 8): the tiles of
  // one XCD are consecutive, consecutive tiles share their A panel
  const int nt = g.tiles_m * g.tiles_n;
  const int item = blockIdx.x;
  const int z = item / nt, lin = item - z * nt;
  const int q8 = nt >> 3, r8 = nt & 7, x8 = lin & 7;
  const int tile = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (lin >> 3);
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  // K range of this slice in 16-k steps: g.tiles_per_split counts 32-k tiles and is even when g.splits > 1, the planes are
  // zero-padded to 64 k: a slice is a whole number of 64-k scale blocks, T (32-k tiles) is even and >= 2
  const int ksteps = (g.K + 15) / 16;
  // (with one slice tiles_per_split = ceil(K / 32) may be odd: the slice is rounded up to the padded end)
  const int s0 = z * g.tiles_per_split * 2, s1 = (min((ksteps + 3) & ~3, s0 + g.tiles_per_split * 2) + 3) & ~3;
  const int T = (s1 - s0) >> 1;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- DMA: wave w moves row block w of A's and of B's tile, 4 KiB (k step x plane) per K tile and operand ----
  const long long rb_stride = static_cast<long long>(pl.KS) * 2 * UNIT;
  const char* a_src = pl.pa + (static_cast<long long>(tm) * 8 + wave) * rb_stride + static_cast<long long>(s0) * 2 * UNIT + lane * 16;
  const char* b_src = pl.pb + (static_cast<long long>(tn) * 8 + wave) * rb_stride + static_cast<long long>(s0) * 2 * UNIT + lane * 16;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_void*)smem));
  const unsigned a_dst = lds0 + wave * 4 * UNIT, b_dst = lds0 + W_ASTAGES * WSTG + wave * 4 * UNIT;
  auto issue_a = [&](int kt, int sa, int part) __attribute__((always_inline)) {     // sa = kt % 3
    if constexpr (ABL != 3) dma_unit(a_src + static_cast<long long>(kt) * (4 * UNIT) + part * UNIT, a_dst + sa * WSTG + part * UNIT);
  };
  auto issue_b = [&](int kt, int sb, int part) __attribute__((always_inline)) {
    if constexpr (ABL != 3) dma_unit(b_src + static_cast<long long>(kt) * (4 * UNIT) + part * UNIT, b_dst + sb * WSTG + part * UNIT);
  };

  const int kbs = pl.KS >> 2;
  // ---- fragments ----
  const char* a_frag0 = smem + wm * 16 * UNIT + lane * 16;                      // + stage * WSTG + (i * 4 + ks * 2 + plane) * UNIT
  const char* b_frag0 = smem + W_ASTAGES * WSTG + wn * 8 * UNIT + lane * 16;    // + stage * WSTG + (j * 4 + ks * 2 + plane) * UNIT
  f16x8 aF[2][2][2];       // [buffer][k step][plane]
  f16x8 bF[2][2][2];       // [k step][j][plane]
  auto read_a = [&](int buf, int sa, int i) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if constexpr (ABL == 2) aF[buf][ks][p] = __builtin_bit_cast(f16x8, make_uint4(sa, i, ks, p));
        else aF[buf][ks][p] = *reinterpret_cast<const f16x8*>(a_frag0 + sa * WSTG + (i * 4 + ks * 2 + p) * UNIT);
      }
  };
  auto read_b = [&](int ks, int sb) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if constexpr (ABL == 2) bF[ks][j][p] = __builtin_bit_cast(f16x8, make_uint4(sb, j, ks, p));
        else bF[ks][j][p] = *reinterpret_cast<const f16x8*>(b_frag0 + sb * WSTG + (j * 4 + ks * 2 + p) * UNIT);
      }
  };

  // largest block scale of every (A row block, B row block) pair over the slice: lanes take the slice's 64-k blocks
  int S[4][2];
  {
    const int nkb = T >> 1;
    int m[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) m[i][j] = -(1 << 20);
    for (int kb = lane; kb < nkb; kb += 64) {
      int ea[4], eb[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) ea[i] = pl.exp_a[static_cast<long long>(tm * 8 + wm * 4 + i) * kbs + (s0 >> 2) + kb];
#pragma unroll
      for (int j = 0; j < 2; ++j) eb[j] = pl.exp_b[static_cast<long long>(tn * 8 + wn * 2 + j) * kbs + (s0 >> 2) + kb];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) m[i][j] = max(m[i][j], ea[i] + eb[j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) S[i][j] = wave_max_int(m[i][j]);
  }

  // ---- prologue: tiles 0 and 1 (T >= 2); the second half of B's tile 1 goes out in phase 0 of tile 0, as in steady state ----
#pragma unroll
  for (int p = 0; p < 4; ++p) issue_b(0, 0, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) issue_a(0, 0, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) issue_a(1, 1, p);
  issue_b(1, 1, 0);
  issue_b(1, 1, 1);
  wait_vm<6>();                                        // tile 0 has landed (this wave's part)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_b(0, 0);
  read_a(0, 0, 0);

  auto issue_phase = [&](int i, int kt1, int kt2, int sb, int sa2) __attribute__((always_inline)) {
    if (i == 3) { issue_b(kt2, sb, 0); issue_b(kt2, sb, 1); }
    else if (i == 0) { issue_b(kt1, sb ^ 1, 2); issue_b(kt1, sb ^ 1, 3); }
    else if (i == 1) { issue_a(kt2, sa2, 0); issue_a(kt2, sa2, 1); }
    else { issue_a(kt2, sa2, 2); issue_a(kt2, sa2, 3); }
  };
  int eaV[4], ebV[2];
  unsigned long long tq[6] = {0, 0, 0, 0, 0, 0};
  // one K tile; branch-free.  Past the end of the slice the DMA re-loads the last tile (clamped index: the stage it lands in
  // is free and is never read), so that the counted wait and the issue pattern are the same for every tile.
  auto body = [&](int kt, int sa) __attribute__((always_inline)) {
    const int sa1 = sa == 2 ? 0 : sa + 1;              // (kt + 1) % 3
    const int sa2 = sa == 0 ? 2 : sa - 1;              // (kt + 2) % 3
    const int sb = kt & 1;
    const int kt1 = min(kt + 1, T - 1), kt2 = min(kt + 2, T - 1);
    // the 64-k block's six exponents: lane kb % 64 of the chunk's exponent registers (no memory operation in the loop: a
    // scalar load here would turn every LDS wait of the tile into lgkmcnt(0))
    const int kl = (kt >> 1) & 63;
    int ea_s[4], eb_s[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) ea_s[i] = __builtin_amdgcn_readlane(eaV[i], kl);
#pragma unroll
    for (int j = 0; j < 2; ++j) eb_s[j] = __builtin_amdgcn_readlane(ebV[j], kl);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x16 P[2];
      unsigned long long tp0 = 0, tb0 = 0, tb1 = 0;
      if constexpr (OPT & 16) tp0 = __builtin_readcyclecounter();
      if (i == 0) read_b(1, sb);
      if (i < 3) {
        read_a((i + 1) & 1, sa, i + 1);
      } else {
        // every fragment of tile kt has been requested (phase 3's in phase 2); once they are here the tile is released
        if constexpr (OPT & 16) tb0 = __builtin_readcyclecounter();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vm<4>();                                  // tile kt+1 has landed; the 4 DMA of A's tile kt+2 may stay in flight
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (OPT & 16) tb1 = __builtin_readcyclecounter();
        read_a(0, sa1, 0);
      }
      phase_mfma<ABL, OPT>(P, aF[i & 1], bF, [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        // two DMA instructions per phase, behind the phase's first six matrix instructions
        if constexpr (!(OPT & 4)) issue_phase(i, kt1, kt2, sb, sa2);
        asm volatile("" ::: "memory");
        if (i == 3) read_b(0, sb ^ 1);                 // rolls: k step 0 of tile kt+1 into the registers just released
      });
#pragma unroll
      for (int j = 0; j < 2; ++j) fold<ABL, OPT>(acc[i][j], P[j], ea_s[i] + eb_s[j] - S[i][j]);
      if constexpr (OPT & 4) {
        asm volatile("" ::: "memory");
        issue_phase(i, kt1, kt2, sb, sa2);
        asm volatile("" ::: "memory");
      }
      if constexpr (OPT & 16) {
        asm volatile("" : : "v"(acc[i][1][15]) : "memory");      // the phase ends when its fold has been issued
        const unsigned long long tp1 = __builtin_readcyclecounter();
        tq[i] += tp1 - tp0;
        if (i == 3) { tq[4] += tb1 - tb0; tq[5] += 1; }
      }
    }
  };
  if constexpr (OPT & 2) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
  {
    int sa = 0;
    const int nkb = T >> 1;
    for (int c0 = 0; c0 < nkb; c0 += 64) {             // chunks of 64 scale blocks = 128 K tiles: lane l holds block c0 + l
      const int kbl = min(c0 + lane, nkb - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) eaV[i] = pl.exp_a[static_cast<long long>(tm * 8 + wm * 4 + i) * kbs + (s0 >> 2) + kbl];
#pragma unroll
      for (int j = 0; j < 2; ++j) ebV[j] = pl.exp_b[static_cast<long long>(tn * 8 + wn * 2 + j) * kbs + (s0 >> 2) + kbl];
      // "used" here: the compiler's wait for these loads (vmcnt(0): once per 128 tiles) stays out of the tile loop
      asm volatile("" : : "v"(eaV[0]), "v"(eaV[1]), "v"(eaV[2]), "v"(eaV[3]), "v"(ebV[0]), "v"(ebV[1]) : "memory");
      const int kt_end = min(T, (c0 + 64) * 2);
      for (int kt = c0 * 2; kt < kt_end; ++kt) {
        body(kt, sa);
        sa = sa == 2 ? 0 : sa + 1;
      }
    }
  }
  if constexpr (OPT & 2) __builtin_amdgcn_s_setprio(0);
  if constexpr (OPT & 16) {
    if (lane == 0 && (wave & 3) == 0) {
#pragma unroll
      for (int q = 0; q < 6; ++q) atomicAdd(&g_x3w_timing[wave >> 2][q], tq[q]);
    }
  }
  wait_vm<0>();          // the clamped re-loads of the last tiles
  // the running result back to its true scale (exact; beyond the fp32 range it over- / underflows as the value does)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = ldexpf(acc[i][j][e], S[i][j]);
  __syncthreads();       // every wave is done with the stages: they become the epilogue's staging blocks
  store_wave_tile(g, acc, smem, wave, lane, tm * 256 + wm * 128, tn * 256 + wn * 64, z);
}


// ---------------------------------------------------------------------------------------------------------------------
// planes x planes, DIRECT accumulation ("rescale" form).  The running result of a 32 x 32 product tile is kept in units of
// the CURRENT scale block's 2^e (e = exponent of A's block + exponent of B's block): the matrix instructions accumulate
// straight into it, and when the next 64-k block has another e the tile is multiplied by 2^(e_prev - e) first -- a power of
// two: exact.  No block-local product set, no fold behind the matrix instructions: the rescale of a tile reads results
// that were finished phases ago, so nothing in the loop waits for a matrix instruction, and the 32 registers of P pay for a
// second set of B fragments (next tile's B is read whole while phase 3 multiplies).
// Rounding: three fp32 accumulations per 16 k against the running sum -- the exact-fp32 MFMA kernel (32x32x2) has eight.
// ---------------------------------------------------------------------------------------------------------------------
template <int OPT>
__global__ __launch_bounds__(512, 1) void gemm_x3d_kernel(const GemmArgs g, const PlaneArgs pl) {
  __shared__ __attribute__((aligned(1024))) char smem[W_SMEM];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = g.tiles_m * g.tiles_n;
  const int item = blockIdx.x;
  const int z = item / nt, lin = item - z * nt;
  const int q8 = nt >> 3, r8 = nt & 7, x8 = lin & 7;
  const int tile = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (lin >> 3);
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int ksteps = (g.K + 15) / 16;
  const int s0 = z * g.tiles_per_split * 2, s1 = (min((ksteps + 3) & ~3, s0 + g.tiles_per_split * 2) + 3) & ~3;
  const int T = (s1 - s0) >> 1;                        // 32-k tiles: even, >= 2

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const long long rb_stride = static_cast<long long>(pl.KS) * 2 * UNIT;
  const char* a_src = pl.pa + (static_cast<long long>(tm) * 8 + wave) * rb_stride + static_cast<long long>(s0) * 2 * UNIT + lane * 16;
  const char* b_src = pl.pb + (static_cast<long long>(tn) * 8 + wave) * rb_stride + static_cast<long long>(s0) * 2 * UNIT + lane * 16;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_void*)smem));
  const unsigned a_dst = lds0 + wave * 4 * UNIT, b_dst = lds0 + W_ASTAGES * WSTG + wave * 4 * UNIT;
  // OPT (development, timing only unless noted): 1 every DMA reads the slice's first tile (L2-hot): is the loop waiting for
  // memory?  2 no barrier;  4 (correct) B's four DMA of a tile right behind the barrier, A's in phases 0 and 1;  8 no fragment
  // reads;  16 no DMA
  auto issue_a = [&](int kt, int sa, int part) __attribute__((always_inline)) {
    if constexpr (OPT & 16) return;
    if constexpr (OPT & 1) kt = 0;
    dma_unit(a_src + static_cast<long long>(kt) * (4 * UNIT) + part * UNIT, a_dst + sa * WSTG + part * UNIT);
  };
  auto issue_b = [&](int kt, int sb, int part) __attribute__((always_inline)) {
    if constexpr (OPT & 16) return;
    if constexpr (OPT & 1) kt = 0;
    dma_unit(b_src + static_cast<long long>(kt) * (4 * UNIT) + part * UNIT, b_dst + sb * WSTG + part * UNIT);
  };
  const int kbs = pl.KS >> 2;
  const char* a_frag0 = smem + wm * 16 * UNIT + lane * 16;
  const char* b_frag0 = smem + W_ASTAGES * WSTG + wn * 8 * UNIT + lane * 16;
  f16x8 aF[2][2][2];       // [buffer][k step][plane]
  f16x8 bF[2][2][2][2];    // [tile parity][k step][j][plane]
  auto read_a = [&](int buf, int sa, int i) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        if constexpr (OPT & 8) aF[buf][ks][p] = __builtin_bit_cast(f16x8, make_uint4(sa, i, ks, p));
        else aF[buf][ks][p] = *reinterpret_cast<const f16x8*>(a_frag0 + sa * WSTG + (i * 4 + ks * 2 + p) * UNIT);
  };
  auto read_b = [&](int set, int sb) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          if constexpr (OPT & 8) bF[set][ks][j][p] = __builtin_bit_cast(f16x8, make_uint4(sb, j, ks, p));
          else bF[set][ks][j][p] = *reinterpret_cast<const f16x8*>(b_frag0 + sb * WSTG + (j * 4 + ks * 2 + p) * UNIT);
  };

  // prologue
#pragma unroll
  for (int p = 0; p < 4; ++p) issue_b(0, 0, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) issue_a(0, 0, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) issue_a(1, 1, p);
  issue_b(1, 1, 0);
  issue_b(1, 1, 1);
  if constexpr (OPT & 4) { issue_b(1, 1, 2); issue_b(1, 1, 3); wait_vm<8>(); }
  else wait_vm<6>();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_b(0, 0);
  read_a(0, 0, 0);

  auto issue_phase = [&](int i, int kt1, int kt2, int sb, int sa2) __attribute__((always_inline)) {
    if constexpr (OPT & 4) {
      // behind barrier(kt): B's tile kt+2 (phase 3); A's tile kt+3 in phases 0 and 1 of tile kt+1 (here: kt2 = this tile + 2,
      // into the stage of this tile - 1 = sa2)
      if (i == 3) { issue_b(kt2, sb, 0); issue_b(kt2, sb, 1); issue_b(kt2, sb, 2); issue_b(kt2, sb, 3); }
      else if (i == 0) { issue_a(kt2, sa2, 0); issue_a(kt2, sa2, 1); }
      else if (i == 1) { issue_a(kt2, sa2, 2); issue_a(kt2, sa2, 3); }
    } else {
      if (i == 3) { issue_b(kt2, sb, 0); issue_b(kt2, sb, 1); }
      else if (i == 0) { issue_b(kt1, sb ^ 1, 2); issue_b(kt1, sb ^ 1, 3); }
      else if (i == 1) { issue_a(kt2, sa2, 0); issue_a(kt2, sa2, 1); }
      else { issue_a(kt2, sa2, 2); issue_a(kt2, sa2, 3); }
    }
  };
  int eaV[4], ebV[2];
  int E[4][2];             // scale exponent the running result of tile (i, j) is held in (scalar)
  // one K tile (parity PAR = kt & 1, compile time: the B fragment set); FIRST: the first tile of a 64-k block -- the running
  // results are brought to the block's scale, acc[0] ahead of phase 0, acc[i + 1] in the shadow of phase i's matrix work
  auto body = [&](int kt, int sa, auto par, auto first) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par)::value;
    constexpr bool FIRST = decltype(first)::value;
    const int sa1 = sa == 2 ? 0 : sa + 1, sa2 = sa == 0 ? 2 : sa - 1;
    const int sb = PAR;
    const int kt1 = min(kt + 1, T - 1), kt2 = min(kt + 2, T - 1);
    float f[4][2];
    if constexpr (FIRST) {
      const int kl = (kt >> 1) & 63;
      int ea_s[4], eb_s[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) ea_s[i] = __builtin_amdgcn_readlane(eaV[i], kl);
#pragma unroll
      for (int j = 0; j < 2; ++j) eb_s[j] = __builtin_amdgcn_readlane(ebV[j], kl);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int e = ea_s[i] + eb_s[j];
          const int d = kt == 0 ? 0 : min(E[i][j] - e, 64);          // TODO(slow path): d > 64 = a block far below the running scale
          f[i][j] = d >= -126 ? __uint_as_float(static_cast<unsigned>(127 + d) << 23) : 0.f;
          E[i][j] = e;
        }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] *= f[0][j];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool late = (OPT & 32) && wave >= 4;       // wave-uniform: the younger wave of every SIMD reads / issues half a phase later
      if (i < 3) {
        if (!late) read_a((i + 1) & 1, sa, i + 1);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wait_vm<4>();
        if constexpr (!(OPT & 2)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_a(0, sa1, 0);
        read_b(PAR ^ 1, sb ^ 1);
      }
      // pinned: hipcc otherwise sinks these reads to just ahead of their first use a phase later (and waits for them on the spot)
      if constexpr (OPT & 64) __builtin_amdgcn_sched_barrier(0);
      const f16x8 (&a)[2][2] = aF[i & 1];
      const f16x8 (&b)[2][2][2] = bF[PAR];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][1], b[ks][j][0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][0], b[ks][j][1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][0], b[ks][j][0], acc[i][j], 0, 0, 0);
        if (ks == 0) {
          asm volatile("" ::: "memory");
          if (!late) issue_phase(i, kt1, kt2, sb, sa2);
          else if (i < 3) read_a((i + 1) & 1, sa, i + 1);
          asm volatile("" ::: "memory");
          if constexpr (FIRST) {
            if (i < 3) {
#pragma unroll
              for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i + 1][j][e] *= f[i + 1][j];
            }
          }
        }
      }
      if (late) {
        asm volatile("" ::: "memory");
        issue_phase(i, kt1, kt2, sb, sa2);
        asm volatile("" ::: "memory");
      }
    }
  };
  {
    int sa = 0;
    const int nkb = T >> 1;
    for (int c0 = 0; c0 < nkb; c0 += 64) {
      const int kbl = min(c0 + lane, nkb - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) eaV[i] = pl.exp_a[static_cast<long long>(tm * 8 + wm * 4 + i) * kbs + (s0 >> 2) + kbl];
#pragma unroll
      for (int j = 0; j < 2; ++j) ebV[j] = pl.exp_b[static_cast<long long>(tn * 8 + wn * 2 + j) * kbs + (s0 >> 2) + kbl];
      asm volatile("" : : "v"(eaV[0]), "v"(eaV[1]), "v"(eaV[2]), "v"(eaV[3]), "v"(ebV[0]), "v"(ebV[1]) : "memory");
      const int kb_end = min(nkb, c0 + 64);
      for (int kb = c0; kb < kb_end; ++kb) {
        body(2 * kb, sa, std::integral_constant<int, 0>{}, std::true_type{});
        sa = sa == 2 ? 0 : sa + 1;
        body(2 * kb + 1, sa, std::integral_constant<int, 1>{}, std::false_type{});
        sa = sa == 2 ? 0 : sa + 1;
      }
    }
  }
  wait_vm<0>();
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = ldexpf(acc[i][j][e], E[i][j]);
  __syncthreads();
  store_wave_tile(g, acc, smem, wave, lane, tm * 256 + wm * 128, tn * 256 + wn * 64, z);
}


// ---------------------------------------------------------------------------------------------------------------------
// Hybrid, 256 x 256: op(A) stays fp32 in memory and is split INSIDE the kernel (gemm_f16x3.hip, gemm_f16x3h_kernel, for the
// reasons), op(B) comes as planes by LDS-DMA; direct accumulation as in gemm_x3d_kernel.  Each of the eight waves converts
// ONE 32-row x 32-k block of the next K tile per tile (the 128 x 128 form converts one per 24 matrix instructions, this one
// per 48, and with N <= 256 an A block is converted once, not once per column tile).  Per K tile kt:
//   phases 0-2   matrix work on tile kt; B's planes of tile kt+2 arrive by DMA (issued behind barrier kt-1 .. phase 0)
//   end of ph 2  vmcnt(0): A's fp32 block of tile kt+1 (requested one tile ago) is here -> block maximum (DPP) -> exponent
//                (kept from the last tile while the maximum still fits: fewer rescales) -> two f16 planes -> LDS, stage (kt+1)&1
//   phase 3      barrier kt: tile kt+1 is complete and tile kt's B stage is free; request A's block of tile kt+2, read
//                tile kt+1's first fragments and exponents while the last matrix instructions of tile kt run
// A's scale block is 32 x 32 (one K tile), B's 32 x 64.  LDS: 2 x 32 KiB of A planes + 2 x 32 KiB of B planes + exponents.
// ARC = false: A element (m, k) at A[m * lda + k] (K % 4 == 0, 16-byte aligned rows); ARC = true: at A[k * lda + m] with
// M % 4 == 0 and 16-byte aligned k rows (float4 loads along m, register transpose).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int H_SMEM = 4 * WSTG + 64;

template <bool ARC>
__global__ __launch_bounds__(512, 1) void gemm_x3dh_kernel(const GemmArgs g, const PlaneArgs pl) {
  __shared__ __attribute__((aligned(1024))) char smem[H_SMEM];
  int* exp_lds = reinterpret_cast<int*>(smem + 4 * WSTG);               // [stage][row block]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = g.tiles_m * g.tiles_n;
  const int item = blockIdx.x;
  const int z = item / nt, lin = item - z * nt;
  const int q8 = nt >> 3, r8 = nt & 7, x8 = lin & 7;
  const int tile = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (lin >> 3);
  const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
  const int ktiles = (g.K + 31) / 32;
  const int kt0 = z * g.tiles_per_split, kt1e = min(ktiles, kt0 + g.tiles_per_split);
  const int T = kt1e - kt0;                            // 32-k tiles of this slice (>= 1)

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- B planes by DMA: wave w moves row block w of B's tile (4 KiB per K tile) ----
  const long long rb_stride = static_cast<long long>(pl.KS) * 2 * UNIT;
  const char* b_src = pl.pb + (static_cast<long long>(tn) * 8 + wave) * rb_stride + static_cast<long long>(kt0) * (4 * UNIT) + lane * 16;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_void*)smem));
  const unsigned b_dst = lds0 + 2 * WSTG + wave * 4 * UNIT;
  auto issue_b = [&](int kt, int part) __attribute__((always_inline)) {
    const int ktc = min(kt, T - 1);                    // past the end: a harmless re-load of the last tile
    dma_unit(b_src + static_cast<long long>(ktc) * (4 * UNIT) + part * UNIT, b_dst + (kt & 1) * WSTG + part * UNIT);
  };

  // ---- A: this wave's 32-row block of a K tile, 16 fp32 per lane, loaded unconditionally from clamped coordinates ----
  const int m_blk = tm * 256 + wave * 32;
  f32x4 va[4];
  auto load_a = [&](int kt_in) __attribute__((always_inline)) {
    const int k0 = (kt0 + min(kt_in, T - 1)) * 32;
    if (!ARC) {           // lane = (row lane / 8 + 8 i, k = 4 (lane % 8) .. + 3)
      const int k = min(k0 + (lane & 7) * 4, g.K - 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = min(m_blk + (lane >> 3) + 8 * i, g.M - 1);
        va[i] = *reinterpret_cast<const f32x4*>(g.A + static_cast<long long>(row) * g.lda + k);
      }
    } else {              // lane = (4 op rows m = 4 (lane % 8) .., k = 4 (lane / 8) + i): 128-byte k-row segments
      const int m = min(m_blk + (lane & 7) * 4, g.M - 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = min(k0 + (lane >> 3) * 4 + i, g.K - 1);
        va[i] = *reinterpret_cast<const f32x4*>(g.A + static_cast<long long>(k) * g.lda + m);
      }
    }
  };
  int e_keep = 1 << 20;                                  // exponent of this wave's previous block (none yet)
  auto store_a = [&](int kt) __attribute__((always_inline)) {      // convert va (tile kt), planes + exponent into stage kt & 1
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
    } else {              // transpose in registers: x[4 j + i] = (m j, k i) -> four consecutive k per op row
      const bool mdead = m_blk + (lane & 7) * 4 >= g.M;       // M % 4 == 0: four rows are in or out together
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool dead = mdead || (k0 + (lane >> 3) * 4 + i >= g.K);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[4 * j + i] = dead ? 0.f : va[i][j]; mx = fmaxf(mx, fabsf(x[4 * j + i])); }
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
      // the last block's scale is kept while this block's maximum still lands in [2^12, 2^15) under it (and always for an
      // all-zero block): two bits of the residual plane's range against a rescale of the running results in all eight waves
      if (bits == 0u || (e - e_keep >= 0 && e - e_keep <= 2)) e = e_keep == (1 << 20) ? e : e_keep;
      e_keep = e;
    }
    const float sc = __uint_as_float(static_cast<unsigned>(127 + e) << 23);
    unsigned h1[8], h2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x2 v = {x[2 * j] * sc, x[2 * j + 1] * sc};
      const f16x2 a = __builtin_convertvector(v, f16x2);
      f32x2 res = {__builtin_fmaf(x[2 * j], sc, -static_cast<float>(a[0])), __builtin_fmaf(x[2 * j + 1], sc, -static_cast<float>(a[1]))};
      if (nonfinite) {    // wave-uniform: an inf keeps a zero residual (inf - inf = NaN would poison the product's inf)
        if (fabsf(v[0]) == __builtin_inff()) res[0] = 0.f;
        if (fabsf(v[1]) == __builtin_inff()) res[1] = 0.f;
      }
      const f16x2 b = __builtin_convertvector(res, f16x2);
      h1[j] = __builtin_bit_cast(unsigned, a);
      h2[j] = __builtin_bit_cast(unsigned, b);
    }
    // 4 consecutive k of row r: half a 16-byte slot of unit (ks, plane), lane slot kg * 32 + r
    char* st = smem + (kt & 1) * WSTG + wave * (4 * UNIT);
    const int kq = ARC ? (lane >> 3) : (lane & 7);
    const int off = ((kq >> 2) * 2) * UNIT + (((kq >> 1) & 1) * 32) * 16 + (kq & 1) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = ARC ? (lane & 7) * 4 + i : (lane >> 3) + 8 * i;
      *reinterpret_cast<uint2*>(st + off + r * 16) = make_uint2(h1[2 * i], h1[2 * i + 1]);
      *reinterpret_cast<uint2*>(st + off + UNIT + r * 16) = make_uint2(h2[2 * i], h2[2 * i + 1]);
    }
    if (lane == 0) exp_lds[(kt & 1) * 8 + wave] = -e;
  };

  // ---- fragments ----
  const char* a_frag0 = smem + wm * 16 * UNIT + lane * 16;                      // + stage * WSTG + (i * 4 + ks * 2 + plane) * UNIT
  const char* b_frag0 = smem + 2 * WSTG + wn * 8 * UNIT + lane * 16;            // + stage * WSTG + (j * 4 + ks * 2 + plane) * UNIT
  f16x8 aF[2][2][2];       // [buffer][k step][plane]
  f16x8 bF[2][2][2][2];    // [tile parity][k step][j][plane]
  auto read_a = [&](int buf, int st, int i) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        aF[buf][ks][p] = *reinterpret_cast<const f16x8*>(a_frag0 + st * WSTG + (i * 4 + ks * 2 + p) * UNIT);
  };
  auto read_b = [&](int set, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          bF[set][ks][j][p] = *reinterpret_cast<const f16x8*>(b_frag0 + st * WSTG + (j * 4 + ks * 2 + p) * UNIT);
  };

  // B's block exponents: lane l of ebV holds 64-k block c0 + l of the current chunk (see gemm_x3w_kernel)
  const int kbs = pl.KS >> 2;
  const int kb0 = kt0 >> 1;                            // first 64-k block this slice touches (a slice may start mid-block)
  const int nkb = ((kt0 + T + 1) >> 1) - kb0;
  int ebV[2];
  auto load_eb = [&](int c0) __attribute__((always_inline)) {
    const int kbl = min(c0 + lane, nkb - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) ebV[j] = pl.exp_b[static_cast<long long>(tn * 8 + wn * 2 + j) * kbs + kb0 + kbl];
    asm volatile("" : : "v"(ebV[0]), "v"(ebV[1]) : "memory");
  };

  // ---- prologue ----
#pragma unroll
  for (int p = 0; p < 4; ++p) issue_b(0, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) issue_b(1, p);
  load_a(0);
  load_eb(0);                                           // the compiler's vmcnt(0) for these loads also covers A's tile 0 and the DMA
  store_a(0);
  load_a(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wait_vm<4>();                                         // B's tile 0 has landed (tile 1 and A's tile 1 may be in flight)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_b(0, 0);
  read_a(0, 0, 0);
  int ea_v[4];                                          // exponents of the current tile's four A row blocks (uniform values in VGPRs)
#pragma unroll
  for (int i = 0; i < 4; ++i) ea_v[i] = exp_lds[wm * 4 + i];

  int E[4][2], Mx[4][2];   // scale exponent the running result of tile (i, j) is held in; largest so far
  int viol = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) { E[i][j] = 0; Mx[i][j] = -(1 << 20); }

  auto body = [&](int kt, auto par) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par)::value;           // kt & 1: LDS stage and B fragment set of this tile
    // ---- this tile's scales: rescale the running results whose scale changes (rare with the kept exponents) ----
    const int kl = (((kt0 + kt) >> 1) - kb0) & 63;
    int e[4][2];
    bool any = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ea = __builtin_amdgcn_readfirstlane(ea_v[i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        e[i][j] = ea + __builtin_amdgcn_readlane(ebV[j], kl);
        any = any || (e[i][j] != E[i][j]);
      }
    }
    if (any) {                                          // wave-uniform
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          Mx[i][j] = max(Mx[i][j], e[i][j]);
          viol |= (Mx[i][j] - e[i][j] > 60) ? 1 : 0;    // a block 2^60 below the running scale: the fallback kernel redoes the product
          const int d = kt == 0 ? 0 : min(E[i][j] - e[i][j], 60);
          const float f = d >= -126 ? __uint_as_float(static_cast<unsigned>(127 + d) << 23) : 0.f;
          E[i][j] = e[i][j];
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] *= f;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < 3) {
        read_a((i + 1) & 1, PAR, i + 1);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // my plane stores and every fragment read of tile kt
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_a(kt + 2);
        read_a(0, PAR ^ 1, 0);
        read_b(PAR ^ 1, PAR ^ 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) ea_v[q] = exp_lds[(PAR ^ 1) * 8 + wm * 4 + q];
      }
      const f16x8 (&a)[2][2] = aF[i & 1];
      const f16x8 (&b)[2][2][2] = bF[PAR];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][1], b[ks][j][0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][0], b[ks][j][1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][0], b[ks][j][0], acc[i][j], 0, 0, 0);
        if (ks == 0) {
          asm volatile("" ::: "memory");
          // B's tile kt+2 into the stage tile kt uses: free behind this tile's barrier (phase 3), two DMA there, two in phase 0
          if (i == 3) { issue_b(kt + 2, 0); issue_b(kt + 2, 1); }
          else if (i == 0) { issue_b(kt + 1, 2); issue_b(kt + 1, 3); }
          asm volatile("" ::: "memory");
        }
      }
      if (i == 2) {
        // A's block of tile kt+1 (requested a tile ago) and B's tile kt+1 have had ~a tile to land
        asm volatile("" ::: "memory");
        store_a(kt + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  };
  // prologue issued all four parts of B's tiles 0 and 1: the phase-0 slot of tile 0 re-loads two parts of tile 1 (harmless)
  for (int c0 = 0; c0 < nkb; c0 += 64) {
    if (c0 > 0) load_eb(c0);
    const int kt_lo = max(0, (kb0 + c0) * 2 - kt0), kt_hi = min(T, (kb0 + c0 + 64) * 2 - kt0);
    int kt = kt_lo;
    if (kt < kt_hi && (kt & 1)) { body(kt, std::integral_constant<int, 1>{}); ++kt; }
    for (; kt + 1 < kt_hi; kt += 2) {
      body(kt, std::integral_constant<int, 0>{});
      body(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < kt_hi) body(kt, std::integral_constant<int, 0>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (viol && pl.flag && lane == 0) atomicOr(pl.flag, 1);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = ldexpf(acc[i][j][e], E[i][j]);
  __syncthreads();
  store_wave_tile(g, acc, smem, wave, lane, tm * 256 + wm * 128, tn * 256 + wn * 64, z);
}


// ---------------------------------------------------------------------------------------------------------------------
// The hybrid kernel, PERSISTENT: one workgroup per CU walks the work items b, b + G, b + 2 G, ... (item = output tile x K
// slice) as ONE stream of K tiles.  The load / convert / DMA stages run one to two tiles ahead of the matrix work straight
// across item boundaries, so an item's first tile is already converted when the previous item's last matrix instruction
// issues: no pipeline fill per item (at K = 256 -- eight tiles per item -- fill + drain + dispatch were 44 % of the
// non-persistent kernel's time), and an item's result is stored while the loads of the next one are in flight.  The
// epilogue has its own LDS (2 KiB per wave: 16 x 32 blocks -> 128-byte row segments) because the stages stay live.
// Every slice must hold at least two K tiles (the launcher checks); G = gridDim.x.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int P_STG_W = 16 * 32 * 4;
constexpr int P_EB_W = 2 * 2 * 64 * 4;               // per wave: [buffer][B row block j][64-k block of the chunk] exponents
constexpr int P_SMEM = 4 * WSTG + 8 * P_STG_W + 64 + 8 * P_EB_W;

struct ItemRef {          // one (output tile, K slice); wave-uniform.  Everything else is derived where it is used (scalar
  int tm, tn, z, T;       // arithmetic is cheap; two of these live across the whole loop)
};

template <bool ARC>
__global__ __launch_bounds__(512, 1) void gemm_x3dp_kernel(const GemmArgs g, const PlaneArgs pl, const int n_items, const int skew) {
  __shared__ __attribute__((aligned(1024))) char smem[P_SMEM];
  int* exp_lds = reinterpret_cast<int*>(smem + 4 * WSTG + 8 * P_STG_W);     // [stage][row block]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nt = g.tiles_m * g.tiles_n;
  const int ktiles = (g.K + 31) / 32;
  const int kbs = pl.KS >> 2;
  const long long rb_stride = static_cast<long long>(pl.KS) * 2 * UNIT;
  const int G = gridDim.x;

  auto setup = [&](int item) __attribute__((always_inline)) {
    ItemRef r;
    const int z = item / nt, lin = item - z * nt;
    const int q8 = nt >> 3, r8 = nt & 7, x8 = lin & 7;
    const int tile = (x8 < r8 ? x8 * (q8 + 1) : r8 * (q8 + 1) + (x8 - r8) * q8) + (lin >> 3);
    r.tm = tile / g.tiles_n;
    r.tn = tile - r.tm * g.tiles_n;
    r.z = z;
    const int kt0 = z * g.tiles_per_split;
    r.T = min(ktiles, kt0 + g.tiles_per_split) - kt0;
    return r;
  };
  auto kt0_of = [&](const ItemRef& r) __attribute__((always_inline)) { return r.z * g.tiles_per_split; };
  auto mblk_of = [&](const ItemRef& r) __attribute__((always_inline)) { return r.tm * 256 + wave * 32; };
  auto kb0_of = [&](const ItemRef& r) __attribute__((always_inline)) { return (r.z * g.tiles_per_split) >> 1; };
  auto nkb_of = [&](const ItemRef& r) __attribute__((always_inline)) {
    const int kt0 = r.z * g.tiles_per_split;
    return ((kt0 + r.T + 1) >> 1) - (kt0 >> 1);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // ---- B planes by DMA ----
  const char* b_lane = pl.pb + lane * 16;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_void*)smem));
  const unsigned b_dst = lds0 + 2 * WSTG + wave * 4 * UNIT;
  auto issue_b = [&](const ItemRef& r, int kt, int st, int part) __attribute__((always_inline)) {
    const long long off = (static_cast<long long>(r.tn) * 8 + wave) * rb_stride + static_cast<long long>(kt0_of(r) + kt) * (4 * UNIT) + part * UNIT;
    dma_unit(b_lane + off, b_dst + st * WSTG + part * UNIT);
  };

  // ---- A: this wave's 32 x 32 block of a K tile, fp32 -> registers ----
  f32x4 va[4];
  auto load_a = [&](const ItemRef& r, int kt) __attribute__((always_inline)) {
    const int k0 = (kt0_of(r) + kt) * 32;
    const int m_blk = mblk_of(r);
    if (!ARC) {
      const int k = min(k0 + (lane & 7) * 4, g.K - 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = min(m_blk + (lane >> 3) + 8 * i, g.M - 1);
        va[i] = *reinterpret_cast<const f32x4*>(g.A + static_cast<long long>(row) * g.lda + k);
      }
    } else {
      const int m = min(m_blk + (lane & 7) * 4, g.M - 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = min(k0 + (lane >> 3) * 4 + i, g.K - 1);
        va[i] = *reinterpret_cast<const f32x4*>(g.A + static_cast<long long>(k) * g.lda + m);
      }
    }
  };
  int e_keep = 1 << 20;
  auto store_a = [&](const ItemRef& r, int kt, int st, bool fresh) __attribute__((always_inline)) {
    const int k0 = (kt0_of(r) + kt) * 32;
    const int m_blk = mblk_of(r);
    float x[16];
    float mx = 0.f;
    if (!ARC) {
      const bool kdead = k0 + (lane & 7) * 4 >= g.K;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool dead = kdead || (m_blk + (lane >> 3) + 8 * i >= g.M);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[4 * i + j] = dead ? 0.f : va[i][j]; mx = fmaxf(mx, fabsf(x[4 * i + j])); }
      }
    } else {
      const bool mdead = m_blk + (lane & 7) * 4 >= g.M;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool dead = mdead || (k0 + (lane >> 3) * 4 + i >= g.K);
#pragma unroll
        for (int j = 0; j < 4; ++j) { x[4 * j + i] = dead ? 0.f : va[i][j]; mx = fmaxf(mx, fabsf(x[4 * j + i])); }
      }
    }
    mx = wave_max_nonneg(mx);
    bool nonfinite = false;
    if (__builtin_expect(!(mx <= 3.402823466e38f), 0)) {
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
      // the previous block's scale is kept while this block's maximum still lands in [2^12, 2^15) under it (always for an
      // all-zero block) -- but never across an item boundary: another item has other rows
      if (!fresh && e_keep != (1 << 20) && (bits == 0u || (e - e_keep >= 0 && e - e_keep <= 2))) e = e_keep;
      e_keep = e;
    }
    const float sc = __uint_as_float(static_cast<unsigned>(127 + e) << 23);
    unsigned h1[8], h2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x2 v = {x[2 * j] * sc, x[2 * j + 1] * sc};
      const f16x2 a = __builtin_convertvector(v, f16x2);
      f32x2 res = {__builtin_fmaf(x[2 * j], sc, -static_cast<float>(a[0])), __builtin_fmaf(x[2 * j + 1], sc, -static_cast<float>(a[1]))};
      if (nonfinite) {
        if (fabsf(v[0]) == __builtin_inff()) res[0] = 0.f;
        if (fabsf(v[1]) == __builtin_inff()) res[1] = 0.f;
      }
      const f16x2 b = __builtin_convertvector(res, f16x2);
      h1[j] = __builtin_bit_cast(unsigned, a);
      h2[j] = __builtin_bit_cast(unsigned, b);
    }
    char* stp = smem + st * WSTG + wave * (4 * UNIT);
    const int kq = ARC ? (lane >> 3) : (lane & 7);
    const int off = ((kq >> 2) * 2) * UNIT + (((kq >> 1) & 1) * 32) * 16 + (kq & 1) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rr = ARC ? (lane & 7) * 4 + i : (lane >> 3) + 8 * i;
      *reinterpret_cast<uint2*>(stp + off + rr * 16) = make_uint2(h1[2 * i], h1[2 * i + 1]);
      *reinterpret_cast<uint2*>(stp + off + UNIT + rr * 16) = make_uint2(h2[2 * i], h2[2 * i + 1]);
    }
    if (lane == 0) exp_lds[st * 8 + wave] = -e;
  };

  // ---- fragments ----
  const char* a_frag0 = smem + wm * 16 * UNIT + lane * 16;
  const char* b_frag0 = smem + 2 * WSTG + wn * 8 * UNIT + lane * 16;
  f16x8 aF[2][2][2];       // [buffer][k step][plane]
  f16x8 bF[2][2][2];       // [k step][j][plane]: ONE set that rolls (k step 0 of the next tile is read in phase 3 behind the
                           // last use of this tile's, k step 1 at the top of the next tile)
  auto read_a = [&](int buf, int st, int i) __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        aF[buf][ks][p] = *reinterpret_cast<const f16x8*>(a_frag0 + st * WSTG + (i * 4 + ks * 2 + p) * UNIT);
  };
  auto read_b = [&](int ks, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        bF[ks][j][p] = *reinterpret_cast<const f16x8*>(b_frag0 + st * WSTG + (j * 4 + ks * 2 + p) * UNIT);
  };

  // ---- B's block exponents travel through the wave's own LDS block: [2 buffers][j][64 blocks].  The chunk (64 blocks =
  // 128 K tiles) after the current one -- of this item, or the first of the next item -- is requested by two 4-byte LDS-DMA
  // instructions one chunk ahead; the vmcnt(0) every tile executes before its conversion retires them long before the chunk
  // is entered, and only this wave reads them (no register of the loop is tied up, nothing for the compiler to wait on)
  int* eb_lds = reinterpret_cast<int*>(smem + 4 * WSTG + 8 * P_STG_W + 64 + wave * P_EB_W);
  const unsigned eb_dst = lds0 + 4 * WSTG + 8 * P_STG_W + 64 + wave * P_EB_W;
  auto request_eb = [&](const ItemRef& r, int c0, int buf) __attribute__((always_inline)) {
    const int kbl = min(c0 + lane, nkb_of(r) - 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int* ptr = pl.exp_b + (static_cast<long long>(r.tn * 8 + wn * 2 + j) * kbs + kb0_of(r) + kbl);
      const unsigned dst = eb_dst + (buf * 2 + j) * 256;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" : : "v"(ptr), "s"(dst) : "memory", "m0");
#pragma clang diagnostic pop
    }
  };

  // ---- start skew: the workgroups of a launch otherwise run in step -- all compute, then all store their 256 KB at once
  // and wait for HBM to take 64 MB -- so each of eight classes (four workgroups per XCD each) starts class * skew / 8 cycles
  // late; the items are equally long, the classes stay apart, and the stores of one class overlap the matrix work of the others
  if (skew > 0 || skew == -4) {
    const int n = (((blockIdx.x >> 3) & 7) * (skew == -4 ? 72000 : skew)) >> 16;       // s_sleep 127 = 8128 cycles ~ 2^13
    for (int q = 0; q < n; ++q) __builtin_amdgcn_s_sleep(127);
  }
  // ---- this workgroup's items ----
  int item = blockIdx.x;
  ItemRef cur = setup(item);
  bool has_next = item + G < n_items;
  ItemRef nxt = setup(has_next ? item + G : item);

  // ---- prologue: tiles 0 and 1 of the first item (T >= 2) ----
#pragma unroll
  for (int p = 0; p < 4; ++p) issue_b(cur, 0, 0, p);
#pragma unroll
  for (int p = 0; p < 4; ++p) issue_b(cur, 1, 1, p);
  load_a(cur, 0);
  request_eb(cur, 0, 0);
  if (nkb_of(cur) > 64) request_eb(cur, 64, 1); else request_eb(nxt, 0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the compiler's own wait for A's tile 0 would be this one, too)
  store_a(cur, 0, 0, true);
  load_a(cur, 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_b(0, 0);
  read_a(0, 0, 0);
  int ea_v[4], eb_v[2];                                 // exponents of the coming tile's row blocks (uniform values in VGPRs)
#pragma unroll
  for (int i = 0; i < 4; ++i) ea_v[i] = exp_lds[wm * 4 + i];
#pragma unroll
  for (int j = 0; j < 2; ++j) eb_v[j] = eb_lds[j * 64];
  int cb = 0;                                           // buffer of the current chunk of B exponents
  bool need_req = false;                                // a new chunk was entered: request the one after it (at the next tile top)

  int E[4][2];
  int viol = 0, e_hi = -(1 << 20);     // e_hi: the largest block scale of the item so far, over the wave's eight product tiles
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) E[i][j] = 0;

  // one K tile of the current item; PAR = parity of the tile in the workgroup's stream = LDS stage and B fragment set
  auto body = [&](int kt, auto par) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par)::value;
    const bool more1 = kt + 1 < cur.T, more2 = kt + 2 < cur.T;
    // the tiles one and two ahead in the stream (past the end of the stream: the last tile again -- never consumed)
    const bool r1n = !more1 && has_next, r2n = !more2 && has_next;
    const ItemRef& r1 = r1n ? nxt : cur;
    const ItemRef& r2 = r2n ? nxt : cur;
    const int kt_r1 = more1 ? kt + 1 : (has_next ? 0 : cur.T - 1);
    const int kt_r2 = more2 ? kt + 2 : (has_next ? kt + 2 - cur.T : cur.T - 1);
    const int ckt0 = kt0_of(cur);
    if (need_req) {                                     // wave-uniform, once per 128 tiles or per item
      const int base = (((ckt0 + kt) >> 1) - (ckt0 >> 1)) & ~63;
      if (base + 64 < nkb_of(cur)) request_eb(cur, base + 64, cb ^ 1); else request_eb(nxt, 0, cb ^ 1);
      need_req = false;
    }
    int e[4][2];
    bool any = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ea = __builtin_amdgcn_readfirstlane(ea_v[i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        e[i][j] = ea + __builtin_amdgcn_readfirstlane(eb_v[j]);
        any = any || (e[i][j] != E[i][j]);
      }
    }
    if (any) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          e_hi = max(e_hi, e[i][j]);
          viol |= (e_hi - e[i][j] > 60) ? 1 : 0;        // (conservative across the eight tiles: at worst an unneeded fallback)
          const int d = kt == 0 ? 0 : min(E[i][j] - e[i][j], 60);
          const float f = d >= -126 ? __uint_as_float(static_cast<unsigned>(127 + d) << 23) : 0.f;
          E[i][j] = e[i][j];
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[i][j][q] *= f;
        }
    }
    // is the next tile of the stream in another chunk of B exponents?
    const int r1kt0 = kt0_of(r1);
    const int kbr1 = ((r1kt0 + kt_r1) >> 1) - (r1kt0 >> 1);
    const bool new_chunk = r1n || (more1 && (kbr1 & 63) == 0 && !((r1kt0 + kt_r1) & 1));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i == 0) read_b(1, PAR);
      if (i < 3) {
        read_a((i + 1) & 1, PAR, i + 1);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_a(r2, kt_r2);
        read_a(0, PAR ^ 1, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) ea_v[q] = exp_lds[(PAR ^ 1) * 8 + wm * 4 + q];
#pragma unroll
        for (int j = 0; j < 2; ++j) eb_v[j] = eb_lds[((new_chunk ? cb ^ 1 : cb) * 2 + j) * 64 + (kbr1 & 63)];
      }
      const f16x8 (&a)[2][2] = aF[i & 1];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][1], bF[ks][j][0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][0], bF[ks][j][1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks][0], bF[ks][j][0], acc[i][j], 0, 0, 0);
        if (ks == 0) {
          asm volatile("" ::: "memory");
          if (i == 3) { issue_b(r2, kt_r2, PAR, 0); issue_b(r2, kt_r2, PAR, 1); }
          else if (i == 0) { issue_b(r1, kt_r1, PAR ^ 1, 2); issue_b(r1, kt_r1, PAR ^ 1, 3); }
          asm volatile("" ::: "memory");
          if (i == 3) read_b(0, PAR ^ 1);               // rolls: k step 0 of the next tile into the registers just released
        }
      }
      if (i == 2) {
        asm volatile("" ::: "memory");
        store_a(r1, kt_r1, PAR ^ 1, !more1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    if (new_chunk) { cb ^= 1; need_req = true; }
  };

  // ---- an item's result: scale, stage 16 x 32 blocks through the wave's own LDS block, store ----
  auto finish = [&]() __attribute__((always_inline)) {
    if (skew == -2) {                                   // development, timing only: no epilogue at all
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          E[i][j] = 0;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        }
      e_hi = -(1 << 20);
      return;
    }
    const int l31 = lane & 31, kh = lane >> 5;
    const bool partial = (g.splits > 1);
    float* out = partial ? g.ws + static_cast<long long>(cur.z) * g.M * g.N : g.C;
    const int row0 = cur.tm * 256 + wm * 128, col0 = cur.tn * 256 + wn * 64;
    const long long ldo = partial ? g.N : g.ldc;
    const bool vec_c = ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && ((ldo & 3) == 0);
    float* stg = reinterpret_cast<float*>(smem + 4 * WSTG + wave * P_STG_W);
    const int c4 = (lane & 7) * 4, r8 = lane >> 3;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + j * 32 + c4;
      float bv[4] = {0.f, 0.f, 0.f, 0.f};
      if (!partial && g.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) bv[q] = (col + q < g.N) ? g.bias[col + q] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int e = 0; e < 8; ++e)
            stg[((e & 3) + 8 * (e >> 2) + 4 * kh) * 32 + l31] = ldexpf(acc[i][j][h * 8 + e], E[i][j]);
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int it = 0; it < 2; ++it) {
            const int r = r8 + 8 * it;
            const int row = row0 + i * 32 + h * 16 + r;
            const float4 t4 = *reinterpret_cast<const float4*>(stg + r * 32 + c4);
            float v[4] = {t4.x, t4.y, t4.z, t4.w};
            if (row < g.M && col < g.N && skew != -1) {   // (skew == -1: development, timing only: no global stores)
              // (skew == -5: development, timing only: every item of a workgroup stores to the same 256 rows -- L2-resident)
              float* o = out + static_cast<long long>(skew == -5 ? (row & 255) + static_cast<int>(blockIdx.x) * 256 : row) * ldo + col;
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
                const f32x4 tt = {v[0], v[1], v[2], v[3]};
                if (!partial && skew != -3 && skew != -4) __builtin_nontemporal_store(tt, reinterpret_cast<f32x4*>(o));
                else *reinterpret_cast<f32x4*>(o) = tt;
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
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        E[i][j] = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
      }
    e_hi = -(1 << 20);
  };

  int par = 0;                                          // parity of the stream position at the start of the current item
  for (;;) {
    int kt = 0;
    if (par) { body(kt, std::integral_constant<int, 1>{}); ++kt; }
    for (; kt + 1 < cur.T; kt += 2) {
      body(kt, std::integral_constant<int, 0>{});
      body(kt + 1, std::integral_constant<int, 1>{});
    }
    if (kt < cur.T) { body(kt, std::integral_constant<int, 0>{}); par = 1; } else par = 0;
    finish();
    if (!has_next) break;
    item += G;
    cur = nxt;
    has_next = item + G < n_items;
    nxt = setup(has_next ? item + G : item);              // (the new item's first tile raised need_req: its next chunk is requested there)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (viol && pl.flag && lane == 0) atomicOr(pl.flag, 1);
}

}  // namespace f16x3

// g.tiles_m / g.tiles_n count 256-wide tiles; the planes and exponents are those of gemm_f16x3.hip's split_kernel
void launch_x3w_planes(const GemmArgs& g, const f16x3::PlaneArgs& pl, int dev_variant, hipStream_t st) {
  const long long items = static_cast<long long>(g.tiles_m) * g.tiles_n * g.splits;
  const dim3 grid(static_cast<unsigned>(items)), block(512);
  using namespace f16x3;
  switch (dev_variant) {      // 1..4: timing-only ablations, 5: scheduling experiment (development)
    case 1: hipLaunchKernelGGL((gemm_x3w_kernel<1, 0>), grid, block, 0, st, g, pl); break;
    case 2: hipLaunchKernelGGL((gemm_x3w_kernel<2, 0>), grid, block, 0, st, g, pl); break;
    case 3: hipLaunchKernelGGL((gemm_x3w_kernel<3, 0>), grid, block, 0, st, g, pl); break;
    case 4: hipLaunchKernelGGL((gemm_x3w_kernel<4, 0>), grid, block, 0, st, g, pl); break;
    case 5: hipLaunchKernelGGL((gemm_x3w_kernel<0, 1>), grid, block, 0, st, g, pl); break;
    case 6: hipLaunchKernelGGL((gemm_x3w_kernel<0, 2>), grid, block, 0, st, g, pl); break;
    case 7: hipLaunchKernelGGL((gemm_x3w_kernel<0, 4>), grid, block, 0, st, g, pl); break;
    case 8: hipLaunchKernelGGL((gemm_x3w_kernel<0, 6>), grid, block, 0, st, g, pl); break;
    case 9: hipLaunchKernelGGL((gemm_x3w_kernel<0, 8>), grid, block, 0, st, g, pl); break;
    case 10: hipLaunchKernelGGL((gemm_x3w_kernel<0, 14>), grid, block, 0, st, g, pl); break;
    case 11: hipLaunchKernelGGL((gemm_x3w_kernel<0, 16>), grid, block, 0, st, g, pl); break;
    case 12: hipLaunchKernelGGL((gemm_x3w_kernel<0, 24>), grid, block, 0, st, g, pl); break;
    case 20: hipLaunchKernelGGL((gemm_x3d_kernel<0>), grid, block, 0, st, g, pl); break;
    case 21: hipLaunchKernelGGL((gemm_x3d_kernel<1>), grid, block, 0, st, g, pl); break;
    case 22: hipLaunchKernelGGL((gemm_x3d_kernel<2>), grid, block, 0, st, g, pl); break;
    case 23: hipLaunchKernelGGL((gemm_x3d_kernel<4>), grid, block, 0, st, g, pl); break;
    case 24: hipLaunchKernelGGL((gemm_x3d_kernel<8>), grid, block, 0, st, g, pl); break;
    case 25: hipLaunchKernelGGL((gemm_x3d_kernel<16>), grid, block, 0, st, g, pl); break;
    case 26: hipLaunchKernelGGL((gemm_x3d_kernel<24>), grid, block, 0, st, g, pl); break;
    case 27: hipLaunchKernelGGL((gemm_x3d_kernel<26>), grid, block, 0, st, g, pl); break;
    case 28: hipLaunchKernelGGL((gemm_x3d_kernel<32>), grid, block, 0, st, g, pl); break;
    case 29: hipLaunchKernelGGL((gemm_x3d_kernel<64>), grid, block, 0, st, g, pl); break;
    default: hipLaunchKernelGGL((gemm_x3w_kernel<0, 0>), grid, block, 0, st, g, pl); break;
  }
}

void launch_x3w_hybrid(const GemmArgs& g, const f16x3::PlaneArgs& pl, bool arc, int persistent_ctas, int skew, hipStream_t st) {
  const long long items = static_cast<long long>(g.tiles_m) * g.tiles_n * g.splits;
  if (persistent_ctas > 0) {
    const dim3 pgrid(static_cast<unsigned>(items < persistent_ctas ? items : persistent_ctas)), pblock(512);
    if (arc) hipLaunchKernelGGL((f16x3::gemm_x3dp_kernel<true>), pgrid, pblock, 0, st, g, pl, static_cast<int>(items), skew);
    else hipLaunchKernelGGL((f16x3::gemm_x3dp_kernel<false>), pgrid, pblock, 0, st, g, pl, static_cast<int>(items), skew);
    return;
  }
  const dim3 grid(static_cast<unsigned>(items)), block(512);
  if (arc) hipLaunchKernelGGL((f16x3::gemm_x3dh_kernel<true>), grid, block, 0, st, g, pl);
  else hipLaunchKernelGGL((f16x3::gemm_x3dh_kernel<false>), grid, block, 0, st, g, pl);
}

}  // namespace sg

// development only (not declared in include/stargcn.h): read and reset the phase counters of gemm_x3w_kernel<., OPT & 16>
extern "C" __attribute__((visibility("default"))) int sg_x3w_timing_read(unsigned long long* out16) {
  unsigned long long z[16] = {0};
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(sg::f16x3::g_x3w_timing), sizeof(z)) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(sg::f16x3::g_x3w_timing), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
