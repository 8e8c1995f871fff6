// gemm_x3w.hip -- the f16x3 product (gemm_f16x3.hip: fp32-accurate GEMM on the f16 matrix cores, three matrix instructions
// per product on block-scaled f16 plane pairs) for the step's BIG products: 256 x 256 output tiles, eight waves, ONE persistent
// workgroup per CU, op(A) fp32 in memory and split inside the kernel, op(B) as planes by LDS-DMA.
//
// What was wrong with 128 x 128 tiles (gemm_f16x3.hip).  Per 16-k step a wave issues 12 matrix instructions (384
// pipe cycles) against 8 fragment reads, 2-3 LDS-DMA instructions and a workgroup barrier; a CU at the matrix-core rate
// would take in 43 B / clk of planes and read 85 B / clk of fragments.  On 256 x 256 a wave owns 128 x 64 of the result: 48
// matrix instructions per 32-k tile against 24 fragment reads, 4 (B only) DMA instructions and one barrier -- and with
// N <= 256 every fp32 block of A is converted ONCE (the 128-wide form converts it once per column tile and per 24 matrix
// instructions, this one per 48).  Measured (profiles/r5_gemm_wide.md): the config-5 products 1 M x 256 x 4160 (K- and
// row-contiguous A), 4096 x 256 x 1 M and 256 x 4160 x 1 M go from 233 / 234 / 211 / 192 to 303-311 / 289-311 / 267-274 /
// 269-275 TFLOP/s-equivalent.
//
// Registers decide the arithmetic.  The running result of a 128 x 64 wave tile is 128 VGPRs of the 256 a wave has with two
// waves per SIMD, so there is no room for the block-local product set P of gemm_f16x3.hip (a second 128).  DIRECT
// accumulation instead: the running result of a 32 x 32 product tile is held in units of the CURRENT scale block's 2^e
// (e = exponent of A's block + exponent of B's block), the matrix instructions accumulate straight into it, and when the
// next block has another e the tile is first multiplied by 2^(e_prev - e) -- a power of two: exact.  Nothing in the loop
// waits for a matrix instruction (the rescale reads results finished phases ago).  Rounding: three fp32 accumulations per
// 16 k into the running sum -- the exact-fp32 MFMA kernel (32x32x2) makes eight; measured against fp64 on random and on
// all-positive operands up to K = 10^6 it stays below the exact kernel's error and inside the same bound as the block-local
// form (tools/x3w_harness acc).  Exactness needs the scale not to DROP too far: a block more than 2^60 below the largest
// scale of the slice so far cannot be brought in by rescaling the running sum up (overflow).  The kernel then raises
// PlaneArgs::flag and the launcher's fallback -- the 128-wide kernel, launched behind it, which returns at once unless the
// flag is up -- redoes the product.  Block scales of real operands move by a few bits.
// A's converter also KEEPS the previous block's exponent while the new block's maximum still lands in [2^12, 2^15) under it
// (two bits of the residual plane's 2^18 range): the running results are rescaled rarely.
//
// The bound this kernel runs into is POWER, not issue slots: with two waves per SIMD the fragment reads, DMA and scalar work
// do hide behind the matrix instructions (tools/mfma_fill.cpp: 1.1 / 3.3 / 0.6 pipe cycles per VALU / ds_read_b128 / SALU
// filler), but removing them (timing-only ablations) raises the CLOCK, not the issue rate -- 1.76 GHz with everything,
// 1.94 without the DMA, 2.24 without DMA and fragment reads (GRBM_GUI_ACTIVE / wall, profiles/r5_gemm_wide.md).  What
// helps is fewer bytes moved per matrix instruction, which is what the 256-wide tile does.
//
// Persistent: see the kernel.  Stores: a CU retires one 64 x 16 B wave store per ~90 cycles whatever the pattern
// (tools/store_rate.cpp: 5.4-6.0 TB/s streaming with 128-byte, 256-byte or 1 KiB row segments); a persistent workgroup
// cannot overlap them with its own matrix work (all eight waves finish an item together), which is why products whose
// OUTPUT is the 16 GB matrix (1 M x 4096 x 256) stay on the 128-wide kernels, three workgroups per CU -- measured 11.4 ms
// here against 9.0 there, 6.2 with the stores removed.
//
// Epilogue / split-K contract: GemmArgs, as in gemm_f16x3.hip (the launcher there splits B and calls in here).
#include "gemm_x3_shared.hpp"

namespace sg {
namespace f16x3 {

constexpr int WSTG = 32 * UNIT;                      // one stage of one operand: 8 row blocks x (2 k steps x 2 planes) units

// LDS-DMA of 64 x 16 bytes: global (per-lane address) -> LDS (wave-uniform base in m0, lane-linear).  Assembly, so that
// the compiler keeps no record of a pending LDS write (it would wait with vmcnt(0) before the next LDS access that may
// alias it); the counted waits below are the only ordering.
__device__ __forceinline__ void dma_unit(const char* src, unsigned lds) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // m0 is reserved: nothing else in these kernels uses it
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds) : "memory", "m0");
#pragma clang diagnostic pop
}


// ---------------------------------------------------------------------------------------------------------------------
// One workgroup per CU walks the work items b, b + G, b + 2 G, ... (item = output tile x K slice; G = gridDim.x) as ONE
// stream of 32-k tiles.  The load / convert / DMA stages run one to two tiles ahead of the matrix work straight across item
// boundaries: an item's first tile is already converted when the previous item's last matrix instruction issues (at
// K = 256 -- eight tiles per item -- pipeline fill, drain and dispatch were 44 % of a one-item-per-workgroup form).
// Per tile g of the stream (stage and fragment-set parity PAR = g & 1), each wave:
//   phase i = 0..3   12 matrix instructions: A row block i of the wave's four x its two B row blocks x two k steps x three
//                    plane pairs, two interleaved accumulation chains; the A fragments of phase i+1 are requested first
//   phase 0, 3       two LDS-DMA instructions each: B's planes of tiles g+1 (second half) and g+2 (first half)
//   end of phase 2   the wave's 32 x 32 fp32 block of A of tile g+1 (requested a tile ago) -> block maximum (DPP) ->
//                    exponent -> two f16 planes -> LDS stage PAR^1; then vmcnt(0): B's tile g+1 has landed
//                    request A's block of tile g+2 (the registers are free again)
//   phase 3          lgkmcnt(0), THE barrier of the tile: tile g+1 is complete and tile g's stages are free; read tile
//                    g+1's first fragments and exponents
// A's scale block is 32 x 32 (one tile), B's 32 x 64.  LDS: 2 x 32 KiB A planes + 2 x 32 KiB B planes + 16 KiB epilogue
// staging (2 KiB per wave: 16 x 32 blocks -> 128-byte row segments; the stages stay live across an item's end) + A's block
// exponents + 8 KiB B exponents (per wave, double-buffered chunks of 64 blocks).
// ARC = false: A element (m, k) at A[m * lda + k] (K % 4 == 0, 16-byte aligned rows); ARC = true: at A[k * lda + m] with
// M % 4 == 0 and 16-byte aligned k rows (float4 loads along m, register transpose).
// Every K slice must hold at least two tiles (the launcher checks).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int P_STG_W = 16 * 32 * 4;
constexpr int P_EB_W = 2 * 2 * 64 * 4;               // per wave: [buffer][B row block j][64-k block of the chunk] exponents
constexpr int P_SMEM = 4 * WSTG + 8 * P_STG_W + 64 + 8 * P_EB_W;

struct ItemRef {          // one (output tile, K slice); wave-uniform.  Everything else is derived where it is used (scalar
  int tm, tn, z, T;       // arithmetic is cheap; two of these live across the whole loop)
};

template <bool ARC>
__global__ __launch_bounds__(512, 1) void gemm_x3dp_kernel(const GemmArgs g, const PlaneArgs pl, const int n_items) {
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
        load_a(r2, kt_r2);      // the registers are free again: A's block of tile g+2 is requested a phase ahead of the barrier
        asm volatile("" ::: "memory");
      }
    }
    if (new_chunk) { cb ^= 1; need_req = true; }
  };

  // ---- an item's result: scale, stage 16 x 32 blocks through the wave's own LDS block, store ----
  auto finish = [&]() __attribute__((always_inline)) {
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

// g.tiles_m / g.tiles_n count 256-wide tiles; pl.pb / pl.exp_b are B's planes and exponents (gemm_f16x3.hip's split_kernel),
// pl.flag the exactness flag (see the head of this file); `ctas` workgroups (the CU count)
void launch_x3w_hybrid(const GemmArgs& g, const f16x3::PlaneArgs& pl, bool arc, int ctas, hipStream_t st) {
  const long long items = static_cast<long long>(g.tiles_m) * g.tiles_n * g.splits;
  const dim3 grid(static_cast<unsigned>(items < ctas ? items : ctas)), block(512);
  if (arc) hipLaunchKernelGGL((f16x3::gemm_x3dp_kernel<true>), grid, block, 0, st, g, pl, static_cast<int>(items));
  else hipLaunchKernelGGL((f16x3::gemm_x3dp_kernel<false>), grid, block, 0, st, g, pl, static_cast<int>(items));
}

}  // namespace sg
