// agg_fused.hip -- aggregate -> contract in ONE kernel: the multi-link graph convolution without its R-expanded intermediate.
//
//   out[i, :] = act( sum_r ( sum_{e in seg(i, r)} w_e x[idx_e, :] ) B_r  +  sum_r rowsum[i, r] b_r )
//
// Reference: MultiLinkGCNAggregator.hybrid_forward (mxgraph/layers/aggregators.py:141-160) computes, per rating level r,
// h_r = FullyConnected(x) (written, n_src x U) and seg_weighted_pool(h_r) and adds the R results; the unfused path of
// this library (multilink.hip) does one gather into an R-expanded matrix (n x R*D floats: 16 GB at the config-5 shard)
// and one GEMM that reads it back.  Here a workgroup owns a TILE of 64 destination rows; for every level r the
// aggregate Z_r (64 x 256 fp32) exists only as two f16 planes in LDS, is multiplied by B_r (256 x 256, f16 planes read
// through L2) on the matrix cores, and only `out` (and, for the backward's weight gradient, optionally Z itself) is written.
//
// Work split inside the workgroup (12 waves = 3 per SIMD, 168 registers each; ONE workgroup per CU, persistent over its tiles;
// profiles/r5_fused_kernel.md -- 8 + 8 waves with 16 rows in flight per gather wave, the first version, is 1.4 % slower):
//   waves 0-7   "G"  gather.  Each owns one EIGHTH of the tile's edges at the current level, cut at any edge (a hub row is
//               summed in pieces by several waves; with cuts at row boundaries the busiest wave of a (tile, level) had 1.45 x
//               / 2.2 x its even share on the config-5 shard graph, into users / into items), streams their source rows --
//               1 KiB per row, one float4 per lane, NB = 32 rows in flight per wave across row, level and tile boundaries: a
//               three-stage stream (plan entries of group s + 2, row loads of group s + 1, FMAs of group s) -- accumulates in
//               fp32 group by group, and when a row is complete scales it by a power of two (row maximum -> [2^14, 2^15)),
//               splits it into an f16 value + f16 residual and writes both to the level's LDS buffer (+ the fp32 row to
//               `zsave`).  A row cut between waves: every wave but the one in whose share the row ends leaves its piece as an
//               fp32 partial row in LDS (one slot per wave: at most one row runs on past a share's end), and the last one adds
//               them to its own piece between the item's two barriers.
//   waves 8-11  "M"  matrix.  Each owns 64 of the 256 output columns (two 32-column blocks, one after the other) for all 64
//               rows: after the barrier that publishes level r's planes it runs, per block, 16 k-steps x 6 v_mfma_f32_32x32x16_f16 (value x value, value x residual, residual x value:
//               fp32 accuracy, gemm_f16x3.hip) into a level-local product P, with B_r's fragments loaded straight from L2 into
//               registers (fragment-major planes, one 1 KiB unit per wave load; in assembly: scalar base + lane offset,
//               counted waits), then folds P * 2^-(e_row + e_B) into the running result.  After the last level: bias term
//               from the tile's rowsum rows, activation, store.
//   Two s_barriers per (tile, level), back to back for the M waves: at the first every piece of level q+1 is in LDS (buffer
//   (q+1)&1 or a partial slot) and M has finished reading level q-1 from that buffer; between the two the cut rows are put
//   together; the second publishes the level.  The G waves never wait for memory at a barrier: loads stay in flight across
//   it (only LDS traffic is drained).
// What binds (measured): the gather side alone runs at 6.4-7.1 TB/s algorithmic on the config-5 shard graph; the matrix
// INSTRUCTIONS cost 0.5 ms of a 23.7 ms launch, the B planes' loads 2.5-3.9 ms -- they share the CU's L1 miss queue with the
// gathered rows, and 4 MB of planes do not survive in a 4 MB L2 that the rows stream through (hit rate 0.17).
//
// Plan ("f-plan"): the edges of a tile reordered level-major inside the tile's own edge range -- 65 absolute edge offsets per
// (launch slot, level) at f_ptr[(slot R + r) 65 ..] -- so that a level's edges of a tile are one contiguous run.  Launch slots
// are the tiles sorted by descending edge count, dealt to the workgroups boustrophedon.  Built on the device (plan_kernel).
//
// Accuracy: Z_r rows carry one scale per (row, level) (256 elements), B_r one per (level, 32 output columns); error model
// as gemm_f16x3.hip (block-relative 3 x 2^-22 per product term); the aggregation itself is plain fp32 FMA, edges of a row
// summed in groups of NB = 32.
#include "gemm_x3_shared.hpp"

#include <mutex>
#include <vector>

namespace sg {
namespace fused {

// measurement aid (bench.py): HIP events around every fused launch on its own stream while sg_agg_fused_profile_enable is on
struct ProfRec { hipEvent_t a, b; int64_t nnz; int zsave; };
static std::mutex g_mu;
static bool g_on = false;
static std::vector<ProfRec> g_recs;
static long prof_begin(hipStream_t st, int64_t nnz, int zsave) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_on) return -1;
  ProfRec r{};
  r.nnz = nnz; r.zsave = zsave;
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  return static_cast<long>(g_recs.size()) - 1;
}
static void prof_end(long rec, hipStream_t st) {
  if (rec < 0) return;
  std::lock_guard<std::mutex> lk(g_mu);
  if (rec < static_cast<long>(g_recs.size())) (void)hipEventRecord(g_recs[rec].b, st);
}

using f16x3::f16x2;
using f16x3::f16x8;
using f16x3::f32x16;
using f16x3::f32x2;
using f16x3::f32x4;
using f16x3::wave_max_nonneg;

constexpr int TM = 64;                    // destination rows per tile
constexpr int KD = 256;                   // width of the gathered rows (contraction length per level)
constexpr int ND = 256;                   // output width
constexpr int KS = KD / 16;               // k steps per level
constexpr int NJB = ND / 32;              // 32-column blocks of the output
#ifndef SG_FUSED_DIRECT
#define SG_FUSED_DIRECT 0                 // 1: the matrix instructions accumulate straight into the running result (UNITS below): no
#endif                                    // level-local product registers, seven B fragment sets in flight instead of three.  Measured
                                          // (profiles/r5_fused_kernel.md): the same launch time -- B's loads and the gathered rows share
                                          // the CU's L1 miss queue, a deeper ring only moves who waits -- and 4e-7 instead of 1e-7 error:
                                          // kept as a build option, not shipped
#ifndef SG_FUSED_ASMLOAD
#define SG_FUSED_ASMLOAD 1                // B fragment loads in assembly (scalar base + lane offset, counted waits)
#endif
#ifndef SG_FUSED_GW
#define SG_FUSED_GW 8
#endif
#ifndef SG_FUSED_NB
#define SG_FUSED_NB 32
#endif
#ifndef SG_FUSED_BRING
#define SG_FUSED_BRING (SG_FUSED_DIRECT ? 7 : 3)
#endif
#ifndef SG_FUSED_MW
#define SG_FUSED_MW 4
#endif
#ifndef SG_FUSED_EVEN
#define SG_FUSED_EVEN 1                   // 0: a level's edges go to the gather waves at ROW boundaries (the first version: a hub row is one wave's)
#endif
#ifndef SG_FUSED_ROWPOLICY
#define SG_FUSED_ROWPOLICY 1              // cache-policy immediate of the rows' buffer loads: 1 sc0 (ships), 0 none, 16 sc1, 2 nt
#endif
#ifndef SG_FUSED_SKIPB
#define SG_FUSED_SKIPB 0                  // timing probe only (see load_b)
#endif
#ifndef SG_FUSED_ADB
#define SG_FUSED_ADB 0                    // 1: the aggregate's fragments double-buffered in registers (fits only with 4 + 4 waves)
#endif
constexpr int GW = SG_FUSED_GW;           // gather waves per workgroup (4 or 8); the matrix waves follow them
constexpr int MW = SG_FUSED_MW;           // matrix waves (4: 64 output columns each, 8: 32)
constexpr int NJ = 8 / MW;                // 32-column blocks per matrix wave
constexpr int NB = SG_FUSED_NB;           // rows in flight per G wave
constexpr int BRING = SG_FUSED_BRING;     // B fragment sets in flight per M wave (k steps of one 32-column block)
constexpr int ZROW = KD * 2 + 16;         // bytes per row and plane in LDS: 528 = 132 words -> rows 4 banks apart
constexpr int ZPLANE = TM * ZROW;
constexpr int ZBUF = 2 * ZPLANE;          // value plane, residual plane
constexpr int SMEM_BASE = 2 * ZBUF + 2 * TM * 4 + 2 * TM * SG_MAX_LINKS * 4 + 4 * TM * 4;
constexpr int SMEM = SMEM_BASE + (SG_FUSED_EVEN == 1 ? GW * 1024 : 0);       // + one fp32 partial row per gather wave
#if SG_FUSED_EVEN == 1 && SG_FUSED_DIRECT
#error "SG_FUSED_EVEN is not written for SG_FUSED_DIRECT"
#endif

struct Args {
  const int32_t* f_ptr;
  const int32_t* f_idx;
  const float* f_w;
  const int32_t* tile_order;   // may be null
  const float* x;
  long long ldx;
  const char* wplanes;         // unit (((r NJB + jb) KS + ks) 2 + plane): lane l holds B_r[n = 32 jb + (l & 31)][k = 16 ks + 8 (l >> 5) ..+7]
  const float* wscale;         // (R NJB) 2^-e of the block
  const int* wexp;             // (R) e of the level (SG_FUSED_DIRECT: one exponent per level)
  const float* bias;           // (R, ND) packed, or null
  const float* rowsum;         // (n_dst, R), or null
  float* out;
  long long ldo;
  float* zsave;                // (n_dst, ldz) fp32 aggregates [r KD + k], or null
  long long ldz;
  int n_dst, n_tiles, R;
  int in_dim;                  // floats per gathered row that exist in memory (multiple of 4, <= KD); the lanes beyond read nothing
  int out_dim;                 // output columns written per level (<= ND); B's planes are zero beyond
  int stack;                   // 1: accum 'stack' -- every level's product (+ its own bias term, activation) goes to its own column block
                               //    [r * out_dim, (r + 1) * out_dim) of `out`; 0: the levels are summed (accum 'sum')
  long long x_lstride;         // level r gathers from x + r * x_lstride (the data gradient of 'stack': level r reads its own column block)
  int act;
  float slope;
  int ablate;                  // timing experiments only (SG_FUSED_ABLATE): 1 = no matrix work, 2 = every row load reads row 0, 64 = every level's B planes = level 0's
};

__device__ __forceinline__ unsigned wave_or(unsigned v) {
  auto step = [&](auto ctrl, auto row_mask) __attribute__((always_inline)) {
    v |= static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), decltype(ctrl)::value,
                                                            decltype(row_mask)::value, 0xf, true));
  };
  using std::integral_constant;
  step(integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});
  step(integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});
  step(integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});
  step(integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});
  step(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});
  step(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});
  return static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63));
}

// maximum of a NON-NEGATIVE float over the wave, on the bit patterns (they order like the values; a NaN pattern would sort above
// infinity, but the callers' fmaxf has dropped NaNs already): unsigned max takes the DPP operand directly -- six instructions,
// where fmaxf on a DPP move costs a move, the max and a canonicalising max per step.  One per emitted row: 20 M per launch.
__device__ __forceinline__ float wave_max_bits(float f) {
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
  step(integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});    // row_mirror
  step(integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});    // row_bcast:15
  step(integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});    // row_bcast:31
  return __uint_as_float(static_cast<unsigned>(__builtin_amdgcn_readlane(static_cast<int>(v), 63)));
}

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long rfl64(unsigned long long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

struct Ctx {                    // one (item = tile x level, G wave): the rows and edges this wave owns
  int pv, pn;                   // per lane j: first / past-the-end edge of tile row j at this level
  unsigned long long rows;      // my non-empty rows
  unsigned long long empt;      // my empty rows
  int e_lo, e_hi, ng;           // my edges; groups of NB (at least one, possibly all padding)
  int tile, r, it;              // tile = launch slot of the tile; it = ordinal of the item in this workgroup's sequence
  int eb;                       // exponent of the level's B planes (SG_FUSED_DIRECT)
  int split;                    // SG_FUSED_EVEN: tail row | head row << 8 | first wave of the head row << 16 (row 64 = none) | empty share << 24
};

// GEN = false: the shape the kernel is built and tuned for -- 256-float rows, 256 output columns, accum 'sum' (in_dim, out_dim,
// stack, x_lstride of Args are not read; the code is the round-5 kernel instruction for instruction).  GEN = true: any supported
// widths, 'stack', level-strided rows (correct and deterministic, not tuned: the compiler spills in this instantiation).
template <bool ZSAVE, bool NT, bool BUF, bool GEN>
__global__ __launch_bounds__(64 * (GW + MW), 1) void agg_contract_kernel(const Args a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* sinv = reinterpret_cast<float*>(smem + 2 * ZBUF);       // [buffer][row] 2^-e of the row's planes
  // UNITS (SG_FUSED_DIRECT).  The matrix waves keep a tile's running result in units of 2^-u[row], u = e_row + e_B of the level
  // last accumulated (e_B: ONE exponent per level of B), and the matrix instructions add a level's products straight into it --
  // no level-local product registers, which pays for six B fragment sets in flight instead of three.  A level whose natural
  // unit lies within 2^8 above the smallest unit the row has seen (or below it) is taken as it is and the running result is
  // first multiplied by fac[row] = 2^(u_new - u_old) <= 2^8 (exact); a level further down -- its products are more than 2^8
  // smaller than those of the row's largest level -- is written to the planes in the CURRENT unit instead (fewer significant
  // bits of its own, the same absolute resolution as the terms that dominate the sum), and an all-zero row keeps the unit.
#if SG_FUSED_DIRECT
  int* urun = reinterpret_cast<int*>(smem + 2 * ZBUF + 2 * TM * 4 + 2 * TM * SG_MAX_LINKS * 4);   // [row] unit exponent after the row's last level
  int* umin = urun + TM;                                          // [row] smallest unit exponent so far (1 << 20: none yet)
  float* fac = reinterpret_cast<float*>(umin + TM);               // [buffer][row] factor for the running result at this level
#endif
  const int t = threadIdx.x, lane = t & 63, wave = rfl(t >> 6);
  const int G = gridDim.x, b = blockIdx.x;
  // launch slots of this workgroup: stratum ti (G consecutive slots) in boustrophedon order -- b, 2G-1-b, 2G+b, ... -- so that
  // with slots sorted by descending work no workgroup collects the heaviest tile of every stratum
  auto slot_of = [&](int ti) __attribute__((always_inline)) { return ti * G + ((ti & 1) ? G - 1 - b : b); };
  const int n_full = a.n_tiles / G;
  const int n_my = n_full + ((a.n_tiles - n_full * G > 0 && slot_of(n_full) < a.n_tiles) ? 1 : 0);
  const int n_items = n_my * a.R;
  if (n_my == 0) return;
  // (Measured and removed, profiles/r5_fused_kernel.md: walking the tiles once per PHASE of a few levels so that the phase's
  // B planes stay in the L2s -- the fabric reads fall from 174 to 146-158 GB per launch, the launch does not get faster, and
  // the partial results written and re-read between phases make the whole step slower; LDS-DMA prefetch of the planes: slower.)
  auto tile_of = [&](int slot) __attribute__((always_inline)) -> int {
    if (!a.tile_order) return slot;
    return *((f16x3::cst_int*)(a.tile_order) + slot);
  };

  if (wave < GW) {
    // ================================================= G: gather =====================================================
    const int gw = wave;
    auto load_ptrs = [&](int it, int& slot, int& r, int& pv, int& pn) __attribute__((always_inline)) {
      const int itc = min(it, n_items - 1);
      const int ti = itc / a.R;
      r = itc - ti * a.R;
      slot = slot_of(ti);
      const long long base = (static_cast<long long>(slot) * a.R + r) * (TM + 1);
      pv = a.f_ptr[base + lane];
      pn = a.f_ptr[base + lane + 1];
    };
    auto make_ctx = [&](int it, int slot, int r, int pv, int pn) __attribute__((always_inline)) {
      Ctx c;
      // real copies (not aliases of the F registers): the next item's pointers are loaded into those while this context lives
      asm volatile("v_mov_b32 %0, %1" : "=v"(c.pv) : "v"(pv));
      asm volatile("v_mov_b32 %0, %1" : "=v"(c.pn) : "v"(pn));
      c.tile = slot; c.r = r; c.it = it; c.split = 64 | (64 << 8);
      c.eb = SG_FUSED_DIRECT ? static_cast<int>(*((f16x3::cst_int*)(a.wexp) + r)) : 0;
      const int p0 = __builtin_amdgcn_readlane(pv, 0), pE = __builtin_amdgcn_readlane(pn, 63);
      const long long total = pE - p0;
      int wj = 0;                                        // the wave whose share of the level's edges holds the row's first edge
      if (total > 0) {
#pragma unroll
        for (int q = 1; q < GW; ++q) wj += (pv >= p0 + static_cast<int>(total * q / GW)) ? 1 : 0;
      }
#if SG_FUSED_EVEN == 1
      // the level's edges cut into GW equal shares at ANY edge: a row that straddles a cut is summed in pieces.  A piece whose row
      // goes on into the next share ("tail": my last row) is left as an fp32 partial row in LDS; the wave in whose share the row ENDS
      // ("head": its first row) adds the partials of the waves before it between the item's two barriers and emits the row.
      const int c_lo = p0 + static_cast<int>(total * gw / GW), c_hi = p0 + static_cast<int>(total * (gw + 1) / GW);
      // (Measured, profiles/r5_fused_kernel.md section 6: leaving rows of <= 4 / 16 / 64 edges whole changes nothing or loses.)
      const int pe = min(pn, c_hi), ps = max(pv, c_lo);
      const bool have = pe > ps;
      const unsigned long long nonempty = __ballot(pn > pv);
      c.rows = __ballot(have);
      c.empt = __ballot(wj == gw) & ~nonempty;
      const unsigned long long tailm = __ballot(have && pn > c_hi);
      const unsigned long long headm = __ballot(have && pv < c_lo && pn <= c_hi);
      const int tj = tailm ? __ffsll(static_cast<long long>(tailm)) - 1 : 64;
      const int hj = headm ? __ffsll(static_cast<long long>(headm)) - 1 : 64;
      const int hu = headm ? __builtin_amdgcn_readlane(wj, hj & 63) : 0;
      // (with fewer than GW edges some shares are empty; one that lies inside a cut row must still hand a partial row -- zeros -- to
      //  the wave that puts the row together: bit 24)
      c.split = tj | (hj << 8) | (hu << 16) | ((total > 0 && c_lo == c_hi) ? (1 << 24) : 0);
      asm volatile("v_mov_b32 %0, %1" : "=v"(c.pn) : "v"(pe));
      c.e_lo = c_lo; c.e_hi = c_hi;
#else
      const unsigned long long mine = __ballot(wj == gw);
      const unsigned long long nonempty = __ballot(pn > pv);
      c.rows = mine & nonempty;
      c.empt = mine & ~nonempty;
      if (mine != 0ull) {
        const int j_lo = __ffsll(static_cast<long long>(mine)) - 1;
        const int j_hi = j_lo + __popcll(mine);
        c.e_lo = __builtin_amdgcn_readlane(pv, j_lo);
        c.e_hi = __builtin_amdgcn_readlane(pn, j_hi - 1);
      } else {
        c.e_lo = 0; c.e_hi = 0;
      }
#endif
      if (it >= n_items) { c.rows = 0ull; c.empt = 0ull; c.e_hi = c.e_lo; c.split = 64 | (64 << 8); }      // past the end of the stream: padding only
      c.ng = max(1, (c.e_hi - c.e_lo + NB - 1) / NB);
      return c;
    };
    // (idx, w) of the NB edges of group gi, one edge per lane (lanes >= NB and edges past the end: clamped, never used)
    auto load_meta = [&](const Ctx& c, int gi, int& m_idx, float& m_w) __attribute__((always_inline)) {
      const int e = max(min(c.e_lo + gi * NB + lane, c.e_hi - 1), 0);
      m_idx = a.f_idx[e];
      m_w = a.f_w[e];
    };
    // bit k: edge k of the group is the last edge of one of my rows
    auto end_mask = [&](const Ctx& c, int gi) __attribute__((always_inline)) {
      const int gb = c.e_lo + gi * NB;
      const int rel = c.pn - 1 - gb;
      const bool mine = ((c.rows >> lane) & 1ull) != 0ull;
      const unsigned bit = (mine && rel >= 0 && rel < NB) ? (1u << rel) : 0u;
      return wave_or(bit);
    };

    const char* xb = reinterpret_cast<const char*>(a.x);
    // BUF: the gathered matrix ends below 4 GB from its first byte -- rows through a buffer resource (wave-uniform base, ONE 32-bit
    // offset per lane instead of a 64-bit address: fewer address instructions on the issue-bound gather waves) with sc0 (no L1
    // allocation for rows nobody re-reads).  Measured, profiles/r5_fused_kernel.md section 8: 21.2 -> 20.5 ms into users; nt 23-25.
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, BUF ? 0xffffffff : 0, 0x00020000);
    // Rows narrower than KD (in_dim < 256): the lanes past the row's end re-read its last float4 (a valid address, no branch around
    // the load) and their sums are zeroed when the row is emitted.  `roff`: byte offset of the level's column block (x_lstride).
    const bool lane_live = GEN ? lane * 4 < a.in_dim : true;
    const unsigned lane_off = static_cast<unsigned>(GEN ? min(lane, (a.in_dim >> 2) - 1) : lane) * 16u;
    const int zdim = GEN ? a.in_dim : KD;             // width of a level's block in zsave
    auto load_row = [&](f32x4& dst, int idx, unsigned roff) __attribute__((always_inline)) {
      if (a.ablate & 2) idx = 0;        // (timing experiment: every load hits the same row -- no control flow around the load,
                                        //  the compiler must keep counting the outstanding loads)
      if (!GEN) roff = 0u;
      if (BUF) {
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        const unsigned voff = static_cast<unsigned>(idx) * static_cast<unsigned>(a.ldx * 4) + lane_off + roff;
        dst = __builtin_bit_cast(f32x4, static_cast<u32x4_t>(__builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, 0, SG_FUSED_ROWPOLICY)));
      } else {
        const f32x4* p = reinterpret_cast<const f32x4*>(xb + static_cast<long long>(idx) * a.ldx * 4 + roff + lane_off);
        if (NT) dst = __builtin_nontemporal_load(p);
        else dst = *p;
      }
    };
    const unsigned lstride4 = GEN ? static_cast<unsigned>(a.x_lstride * 4) : 0u;

#if SG_FUSED_DIRECT
    // the exponent a row's planes are written with at this level, and the bookkeeping of the row's unit (see UNITS above)
    auto unit_of = [&](const Ctx& c, int j, int e_nat, bool zero) __attribute__((always_inline)) {
      const int eb = c.eb;
      const bool first = c.r == 0;
      const int u_old = first ? 0 : rfl(urun[j]);
      const int um_old = first ? (1 << 20) : rfl(umin[j]);
      int u;
      if (zero) u = first ? eb : u_old;
      else if (e_nat + eb <= um_old + 8) u = e_nat + eb;
      else u = u_old;
      const int e_use = min(max(u - eb, -126), 126);
      u = e_use + eb;
      const int d = min(max(u - u_old, -127), 8);
      if (lane == 0) {
        urun[j] = u;
        umin[j] = zero ? um_old : min(um_old, u);
        fac[(c.it & 1) * TM + j] = d <= -127 ? 0.f : __uint_as_float(static_cast<unsigned>(127 + d) << 23);
      }
      return e_use;
    };
#endif
    // a finished row: row maximum -> scale -> two f16 planes in the item's LDS buffer (+ the fp32 row to zsave)
    auto emit = [&](const Ctx& c, int j, const f32x4& acc_in) __attribute__((always_inline)) {
      f32x4 acc = acc_in;
      if (GEN && !lane_live) acc = f32x4{0.f, 0.f, 0.f, 0.f};
      float mx = fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3])));
      mx = wave_max_bits(mx);
      bool nonfinite = false;
      if (__builtin_expect(!(mx <= 3.402823466e38f), 0)) {
        nonfinite = true;
        float mf = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float v = fabsf(acc[i]); mf = fmaxf(mf, v <= 3.402823466e38f ? v : 0.f); }
        mx = wave_max_bits(mf);
      }
      int e = 0;
      {
        const unsigned bits = __float_as_uint(mx);
        const int ex = static_cast<int>((bits >> 23) & 0xffu);
        if (bits != 0u && ex != 0xff) e = 14 - (max(ex, 1) - 127);
        e = min(max(e, -126), 126);
      }
#if SG_FUSED_DIRECT
      e = unit_of(c, j, e, mx == 0.f);
#endif
      const float sc = __uint_as_float(static_cast<unsigned>(127 + e) << 23);
      unsigned h1[2], h2[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f32x2 v = {acc[2 * i] * sc, acc[2 * i + 1] * sc};
        const f16x2 hi = __builtin_convertvector(v, f16x2);
        f32x2 res = {__builtin_fmaf(acc[2 * i], sc, -static_cast<float>(hi[0])),
                     __builtin_fmaf(acc[2 * i + 1], sc, -static_cast<float>(hi[1]))};
        if (nonfinite) {
          if (fabsf(v[0]) == __builtin_inff()) res[0] = 0.f;
          if (fabsf(v[1]) == __builtin_inff()) res[1] = 0.f;
        }
        const f16x2 lo = __builtin_convertvector(res, f16x2);
        h1[i] = __builtin_bit_cast(unsigned, hi);
        h2[i] = __builtin_bit_cast(unsigned, lo);
      }
      char* zb = smem + (c.it & 1) * ZBUF + j * ZROW + lane * 8;
      *reinterpret_cast<uint2*>(zb) = make_uint2(h1[0], h1[1]);
      *reinterpret_cast<uint2*>(zb + ZPLANE) = make_uint2(h2[0], h2[1]);
      if (lane == 0) sinv[(c.it & 1) * TM + j] = __uint_as_float(static_cast<unsigned>(127 - e) << 23);
      if (ZSAVE) {
        const long long row = static_cast<long long>(tile_of(c.tile)) * TM + j;
        if (row < a.n_dst && lane_live)
          __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(a.zsave + row * a.ldz + static_cast<long long>(c.r) * zdim) + lane);
      }
    };
    auto emit_zero = [&](const Ctx& c, int j) __attribute__((always_inline)) {
      char* zb = smem + (c.it & 1) * ZBUF + j * ZROW + lane * 8;
      *reinterpret_cast<uint2*>(zb) = make_uint2(0u, 0u);
      *reinterpret_cast<uint2*>(zb + ZPLANE) = make_uint2(0u, 0u);
#if SG_FUSED_DIRECT
      const int ez = unit_of(c, j, 0, true);
      if (lane == 0) sinv[(c.it & 1) * TM + j] = __uint_as_float(static_cast<unsigned>(127 - ez) << 23);
#else
      if (lane == 0) sinv[(c.it & 1) * TM + j] = 1.f;
#endif
      if (ZSAVE) {
        const long long row = static_cast<long long>(tile_of(c.tile)) * TM + j;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (row < a.n_dst && lane_live)
          __builtin_nontemporal_store(z, reinterpret_cast<f32x4*>(a.zsave + row * a.ldz + static_cast<long long>(c.r) * zdim) + lane);
      }
    };

    // ---- the stream: three stages one group apart -- M loads (idx, w) of group s + 2, I issues the row loads of group
    // s + 1, C consumes group s.  Every item has at least one group, so the three stages span at most three items.
    int tF, rF, pvF, pnF, itF = 0;                       // F: pointers of the item after stage M's
    load_ptrs(0, tF, rF, pvF, pnF);
    Ctx cM = make_ctx(0, tF, rF, pvF, pnF);
    int giM = 0;
    itF = 1;
    load_ptrs(itF, tF, rF, pvF, pnF);
    auto advance_m = [&]() __attribute__((always_inline)) {
      if (giM + 1 < cM.ng) { ++giM; return; }
      cM = make_ctx(itF, tF, rF, pvF, pnF);
      giM = 0;
      ++itF;
      load_ptrs(itF, tF, rF, pvF, pnF);
    };

    int mI_idx, mM_idx;
    float mI_w, mM_w;
    f32x4 x[NB];
    float wv_cur, wv_nxt;                                // lane k: weight of edge k of the group (0 past the end of the item)
    unsigned end_cur, end_nxt;

    // fill: group 0 -> meta; group 1 -> meta, group 0 -> rows
    load_meta(cM, giM, mI_idx, mI_w);
    Ctx cI = cM;
    int giI = giM;
    advance_m();
    load_meta(cM, giM, mM_idx, mM_w);
    {
      const int gb = cI.e_lo + giI * NB;
      wv_cur = (gb + lane < cI.e_hi) ? mI_w : 0.f;
#pragma unroll
      for (int k = 0; k < NB; ++k) load_row(x[k], __builtin_amdgcn_readlane(mI_idx, k), static_cast<unsigned>(cI.r) * lstride4);
      end_cur = end_mask(cI, giI);
    }
    Ctx cC = cI;
    int giC = giI;
    cI = cM; giI = giM;
    mI_idx = mM_idx; mI_w = mM_w;
    advance_m();

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    f32x4 hacc = {0.f, 0.f, 0.f, 0.f};                   // SG_FUSED_EVEN: my piece of the head row
    unsigned long long rem = cC.rows;                    // my rows of the consume item that are still open

    for (;;) {
      // stage M
      load_meta(cM, giM, mM_idx, mM_w);
      // stage C enters an item: its empty rows (the barrier that freed this buffer ended the previous item)
      if (giC == 0) {
        unsigned long long em = cC.empt;
        while (em != 0ull) {
          const int j = __ffsll(static_cast<long long>(em)) - 1;
          em &= em - 1ull;
          emit_zero(cC, j);
        }
        rem = cC.rows;
#if SG_FUSED_EVEN == 1
        if ((cC.split >> 24) & 1) *reinterpret_cast<f32x4*>(smem + SMEM_BASE + gw * 1024 + lane * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
      }
      const int gbI = cI.e_lo + giI * NB;
      const unsigned roffI = static_cast<unsigned>(cI.r) * lstride4;
      wv_nxt = (gbI + lane < cI.e_hi) ? mI_w : 0.f;
      // a row's edges are summed group by group (NB edges in `part`, then `part` into the row's sum): the rounding error of a
      // 50 000-edge row grows with sqrt(edges / NB) instead of sqrt(edges) (the chunked gather of seg_gather.hip does the same)
      f32x4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        // C: edge k of group giC
        const float wk = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv_cur), k));
#pragma unroll
        for (int v = 0; v < 4; ++v) part[v] = __builtin_fmaf(wk, x[k][v], part[v]);
        if ((end_cur >> k) & 1u) {
          const int j = __ffsll(static_cast<long long>(rem)) - 1;
          rem &= rem - 1ull;
          acc += part;
#if SG_FUSED_EVEN == 1
          if (j == (cC.split & 0xff)) *reinterpret_cast<f32x4*>(smem + SMEM_BASE + gw * 1024 + lane * 16) = acc;
          else if (j == ((cC.split >> 8) & 0xff)) hacc = acc;
          else emit(cC, j, acc);
#else
          emit(cC, j, acc);
#endif
          acc = f32x4{0.f, 0.f, 0.f, 0.f};
          part = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // I: edge k of group giI into the registers just released
        load_row(x[k], __builtin_amdgcn_readlane(mI_idx, k), roffI);
      }
      acc += part;
      end_nxt = end_mask(cI, giI);
      // the consume item ends with this group: publish it
      const bool item_done = (giC + 1 == cC.ng);
      const bool last = item_done && (cC.it + 1 >= n_items);
      if (item_done) {
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#if SG_FUSED_EVEN == 1
        {     // every piece of the item is in LDS: the rows that END in my share but began in an earlier one
          const int hj = (cC.split >> 8) & 0xff;
          if (hj < 64) {
            f32x4 sum = hacc;
            for (int v = (cC.split >> 16) & 0xff; v < gw; ++v)
              sum += *reinterpret_cast<const f32x4*>(smem + SMEM_BASE + v * 1024 + lane * 16);
            emit(cC, hj, sum);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
#endif
      }
      if (last) break;
      // shift the stages
      cC = cI; giC = giI;
      cI = cM; giI = giM;
      wv_cur = wv_nxt;
      end_cur = end_nxt;
      mI_idx = mM_idx; mI_w = mM_w;
      advance_m();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // =================================================== M: matrix ======================================================
  const int wn = wave - GW;                                // output columns [32 NJ wn, 32 NJ (wn + 1))
  const int l31 = lane & 31, kh = lane >> 5;
  float* rs_lds = reinterpret_cast<float*>(smem + 2 * ZBUF + 2 * TM * 4);      // [tile parity][row][r] support row sums
  f32x16 acc[NJ][2];          // [column block][row block]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][i][q] = 0.f;

  // B fragment loads in assembly: wave-uniform 64-bit base in scalar registers + the lane's 32-bit offset.  (Written in C++ the
  // compiler forms one 64-bit VGPR address per unit, hoists all 32 of a level out of the loops and spills them.)  The compiler
  // does not see these loads: wait_b<N>() is the counted wait before a fragment set's first use -- N = loads issued after it.
  const unsigned lane16 = static_cast<unsigned>(lane) * 16u;
#if SG_FUSED_SKIPB
  bool skip_b = false;
#endif
  auto load_b = [&](f16x8 (&bf)[2], int r, int j, int ks) __attribute__((always_inline)) {
    if (a.ablate & 64) r = 0;       // (timing: every level multiplies by level 0's planes -- 256 KB that stay in the L2s)
    const char* ub = a.wplanes + ((((static_cast<long long>(r) * NJB + wn * NJ + j) * KS + ks) * 2) << 10);
#if SG_FUSED_SKIPB                  // (timing probe, tools/r6_fused_skipb.sh: every second tile of a workgroup reads ONE 2 KB unit for all its
    if (skip_b) ub = a.wplanes;     //  fragments -- what a 128-row tile, i.e. half the plane bytes per row, could save at best; results are wrong)
#endif
#if SG_FUSED_ASMLOAD
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(bf[0]) : "v"(lane16), "s"(ub) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=&v"(bf[1]) : "v"(lane16), "s"(ub) : "memory");
#else
    bf[0] = *reinterpret_cast<const f16x8*>(ub + lane16);
    bf[1] = *reinterpret_cast<const f16x8*>(ub + 1024 + lane16);
#endif
  };
  auto wait_b = [&](f16x8 (&bf)[2], int n) __attribute__((always_inline)) {      // n is a constant after unrolling
#if SG_FUSED_ASMLOAD
#define SG_WAITB(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(bf[0]), "+v"(bf[1]) : : "memory")
#else
#define SG_WAITB(N) (void)bf
#endif
    if (n <= 0) SG_WAITB(0);
    else if (n == 2) SG_WAITB(2);
    else if (n == 4) SG_WAITB(4);
    else if (n == 6) SG_WAITB(6);
    else if (n == 8) SG_WAITB(8);
    else if (n == 10) SG_WAITB(10);
    else if (n == 12) SG_WAITB(12);
    else SG_WAITB(14);
#undef SG_WAITB
  };
  auto read_a = [&](f16x8 (&af)[2][2], int buf, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int p = 0; p < 2; ++p)
        af[i][p] = *reinterpret_cast<const f16x8*>(smem + buf * ZBUF + p * ZPLANE + (32 * i + l31) * ZROW + (ks * 2 + kh) * 16);
  };
  const bool has_bias = a.bias && a.rowsum;
  // ---- a finished result: the tile's (accum 'sum': after the last level) or the level's ('stack': after every level) ----
  auto finish = [&](int r, int it, int ti, long long row0) __attribute__((always_inline)) {
#if SG_FUSED_DIRECT
      {   // out of the last level's unit: 2^-e_row (its planes' scale) * 2^-e_B.  (SG_FUSED_DIRECT is written for 'sum' only.)
        const int buf = it & 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const float sb = a.wscale[(a.R - 1) * NJB + wn * NJ + j];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinv + buf * TM + 32 * i + 8 * g4 + 4 * kh);
#pragma unroll
              for (int v = 0; v < 4; ++v) acc[j][i][4 * g4 + v] = (acc[j][i][4 * g4 + v] * s4[v]) * sb;
            }
        }
      }
#endif
      // bias term, activation, store.
      // (the assembly loads above are invisible to the compiler: none may be in flight when it re-uses their registers for the
      //  addresses below.  The last k step's wait already drained them on this path; a build whose compiler peeled the level loop
      //  -- SG_FUSED_DIRECT with `if (r > 0)` around the rescale -- faulted here without this wait.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (has_bias) {
        const float* rs = rs_lds + (ti & 1) * TM * a.R + (4 * kh) * a.R;
        for (int rb = (GEN && a.stack) ? r : 0; rb <= r; ++rb) {        // 'sum': every level's bias rides on its support row sums; 'stack': this level's
          float bv[NJ];
#pragma unroll
          for (int j = 0; j < NJ; ++j) bv[j] = a.bias[rb * ND + (wn * NJ + j) * 32 + l31];
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const float sv = rs[(32 * i + (q & 3) + 8 * (q >> 2)) * a.R + rb];
#pragma unroll
              for (int j = 0; j < NJ; ++j) acc[j][i][q] = __builtin_fmaf(sv, bv[j], acc[j][i][q]);
            }
        }
      }
      const long long coff = (GEN && a.stack) ? static_cast<long long>(r) * a.out_dim : 0;
      auto store_tile = [&](auto actc) __attribute__((always_inline)) {
        constexpr int ACT = decltype(actc)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const long long row = row0 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * kh;
            if (row < a.n_dst) {
#pragma unroll
              for (int j = 0; j < NJ; ++j) {
                const int col = (wn * NJ + j) * 32 + l31;
                if (!GEN || col < a.out_dim) a.out[row * a.ldo + coff + col] = f16x3::act_fn(acc[j][i][q], ACT, a.slope);
              }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[j][i][q] = 0.f;
          }
      };
      switch (a.act) {
        case SG_ACT_LEAKY: store_tile(std::integral_constant<int, SG_ACT_LEAKY>{}); break;
        case SG_ACT_RELU: store_tile(std::integral_constant<int, SG_ACT_RELU>{}); break;
        case SG_ACT_SIGMOID: store_tile(std::integral_constant<int, SG_ACT_SIGMOID>{}); break;
        case SG_ACT_TANH: store_tile(std::integral_constant<int, SG_ACT_TANH>{}); break;
        default: store_tile(std::integral_constant<int, SG_ACT_NONE>{}); break;
      }
  };
  int it = 0;
  for (int ti = 0; ti < n_my; ++ti) {
    const int tile = tile_of(slot_of(ti));
    const long long row0 = static_cast<long long>(tile) * TM;
#if SG_FUSED_SKIPB
    skip_b = (ti & 1) != 0;
#endif
    if (has_bias) {       // this tile's support row sums -> LDS (each M wave a share; the level barriers publish them)
      float* dst = rs_lds + (ti & 1) * TM * a.R;
      const int cnt = TM * a.R;
      for (int e = wn * 64 + lane; e < cnt; e += 64 * MW) {
        const long long row = row0 + e / a.R;
        dst[e] = row < a.n_dst ? a.rowsum[row0 * a.R + e] : 0.f;
      }
    }
    for (int r = 0; r < a.R; ++r, ++it) {
      f16x8 bF[BRING][2];
#pragma unroll
      for (int p = 0; p < BRING - 1; ++p) load_b(bF[p], r, p / KS, p % KS);      // B does not depend on the gather: requested ahead of the barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // item `it` is published
#if SG_FUSED_EVEN == 1
      __builtin_amdgcn_s_barrier();                         // ... after the rows cut between gather waves are put together
#endif
      asm volatile("" ::: "memory");
      const int buf = it & 1;
      if (__builtin_expect(a.ablate & 1, 0)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); continue; }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#if SG_FUSED_DIRECT
        f32x16 (&P)[2] = acc[j];                            // the running result itself, brought to the level's unit
        {     // (also at a tile's first level: the running result is zero there and the factor finite)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const f32x4 f4 = *reinterpret_cast<const f32x4*>(fac + buf * TM + 32 * i + 8 * g4 + 4 * kh);
#pragma unroll
              for (int v = 0; v < 4; ++v) P[i][4 * g4 + v] *= f4[v];
            }
        }
#else
        f32x16 P[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 16; ++q) P[i][q] = 0.f;
#endif
#if SG_FUSED_ADB
        f16x8 aF[2][2][2];
        read_a(aF[0], buf, 0);
#else
        f16x8 aF[1][2][2];         // one set: with two matrix waves per SIMD the other wave covers the LDS latency
#endif
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          // B fragments BRING - 1 k steps ahead (straight into the next column block at the end of this one)
          const int s2 = j * KS + ks + BRING - 1;
          if (s2 < NJ * KS) load_b(bF[s2 % BRING], r, s2 / KS, s2 % KS);
#if SG_FUSED_ADB
          if (ks + 1 < KS) read_a(aF[(ks + 1) & 1], buf, ks + 1);
#else
          read_a(aF[0], buf, ks);
#endif
          asm volatile("" ::: "memory");                      // the requests stay HERE: ahead of their use
          __builtin_amdgcn_sched_barrier(0);
          const f16x8 (&af)[2][2] = aF[SG_FUSED_ADB ? (ks & 1) : 0];
          f16x8 (&bf)[2] = bF[(j * KS + ks) % BRING];
          // fragment sets requested after this one: BRING - 1, fewer at the end of the level
          wait_b(bf, 2 * min(NJ * KS - 1 - (j * KS + ks), BRING - 1));
#pragma unroll
          for (int i = 0; i < 2; ++i) P[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], bf[0], P[i], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i) P[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[1], P[i], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < 2; ++i) P[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[0], P[i], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#if !SG_FUSED_DIRECT
        // fold: acc += P * 2^-e_row * 2^-e_B
        const float sb = a.wscale[r * NJB + wn * NJ + j];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(sinv + buf * TM + 32 * i + 8 * g4 + 4 * kh);
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[j][i][4 * g4 + v] = __builtin_fmaf(P[i][4 * g4 + v] * s4[v], sb, acc[j][i][4 * g4 + v]);
          }
#endif
      }
      if (GEN && (a.stack || r + 1 == a.R)) finish(r, it, ti, row0);
    }
    if (!GEN) finish(a.R - 1, it - 1, ti, row0);
  }
}

// ---- B planes: one workgroup per (level r, 32-column block jb): block maximum -> scale -> fragment-major f16 planes.
// trans = 0: B_r[n][k] = W_r[n * ldw + k] (forward: W_r is (units, in_dim)); trans = 1: B_r[n][k] = W_r[k * ldw + n]
// (data gradient: contraction over the units).  Also packs the level biases into (R, ND).
struct WTable {
  const float* w[SG_MAX_LINKS];
  const float* b[SG_MAX_LINKS];
};
// kdim / ndim: the contraction length and the output width that exist (<= KD / ND); the planes are zero beyond them.
__global__ __launch_bounds__(256) void split_w_kernel(char* __restrict__ planes, float* __restrict__ wscale, int* __restrict__ wexp,
                                                      float* __restrict__ bias_pack, const WTable tab, long long ldw, int trans,
                                                      int kdim, int ndim) {
  __shared__ float tile[32][KD + 1];
  __shared__ float wmax[4];
  const int r = blockIdx.x / NJB, jb = blockIdx.x - r * NJB;
  const int t = threadIdx.x;
  const float* W = tab.w[r];
  float m = 0.f;
  if (!trans) {
    for (int e = t; e < 32 * KD; e += 256) {
      const int n = e / KD, k = e - n * KD;
      const float v = (32 * jb + n < ndim && k < kdim) ? W[static_cast<long long>(32 * jb + n) * ldw + k] : 0.f;
      tile[n][k] = v;
      m = fmaxf(m, fabsf(v) <= 3.402823466e38f ? fabsf(v) : 0.f);
    }
  } else {
    for (int e = t; e < 32 * KD; e += 256) {
      const int k = e / 32, n = e - k * 32;
      const float v = (32 * jb + n < ndim && k < kdim) ? W[static_cast<long long>(k) * ldw + 32 * jb + n] : 0.f;
      tile[n][k] = v;
      m = fmaxf(m, fabsf(v) <= 3.402823466e38f ? fabsf(v) : 0.f);
    }
  }
#if SG_FUSED_DIRECT
  // one exponent per LEVEL (the matrix waves accumulate across the column blocks' products in one unit per row): the maximum of
  // the whole 256 x 256 matrix, recomputed by each of the level's eight workgroups
  for (int e2 = t; e2 < KD * ND; e2 += 256) {      // (SG_FUSED_DIRECT is written for full 256 x 256 levels)
    const float v = fabsf(W[static_cast<long long>(e2 >> 8) * ldw + (e2 & 255)]);
    m = fmaxf(m, v <= 3.402823466e38f ? v : 0.f);
  }
#endif
  m = wave_max_nonneg(m);
  if ((t & 63) == 0) wmax[t >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  int e = 0;
  {
    const unsigned bits = __float_as_uint(m);
    const int ex = static_cast<int>((bits >> 23) & 0xffu);
    if (bits != 0u && ex != 0xff) e = 14 - (max(ex, 1) - 127);
    e = min(max(e, -126), 126);
  }
  const float sc = __uint_as_float(static_cast<unsigned>(127 + e) << 23);
  // unit (ks, plane): lane l -> row n = l & 31, k = 16 ks + 8 (l >> 5) .. + 7; thread handles (ks = t / 64 + 4 pass, lane = t % 64)
  const int lane = t & 63;
  const int n = lane & 31, kg = lane >> 5;
  for (int ks = t >> 6; ks < KS; ks += 4) {
    unsigned h1[4], h2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x0 = tile[n][16 * ks + 8 * kg + 2 * i], x1 = tile[n][16 * ks + 8 * kg + 2 * i + 1];
      const f32x2 v = {x0 * sc, x1 * sc};
      const f16x2 hi = __builtin_convertvector(v, f16x2);
      f32x2 res = {__builtin_fmaf(x0, sc, -static_cast<float>(hi[0])), __builtin_fmaf(x1, sc, -static_cast<float>(hi[1]))};
      if (fabsf(v[0]) == __builtin_inff()) res[0] = 0.f;
      if (fabsf(v[1]) == __builtin_inff()) res[1] = 0.f;
      const f16x2 lo = __builtin_convertvector(res, f16x2);
      h1[i] = __builtin_bit_cast(unsigned, hi);
      h2[i] = __builtin_bit_cast(unsigned, lo);
    }
    char* u = planes + ((((static_cast<long long>(r) * NJB + jb) * KS + ks) * 2) << 10) + lane * 16;
    *reinterpret_cast<uint4*>(u) = make_uint4(h1[0], h1[1], h1[2], h1[3]);
    *reinterpret_cast<uint4*>(u + 1024) = make_uint4(h2[0], h2[1], h2[2], h2[3]);
  }
  if (t == 0) wscale[r * NJB + jb] = __uint_as_float(static_cast<unsigned>(127 - e) << 23);
  if (t == 0 && jb == 0 && wexp) wexp[r] = e;
  if (bias_pack && t < 32) bias_pack[r * ND + 32 * jb + t] = (tab.b[r] && 32 * jb + t < ndim) ? tab.b[r][32 * jb + t] : 0.f;
}

// ---- f-plan: one workgroup per launch slot (tile = tile_order[slot], or the slot itself).  The tile's edges keep their
// range of the (row, level)-major CSR (a tile's rows are contiguous there) and are permuted inside it to level-major:
// segment (r, j) of the tile = source segment (tile * 64 + j) * R + r.  f_ptr: 65 absolute edge offsets per (slot, level).
__global__ __launch_bounds__(256) void plan_kernel(int32_t* __restrict__ f_ptr, int32_t* __restrict__ f_idx,
                                                   float* __restrict__ f_w, int32_t* __restrict__ f_pos,
                                                   const int32_t* __restrict__ tile_order,
                                                   const int32_t* __restrict__ indptr, const int32_t* __restrict__ idx,
                                                   const float* __restrict__ w, int n_dst, int R) {
  __shared__ int s_len[SG_MAX_LINKS * TM + 1];
  __shared__ int s_src[SG_MAX_LINKS * TM];
  __shared__ int s_part[256];
  const int slot = blockIdx.x, t = threadIdx.x;
  const int tile = tile_order ? tile_order[slot] : slot;
  const int nseg = R * TM;
  const int row0 = tile * TM;
  const int base = indptr[static_cast<long long>(row0) * R];
  for (int s = t; s < nseg; s += 256) {
    const int r = s / TM, j = s - r * TM;
    const int row = row0 + j;
    int len = 0, src = 0;
    if (row < n_dst) {
      const long long cs = static_cast<long long>(row) * R + r;
      src = indptr[cs];
      len = indptr[cs + 1] - src;
    }
    s_len[s] = len;
    s_src[s] = src;
  }
  __syncthreads();
  // exclusive scan of s_len (nseg <= 2048): thread t owns segments [t * per, (t + 1) * per)
  const int per = (nseg + 255) / 256;
  int sum = 0;
  for (int s = t * per; s < min(nseg, (t + 1) * per); ++s) sum += s_len[s];
  s_part[t] = sum;
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int i = 0; i < 256; ++i) { const int v = s_part[i]; s_part[i] = run; run += v; }
    s_len[nseg] = run;
  }
  __syncthreads();
  int run = s_part[t];
  for (int s = t * per; s < min(nseg, (t + 1) * per); ++s) { const int v = s_len[s]; s_len[s] = run; run += v; }
  __syncthreads();
  for (int s = t; s < nseg; s += 256) {
    const int r = s / TM, j = s - r * TM;
    int32_t* row_ptr = f_ptr + (static_cast<long long>(slot) * R + r) * (TM + 1);
    row_ptr[j] = base + s_len[s];
    if (j == TM - 1) row_ptr[TM] = base + s_len[s + 1];
  }
  // copy the edges: one wave per segment, round-robin
  const int lane = t & 63, wv = t >> 6;
  for (int s = wv; s < nseg; s += 4) {
    const int dst = base + s_len[s], src = s_src[s];
    const int len = s_len[s + 1] - s_len[s];
    for (int e = lane; e < len; e += 64) {
      f_idx[dst + e] = idx[src + e];
      f_w[dst + e] = w[src + e];
      if (f_pos) f_pos[dst + e] = src + e;
    }
  }
}

__global__ void refresh_kernel(float* __restrict__ f_w, const int32_t* __restrict__ f_pos, const float* __restrict__ w, long long nnz) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < nnz) f_w[i] = w[f_pos[i]];
}

inline size_t al256(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }

}  // namespace fused
}  // namespace sg

using namespace sg;

SG_API int64_t sg_agg_fused_tiles(int64_t n_dst) { return (n_dst + fused::TM - 1) / fused::TM; }

// Widths the kernel handles: gathered rows of 4 .. 256 floats in steps of 4 (16-byte lanes), 1 .. 256 output columns per level.
// It is BUILT for 256 x 256 (KD, ND): narrower rows leave gather lanes idle and narrower outputs multiply zero-padded planes, so
// `auto` routes only the 256-wide 'sum' case here (multilink.hip); the other widths are for callers who ask for the order.
SG_API int sg_agg_fused_supported(int64_t in_dim, int64_t out_dim, int32_t num_links) {
  return in_dim >= 4 && in_dim <= fused::KD && in_dim % 4 == 0 && out_dim >= 1 && out_dim <= fused::ND && num_links >= 1 &&
         num_links <= SG_MAX_LINKS;
}

// f_ptr: tiles * R * 65 entries; f_idx, f_w (, f_pos: position of every edge in the source order, may be null): nnz.
// tile_order (tiles, may be null): the tile of every launch slot -- a permutation of 0 .. tiles - 1, e.g. by descending work.
SG_API int sg_agg_fused_plan_build_hip(int32_t* f_ptr, int32_t* f_idx, float* f_w, int32_t* f_pos, const int32_t* tile_order,
                                       const int32_t* indptr,
                                       const int32_t* indices, const float* weights, int64_t n_dst, int32_t num_links,
                                       int64_t nnz, void* stream) {
  if (num_links < 1 || num_links > SG_MAX_LINKS || n_dst < 0 || nnz < 0) return fail(SG_ERR_INVALID, "bad size");
  if (n_dst == 0) return SG_OK;
  if (!f_ptr || !indptr || (nnz > 0 && (!f_idx || !f_w || !indices || !weights))) return fail(SG_ERR_INVALID, "null pointer argument");
  const int64_t tiles = sg_agg_fused_tiles(n_dst);
  if (tiles * num_links * (fused::TM + 1) >= (1ll << 31) - 1 || nnz >= (1ll << 31) - 64) return fail(SG_ERR_INVALID, "plan too large for int32 indices");
  hipLaunchKernelGGL(fused::plan_kernel, dim3(static_cast<unsigned>(tiles)), dim3(256), 0, static_cast<hipStream_t>(stream), f_ptr,
                     f_idx, f_w, f_pos, tile_order, indptr, indices, weights, static_cast<int>(n_dst), num_links);
  return check_launch("fused::plan_kernel");
}

SG_API int sg_agg_fused_refresh_hip(float* f_w, const int32_t* f_pos, const float* weights, int64_t nnz, void* stream) {
  if (nnz <= 0) return SG_OK;
  if (!f_w || !f_pos || !weights) return fail(SG_ERR_INVALID, "null pointer argument");
  hipLaunchKernelGGL(fused::refresh_kernel, dim3(static_cast<unsigned>((nnz + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), f_w, f_pos, weights, static_cast<long long>(nnz));
  return check_launch("fused::refresh_kernel");
}

SG_API size_t sg_agg_fused_workspace_bytes(int32_t num_links) {
  const size_t R = static_cast<size_t>(num_links);
  return fused::al256(R * fused::NJB * fused::KS * 2 * 1024) + fused::al256(R * fused::NJB * 4) + fused::al256(R * fused::ND * 4) + fused::al256(R * 4) + 256;
}

// out (n_dst, ldo) = act( sum_r (A_r x) B_r + sum_r rowsum[:, r] b_r ),  x (n_src, ldx) of width 256, out width 256.
// weights[r]: trans_w = 0 -> (256 out, 256 in) row-major with leading dimension ldw (B_r = W_r^T); trans_w = 1 -> (256 in, 256 out)
// (B_r = W_r).  biases (host array of R device pointers) and rowsum may be null (no bias term).  zsave (n_dst, ldz) receives the
// fp32 aggregates [r * 256 + k] when not null.  f_*: the level-major plan of sg_agg_fused_plan_build_hip; tile_order may be null.
// n_src: rows of x (every index of the plan is below it), or 0 when unknown.
SG_API int sg_agg_fused_hip(float* out, int64_t ldo, float* zsave, int64_t ldz, const float* x, int64_t ldx,
                            const float* const* weights, int64_t ldw, int trans_w, const float* const* biases,
                            const float* rowsum, const int32_t* f_ptr, const int32_t* f_idx, const float* f_w,
                            const int32_t* tile_order, int64_t n_dst, int64_t n_src, int32_t num_links, int64_t nnz,
                            int64_t in_dim, int64_t out_dim, int act, float slope, int nt_loads, void* workspace,
                            size_t workspace_bytes, void* stream) {
  return sg_agg_fused2_hip(out, ldo, zsave, ldz, x, ldx, 0, weights, ldw, trans_w, in_dim, biases, rowsum, f_ptr, f_idx, f_w, tile_order,
                           n_dst, n_src, num_links, nnz, in_dim, out_dim, SG_ACCUM_SUM, act, slope, nt_loads, workspace, workspace_bytes,
                           stream);
}

// The general form.  accum = SG_ACCUM_STACK: out (n_dst, ldo >= R * out_dim) receives act( (A_r x) B_r + rowsum[:, r] b_r ) in column
// block r.  x_level_stride: level r gathers its rows from x + r * x_level_stride (floats) -- the data gradient of 'stack', where level
// r reads column block r of the output gradient.  in_dim: floats per gathered row in memory (multiple of 4, <= 256); k_valid <= in_dim:
// the contraction length that exists in the weights (a caller that pads its rows to a multiple of 4 passes the true width here; the
// padding columns of x must be finite).  zsave (n_dst, ldz): level r's aggregate at [r * in_dim, (r + 1) * in_dim).
SG_API int sg_agg_fused2_hip(float* out, int64_t ldo, float* zsave, int64_t ldz, const float* x, int64_t ldx, int64_t x_level_stride,
                             const float* const* weights, int64_t ldw, int trans_w, int64_t k_valid, const float* const* biases,
                             const float* rowsum, const int32_t* f_ptr, const int32_t* f_idx, const float* f_w,
                             const int32_t* tile_order, int64_t n_dst, int64_t n_src, int32_t num_links, int64_t nnz,
                             int64_t in_dim, int64_t out_dim, int accum, int act, float slope, int nt_loads, void* workspace,
                             size_t workspace_bytes, void* stream) {
  if (!sg_agg_fused_supported(in_dim, out_dim, num_links))
    return fail(SG_ERR_UNSUPPORTED, "fused aggregation handles rows of 4 .. 256 floats (multiple of 4) and 1 .. 256 output columns per "
                                    "level (got %lld, %lld)", (long long)in_dim, (long long)out_dim);
  if (accum != SG_ACCUM_SUM && accum != SG_ACCUM_STACK) return fail(SG_ERR_INVALID, "accum %d", accum);
  if (k_valid < 1 || k_valid > in_dim || x_level_stride < 0 || (x_level_stride & 3)) return fail(SG_ERR_INVALID, "k_valid / x_level_stride");
#if SG_FUSED_DIRECT
  if (accum != SG_ACCUM_SUM || in_dim != fused::KD || out_dim != fused::ND) return fail(SG_ERR_UNSUPPORTED, "SG_FUSED_DIRECT build: 256 x 256 'sum' only");
#endif
  if (n_dst == 0) return SG_OK;
  if (nnz < 1) return fail(SG_ERR_UNSUPPORTED, "fused aggregation needs at least one edge");
  if (!out || !x || !weights || !f_ptr || !f_idx || !f_w) return fail(SG_ERR_INVALID, "null pointer argument");
  if (act < SG_ACT_NONE || act > SG_ACT_TANH) return fail(SG_ERR_INVALID, "bad activation %d", act);
  if ((ldx & 3) || !aligned(x, 16) || (zsave && ((ldz & 3) || !aligned(zsave, 16))))
    return fail(SG_ERR_INVALID, "rows must be 16-byte aligned");
  if (!workspace || workspace_bytes < sg_agg_fused_workspace_bytes(num_links)) return fail(SG_ERR_WORKSPACE, "fused aggregation workspace too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~static_cast<uintptr_t>(255));
  const size_t R = static_cast<size_t>(num_links);
  char* planes = base;
  float* wscale = reinterpret_cast<float*>(base + fused::al256(R * fused::NJB * fused::KS * 2 * 1024));
  float* bias_pack = reinterpret_cast<float*>(reinterpret_cast<char*>(wscale) + fused::al256(R * fused::NJB * 4));
  int* wexp = reinterpret_cast<int*>(reinterpret_cast<char*>(bias_pack) + fused::al256(R * fused::ND * 4));
  fused::WTable tab{};
  for (int r = 0; r < num_links; ++r) {
    if (!weights[r]) return fail(SG_ERR_INVALID, "weights[%d] is null", r);
    tab.w[r] = weights[r];
    tab.b[r] = biases ? biases[r] : nullptr;
  }
  const bool has_bias = biases && rowsum;
  hipLaunchKernelGGL(fused::split_w_kernel, dim3(static_cast<unsigned>(R * fused::NJB)), dim3(256), 0, st, planes, wscale, wexp,
                     has_bias ? bias_pack : static_cast<float*>(nullptr), tab, static_cast<long long>(ldw), trans_w,
                     static_cast<int>(k_valid), static_cast<int>(out_dim));
  if (check_launch("fused::split_w_kernel") != SG_OK) return SG_ERR_HIP;

  fused::Args a{};
  a.f_ptr = f_ptr; a.f_idx = f_idx; a.f_w = f_w; a.tile_order = tile_order;
  a.x = x; a.ldx = ldx;
  a.wplanes = planes; a.wscale = wscale; a.wexp = wexp;
  a.bias = has_bias ? bias_pack : nullptr; a.rowsum = has_bias ? rowsum : nullptr;
  a.out = out; a.ldo = ldo; a.zsave = zsave; a.ldz = ldz;
  a.n_dst = static_cast<int>(n_dst); a.n_tiles = static_cast<int>(sg_agg_fused_tiles(n_dst)); a.R = num_links;
  a.in_dim = static_cast<int>(in_dim); a.out_dim = static_cast<int>(out_dim);
  a.stack = accum == SG_ACCUM_STACK ? 1 : 0;
  a.x_lstride = x_level_stride;
  a.act = act; a.slope = slope;
  static const int ablate = [] { const char* e = getenv("SG_FUSED_ABLATE"); return e ? atoi(e) : 0; }();
  a.ablate = ablate;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
    return fail(SG_ERR_HIP, "device query");
  const unsigned grid = static_cast<unsigned>(a.n_tiles < cus ? a.n_tiles : cus);
  auto launch = [&](auto kern) {
    static_cast<void>(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, fused::SMEM));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (fused::GW + fused::MW)), fused::SMEM, st, a);
  };
  const long rec = fused::prof_begin(st, nnz, zsave ? 1 : 0);
  // rows of x through 32-bit buffer offsets when every row ends below 4 GB (n_src = 0: extent unknown, 64-bit addresses)
  const bool gen_nt = nt_loads && (in_dim != fused::KD || out_dim != fused::ND || accum == SG_ACCUM_STACK || x_level_stride != 0);
  const bool buf = (!nt_loads || gen_nt) && n_src > 0 && ((n_src - 1) * ldx + (num_links - 1) * x_level_stride + in_dim) * 4 <= 0xffffffffll;
  const bool gen = in_dim != fused::KD || out_dim != fused::ND || a.stack || x_level_stride != 0;
  if (gen) {          // (nt_loads is a tuning aid of the 256-wide instantiation only)
    if (zsave) { if (buf) launch(fused::agg_contract_kernel<true, false, true, true>); else launch(fused::agg_contract_kernel<true, false, false, true>); }
    else { if (buf) launch(fused::agg_contract_kernel<false, false, true, true>); else launch(fused::agg_contract_kernel<false, false, false, true>); }
  } else if (zsave) {
    if (nt_loads) launch(fused::agg_contract_kernel<true, true, false, false>);
    else if (buf) launch(fused::agg_contract_kernel<true, false, true, false>);
    else launch(fused::agg_contract_kernel<true, false, false, false>);
  } else {
    if (nt_loads) launch(fused::agg_contract_kernel<false, true, false, false>);
    else if (buf) launch(fused::agg_contract_kernel<false, false, true, false>);
    else launch(fused::agg_contract_kernel<false, false, false, false>);
  }
  fused::prof_end(rec, st);
  return check_launch("fused::agg_contract_kernel");
}

// measurement aid: enable(1) clears old records and returns the previous state; read() synchronises the recorded events and
// returns up to `capacity` (elapsed ms, edges, 1 if the launch also wrote the aggregates) triples in launch order
SG_API int sg_agg_fused_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(fused::g_mu);
  const int was = fused::g_on ? 1 : 0;
  if (on && !was) {
    for (auto& r : fused::g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    fused::g_recs.clear();
  }
  fused::g_on = on != 0;
  return was;
}
SG_API int64_t sg_agg_fused_profile_read(float* ms, int64_t* nnz, int32_t* zsave, int64_t capacity) {
  std::lock_guard<std::mutex> lk(fused::g_mu);
  int64_t n = 0;
  for (auto& r : fused::g_recs) {
    if (n < capacity) {
      float t = 0.f;
      (void)hipEventSynchronize(r.b);
      if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) t = -1.f;
      if (ms) ms[n] = t;
      if (nnz) nnz[n] = r.nnz;
      if (zsave) zsave[n] = r.zsave;
      ++n;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  fused::g_recs.clear();
  return n;
}
