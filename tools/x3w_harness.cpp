// x3w_harness.cpp -- development harness for the f16x3 GEMM variants (no torch: starts in a second on the GPU box).
//   x3w_harness check            correctness of the listed variants on small / ragged shapes vs an fp64 host reference
//   x3w_harness bench [iters]    timing of the step's big shapes, variants side by side, sampled fp64 check of each
// Operands are a hash of the element index (host and device agree without a copy): uniform in [-1, 1) times a per-row scale.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/x3w_harness tools/x3w_harness.cpp -Lstar-gcn_amd/csrc -lstargcn_hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/stargcn.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

extern "C" int sg_gemm_x3_variant(int);

__device__ int d_pos = 0;      // 1: operands are |value| -- every product positive, the running sums grow linearly
static int h_pos = 0;
__host__ __device__ inline float elem(uint64_t seed, uint64_t r, uint64_t c, uint64_t rows);
// pos >= 2 ("drop" mode, A = seed 11, stored (M x K)): columns k >= pos of A are 2^-70 of the rest and rows m % 32 == 5 are zero
// before that -- the second half's scale blocks lie 2^70 below the running scale of their row block, and rows 5, 37, .. see
// only them: the direct-accumulation kernel must raise its flag and the fallback must redo the product
__host__ __device__ inline float elem_p(uint64_t seed, uint64_t r, uint64_t c, uint64_t rows, int pos) {
  const float v = elem(seed, r, c, rows);
  if (pos >= 2) {
    if (seed != 11) return v;
    if (c >= static_cast<uint64_t>(pos)) return v * 8.470329472543003e-22f;      // 2^-70
    return (r % 32 == 5) ? 0.f : v;
  }
  return pos ? fabsf(v) : v;
}
__host__ __device__ inline float elem(uint64_t seed, uint64_t r, uint64_t c, uint64_t rows) {
  uint64_t x = seed * 0x9E3779B97F4A7C15ull + r * 0xD1B54A32D192ED03ull + c * 0x8CB92BA72F3D8DD7ull;
  x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32; x *= 0xD6E8FEB86659FD93ull; x ^= x >> 32;
  const float u = static_cast<float>(static_cast<int32_t>(x & 0xffffffu) - 0x800000) * (1.0f / 8388608.0f);
  // rows span three decades (the per-block scales differ from block to block)
  const float sc = (seed & 1) ? 1.0f : exp2f(static_cast<float>(static_cast<int>((r * 2654435761ull) % 21) - 10) * 0.5f);
  (void)rows;
  return u * sc;
}

// stored matrix of `rows` x `cols` (leading dimension ld); logical element (r, c)
__global__ void fill_kernel(float* p, long long rows, long long cols, long long ld, uint64_t seed) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= rows * ld) return;
  const long long r = i / ld, c = i - r * ld;
  p[i] = c < cols ? elem_p(seed, r, c, rows, d_pos) : 0.f;
}

struct Case { long long M, N, K; int ta, tb; const char* name; };

static double ref_entry(const Case& c, long long i, long long j) {
  // op(A)(i, k): A stored (M x K) or (K x M) when ta; op(B)(k, j): B stored (K x N) or (N x K) when tb
  double s = 0.0;
  for (long long k = 0; k < c.K; ++k) {
    const float a = c.ta ? elem_p(11, k, i, c.K, h_pos) : elem_p(11, i, k, c.M, h_pos);
    const float b = c.tb ? elem_p(12, j, k, c.N, h_pos) : elem_p(12, k, j, c.K, h_pos);
    s += static_cast<double>(a) * static_cast<double>(b);
  }
  return s;
}
static double mag_entry(const Case& c, long long i, long long j) {
  double s = 0.0;
  for (long long k = 0; k < c.K; ++k) {
    const float a = c.ta ? elem_p(11, k, i, c.K, h_pos) : elem_p(11, i, k, c.M, h_pos);
    const float b = c.tb ? elem_p(12, j, k, c.N, h_pos) : elem_p(12, k, j, c.K, h_pos);
    s += std::fabs(static_cast<double>(a) * static_cast<double>(b));
  }
  return s;
}

struct Run { double ms; double max_rel; };

static Run run_case(const Case& c, int variant, int iters, int samples, bool full_check) {
  const long long ar = c.ta ? c.K : c.M, ac = c.ta ? c.M : c.K, br = c.tb ? c.N : c.K, bc = c.tb ? c.K : c.N;
  float *A, *B, *C;
  CK(hipMalloc(&A, ar * ac * 4)); CK(hipMalloc(&B, br * bc * 4)); CK(hipMalloc(&C, c.M * c.N * 4));
  fill_kernel<<<static_cast<unsigned>((ar * ac + 255) / 256), 256>>>(A, ar, ac, ac, 11);
  fill_kernel<<<static_cast<unsigned>((br * bc + 255) / 256), 256>>>(B, br, bc, bc, 12);
  CK(hipMemset(C, 0xff, c.M * c.N * 4));
  sg_gemm_backend(variant == 100 ? 0 : 3);      // variant 100: the exact-fp32 MFMA kernel
  sg_gemm_x3_variant(variant == 100 ? 0 : variant);
  const size_t wsb = sg_gemm_f32_workspace_bytes(c.M, c.N, c.K, c.ta);
  void* ws = nullptr;
  if (wsb) CK(hipMalloc(&ws, wsb));
  auto call = [&]() {
    const int rc = sg_gemm_f32_hip(C, c.N, A, ac, c.ta, B, bc, c.tb, c.M, c.N, c.K, nullptr, SG_ACT_NONE, 0.f, 0, ws, wsb, nullptr);
    if (rc != SG_OK) { fprintf(stderr, "sg_gemm_f32_hip failed: %s\n", sg_last_error()); exit(3); }
  };
  call();
  CK(hipDeviceSynchronize());
  Run r{0.0, 0.0};
  if (iters > 0) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int it = 0; it < iters; ++it) {
      CK(hipEventRecord(e0, nullptr));
      call();
      CK(hipEventRecord(e1, nullptr));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    r.ms = ts[ts.size() / 2];
  }
  // check
  if (full_check) {
    std::vector<float> h(c.M * c.N);
    CK(hipMemcpy(h.data(), C, h.size() * 4, hipMemcpyDeviceToHost));
    for (long long i = 0; i < c.M; ++i)
      for (long long j = 0; j < c.N; ++j) {
        const double ref = ref_entry(c, i, j), mag = mag_entry(c, i, j);
        const double e = std::fabs(static_cast<double>(h[i * c.N + j]) - ref) / (mag + 1e-30);
        if (!(e <= r.max_rel)) r.max_rel = std::isnan(e) ? 1e30 : e;
      }
  } else {
    uint64_t s = 12345;
    for (int q = 0; q < samples; ++q) {
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      long long i = static_cast<long long>((s >> 20) % static_cast<uint64_t>(c.M));
      s = s * 6364136223846793005ull + 1442695040888963407ull;
      long long j = static_cast<long long>((s >> 20) % static_cast<uint64_t>(c.N));
      if (q < 8) { i = (q & 1) ? c.M - 1 - (q >> 1) : (q >> 1); j = (q & 2) ? c.N - 1 : 0; }      // corners
      else if (h_pos >= 2 && (q & 1)) i = std::min(c.M - 1, i / 32 * 32 + 5);
      float got;
      CK(hipMemcpy(&got, C + i * c.N + j, 4, hipMemcpyDeviceToHost));
      const double ref = ref_entry(c, i, j), mag = mag_entry(c, i, j);
      const double e = std::fabs(static_cast<double>(got) - ref) / (mag + 1e-30);
      if (!(e <= r.max_rel)) r.max_rel = std::isnan(e) ? 1e30 : e;
    }
  }
  if (ws) CK(hipFree(ws));
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
  sg_gemm_x3_variant(-1);
  sg_gemm_backend(-1);
  return r;
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "check";
  std::vector<int> variants;
  if (const char* v = getenv("X3W_VARIANTS")) {
    for (const char* p = v; *p;) { variants.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
  } else {
    variants = {0, 8};
  }
  int fails = 0;
  if (getenv("X3W_POS")) { h_pos = 1; CK(hipMemcpyToSymbol(HIP_SYMBOL(d_pos), &h_pos, sizeof(int))); }
  if (mode == "acc") {        // accuracy at long K: max and rms of err / mag over sampled entries, variants side by side
    const Case cases[] = {{512, 512, 4096, 0, 1, "K 4096"}, {512, 512, 65536, 0, 1, "K 65536"}, {256, 512, 1000000, 1, 0, "K 1M TN"}};
    for (const Case& c : cases) {
      printf("acc %-10s", c.name);
      for (int v : variants) { const Run r = run_case(c, v, 0, 400, false); printf("  v%-3d max %.2e", v, r.max_rel); fflush(stdout); }
      printf("\n");
    }
    return 0;
  }
  if (mode == "drop") {       // a scale drop of 2^70 inside the K range: exact only through the flag + fallback
    int bad = 0;
    for (const Case& c : {Case{3000, 256, 4096, 0, 1, "drop NT"}, Case{70000, 200, 1024, 0, 0, "drop NN, many items"}}) {
      h_pos = static_cast<int>(c.K / 2);
      CK(hipMemcpyToSymbol(HIP_SYMBOL(d_pos), &h_pos, sizeof(int)));
      for (int v : variants) {
        const Run r = run_case(c, v, 0, 3000, false);
        const double bound = 4e-7 * std::sqrt(static_cast<double>(c.K));
        printf("drop  %-24s variant %2d  max err / mag %.3e (bound %.1e) %s\n", c.name, v, r.max_rel, bound, r.max_rel <= bound ? "ok" : "FAIL");
        bad += !(r.max_rel <= bound);
      }
    }
    return bad ? 1 : 0;
  }
  if (mode == "check") {
    const Case cases[] = {
        {256, 256, 64, 0, 1, "one tile, one block"}, {256, 256, 128, 0, 1, "one tile"}, {300, 260, 200, 0, 1, "ragged"}, {300, 260, 200, 1, 0, "ragged TN"}, {513, 257, 96, 0, 1, "ragged 3x2"},
        {256, 512, 160, 0, 0, "odd 32-k tile count"},
        {130, 250, 96, 0, 0, "NN ragged"}, {257, 64, 2570, 1, 0, "TN long K"}, {700, 515, 1027, 1, 1, "TT"},
        {512, 768, 8256, 0, 1, "129 blocks (two exponent chunks)"}, {1000, 76, 252, 0, 1, "narrow"}, {5, 300, 1028, 0, 1, "short M"},
        {200, 130, 40000, 1, 0, "split-K"}, {2304, 2560, 256, 0, 1, "many tiles"},
        {3000, 256, 128, 0, 1, "hybrid NT"}, {3000, 200, 2052, 0, 0, "hybrid NN"}, {1028, 256, 3000, 1, 0, "hybrid TN (row-contiguous A)"},
        {96, 2000, 40004, 1, 0, "swapped hybrid"}, {256, 4160, 20000, 1, 0, "swapped hybrid, dWext-like"}, {700, 130, 100, 0, 1, "hybrid, odd tile count"},
        {153600, 256, 256, 0, 1, "600 items"}, {70000, 300, 160, 0, 0, "548 items, 5 tiles each"}, {76800, 256, 8260, 0, 1, "300 items, two chunks"},
        {2052, 70000, 96, 1, 0, "2200 items, TN"}, {256, 4160, 100000, 1, 0, "swapped, split-K"}};
    for (const Case& c : cases)
      for (int v : variants) {
        const bool full = static_cast<double>(c.M) * c.N * c.K < 2e8;
        const Run r = run_case(c, v, 0, 3000, full);
        const double bound = 4e-7 * std::max(1.0, std::sqrt(static_cast<double>(c.K)));
        const bool ok = r.max_rel <= bound;
        fails += !ok;
        printf("check %-34s M=%5lld N=%5lld K=%6lld %c%c variant %2d  max err / mag %.3e (bound %.1e) %s\n", c.name, c.M, c.N, c.K,
               c.ta ? 'T' : 'N', c.tb ? 'T' : 'N', v, r.max_rel, bound, ok ? "ok" : "FAIL");
        fflush(stdout);
      }
  } else {
    const int iters = argc > 2 ? atoi(argv[2]) : 7;
    const long long nu = 1000000, RD = 4096, D = 256, ld = 4160;
    std::vector<Case> cases = {
        {4096, 4096, 4096, 0, 1, "square 4096"},
        {nu, RD, D, 0, 1, "c5 TF fwd 1Mx4096x256 NT"},
        {nu, ld, D, 0, 0, "c5 dZ 1Mx4160x256 NN"},
        {nu, D, ld, 0, 1, "c5 AF fwd 1Mx256x4160 NT"},
        {nu, D, RD, 0, 0, "c5 dX 1Mx256x4096 NN"},
        {RD, D, nu, 1, 0, "c5 dW 4096x256x1M TN"},
        {D, ld, nu, 1, 0, "c5 dWext 256x4160x1M TN"},
        {1250000, D, D, 0, 1, "c5 out_fc 1.25Mx256x256 NT"},
        {10677, 2560, 256, 0, 1, "ml10m TF fwd"},
        {69878, 256, 256, 0, 1, "ml10m out_fc"},
        {10677, 256, 2624, 0, 1, "ml10m AF fwd"},
        {10677, 256, 2560, 0, 0, "ml10m dX"},
        {2560, 256, 10677, 1, 0, "ml10m dW"},
        {256, 2624, 10677, 1, 0, "ml10m dWext"},
        {69878, 256, 256, 0, 0, "ml10m out_fc dX"},
        {256, 256, 69878, 1, 0, "ml10m out_fc dW"}};
    if (const char* only = getenv("X3W_ONLY")) {
      std::vector<Case> sel;
      for (const Case& c : cases) if (strstr(c.name, only)) sel.push_back(c);
      cases = sel;
    }
    for (const Case& c : cases) {
      printf("bench %-28s", c.name);
      for (int v : variants) {
        const Run r = run_case(c, v, iters, 24, false);
        const double tf = 2.0 * c.M * c.N * c.K / (r.ms * 1e-3) / 1e12;
        const double bound = 4e-7 * std::max(1.0, std::sqrt(static_cast<double>(c.K)));
        fails += !(r.max_rel <= bound);
        printf("  v%-2d %8.3f ms %6.1f TF err %.1e%s", v, r.ms, tf, r.max_rel, r.max_rel <= bound ? "" : " FAIL");
        fflush(stdout);
      }
      printf("\n");
    }
  }
  return fails ? 1 : 0;
}
