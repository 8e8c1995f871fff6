// prof_harness.cpp -- torch-free driver of libstargcn_hip.so for rocprofv3 (kernel trace / PMC passes).
// Builds an ML-10M-shaped random multi-link plan on the host (native helpers of the library), uploads it, and launches
// the two forward aggregation gathers + the two contractions of one layer `reps` times.
//
//   hipcc --offload-arch=gfx950 -O2 tools/prof_harness.cpp -Iinclude -Lstar-gcn_amd/csrc -lstargcn_hip \
//         -Wl,-rpath,$PWD/star-gcn_amd/csrc -o tools/prof_harness
//   rocprofv3 --kernel-trace --stats -d out -- tools/prof_harness [n_dst n_src nnz R D reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "stargcn.h"

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } \
  } while (0)
#define SG(x)                                                                     \
  do {                                                                            \
    int rc_ = (x);                                                                \
    if (rc_ != 0) { printf("sg error %d: %s at line %d\n", rc_, sg_last_error(), __LINE__); return 1; } \
  } while (0)

template <typename T> static T* upload(const std::vector<T>& v) {
  T* d = nullptr;
  if (hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
  hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  return d;
}

int main(int argc, char** argv) {
  int64_t n_dst = argc > 1 ? atoll(argv[1]) : 69878, n_src = argc > 2 ? atoll(argv[2]) : 10677;
  int64_t nnz = argc > 3 ? atoll(argv[3]) : 10000000;
  int R = argc > 4 ? atoi(argv[4]) : 10, D = argc > 5 ? atoi(argv[5]) : 256, reps = argc > 6 ? atoi(argv[6]) : 5;
  std::mt19937_64 rng(1234);
  // log-normal-ish row lengths
  std::lognormal_distribution<double> ln(0.0, 1.0);
  std::vector<double> wrow(n_dst);
  double tot = 0;
  for (auto& w : wrow) { w = ln(rng); tot += w; }
  std::vector<int32_t> indptr(n_dst + 1, 0);
  for (int64_t i = 0; i < n_dst; ++i) indptr[i + 1] = indptr[i] + (int32_t)std::max<double>(1.0, wrow[i] / tot * nnz);
  nnz = indptr[n_dst];
  std::vector<int32_t> ep(nnz);
  std::vector<float> vals(nnz), sup(nnz);
  std::uniform_int_distribution<int32_t> ui(0, (int32_t)n_src - 1), ul(0, R - 1);
  std::uniform_real_distribution<float> uf(0.01f, 0.1f);
  for (int64_t j = 0; j < nnz; ++j) { ep[j] = ui(rng); vals[j] = (float)(ul(rng) + 1); sup[j] = uf(rng); }
  std::vector<float> levels(R);
  for (int r = 0; r < R; ++r) levels[r] = (float)(r + 1);
  // per-level split + fuse (native host helpers)
  std::vector<int32_t> pos(nnz), lip((size_t)R * (n_dst + 1));
  std::vector<int64_t> off(R + 1);
  SG(sg_multi_link_split_cpu(pos.data(), lip.data(), off.data(), vals.data(), indptr.data(), levels.data(), n_dst, R));
  std::vector<std::vector<int32_t>> epl(R);
  std::vector<std::vector<float>> spl(R);
  std::vector<const int32_t*> epp(R), ipp(R);
  std::vector<const float*> spp(R);
  for (int r = 0; r < R; ++r) {
    for (int64_t p = off[r]; p < off[r + 1]; ++p) { epl[r].push_back(ep[pos[p]]); spl[r].push_back(sup[pos[p]]); }
    if (epl[r].empty()) { epl[r].push_back(0); spl[r].push_back(0.f); }
    epp[r] = epl[r].data(); spp[r] = spl[r].data(); ipp[r] = lip.data() + (size_t)r * (n_dst + 1);
  }
  std::vector<int32_t> c_indptr(n_dst * R + 1), t_indptr(n_src * R + 1), c_idx(nnz), c_q(nnz), t_idx(nnz), t_q(nnz);
  std::vector<float> c_w(nnz), t_w(nnz);
  SG(sg_multilink_fuse_cpu(c_indptr.data(), c_idx.data(), c_q.data(), c_w.data(), t_indptr.data(), t_idx.data(),
                           t_q.data(), t_w.data(), epp.data(), ipp.data(), spp.data(), R, n_dst, n_src));
  std::vector<int32_t> d_indptr_h(n_dst + 1);
  for (int64_t i = 0; i <= n_dst; ++i) d_indptr_h[i] = c_indptr[i * R];

  int32_t *d_cip = upload(c_indptr), *d_dip = upload(d_indptr_h), *d_idx = upload(c_idx), *d_q = upload(c_q);
  float* d_w = upload(c_w);
  const int64_t ld = (int64_t)R * D + 16;
  std::vector<float> hx((size_t)n_src * D), hh((size_t)n_src * R * D), hw((size_t)D * ld);
  std::uniform_real_distribution<float> un(-1.f, 1.f);
  for (auto& v : hx) v = un(rng);
  for (auto& v : hh) v = un(rng);
  for (auto& v : hw) v = un(rng) * 0.05f;
  float *d_x = upload(hx), *d_h = upload(hh), *d_wext = upload(hw), *d_out, *d_zext, *d_pre;
  CK(hipMalloc(&d_out, (size_t)n_dst * D * 4));
  CK(hipMalloc(&d_zext, (size_t)n_dst * ld * 4));
  CK(hipMalloc(&d_pre, (size_t)n_dst * D * 4));
  CK(hipMemset(d_zext, 0, (size_t)n_dst * ld * 4));
  size_t wsb = sg_seg_weighted_pool_workspace_bytes(1, n_dst * R, nnz, D);
  void* ws;
  CK(hipMalloc(&ws, wsb + 16));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms_tf = 0, ms_af = 0, ms_g1 = 0, ms_g2 = 0, ms;
  for (int it = 0; it < reps + 1; ++it) {
    CK(hipEventRecord(e0, 0));   // transform-first forward gather: grouped source rows, un-split destination CSR
    SG(sg_seg_gather_sum_hip(d_out, 1, D, d_h, R, (int64_t)R * D, d_w, d_q, d_dip, n_dst, nnz, D, SG_REQ_WRITE,
                             SG_ACT_LEAKY, 0.1f, ws, wsb + 16, nullptr));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (it) ms_tf += ms;
    CK(hipEventRecord(e0, 0));   // aggregate-first forward gather: grouped destination rows
    SG(sg_seg_gather_sum_hip(d_zext, R, ld, d_x, 1, D, d_w, d_idx, d_cip, n_dst * R, nnz, D, SG_REQ_WRITE, SG_ACT_NONE,
                             0.f, ws, wsb + 16, nullptr));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (it) ms_af += ms;
    CK(hipEventRecord(e0, 0));   // contraction after aggregation: pre = Zext * Wext^T
    SG(sg_gemm_f32_hip(d_pre, D, d_zext, ld, 0, d_wext, ld, 1, n_dst, D, ld, nullptr, SG_ACT_LEAKY, 0.1f, 0, nullptr, 0,
                       nullptr));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (it) ms_g1 += ms;
    CK(hipEventRecord(e0, 0));   // transform before aggregation: H = X * Wcat^T  (n_src x R*D)
    SG(sg_gemm_f32_hip(d_h, (int64_t)R * D, d_x, D, 0, d_wext, D, 1, n_src, (int64_t)R * D, D, nullptr, SG_ACT_NONE, 0.f,
                       0, nullptr, 0, nullptr));
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (it) ms_g2 += ms;
  }
  // ---- overlap experiment: the fabric-bound gather and the MFMA-bound contraction on ONE stream vs TWO streams ----
  {
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa));
    CK(hipStreamCreate(&sb));
    void* ws2;
    CK(hipMalloc(&ws2, wsb + 16));
    float* d_out2;
    CK(hipMalloc(&d_out2, (size_t)n_dst * D * 4));
    hipEvent_t f0, f1, g1;
    CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1)); CK(hipEventCreate(&g1));
    float serial = 0, par = 0;
    for (int it = 0; it < reps + 1; ++it) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(f0, sa));
      SG(sg_seg_gather_sum_hip(d_out, 1, D, d_h, R, (int64_t)R * D, d_w, d_q, d_dip, n_dst, nnz, D, SG_REQ_WRITE,
                               SG_ACT_LEAKY, 0.1f, ws, wsb + 16, sa));
      SG(sg_gemm_f32_hip(d_pre, D, d_zext, ld, 0, d_wext, ld, 1, n_dst, D, ld, nullptr, SG_ACT_LEAKY, 0.1f, 0, nullptr, 0, sa));
      CK(hipEventRecord(f1, sa)); CK(hipEventSynchronize(f1)); CK(hipEventElapsedTime(&ms, f0, f1)); if (it) serial += ms;
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(f0, sa));
      CK(hipStreamWaitEvent(sb, f0, 0));
      SG(sg_seg_gather_sum_hip(d_out2, 1, D, d_h, R, (int64_t)R * D, d_w, d_q, d_dip, n_dst, nnz, D, SG_REQ_WRITE,
                               SG_ACT_LEAKY, 0.1f, ws2, wsb + 16, sa));
      SG(sg_gemm_f32_hip(d_pre, D, d_zext, ld, 0, d_wext, ld, 1, n_dst, D, ld, nullptr, SG_ACT_LEAKY, 0.1f, 0, nullptr, 0, sb));
      CK(hipEventRecord(g1, sb));
      CK(hipStreamWaitEvent(sa, g1, 0));
      CK(hipEventRecord(f1, sa)); CK(hipEventSynchronize(f1)); CK(hipEventElapsedTime(&ms, f0, f1)); if (it) par += ms;
    }
    printf("overlap: gather + gemm serial %.3f ms, on two streams %.3f ms\n", serial / reps, par / reps);
  }
  // ---- same pair on CU-MASKED streams: the gather gets `gcu` CUs (spread evenly over the 8 XCDs), the GEMM the rest ----
  for (int gemm_per_xcd : {4, 6, 8, 12, 16}) {
    // physical layout assumption: CU index c -> XCD c % 8 (round-robin across XCDs in the mask bit order)
    const int total_cu = 256;
    std::vector<uint32_t> mg(total_cu / 32, 0u), mm(total_cu / 32, 0u);
    for (int c = 0; c < total_cu; ++c) {
      const bool to_gemm = (c / 8) < gemm_per_xcd;        // the first gemm_per_xcd CUs of every XCD
      (to_gemm ? mm : mg)[c / 32] |= 1u << (c % 32);
    }
    hipStream_t sa, sb;
    if (hipExtStreamCreateWithCUMask(&sa, mg.size(), mg.data()) != hipSuccess ||
        hipExtStreamCreateWithCUMask(&sb, mm.size(), mm.data()) != hipSuccess) {
      printf("CU-masked streams unavailable\n");
      break;
    }
    void* ws2;
    CK(hipMalloc(&ws2, wsb + 16));
    hipEvent_t f0, f1, g1;
    CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1)); CK(hipEventCreate(&g1));
    float par = 0, galone = 0, malone = 0;
    for (int it = 0; it < reps + 1; ++it) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(f0, sa));
      SG(sg_seg_gather_sum_hip(d_out, 1, D, d_h, R, (int64_t)R * D, d_w, d_q, d_dip, n_dst, nnz, D, SG_REQ_WRITE,
                               SG_ACT_LEAKY, 0.1f, ws2, wsb + 16, sa));
      CK(hipEventRecord(f1, sa)); CK(hipEventSynchronize(f1)); CK(hipEventElapsedTime(&ms, f0, f1)); if (it) galone += ms;
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(f0, sb));
      SG(sg_gemm_f32_hip(d_pre, D, d_zext, ld, 0, d_wext, ld, 1, n_dst, D, ld, nullptr, SG_ACT_LEAKY, 0.1f, 0, nullptr, 0, sb));
      CK(hipEventRecord(f1, sb)); CK(hipEventSynchronize(f1)); CK(hipEventElapsedTime(&ms, f0, f1)); if (it) malone += ms;
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(f0, sa));
      CK(hipStreamWaitEvent(sb, f0, 0));
      SG(sg_seg_gather_sum_hip(d_out, 1, D, d_h, R, (int64_t)R * D, d_w, d_q, d_dip, n_dst, nnz, D, SG_REQ_WRITE,
                               SG_ACT_LEAKY, 0.1f, ws2, wsb + 16, sa));
      SG(sg_gemm_f32_hip(d_pre, D, d_zext, ld, 0, d_wext, ld, 1, n_dst, D, ld, nullptr, SG_ACT_LEAKY, 0.1f, 0, nullptr, 0, sb));
      CK(hipEventRecord(g1, sb));
      CK(hipStreamWaitEvent(sa, g1, 0));
      CK(hipEventRecord(f1, sa)); CK(hipEventSynchronize(f1)); CK(hipEventElapsedTime(&ms, f0, f1)); if (it) par += ms;
    }
    printf("cu-mask: gemm %3d CUs / gather %3d CUs: gather alone %.3f ms, gemm alone %.3f ms, concurrent %.3f ms\n",
           gemm_per_xcd * 8, total_cu - gemm_per_xcd * 8, galone / reps, malone / reps, par / reps);
    CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sb));
  }
  const double bytes = (8.0 + 4.0 * D) * nnz;
  printf("n_dst %lld n_src %lld nnz %lld R %d D %d reps %d\n", (long long)n_dst, (long long)n_src, (long long)nnz, R, D, reps);
  printf("gather transform-first : %.3f ms  %.1f GB/s algorithmic\n", ms_tf / reps, bytes / (ms_tf / reps) / 1e6);
  printf("gather aggregate-first : %.3f ms  %.1f GB/s algorithmic\n", ms_af / reps, bytes / (ms_af / reps) / 1e6);
  printf("gemm Zext*Wext^T (%lldx%dx%lld): %.3f ms  %.1f TF/s\n", (long long)n_dst, D, (long long)ld, ms_g1 / reps,
         2.0 * n_dst * D * ld / (ms_g1 / reps) / 1e9);
  printf("gemm X*Wcat^T   (%lldx%lldx%d): %.3f ms  %.1f TF/s\n", (long long)n_src, (long long)R * D, D, ms_g2 / reps,
         2.0 * n_src * R * D * D / (ms_g2 / reps) / 1e9);
  return 0;
}
