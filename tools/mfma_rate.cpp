// mfma_rate.cpp -- sustained fp32 MFMA issue rate on gfx950 for the two fp32 shapes (32x32x2 and 16x16x4), 1..4 waves per
// SIMD, independent accumulator chains.  hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.cpp -o tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int NACC>
__global__ void k32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static void run(const char* name, F launch, double flops_per_launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch(); hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int r = 0; r < 5; ++r) launch();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms / 5, flops_per_launch / (ms / 5) / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 256 * 16 * 256 * 4);
  const int iters = 2000;
  for (int wg_per_cu : {1, 2, 4}) {     // 256-thread workgroups = 1 wave per SIMD each
    const int grid = 256 * wg_per_cu;
    char nm[96];
    snprintf(nm, 96, "32x32x2  4 acc chains, %d wave(s)/SIMD", wg_per_cu);
    run(nm, [&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, out, iters); }, 4096.0 * 8 * 4 * iters * grid * 4);
    snprintf(nm, 96, "32x32x2  2 acc chains, %d wave(s)/SIMD", wg_per_cu);
    run(nm, [&] { hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(256), 0, 0, out, iters); }, 4096.0 * 8 * 2 * iters * grid * 4);
    snprintf(nm, 96, "16x16x4  8 acc chains, %d wave(s)/SIMD", wg_per_cu);
    run(nm, [&] { hipLaunchKernelGGL(k16<8>, dim3(grid), dim3(256), 0, 0, out, iters); }, 2048.0 * 8 * 8 * iters * grid * 4);
    snprintf(nm, 96, "16x16x4  4 acc chains, %d wave(s)/SIMD", wg_per_cu);
    run(nm, [&] { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(256), 0, 0, out, iters); }, 2048.0 * 8 * 4 * iters * grid * 4);
  }
  return 0;
}
