// Raw issue rate of v_mfma_f32_32x32x16_bf16 on gfx950: W waves per workgroup (256 workgroups = 1 per CU), each wave
// issues back-to-back MFMAs on 4 independent accumulators.  hipcc --offload-arch=gfx950 -O3 -o mfma_bf16_rate mfma_bf16_rate.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
__global__ void k(float* out, int iters, int valu) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  bf16x8 a = __builtin_bit_cast(bf16x8, make_uint4(threadIdx.x, 1, 2, 3)), b = __builtin_bit_cast(bf16x8, make_uint4(5, threadIdx.x, 7, 9));
  float x = threadIdx.x * 1e-3f, y = 1.0001f;
  const bool mfma_wave = (valu == 0) || ((threadIdx.x >> 6) < 4);
  if (mfma_wave) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 12; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
  } else {       // VALU-only partner waves (the split work of a producer): ~valu fma per 48 MFMAs of the partner
    for (int it = 0; it < iters; ++it)
      for (int r = 0; r < valu; ++r) { x = __builtin_fmaf(x, y, 0.5f); y = __builtin_fmaf(y, x, 0.25f); }
  }
  float s = x + y;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 12345.f) out[0] = s;
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {4, 8}) for (int valu : {0, 100, 200, 300}) {
    if (waves == 4 && valu) continue;
    const int iters = 4000;
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, out, 100, valu);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(64 * waves), 0, 0, out, iters, valu); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int mw = valu ? 4 : waves;
    double flops = 256.0 * mw * iters * 48.0 * 32 * 32 * 16 * 2;
    printf("waves/CU %d (%d MFMA waves%s)  %.3f ms  %.0f TF/s bf16  = %.1f TF/s-equivalent at 6 MFMAs per product\n", waves, mw,
           valu ? ", 4 VALU waves" : "", ms, flops / ms / 1e9, flops / ms / 1e9 / 6);
    if (valu) printf("    VALU partner: %d dependent fma pairs per 48 MFMAs\n", valu);
  }
  return 0;
}
