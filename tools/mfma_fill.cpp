// mfma_fill.cpp -- do VALU / LDS instructions placed between v_mfma_f32_32x32x16_f16 issue in the shadow of the matrix
// instructions, and does it matter whether the accumulators are arch VGPRs or AGPRs?  One workgroup per CU, WAVES waves,
// two independent accumulator chains per wave, FILL filler instructions after every matrix instruction.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/mfma_fill tools/mfma_fill.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

template <int FILL, int KIND, bool AG>
__global__ __launch_bounds__(512) void k(float* out, int iters, float c, float d) {
  __shared__ float lds[8192];
  f32x16 a0, a1;
  for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
  f16x8 fa, fb;
  for (int e = 0; e < 8; ++e) { fa[e] = (_Float16)(threadIdx.x * 0.001f + e); fb[e] = (_Float16)(e * 0.5f); }
  float x[8];
  for (int q = 0; q < 8; ++q) x[q] = threadIdx.x + q;
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const float* lp = lds + (threadIdx.x & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (AG) {
        if (m & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(a1) : "v"(fa), "v"(fb));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(a0) : "v"(fa), "v"(fb));
      } else {
        if (m & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a1) : "v"(fa), "v"(fb));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(a0) : "v"(fa), "v"(fb));
      }
#pragma unroll
      for (int q = 0; q < FILL; ++q) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[q & 7]) : "v"(c), "v"(d));
        else if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(iters) : : "scc");
        else { __attribute__((ext_vector_type(4))) float v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)lp)); asm volatile("" :: "v"(v)); }
      }
    }
    if (KIND == 1) iters -= 8 * FILL;
    if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  asm volatile("s_nop 15\n\ts_nop 15");
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e];
  for (int q = 0; q < 8; ++q) s += x[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FILL, int KIND, bool AG>
void run(int waves) {
  float* out; CK(hipMalloc(&out, 256 * 512 * 4));
  const int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<FILL, KIND, AG><<<256, waves * 64>>>(out, 100, 1.0001f, 0.5f);
  CK(hipEventRecord(e0));
  k<FILL, KIND, AG><<<256, waves * 64>>>(out, iters, 1.0001f, 0.5f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfma_per_simd = (double)iters * 8 * (waves / 4.0);
  printf("waves %d fill %2d kind %s acc %s: %.3f ms  %.1f ns per MFMA per SIMD (32 cyc @2.4GHz = 13.3 ns)\n", waves, FILL,
         KIND == 0 ? "valu" : KIND == 1 ? "salu" : "ds_read", AG ? "AGPR" : "VGPR", ms, ms * 1e6 / mfma_per_simd);
  CK(hipFree(out));
}
int main() {
  for (int waves : {4, 8}) {
    run<0, 0, false>(waves); run<0, 0, true>(waves);
    run<2, 0, false>(waves); run<2, 0, true>(waves);
    run<4, 0, false>(waves); run<4, 0, true>(waves);
    run<6, 0, false>(waves); run<6, 0, true>(waves);
    run<4, 1, false>(waves); run<4, 1, true>(waves);
    run<1, 2, false>(waves); run<1, 2, true>(waves);
    run<2, 2, false>(waves); run<2, 2, true>(waves);
  }
  return 0;
}
