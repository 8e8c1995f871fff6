// Issue rate of the VALU instructions the x6v2 producers use (one wave per SIMD, 4 independent chains):
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate tools/valu_rate.cpp && tools/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int OP>
__global__ void k(unsigned* out, int iters) {
  unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, g0 = 1.5f, g1 = 2.5f, g2 = 3.5f, g3 = 4.5f;
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v p0 = {f0, f1}, p1 = {f2, f3}, p2 = {g0, g1}, p3 = {g2, g3};
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < REP; ++r) {
      if (OP == 0) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a0) : "v"(f0), "v"(f1)); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a1) : "v"(f2), "v"(f3));
                     asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a2) : "v"(g0), "v"(g1)); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a3) : "v"(g2), "v"(g3)); }
      if (OP == 1) { asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a0) : "v"(a1)); asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a1) : "v"(a2));
                     asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a2) : "v"(a3)); asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(a3) : "v"(a0)); }
      if (OP == 2) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p0) : "v"(p1), "v"(p2)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p1) : "v"(p2), "v"(p3));
                     asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p2) : "v"(p3), "v"(p0)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p3) : "v"(p0), "v"(p1)); }
      if (OP == 3) { asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a0) : "v"(a1), "v"(a2)); asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a1) : "v"(a2), "v"(a3));
                     asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a2) : "v"(a3), "v"(a0)); asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a3) : "v"(a0), "v"(a1)); }
      if (OP == 4) { asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a0) : "v"(a1), "v"(a2), "v"(a3)); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a1) : "v"(a2), "v"(a3), "v"(a0));
                     asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a2) : "v"(a3), "v"(a0), "v"(a1)); asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(a3) : "v"(a0), "v"(a1), "v"(a2)); }
      if (OP == 5) { asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f0) : "v"(f1), "v"(f2)); asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f1) : "v"(f2), "v"(f3));
                     asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f2) : "v"(f3), "v"(f0)); asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f3) : "v"(f0), "v"(f1)); }
      if (OP == 6) { asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a0) : "v"(a1)); asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a1) : "v"(a2));
                     asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a2) : "v"(a3)); asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(a3) : "v"(a0)); }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (unsigned)(f0 + f1 + f2 + f3 + p0.x + p1.y + p2.x + p3.y);
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<long long*>(out)[4096] = t1 - t0;
}
template <int OP> void run(const char* name, unsigned* d) {
  const int iters = 200;
  hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256), 0, 0, d, iters);     // 4 waves = one per SIMD
  hipDeviceSynchronize();
  long long cyc; hipMemcpy(&cyc, reinterpret_cast<long long*>(d) + 4096, 8, hipMemcpyDeviceToHost);
  printf("%-22s %6.2f clock ticks per instruction (one wave per SIMD)\n", name, double(cyc) / (double(iters) * REP * 4));
}
int main() {
  unsigned* d; hipMalloc(&d, 1 << 20);
  run<5>("v_sub_f32", d); run<1>("v_and_b32", d); run<6>("v_lshlrev_b32", d); run<4>("v_perm_b32", d);
  run<0>("v_cvt_pk_bf16_f32", d); run<2>("v_pk_add_f32", d); run<3>("v_mul_lo_u32", d);
  return 0;
}
