// store_rate.cpp -- what does a 64 x 16 B wave store cost by address pattern?  256 workgroups x 8 waves, each wave writes
// ROUNDS x 32 store instructions the way a GEMM epilogue does: SEG bytes contiguous per row, rows `stride` bytes apart.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/store_rate tools/store_rate.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

// tile of 256 rows x 1 KiB per workgroup and round (the 256 x 256 fp32 output tile); wave w owns rows [128 (w / 4), +128), 256-byte
// column block w % 4 (SEG <= 256) -- or whole rows (SEG = 1024)
template <int SEG, bool NT>
__global__ __launch_bounds__(512) void k(char* out, long long stride, long long tile_bytes, int rounds, int tiles_per_row, int wrap) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f4 v = {1.f, 2.f, 3.f, (float)lane};
  for (int r = 0; r < rounds; ++r) {
    const long long t = ((long long)r * gridDim.x + blockIdx.x) % wrap;      // tile index; tiles_per_row tiles side by side
    char* base = out + (t / tiles_per_row) * 256 * stride + (t % tiles_per_row) * 1024;
    (void)tile_bytes;
#pragma unroll 4
    for (int q = 0; q < 32; ++q) {
      long long off;
      if (SEG == 1024) {         // one full 1 KiB row per instruction: rows q * 8 + wave
        off = (long long)(q * 8 + wave) * stride + lane * 16;
      } else {
        constexpr int LPR = SEG / 16, RPI = 64 / LPR;          // lanes per row, rows per instruction
        const int row = (wave >> 2) * 128 + (q * RPI) % 128 + lane / LPR;
        const int cb = (wave & 3) * 256 + ((q * RPI) / 128) * SEG + (lane % LPR) * 16;
        off = (long long)row * stride + cb;
      }
      if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(base + off));
      else *reinterpret_cast<f4*>(base + off) = v;
    }
  }
}

template <int SEG, bool NT>
void run(char* buf, long long stride, int tiles_per_row, int wrap, const char* what) {
  const int rounds = 64;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<SEG, NT><<<256, 512>>>(buf, stride, 0, 2, tiles_per_row, wrap);
  CK(hipEventRecord(e0));
  k<SEG, NT><<<256, 512>>>(buf, stride, 0, rounds, tiles_per_row, wrap);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = 256.0 * rounds * 8 * 32 * 1024;
  printf("%-34s seg %4d B %s stride %6lld B: %7.3f ms  %5.2f TB/s  %5.1f B/clk/CU@2GHz  %4.0f cyc per wave-store per CU\n", what, SEG, NT ? "nt   " : "plain",
         stride, ms, bytes / ms / 1e9, bytes / 256 / (ms * 1e-3) / 2e9, ms * 1e-3 * 2e9 / (rounds * 8 * 32));
}

int main() {
  char* buf; const long long cap = 20ll << 30; CK(hipMalloc(&buf, cap));
  // (a) streaming: 1M x 4096 fp32 output (row stride 16 KiB, 16 tiles per row); 64 rounds x 256 tiles = 16384 tiles = 4 GB
  run<128, true>(buf, 16384, 16, 1 << 30, "stream, N=4096");
  run<256, true>(buf, 16384, 16, 1 << 30, "stream, N=4096");
  run<1024, true>(buf, 16384, 16, 1 << 30, "stream, N=4096");
  run<128, false>(buf, 16384, 16, 1 << 30, "stream, N=4096");
  run<256, false>(buf, 16384, 16, 1 << 30, "stream, N=4096");
  run<1024, false>(buf, 16384, 16, 1 << 30, "stream, N=4096");
  // (b) N = 256 output (row stride 1 KiB: a tile is 256 KB contiguous)
  run<128, true>(buf, 1024, 1, 1 << 30, "stream, N=256");
  run<256, true>(buf, 1024, 1, 1 << 30, "stream, N=256");
  run<1024, true>(buf, 1024, 1, 1 << 30, "stream, N=256");
  // (c) L2-resident: every workgroup rewrites its own tile
  run<128, false>(buf, 16384, 16, 256, "L2-resident, N=4096");
  run<256, false>(buf, 16384, 16, 256, "L2-resident, N=4096");
  run<1024, false>(buf, 16384, 16, 256, "L2-resident, N=4096");
  run<1024, false>(buf, 1024, 1, 256, "L2-resident, N=256");
  return 0;
}
