// How hipExtStreamCreateWithCUMask numbers the CUs of an MI355X (8 XCDs x 32 CUs): for a few masks, launch many one-wave
// workgroups on the masked stream and histogram the XCC id and the (SE, CU) id the hardware reports per workgroup.
//   hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.cpp -o tools/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>

__global__ void probe(uint32_t* out) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // spin a little so that all slots fill
  unsigned long long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 20000) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create failed\n", name); return; }
  const int n = 65536;
  uint32_t* d;
  hipMalloc(&d, 2 * n * sizeof(uint32_t));
  hipLaunchKernelGGL(probe, dim3(n), dim3(64), 0, st, d);
  hipStreamSynchronize(st);
  std::vector<uint32_t> h(2 * n);
  hipMemcpy(h.data(), d, 2 * n * sizeof(uint32_t), hipMemcpyDeviceToHost);
  int per_xcc[16] = {0};
  std::set<uint32_t> cus;
  for (int i = 0; i < n; ++i) {
    const uint32_t xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    per_xcc[xcc]++;
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;   // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
    cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
  }
  printf("%-28s distinct (xcc, se, sh, cu): %3zu   workgroups per XCC:", name, cus.size());
  for (int x = 0; x < 8; ++x) printf(" %6d", per_xcc[x]);
  printf("\n");
  hipFree(d);
  hipStreamDestroy(st);
}

int main() {
  auto first = [](int k) { std::vector<uint32_t> m(8, 0); for (int i = 0; i < k; ++i) m[i / 32] |= 1u << (i % 32); return m; };
  auto last = [](int k) { std::vector<uint32_t> m(8, 0); for (int i = 256 - k; i < 256; ++i) m[i / 32] |= 1u << (i % 32); return m; };
  run("all 256 bits", first(256));
  run("first 8 bits", first(8));
  run("first 32 bits", first(32));
  run("first 64 bits", first(64));
  run("first 128 bits", first(128));
  run("last 64 bits", last(64));
  run("last 192 bits", last(192));
  std::vector<uint32_t> ev(8, 0x55555555u);
  run("even bits", ev);
  return 0;
}
