// stream_read.hip -- measurement aid only (bench.py roofline, tools/mall_sweep.py): best-case streaming reads with the
// gather's launch geometry.  Kept out of seg_gather.hip so that the PMC records stamped with that file's sha
// (profiles/pmc_traffic.json) stay valid when only the measurement aid changes.
#include "common.hpp"

// ---- measurement aid: best-case streaming read with the gather's launch geometry ---------------------------------------
// One 64-lane wavefront per workgroup (as in the gather); wave w reads `bursts` consecutive 1 KiB bursts (float4 per
// lane, 4 in flight -- the gather's row reads at width 256 with a perfectly regular index stream) starting at burst
// w * bursts, wrapping around the buffer.  With a buffer inside the Infinity Cache but several times the aggregate L2
// the waves in flight are spread over the whole buffer, so (almost) every burst misses L2 and hits the Infinity Cache;
// with a buffer of a few MB every burst hits L2.  bench.py measures both IN THE SAME RUN to price cache-resident shapes.
namespace sg {
__global__ __launch_bounds__(kWave) void stream_read_kernel(const float4* __restrict__ buf, long long n_bursts, int bursts,
                                                            float* __restrict__ sink, int strided, long long step_in) {
  const int lane = threadIdx.x;
  // stride == 1: wave w reads `bursts` CONSECUTIVE bursts from burst w * bursts (the gather's row reads with a perfectly
  // regular index stream).  stride > 1 (normally = the grid size): wave w reads bursts w, w + stride, w + 2 stride, ...
  // -- the waves resident at one time then sweep a contiguous window of about (resident waves) KiB through the buffer and
  // no two of them ask for the same burst, so a buffer larger than the L2s is served by the Infinity Cache (or HBM)
  // without sibling-wave L2 hits: the clean bandwidth of that level.
  // (the step must not be 0 and should share no factor with n_bursts, or waves re-read / alias the same bursts and L1 / L2
  // hits inflate the rate -- ADVICE r4: a grid that is a multiple of n_bursts gave step 0: the launcher adjusts it)
  const long long step = strided ? step_in : 1;
  long long b = (strided ? static_cast<long long>(blockIdx.x) : static_cast<long long>(blockIdx.x) * bursts) % n_bursts;
  auto next = [&](long long v) { v += step; return v >= n_bursts ? v - n_bursts : v; };
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = 0;
  for (; k + 3 < bursts; k += 4) {
    const long long b1 = next(b), b2 = next(b1), b3 = next(b2);
    const float4 x0 = buf[b * kWave + lane], x1 = buf[b1 * kWave + lane], x2 = buf[b2 * kWave + lane],
                 x3 = buf[b3 * kWave + lane];
    acc.x += x0.x + x1.x + x2.x + x3.x; acc.y += x0.y + x1.y + x2.y + x3.y;
    acc.z += x0.z + x1.z + x2.z + x3.z; acc.w += x0.w + x1.w + x2.w + x3.w;
    b = next(b3);
  }
  for (; k < bursts; ++k) {
    const float4 x0 = buf[b * kWave + lane];
    acc.x += x0.x; acc.y += x0.y; acc.z += x0.z; acc.w += x0.w;
    b = next(b);
  }
  const float v = acc.x + acc.y + acc.z + acc.w;
  if (v == 12345.678f) sink[0] = v;   // keeps the loads alive without a store per wave
}
}  // namespace sg

// `workgroups` single-wave workgroups each read `bursts` 1 KiB bursts of the `bytes`-long buffer (multiple of 1024,
// 16-byte aligned): workgroups * bursts KiB in total.
static int stream_read_launch(const void* buf, int64_t bytes, int bursts, int64_t workgroups, float* sink, void* stream,
                              int64_t stride) {
  if (!buf || !sink || bytes < 1024 || bytes % 1024 || bursts < 1 || workgroups < 1 || workgroups >= (1ll << 31) || stride < 1)
    return sg::fail(SG_ERR_INVALID, "bad stream-read arguments");
  if (!sg::aligned(buf, 16)) return sg::fail(SG_ERR_INVALID, "buffer must be 16-byte aligned");
  const long long n_bursts = bytes / 1024;
  long long step = stride;
  if (stride > 1) {      // reduce mod n_bursts, then bump to the next value coprime with n_bursts (every burst visited once per sweep)
    step = stride % n_bursts;
    if (step == 0) step = 1;
    auto gcd = [](long long a, long long b) { while (b) { const long long t = a % b; a = b; b = t; } return a; };
    while (step < n_bursts && gcd(step, n_bursts) != 1) ++step;
    if (step >= n_bursts) step = 1;
  }
  hipLaunchKernelGGL(sg::stream_read_kernel, dim3(static_cast<unsigned>(workgroups)), dim3(sg::kWave), 0,
                     static_cast<hipStream_t>(stream), static_cast<const float4*>(buf), n_bursts, bursts, sink, stride > 1 ? 1 : 0, step);
  return sg::check_launch("stream_read");
}
SG_API int sg_stream_read_hip(const void* buf, int64_t bytes, int bursts, int64_t workgroups, float* sink, void* stream) {
  return stream_read_launch(buf, bytes, bursts, workgroups, sink, stream, 1);
}
// the same with a burst stride per wave (see the kernel): wave w reads bursts w, w + stride, ... (mod bytes / 1024)
SG_API int sg_stream_read_strided_hip(const void* buf, int64_t bytes, int bursts, int64_t workgroups, int64_t stride, float* sink,
                                      void* stream) {
  return stream_read_launch(buf, bytes, bursts, workgroups, sink, stream, stride);
}

