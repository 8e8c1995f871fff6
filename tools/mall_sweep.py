"""Size sweep of the CLEAN streaming-read bandwidth of each level of the memory hierarchy, for pricing cache-resident gather
launches (bench.py `roofline`, VERDICT r3 #3 iii): sg_stream_read_strided_hip with stride = grid size -- the resident waves
sweep ONE contiguous window (about 7 MB: 7 168 resident single-wave workgroups x 1 KiB) through the buffer, no two waves
ever ask for the same burst, and a pass over the buffer re-reads a byte only after (buffer size) other bytes.  So:

  buffer <= 32 MB (8 x 4 MB L2)    -> the L2 rate for this launch geometry
  32 MB < buffer <= 256 MB         -> the Infinity-Cache (MALL) rate: every L2 sees buffer / 8 > 4 MB per pass, LRU keeps nothing
  buffer >> 256 MB                 -> the HBM streaming rate

Round 3's ceiling (consecutive bursts per wave, buffer wrapped) was unstable: 26.5 TB/s at 52 MB, 15.8 TB/s at 53 MB.  Cause:
with 256 consecutive KiB per wave and 7 168 resident waves the waves in flight cover 1.8 GB of addresses -- 34 laps of the
buffer -- so wave w and wave w + (buffer / 256 KiB) read the SAME bursts at the same time; when that distance is a multiple
of 8 both run on the same XCD and the second one hits L2.  52 MB = 208 x 256 KiB (208 % 8 == 0: L2 hits), 53 MB = 212 (212 % 8
== 4: another XCD, Infinity Cache).  Both forms are printed here for the record.

    python tools/mall_sweep.py > profiles/r4_mall_sweep.txt
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import _lib as L  # noqa: E402


def rate(buf, n_bytes, workgroups, bursts, stride, reps=5):
    lib, st = L.lib(), L.stream_ptr()
    sink = torch.zeros(4, dtype=torch.float32, device=buf.device)

    def launch():
        if stride:
            L.check(lib.sg_stream_read_strided_hip(L.ptr(buf), n_bytes, bursts, workgroups, stride, L.ptr(sink), st), "strided")
        else:
            L.check(lib.sg_stream_read_hip(L.ptr(buf), n_bytes, bursts, workgroups, L.ptr(sink), st), "consecutive")
    launch()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    gb = workgroups * bursts * 1024 / 1e9
    return gb / ts[len(ts) // 2], gb / ts[0], gb / ts[-1]


if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    wg, bursts = 39063, 256           # the gather's grid at 10 M edges: 10.24 GB read per launch
    print("# clean = sg_stream_read_strided_hip(stride = grid): median / best / worst of 5 launches, GB/s")
    print("# wrapped = round 3's sg_stream_read_hip (256 consecutive KiB per wave): median")
    for mb in (4, 8, 16, 24, 28, 32, 36, 40, 48, 52, 53, 56, 64, 68, 72, 80, 96, 104, 109, 112, 128, 144, 160, 176, 192, 208,
               224, 240, 256, 288, 320, 384, 512, 768, 1024, 2048, 4096):
        n = mb << 20
        buf = torch.empty(n // 4, dtype=torch.float32, device=dev).normal_()
        med, best, worst = rate(buf, n, wg, bursts, wg)
        wrapped = rate(buf, n, wg, bursts, 0)[0]
        print("buffer %5d MB   clean %8.0f / %8.0f / %8.0f   wrapped %8.0f" % (mb, med, best, worst, wrapped), flush=True)
        del buf
