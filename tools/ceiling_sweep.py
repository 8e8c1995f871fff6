"""Sweep of the best-case streaming-read ceiling (sg_stream_read_hip) over buffer sizes: where the L2 / Infinity Cache /
HBM plateaus sit for the gather's launch geometry (one wave per workgroup, 1 KiB bursts, 4 in flight)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
for wg in (39063,):
    for mb in (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 224, 256, 320, 512, 1024, 4096):
        for bursts in (64, 256):
            r = bench.measure_stream_ceiling(dev, mb << 20, wg, bursts)
            print("buffer %5d MB  workgroups %d  bursts/wave %3d  %8.0f GB/s" % (mb, wg, bursts, r), flush=True)
