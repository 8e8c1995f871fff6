"""one clean streaming-read configuration, a few launches (profiling target of tools/pmc_mall.sh): python tools/one_stream_read.py MB
sg_stream_read_strided_hip over a buffer of MB megabytes, the gather's grid (39 063 single-wave workgroups x 256 bursts)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from star_gcn_amd import _lib as L
mb = int(sys.argv[1])
n = mb << 20
buf = torch.empty(n // 4, dtype=torch.float32, device="cuda").normal_()
sink = torch.zeros(4, dtype=torch.float32, device="cuda")
for _ in range(4):
    L.check(L.lib().sg_stream_read_strided_hip(L.ptr(buf), n, 256, 39063, 39063, L.ptr(sink), L.stream_ptr()), "stream")
torch.cuda.synchronize()
