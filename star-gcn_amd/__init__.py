"""star-gcn_amd: MI355X-native (gfx950) implementation of the STAR-GCN multi-link graph-conv hot path.

Layout
  csrc/            hand-written HIP kernels + the C ABI (include/stargcn.h) -> libstargcn_hip.so
  _lib.py          ctypes binding of that library (fails loudly if it is missing; no CPU fallback)
  contrib.py       the `F.contrib.seg_*` operator surface of the reference on torch CUDA tensors
  plan.py          cached aggregation plans (fused multi-link CSR + transpose), built by native host code
  dense.py         Dense / FullyConnected on the fp32 MFMA GEMM
  mxgraph/layers/  drop-in mirror of reference mxgraph/layers (aggregators, HeterGCNLayer, ...)
"""
__version__ = "0.1.0"
