"""star-gcn_amd: MI355X-native (gfx950) implementation of the STAR-GCN multi-link graph-conv hot path.

Layout
  csrc/            hand-written HIP kernels + the C ABI (include/stargcn.h) -> libstargcn_hip.so
  _lib.py          ctypes binding of that library (fails loudly if it is missing; no CPU fallback)
  contrib.py       the `F.contrib.seg_*` operator surface of the reference on torch CUDA tensors
  ops.py           thin wrappers of the C-ABI entry points (gather, seg ops, GEMM, fused aggregator, ...)
  functional.py    autograd glue: linear (Dense / FullyConnected on the matrix-core GEMM), multilink_aggregate, take_rows, ...
  plan.py          cached aggregation plans (fused multi-link CSR + transpose, source-range phases), host- or device-built
  device_graph.py  rating graph resident in HBM, plans built on the device; device_sampler.py / resident.py: per-batch work
  model.py         the STAR-GCN network on this API (Net, losses, deterministic init, scale calibration)
  dist.py          1-D user-block node partition: crossings, in-place all-reduce on a communication stream
  datasets.py      MovieLens ETL;  synthetic.py: the SURVEY 8(d) synthetic graphs
  mxgraph/         drop-in mirror of the reference's mxgraph.graph / iterators / layers (aggregators, HeterGCNLayer, ...)
"""
__version__ = "0.1.0"
