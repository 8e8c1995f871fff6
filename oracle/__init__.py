"""CPU oracle for the STAR-GCN hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product (star-gcn_amd/) never does; a product path routed through here would void parity.

`oracle.seg` wraps oracle/seg_oracle.c (plain-C restatement of reference seg_op.cc CPU kernels,
pinned against the reference's numpy models via tests/golden/).  `oracle.model` restates the
Python layers (aggregators.py / layers.py / STAR-GCN.py Net) on numpy/torch-CPU float64 in the
REFERENCE's operation order (dense FullyConnected first, then seg_weighted_pool): parity unpinned
for that part, because MXNet/Gluon cannot be imported in this image.
"""
