"""Drop-in mirror of the reference's `mxgraph` package for the hot path: `mxgraph.layers` (operator/layer API)
and `mxgraph.graph` (CSRMat / HeterGraph plan producers), on torch tensors + libstargcn_hip."""
