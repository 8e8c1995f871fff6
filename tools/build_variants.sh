#!/bin/bash
# Development builds of the library for tools/exp_x6v2_ablate.py (tools/ablate/{1,2,3}: SG_X6V2_ABLATE timing ablations) and
# tools/exp_x6v2_timing.py (tools/ablate/t: per-phase cycle counters).  Loaded through SG_LIB_OVERRIDE; tools/ablate/ is
# git-ignored and should be deleted afterwards (it travels to the GPU box with every gpurun snapshot).
set -e
cd "$(dirname "$0")/../star-gcn_amd/csrc"
make -j8 > /dev/null
OBJS="seg_gather.o seg_ops.o gemm_f32.o gemm_bf16x6.o multilink.o edge_mask.o embed.o plan_build.o graph_host.o"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -fopenmp"
for v in 1 2 3 4 t; do
  mkdir -p ../../tools/ablate/$v
  if [ $v = t ]; then D="-DSG_X6V2_TIMING=1"; else D="-DSG_X6V2_ABLATE=$v"; fi
  /opt/rocm/bin/hipcc $FLAGS $D -c gemm_x6v2.hip -o /tmp/gemm_x6v2_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fopenmp -o ../../tools/ablate/$v/libstargcn_hip.so $OBJS /tmp/gemm_x6v2_$v.o
done
ls -la ../../tools/ablate/*/libstargcn_hip.so
