#!/bin/bash
# Development builds of the library with other gather parameters (tools/ablate/g_<name>/libstargcn_hip.so, loaded through
# SG_LIB_OVERRIDE by tools/exp_r4_gather_variants.py): chunk size (edges per wave) and rows in flight per edge group.
set -e
cd "$(dirname "$0")/../star-gcn_amd/csrc"
make -j8 > /dev/null
OBJS="stream_read.o seg_ops.o gemm_f32.o gemm_bf16x6.o gemm_x6v2.o gemm_f16x3.o multilink.o edge_mask.o embed.o plan_build.o graph_host.o"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden"
for v in "chunk128:-DSG_GATHER_CHUNK=128" "chunk512:-DSG_GATHER_CHUNK=512" "u2:-DSG_GATHER_U=2" "u8:-DSG_GATHER_U=8" "chunk512u8:-DSG_GATHER_CHUNK=512 -DSG_GATHER_U=8"; do
  name=${v%%:*}; defs=${v#*:}
  mkdir -p ../../tools/ablate/g_$name
  /opt/rocm/bin/hipcc $FLAGS $defs -c seg_gather.hip -o /tmp/seg_gather_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/ablate/g_$name/libstargcn_hip.so $OBJS /tmp/seg_gather_$name.o
done
ls -la ../../tools/ablate/g_*/libstargcn_hip.so
