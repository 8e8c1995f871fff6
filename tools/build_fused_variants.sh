#!/bin/bash
# Builds of csrc/agg_fused.hip with other wave splits / pipeline depths (SG_FUSED_GW gather waves, SG_FUSED_MW matrix waves,
# FUSED_SRC = another source file (a development copy, path relative to csrc/),
# SG_FUSED_NB rows in flight per gather wave, SG_FUSED_BRING B fragment sets in flight, SG_FUSED_ADB double-buffered A
# fragments) into tools/ablate/fv_<name>/libstargcn_hip.so (git-ignored); run with SG_LIB_OVERRIDE (tools/exp_r5_fused.py).
set -e
cd "$(dirname "$0")/../star-gcn_amd/csrc"
make -j8 > /dev/null
OBJS=$(ls *.o | grep -v agg_fused.o)
while read name gw mw nb br adb extra; do
  [ -z "$name" ] && continue
  d=../../tools/ablate/fv_$name; mkdir -p $d
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-parameter \
    -DSG_FUSED_GW=$gw -DSG_FUSED_MW=$mw -DSG_FUSED_NB=$nb -DSG_FUSED_BRING=$br -DSG_FUSED_ADB=$adb $extra -I. -c ${FUSED_SRC:-agg_fused.hip} -o $d/agg_fused.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $d/libstargcn_hip.so $OBJS $d/agg_fused.o
  echo "built $name: GW $gw MW $mw NB $nb BRING $br ADB $adb"
done <<LIST
${FUSED_VARIANTS:-g4m4 4 4 32 6 1
g8m8b3 8 8 16 3 0
g8m8b4 8 8 16 4 0
g8m4 8 4 16 3 0}
LIST
