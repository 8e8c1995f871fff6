#!/bin/bash
# Round 6, VERDICT r5 "next" #3: upper bound of what halving the B-plane bytes per gathered row (a 128-row tile) could save in
# the fused aggregate -> contract kernel.  Variant build -DSG_FUSED_SKIPB=1: every second tile of a workgroup reads one 2 KB unit
# for all of its B fragments (L1-resident), everything else -- tiles, barriers, matrix instructions, double buffering --
# unchanged.  Timed on the config-5 shard graph, forward of either direction (tools/exp_r5_fused.py bench-graph), beside the
# shipped build and SG_FUSED_ABLATE=64 (ALL tiles read level 0's planes: 256 KB, L2-resident).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  python tools/exp_r5_fused.py bench-graph 2>&1 | grep "into" | cut -c1-60 | sed "s/^/shipped        /"
  SG_LIB_OVERRIDE=$PWD/tools/ablate/fv_skipb/libstargcn_hip.so python tools/exp_r5_fused.py bench-graph 2>&1 | grep "into" | cut -c1-60 | sed "s/^/skipb (half B) /"
  SG_FUSED_ABLATE=64 python tools/exp_r5_fused.py bench-graph 2>&1 | grep "into" | cut -c1-60 | sed "s/^/B = level 0    /"
done
