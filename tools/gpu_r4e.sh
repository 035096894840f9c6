#!/bin/bash
# A/B of the round-3 one-wave-per-SIMD screening kernel (commit 867c888, worktree tools/_old867): top-3 fold / no fold / top-2 pairs
set -x
O=$PWD/gpurun_out/r4e; mkdir -p $O
cd tools/_old867
for v in base nofold top2; do
  VQHIP_SCREEN_PERSIST=1 VQHIP_SO=$PWD/tools/variants/libvqhip_$v.so timeout 120 python tools/time_assign.py > $O/p1_$v.txt 2>&1
done
VQHIP_SO=$PWD/tools/variants/libvqhip_base.so timeout 120 python tools/time_assign.py > $O/p0_base.txt 2>&1
cd ../..
tail -3 $O/p1_*.txt $O/p0_base.txt
