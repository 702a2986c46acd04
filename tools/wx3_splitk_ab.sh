#!/bin/bash
# split-K of az_conv2d_winograd_x3_f32 on the small maps of C2 / ADM (batch 4): the suggestion (first line of a group) against overrides
for shp in "4 32 32 512 512" "4 16 16 1024 1024" "4 8 8 1024 1024" "4 32 32 1024 1024" "4 16 16 2048 1024" "4 8 8 2048 1024"; do
  echo -n "suggested : "; AZ_WINO=wx3 python tools/conv_micro.py $shp 3 1 40 2>&1 | grep -v amdgpu
  for sk in 1 2 4 8 16 32; do
    echo -n "splitk $sk : "; AZ_SPLITK=$sk AZ_WINO=wx3 python tools/conv_micro.py $shp 3 1 40 2>&1 | grep -v amdgpu | grep "^conv"
  done
done
