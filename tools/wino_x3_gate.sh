#!/bin/bash
# Winograd frequency GEMMs on the bf16 pipe (az_conv2d_winograd_x3_f32) against the fp32 stream (az_conv2d_winograd_f32),
# interleaved, on the gate layers:   bash tools/wino_x3_gate.sh [variant library names under azula_amd/csrc/_ab ...]
shapes=("4 256 256 256 256" "4 64 64 512 512" "32 128 128 256 256" "4 128 128 512 512" "4 32 32 1024 1024")
libs=("main" "$@")
for r in 1 2; do
  for shp in "${shapes[@]}"; do
    echo -n "f32   : "; AZ_WINO=1 python tools/conv_micro.py $shp 3 1 30 2>&1 | grep -v amdgpu.ids
    for v in "${libs[@]}"; do
      lib=azula_amd/csrc/_ab/libazula_amd_$v.so; [ "$v" = main ] && lib=azula_amd/csrc/libazula_amd.so
      echo -n "wx3 $v: "; AZ_WINO=wx3 AZULA_AMD_LIB=$lib python tools/conv_micro.py $shp 3 1 30 2>&1 | grep -v amdgpu.ids
    done
  done
done
