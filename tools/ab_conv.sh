#!/bin/bash
# A/B of variant libraries (azula_amd/csrc/_ab/libazula_amd_<name>.so; "main" = the tree's library) on Winograd layers,
# interleaved, 3 rounds:   bash tools/ab_conv.sh name1 name2 ...
shapes=("4 256 256 256 256" "4 128 128 512 512" "4 64 64 512 512" "4 32 32 1024 1024" "4 256 256 64 256")
for r in 1 2 3; do
for v in "$@"; do
  lib=azula_amd/csrc/_ab/libazula_amd_$v.so; [ "$v" = main ] && lib=azula_amd/csrc/libazula_amd.so
  for shp in "${shapes[@]}"; do
    echo -n "$v: "; AZ_WINO=1 AZULA_AMD_LIB=$lib python tools/conv_micro.py $shp 3 1 30 2>&1 | grep -v amdgpu.ids
  done
done
done
