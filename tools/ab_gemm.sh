#!/bin/bash
# A/B of variant libraries on token GEMM shapes (az_conv2d_x3_f32), interleaved:  bash tools/ab_gemm.sh name1 name2 ...
shapes=("64 256 1 768 2304" "64 256 1 768 768" "64 256 1 768 3072" "64 256 1 3072 768" "32 288 1 768 4096")
for r in 1 2; do
for v in "$@"; do
  lib=azula_amd/csrc/_ab/libazula_amd_$v.so; [ "$v" = main ] && lib=azula_amd/csrc/libazula_amd.so
  for shp in "${shapes[@]}"; do
    echo -n "$v: "; AZ_ACT=0 AZULA_AMD_LIB=$lib python tools/conv_micro.py $shp 1 1 30 2>&1 | grep -v amdgpu.ids
  done
done
done
