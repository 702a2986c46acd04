#!/bin/bash
# Timing ablations of az_conv2d_winograd_x3_f32 (libraries from tools/ablate.py; WRONG results, timing only), interleaved with the product library.
shapes=("4 256 256 256 256" "4 64 64 512 512")
for r in 1 2; do
  for shp in "${shapes[@]}"; do
    for v in main "$@"; do
      lib=azula_amd/csrc/_ab/libazula_amd_$v.so; [ "$v" = main ] && lib=azula_amd/csrc/libazula_amd.so
      echo -n "$v: "; AZ_WINO=wx3 AZULA_AMD_LIB=$lib python tools/conv_micro.py $shp 3 1 30 2>&1 | grep -v amdgpu.ids
    done
  done
done
