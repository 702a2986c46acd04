#!/bin/bash
# A/B of variant libraries on two layers, interleaved, 3 rounds
for r in 1 2 3; do
for v in base abl_us abl_af abl_usaf abl_all; do
  for shp in "4 256 256 256 256" "4 64 64 512 512"; do
    echo -n "$v: "; AZ_WINO=1 AZULA_AMD_LIB=azula_amd/csrc/_ab/libazula_amd_$v.so python tools/conv_micro.py $shp 3 1 30
  done
done
done
