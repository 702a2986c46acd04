for r in 1 2; do for v in r03stream main; do lib=azula_amd/csrc/_ab/libazula_amd_$v.so; [ "$v" = main ] && lib=azula_amd/csrc/libazula_amd.so
for shp in "64 256 1 768 2304 1 1" "4 256 256 256 256 3 2" "4 128 128 512 256 1 1" "4 64 64 512 512 3 2"; do echo -n "$v: "; AZ_WINO=0 AZ_ACT=0 AZULA_AMD_LIB=$lib python tools/conv_micro.py $shp 30 2>&1 | grep -v amdgpu.ids; done; done; done
