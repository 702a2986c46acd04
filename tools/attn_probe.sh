# attention: fp32 MFMA kernel (AZ_ATTN_X3=0) vs bf16x3 contractions (default), interleaved; extra arguments = A/B library names
for r in 1 2; do
for shp in "64 12 256 64" "32 12 288 64" "4 8 1024 64" "8 6 256 80" "16 8 256 32"; do
  for opts in ""; do
    echo -n "fp32: "; AZ_ATTN_OPTS=$opts AZ_ATTN_X3=0 python tools/attn_micro.py $shp 100 2>&1 | grep -v amdgpu.ids
    for v in main "$@"; do
      lib=azula_amd/csrc/_ab/libazula_amd_$v.so; [ "$v" = main ] && lib=azula_amd/csrc/libazula_amd.so
      echo -n "x3[$v]: "; AZULA_AMD_LIB=$lib AZ_ATTN_OPTS=$opts AZ_ATTN_X3=1 python tools/attn_micro.py $shp 100 2>&1 | grep -v amdgpu.ids
    done
  done
done; done
