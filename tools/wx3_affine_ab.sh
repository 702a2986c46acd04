for r in 1 2; do
for shp in "4 256 256 256 256" "4 64 64 512 512"; do
  echo -n "plain v5      : "; AZ_WINO=wx3 python tools/conv_micro.py $shp 3 1 30 2>&1 | grep -v amdgpu
  echo -n "affine v5     : "; AZ_AFFINE=1 AZ_WINO=wx3 AZULA_AMD_LIB=azula_amd/csrc/_ab/libazula_amd_wx3v5.so python tools/conv_micro.py $shp 3 1 30 2>&1 | grep -v amdgpu
  echo -n "affine early  : "; AZ_AFFINE=1 AZ_WINO=wx3 python tools/conv_micro.py $shp 3 1 30 2>&1 | grep -v amdgpu
  echo -n "affine f32    : "; AZ_AFFINE=1 AZ_WINO=1 python tools/conv_micro.py $shp 3 1 30 2>&1 | grep -v amdgpu
done; done
