# bf16x3 token GEMMs: 128 x 128 tile (AZ_X3_BIG=0) vs 256 x 256 everywhere (1) vs 192-cout x 256 tiles (3) vs the library's plan (unset)
for r in 1 2; do
for shp in "64 256 1 768 2304" "64 256 1 768 768" "64 256 1 768 3072" "64 256 1 3072 768" "32 288 1 768 4096" "32 288 1 768 2304" "32 288 1 2048 768" "4 128 128 256 256" "4 64 64 512 512"; do
  for big in 0 1 3 ""; do
    echo -n "big=$big: "; AZ_DEBUG_AB=1 AZ_X3_BIG=$big AZ_WINO=x3 AZ_ACT=0 python tools/conv_micro.py $shp 1 1 200 2>&1 | grep -v amdgpu.ids | tail -1
  done
done; done
