for r in 1 2; do
for shp in "4 256 256 256 256" "4 128 128 256 256" "4 64 64 512 512" "4 128 128 512 512"; do
  for rect in "8,4" "16,2" "32,1" "4,4" "8,2"; do
    echo -n "rect $rect: "; AZ_DEBUG_AB=1 AZ_WINO_RECT=$rect AZ_WINO=wx3 python tools/conv_micro.py $shp 3 1 30 2>&1 | grep -v amdgpu
  done
done; done
