# bf16x3 token GEMMs on real and on zero operands (the power cap's share):  bash tools/x3_probe.sh
for r in 1 2; do
for z in "" 1; do
for shp in "64 256 1 768 2304" "64 256 1 768 768" "64 256 1 768 3072" "64 256 1 3072 768"; do
  echo -n "zero=$z: "; AZ_ZERO=$z AZ_WINO=x3 AZ_ACT=0 python tools/conv_micro.py $shp 1 1 200 2>&1 | grep -v amdgpu.ids
done; done; done
