mkdir -p gpurun_out; AB=azula_amd/csrc/_ab
{
for shape in "4 8 8 1024 1024" "4 16 16 1024 1024" "4 32 32 1024 1024" "4 32 32 512 512" "4 64 64 512 512" "4 256 256 256 256"; do
  for v in base wx3_unt; do
    if [ $v = base ]; then unset AZULA_AMD_LIB; else export AZULA_AMD_LIB=$PWD/$AB/libazula_amd_$v.so; fi
    echo -n "$v wx3 "; AZ_WINO=wx3 AZ_GN=1 timeout 120 python tools/conv_micro.py $shape 3 1 50 2>&1 | tail -2 | tr '\n' ' '; echo
  done
  unset AZULA_AMD_LIB
  echo -n "base direct-x3 "; AZ_WINO=x3 AZ_GN=1 timeout 120 python tools/conv_micro.py $shape 3 1 50 2>&1 | tail -2 | tr '\n' ' '; echo
done
} > gpurun_out/s7_smallmap.txt 2>&1
cat gpurun_out/s7_smallmap.txt
