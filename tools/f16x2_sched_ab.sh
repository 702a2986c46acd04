#!/bin/bash
# issue-order experiments on the f16x2 kernels (tools/ablate.py variants; results stay correct), same box, alternating
set -u
OUT=gpurun_out/f16x2_sched_ab.txt
mkdir -p gpurun_out
: > $OUT
AB=azula_amd/csrc/_ab
echo "== K3x f16x2: correctness of the variants" | tee -a $OUT
for v in wx3h_2mf wx3h_earlygl wx3h_nofence; do
  echo -n "$v: " | tee -a $OUT; AZULA_AMD_LIB=$AB/libazula_amd_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "wh2" 2>&1 | tail -1 | tee -a $OUT
done
echo "== K3x f16x2: slot orders" | tee -a $OUT
for shape in "4 256 256 256 256" "4 64 64 512 512" "4 128 128 512 512" "4 32 32 1024 1024"; do
  for rep in 1 2; do
    echo -n "tree     " | tee -a $OUT; AZ_WINO=wh2 python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT
    for v in wx3h_2mf wx3h_earlygl wx3h_nofence; do
      echo -n "$v " | tee -a $OUT; AZULA_AMD_LIB=$AB/libazula_amd_$v.so AZ_WINO=wh2 python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT
    done
  done
done
echo "== f16x2 GEMM 256 x 256 tile: vector instructions per gap" | tee -a $OUT
for shape in "64 256 1 768 3072" "64 256 1 3072 768" "64 256 1 768 768" "32 288 1 768 2304"; do
  for rep in 1 2; do
    echo -n "tree (3)    " | tee -a $OUT; AZ_WINO=h2 AZ_ACT=0 python tools/conv_micro.py $shape 1 1 30 2>&1 | tail -1 | tee -a $OUT
    for v in h2big_valu2 h2big_valu4; do
      echo -n "$v " | tee -a $OUT; AZULA_AMD_LIB=$AB/libazula_amd_$v.so AZ_WINO=h2 AZ_ACT=0 python tools/conv_micro.py $shape 1 1 30 2>&1 | tail -1 | tee -a $OUT
    done
  done
done
