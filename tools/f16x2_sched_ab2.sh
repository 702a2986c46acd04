#!/bin/bash
# second round of slot-order experiments on the f16x2 Winograd kernel (tree = the early-load order)
set -u
OUT=gpurun_out/f16x2_sched_ab2.txt
mkdir -p gpurun_out
: > $OUT
AB=azula_amd/csrc/_ab
echo "== correctness: tree and variants" | tee -a $OUT
echo -n "tree: " | tee -a $OUT; timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "wh2" 2>&1 | tail -1 | tee -a $OUT
for v in wx3h_gl1 wx3h_p1 wx3h_p1nofence; do
  echo -n "$v: " | tee -a $OUT; AZULA_AMD_LIB=$AB/libazula_amd_$v.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "wh2" 2>&1 | tail -1 | tee -a $OUT
done
for shape in "4 256 256 256 256" "4 64 64 512 512" "4 128 128 512 512" "4 32 32 1024 1024"; do
  for rep in 1 2; do
    echo -n "tree     " | tee -a $OUT; AZ_WINO=wh2 python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT
    for v in wx3h_lategl wx3h_gl1 wx3h_p1 wx3h_p1nofence; do
      echo -n "$v " | tee -a $OUT; AZULA_AMD_LIB=$AB/libazula_amd_$v.so AZ_WINO=wh2 python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT
    done
  done
done
