#!/bin/bash
# second f16x2 run: the whole GPU suite with f16x2 as the mode of fp32 modules, attention x3 vs f16x2, two kernel A/Bs
set -u
OUT=gpurun_out/f16x2_gate2.txt
mkdir -p gpurun_out
: > $OUT
echo "== whole GPU suite under AZ_FP32_MFMA=f16x2" | tee -a $OUT
AZ_FP32_MFMA=f16x2 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -40 | tee -a $OUT
echo "== attention: bf16x3 against f16x2" | tee -a $OUT
for shape in "64 12 256 64" "32 12 288 64" "4 8 1024 64" "4 16 256 64" "4 4 4096 64"; do
  for rep in 1 2; do
    for m in bf16x3 f16x2; do echo -n "$m " | tee -a $OUT; AZ_FP32_MFMA=$m python tools/attn_micro.py $shape 2>&1 | tail -1 | tee -a $OUT; done
  done
done
echo "== K3x f16x2: low piece by v_fma_mix (variant wx3h_mix) -- correctness first" | tee -a $OUT
AZULA_AMD_LIB=azula_amd/csrc/_ab/libazula_amd_wx3h_mix.so timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "wh2 or f16x2_domain or accuracy" 2>&1 | tail -3 | tee -a $OUT
for shape in "4 256 256 256 256" "4 64 64 512 512" "4 128 128 512 512" "4 32 32 1024 1024"; do
  for rep in 1 2; do
    echo -n "tree    " | tee -a $OUT; AZ_WINO=wh2 python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT
    echo -n "fma_mix " | tee -a $OUT; AZULA_AMD_LIB=azula_amd/csrc/_ab/libazula_amd_wx3h_mix.so AZ_WINO=wh2 python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT
  done
done
echo "== f16x2 GEMM 256 x 256 tile: the pinned issue pattern against the compiler's own order (variant h2big_nosched)" | tee -a $OUT
for shape in "64 256 1 768 3072" "64 256 1 3072 768" "64 256 1 768 768"; do
  for rep in 1 2; do
    echo -n "tree    " | tee -a $OUT; AZ_WINO=h2 AZ_ACT=0 python tools/conv_micro.py $shape 1 1 30 2>&1 | tail -1 | tee -a $OUT
    echo -n "nosched " | tee -a $OUT; AZULA_AMD_LIB=azula_amd/csrc/_ab/libazula_amd_h2big_nosched.so AZ_WINO=h2 AZ_ACT=0 python tools/conv_micro.py $shape 1 1 30 2>&1 | tail -1 | tee -a $OUT
  done
done
echo "== bench lines c5 / c6 (no CPU baseline / PMC / native line)" | tee -a $OUT
for cfg in c5 c6; do
  for m in bf16x3 f16x2; do
    python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-native-line --fp32-mfma $m 2>gpurun_out/bench_${cfg}_${m}.err | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$cfg $m', d['value'], d['unit'], 'ms/denoise', round(d['ms_per_step']/d['config']['denoise_steps'],3), 'dominant', r['entry'], r['avg_us'], 'us frac', r['frac'])
" | tee -a $OUT
  done
done
