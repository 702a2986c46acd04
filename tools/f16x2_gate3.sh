#!/bin/bash
# third f16x2 run: f16x2 only on bounded inputs (Act.bounded), the third filter plane derived in the kernel (A/B against HEAD's loads)
set -u
OUT=gpurun_out/f16x2_gate3.txt
mkdir -p gpurun_out
: > $OUT
HEADLIB=azula_amd/csrc/_ab/libazula_amd_head.so
echo "== whole GPU suite under AZ_FP32_MFMA=f16x2" | tee -a $OUT
AZ_FP32_MFMA=f16x2 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -15 | tee -a $OUT
echo "== third filter plane derived in the kernel (tree) against loaded (HEAD's conv.hip / wino_x3.hip)" | tee -a $OUT
for shape in "4 256 256 256 256" "4 64 64 512 512" "4 128 128 512 512" "32 128 128 256 256" "4 32 32 1024 1024"; do
  for rep in 1 2; do
    echo -n "loaded  " | tee -a $OUT; AZULA_AMD_LIB=$HEADLIB AZ_WINO=wh2 python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT
    echo -n "derived " | tee -a $OUT; AZ_WINO=wh2 python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT
  done
done
for shape in "64 256 1 768 3072" "64 256 1 3072 768" "64 256 1 768 768" "32 288 1 768 2304" "4 64 64 512 512"; do
  for rep in 1 2; do
    echo -n "loaded  " | tee -a $OUT; AZULA_AMD_LIB=$HEADLIB AZ_WINO=h2 AZ_ACT=0 python tools/conv_micro.py $shape 1 1 30 2>&1 | tail -1 | tee -a $OUT
    echo -n "derived " | tee -a $OUT; AZ_WINO=h2 AZ_ACT=0 python tools/conv_micro.py $shape 1 1 30 2>&1 | tail -1 | tee -a $OUT
  done
done
echo "== bench lines (no CPU baseline / PMC / native line): bf16x3, f16x2" | tee -a $OUT
for cfg in c2 c3 c5 c6; do
  for m in bf16x3 f16x2; do
    python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-native-line --fp32-mfma $m 2>gpurun_out/bench_${cfg}_${m}.err | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$cfg $m', d['value'], d['unit'], 'ms/denoise', round(d['ms_per_step']/d['config']['denoise_steps'],3), 'dominant', r['entry'], r['avg_us'], 'us frac', r['frac'], {k: (v['ms_per_denoise_step'], v['frac']) for k, v in d['roofline_kernels'].items()})
" | tee -a $OUT
  done
done
