#!/bin/bash
# f16x2 against bf16x3 on one box: the new kernel tests, layer timings (same process order, alternating), short bench lines.
#   gpurun -- 'bash tools/f16x2_gate.sh'   ->  gpurun_out/f16x2_gate.txt
set -u
OUT=gpurun_out/f16x2_gate.txt
mkdir -p gpurun_out
: > $OUT
echo "== kernel tests (f16x2 parametrisations, accuracy, domain)" | tee -a $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "h2 or accuracy or f16x2_domain" -s 2>&1 | grep -v "^$" | tail -40 | tee -a $OUT
echo "== layers: x3 Winograd kernel, bf16x3 (wx3) against f16x2 (wh2)" | tee -a $OUT
for shape in "4 256 256 256 256" "4 64 64 512 512" "32 128 128 256 256" "4 128 128 512 512" "4 32 32 1024 1024"; do
  for rep in 1 2; do
    for m in wx3 wh2; do AZ_WINO=$m python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT; done
  done
done
echo "== token GEMMs / 1x1: bf16x3 (x3) against f16x2 (h2)" | tee -a $OUT
for shape in "64 256 1 768 3072" "64 256 1 3072 768" "64 256 1 768 2304" "64 256 1 768 768" "32 288 1 768 2304" "4 64 64 512 512" "4 128 128 256 256"; do
  for rep in 1 2; do
    for m in x3 h2; do AZ_WINO=$m AZ_ACT=0 python tools/conv_micro.py $shape 1 1 30 2>&1 | tail -1 | tee -a $OUT; done
  done
done
echo "== strided 3x3 (big tile with taps)" | tee -a $OUT
for m in x3 h2; do AZ_WINO=$m python tools/conv_micro.py 4 256 256 256 256 3 2 30 2>&1 | tail -1 | tee -a $OUT; done
echo "== bench lines (no CPU baseline / PMC / native line)" | tee -a $OUT
for cfg in c2 c3; do
  for m in bf16x3 f16x2 bf16x3 f16x2; do
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
