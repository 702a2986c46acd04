#!/bin/bash
# f16x2 with the activation scale measured from unbounded sources (AZ_F16X2_DYNAMIC): kernel tests, whole suite, A/B of the bench lines
set -u
OUT=gpurun_out/f16x2_gate5.txt
mkdir -p gpurun_out
: > $OUT
echo "== new kernel tests" | tee -a $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "dynamic_scale or absmax or f16x2_domain or accuracy" -s 2>&1 | grep -E "relative error|passed|failed|Error" | tail -50 | tee -a $OUT
echo "== whole GPU suite, default mode" | tee -a $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -8 | tee -a $OUT
echo "== bench lines: AZ_F16X2_DYNAMIC=0 against 1 (no CPU baseline / PMC / other-mode lines)" | tee -a $OUT
for cfg in c2 c5 c3 c6; do
  for d in 0 1 0 1; do
    AZ_F16X2_DYNAMIC=$d python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-native-line 2>gpurun_out/bench_${cfg}_dyn$d.err | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('$cfg dynamic=$d', d['value'], d['unit'], 'ms/denoise', round(d['ms_per_step']/d['config']['denoise_steps'],3), {k: (v['launches'], v['ms_per_denoise_step'], v['frac']) for k, v in d['roofline_kernels'].items()}, {k: v for k, v in d.get('step_breakdown', {}).get('other_ms', {}).items() if 'absmax' in k} if isinstance(d.get('step_breakdown', {}).get('other_ms'), dict) else '')
" | tee -a $OUT
  done
done
