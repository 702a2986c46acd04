#!/bin/bash
# the activation scale of f16x2 launches on the stream from the producers' GroupNorm moments (AZ_F16X2_MOMENTS) instead of a pass
set -u
OUT=gpurun_out/f16x2_gate6.txt
mkdir -p gpurun_out
: > $OUT
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "moments or absmax or dynamic" -s 2>&1 | grep -E "bound from|passed|failed|rror" | tee -a $OUT
echo "== whole GPU suite, default mode" | tee -a $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee -a $OUT
echo "== bench lines: AZ_F16X2_MOMENTS=0 against 1" | tee -a $OUT
for cfg in c5 c2 c4; do
  extra=""; [ $cfg = c4 ] && extra="--denoise-steps 4 --warmup 0 --steps 1"
  [ $cfg = c4 ] || extra="--steps 2 --warmup 1"
  for d in 0 1 0 1; do
    AZ_F16X2_MOMENTS=$d python bench.py --config $cfg $extra --no-cpu-baseline --no-pmc --no-native-line 2>gpurun_out/bench_${cfg}_mom$d.err | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l)
        print('$cfg moments=$d', d['value'], d['unit'], 'ms/denoise', round(d['ms_per_step']/d['config']['denoise_steps'],3), {k: (v['launches'], v['ms_per_denoise_step']) for k, v in d['roofline_kernels'].items()}, {k: v for k, v in d['step_breakdown']['other_ms'].items() if 'absmax' in k})
" | tee -a $OUT
  done
done
