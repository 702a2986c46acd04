#!/bin/bash
# fourth f16x2 run: f16x2 is the default mode.  Whole GPU suite (default), the oracle tests' printed errors, whole suite under bf16x3, bench lines.
set -u
OUT=gpurun_out/f16x2_gate4.txt
mkdir -p gpurun_out
: > $OUT
echo "== whole GPU suite, default mode (f16x2)" | tee -a $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -15 | tee -a $OUT
echo "== errors printed by the oracle tests, default mode (f16x2)" | tee -a $OUT
timeout 1500 python -m pytest tests/test_gpu_fullwidth.py tests/test_gpu_fullres.py tests/test_gpu_adm.py tests/test_gpu_vit.py tests/test_gpu_unet.py tests/test_gpu_jit.py -m gpu -q -s 2>&1 | grep -E "max\|d\||DDIM|DDPM|f16x2 vs" | cut -c1-260 | tee -a $OUT
echo "== whole GPU suite under AZ_FP32_MFMA=bf16x3" | tee -a $OUT
AZ_FP32_MFMA=bf16x3 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^$" | tail -8 | tee -a $OUT
echo "== bench lines (no CPU baseline / PMC / other-mode lines): bf16x3, f16x2" | tee -a $OUT
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
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee -a $OUT
