#!/bin/bash
# staged pixel slots of the x3 Winograd kernel at 80 B (tree) against 64 B (variant wx3_slot64): correctness, timing (both piece modes), bank conflicts
set -u
OUT=gpurun_out/wx3_slot_ab.txt
mkdir -p gpurun_out
: > $OUT
R=$PWD
AB=azula_amd/csrc/_ab/libazula_amd_wx3_slot64.so
echo "== correctness (tree): every Winograd-x3 / f16x2 kernel test" | tee -a $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -k "wx3 or wh2 or dynamic or accuracy" 2>&1 | tail -1 | tee -a $OUT
for shape in "4 256 256 256 256" "4 64 64 512 512" "4 128 128 512 512" "4 32 32 1024 1024" "32 128 128 256 256"; do
  for rep in 1 2; do
    for m in wh2 wx3; do
      echo -n "64 B $m " | tee -a $OUT; AZULA_AMD_LIB=$AB AZ_WINO=$m python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT
      echo -n "80 B $m " | tee -a $OUT; AZ_WINO=$m python tools/conv_micro.py $shape 3 1 30 2>&1 | tail -1 | tee -a $OUT
    done
  done
done
echo "== LDS counters, f16x2 form, 4 x 256^2 256 -> 256 (80 B slots)" | tee -a $OUT
export TMPDIR=/tmp
for pair in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"; do
  d=/tmp/pmc_slot; rm -rf $d
  (cd /tmp && AZ_WINO=wh2 rocprofv3 --pmc $pair --kernel-trace -d $d -o run --output-format csv -- python $R/tools/conv_micro.py 4 256 256 256 256 3 1 5 > /dev/null 2>&1)
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" $pair <<'PY' | tee -a $OUT
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if "conv_winograd_x3" in r["Kernel_Name"]: acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for n in sys.argv[2:]:
    v = sorted(acc[n].values()); print(f"{n:26s} per launch median {v[len(v)//2]:14.0f}" if v else f"{n} (no rows)")
PY
done
echo "== bench c2 / c5 (tree)" | tee -a $OUT
for cfg in c2 c5; do
  python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-native-line 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('$cfg', d['value'], d['unit'], 'ms/denoise', round(d['ms_per_step']/d['config']['denoise_steps'],3))
" | tee -a $OUT
done
