# End-to-end A/B of an environment switch on one box, interleaved:  bash tools/ab_bench.sh CONFIG VAR VALUE_A VALUE_B [rounds]
# (a value of "unset" runs with the variable absent)
cfg=$1; var=$2; a=$3; b=$4; n=${5:-2}
for r in $(seq $n); do for v in "$a" "$b"; do
  if [ "$v" = unset ]; then pre="env -u $var"; else pre="env $var=$v"; fi
  $pre python bench.py --config $cfg --steps 4 --warmup 1 --no-native-line --no-pmc --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg $var=$v:', d['value'], d['unit'], d['ms_per_denoise_step'], 'ms per denoise step')"
done; done
