# End-to-end A/B of an environment switch on one box, interleaved:  bash tools/ab_bench.sh CONFIG VAR VALUE_A VALUE_B [rounds]
cfg=$1; var=$2; a=$3; b=$4; n=${5:-2}
for r in $(seq $n); do for v in $a $b; do
  env $var=$v python bench.py --config $cfg --steps 4 --warmup 1 --no-native-line --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg $var=$v:', d['value'], d['unit'], round(d['ms_per_step']/d['config'].get('steps',1),3) if 'steps' in d['config'] else d['ms_per_step'])"
done; done
