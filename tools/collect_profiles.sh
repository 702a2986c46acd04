#!/bin/bash
# End-of-round evidence, on the GPU box:  bash tools/collect_profiles.sh r04   ->  gpurun_out/<tag>_*  (copy into profiles/)
#   bench lines of every configuration, rocprofv3 --kernel-trace --stats tables of C2 / C3 / C5 / C6 (same command as the
#   bench line, fewer steps, ONE mode per table: --no-native-line), the PMC traffic collections (tools/pmc_traffic.py, also run live by
#   the default bench.py for its own configuration).
TAG=${1:-r06}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err
for c in c2 c3 c5 c6; do
  extra=""; [ $c = c5 ] && extra="--denoise-steps 16"   # (rocprofv3 segfaults on the full C5 trace: 16 of the 64 denoise steps)
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o run -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-pmc --no-cpu-baseline --no-native-line $extra > $R/gpurun_out/prof_$c.json 2> $R/gpurun_out/prof_$c.err)
  db=$(find gpurun_out/prof_$c -name "*.db" | head -1)
  python profiles/summarize.py $db 40 > gpurun_out/${TAG}_${c}_kernel_stats.txt
  rm -rf gpurun_out/prof_$c
done
for c in c2 c3 c5 c6; do  # live PMC traffic of every configuration's dominant family (separate --pmc passes, --kernel-trace only)
  python tools/pmc_traffic.py collect --config $c > gpurun_out/${TAG}_traffic_$c.json 2> gpurun_out/${TAG}_traffic_$c.err
done
python bench.py --config c3 --steps 3 --warmup 1 --no-pmc > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/b_c3.err
python bench.py --config c5 --steps 2 --warmup 1 --no-pmc > gpurun_out/${TAG}_bench_c5.json 2> gpurun_out/b_c5.err
python bench.py --config c5cfg32 --steps 1 --warmup 1 --no-pmc > gpurun_out/${TAG}_bench_c5cfg32.json 2> gpurun_out/b_c5cfg32.err
python bench.py --config c6 --steps 2 --warmup 1 --no-pmc > gpurun_out/${TAG}_bench_c6.json 2> gpurun_out/b_c6.err
python bench.py --config c4 --denoise-steps 4 --steps 1 --warmup 0 --no-pmc > gpurun_out/${TAG}_bench_c4_4steps.json 2> gpurun_out/b_c4.err
python bench.py --config vol --steps 3 --warmup 1 --no-pmc > gpurun_out/${TAG}_bench_vol.json 2> gpurun_out/b_vol.err
# modules cast to half precision, activations in HBM in the module's type (round 6)
for c in c2 c3 c5 c6; do python bench.py --config $c --steps 2 --warmup 1 --no-pmc --half bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_${c}_half.json 2> gpurun_out/b_${c}h.err; done
[ "$2" = "c4full" ] && python bench.py --config c4 --steps 1 --warmup 0 --no-pmc --no-native-line > gpurun_out/${TAG}_bench_c4_full.json 2> gpurun_out/b_c4full.err
head -6 gpurun_out/${TAG}_c2_kernel_stats.txt
for c in c2 c3 c5 c5cfg32 c6 c4_4steps vol c4_full c2_half c3_half c5_half c6_half; do python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${TAG}_bench_$c.json"))
    print("$c", d["value"], d["unit"], d["ms_per_denoise_step"], "ms/denoise step", {k: (v["achieved"], v["frac"]) for k, v in d["roofline_kernels"].items()})
except Exception as e:
    print("$c", "FAILED", e)
PY
done
