#!/bin/bash
# the lines that moved after the r06b collection (moment-bounded scale on ADM's skip projections, volumes on f16x2), on the final tree
mkdir -p gpurun_out
python bench.py > gpurun_out/r06c_bench_c2.json 2> gpurun_out/r06c_c2.err
python bench.py --config c5 --steps 2 --warmup 1 --no-pmc > gpurun_out/r06c_bench_c5.json 2> gpurun_out/r06c_c5.err
python bench.py --config vol --steps 3 --warmup 1 --no-pmc > gpurun_out/r06c_bench_vol.json 2> gpurun_out/r06c_vol.err
python bench.py --config c5cfg32 --steps 1 --warmup 1 --no-pmc --no-cpu-baseline > gpurun_out/r06c_bench_c5cfg32.json 2> gpurun_out/r06c_c5cfg32.err
python bench.py --config c4 --steps 1 --warmup 0 --no-pmc --no-native-line > gpurun_out/r06c_bench_c4_full.json 2> gpurun_out/r06c_c4.err
for c in c2 c5 vol c5cfg32 c4_full; do python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06c_bench_$c.json"))
    print("$c", d["value"], d["unit"], d["ms_per_denoise_step"], "ms/denoise step", {k: (v["launches"], v["ms_per_denoise_step"], v["frac"]) for k, v in d["roofline_kernels"].items()}, {k: d[k]["value"] for k in ("bf16x3_mode", "native_fp32_mfma") if k in d})
except Exception as e:
    print("$c", "FAILED", e)
PY
done
