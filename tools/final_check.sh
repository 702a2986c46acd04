#!/bin/bash
# end-of-round check on the final tree: the whole GPU suite (default mode), smoke(), the default bench.py with its wall time
set -u
OUT=gpurun_out/final_check.txt
mkdir -p gpurun_out
: > $OUT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee -a $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT
SECONDS=0
python bench.py > gpurun_out/final_bench_c2.json 2> gpurun_out/final_bench_c2.err
echo "default bench.py: ${SECONDS} s wall" | tee -a $OUT
python - <<'PY' | tee -a $OUT
import json
d = json.load(open("gpurun_out/final_bench_c2.json"))
r = d["roofline"]
print(d["value"], d["unit"], d["ms_per_denoise_step"], "ms per denoise step;", "roofline", r["entry"], r["achieved"], r["peak"], r["frac"], "traffic", r.get("traffic"))
print({k: d[k]["value"] for k in ("bf16x3_mode", "native_fp32_mfma") if k in d}, "cpu", d.get("cpu_baseline", {}).get("value"), d.get("speedup_vs_cpu"))
print("tolerance:", d["tolerance"].get("measured_max_abs_err"), d["tolerance"]["mode"][:80])
PY
