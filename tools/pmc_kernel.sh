#!/bin/bash
# Hardware counters of one kernel, one counter pair per pass (--pmc with --kernel-trace only):
#   bash tools/pmc_kernel.sh <kernel name substring> <output file> <command ...>
# e.g. AZ_WINO=wx3 bash tools/pmc_kernel.sh conv_winograd_x3 gpurun_out/pmc_wx3.txt python tools/conv_micro.py 4 256 256 256 256 3 1 5
export TMPDIR=/tmp
R=$PWD
kern=$1; out=$R/$2; shift 2
: > $out
for pair in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
            "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
            "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  d=/tmp/pmc_$$; rm -rf $d
  (cd /tmp && rocprofv3 --pmc $pair --kernel-trace -d $d -o run --output-format csv -- "$@" > /dev/null 2>&1)
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" "$kern" $pair >> $out <<'PY'
import csv, sys, collections
f, kern, names = sys.argv[1], sys.argv[2], sys.argv[3:]
try:
    rows = list(csv.DictReader(open(f)))
except Exception as e:
    print(f"{' '.join(names)}: no counter file ({e!r})"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if kern not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
for n in names:
    d = acc.get(n, {})
    if not d: print(f"{n:30s} (no rows)"); continue
    vals = sorted(d.values())
    print(f"{n:30s} per launch median {vals[len(vals)//2]:16.0f}   launches {len(vals)}")
PY
  rm -rf $d
done
cat $out
