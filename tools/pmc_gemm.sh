# Hardware counters of conv_gemm_x3_big_kernel on the DiT-B MLP projection (16384 tokens, 768 -> 3072), one counter pair per pass:
#   bash tools/pmc_gemm.sh  ->  gpurun_out/pmc_gemm.txt
export TMPDIR=/tmp
R=$PWD
out=$R/gpurun_out/pmc_gemm.txt; : > $out
for pair in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_WAIT_INST_ANY"; do
  d=/tmp/pmc_$$; rm -rf $d
  (cd /tmp && AZ_WINO=x3 AZ_ACT=0 rocprofv3 --pmc $pair --kernel-trace -d $d -o run --output-format csv -- python $R/tools/conv_micro.py 64 256 1 768 3072 1 1 5 > /dev/null 2>&1)
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python - "$f" $pair >> $out <<'PY'
import csv, sys, collections
f, names = sys.argv[1], sys.argv[2:]
rows = list(csv.DictReader(open(f)))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in rows:
    if "conv_gemm_x3_big_kernel" not in r["Kernel_Name"]: continue
    acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]][r["Dispatch_Id"]] += 1
for n in names:
    d = acc.get(n, {})
    if not d: print(f"{n:28s} (no rows)"); continue
    vals = sorted(d.values()); k = sorted(d)[len(d) // 2]
    print(f"{n:28s} per launch (sum of its {cnt[n][k]} rows) median {vals[len(vals)//2]:16.0f}   launches {len(vals)}")
PY
  rm -rf $d
done
cat $out
