set -x
R=$PWD
export TMPDIR=/tmp
./tools/dbg/mfma_valu.bin > gpurun_out/mfma_valu.txt 2>&1
./tools/dbg/mfma_valu2.bin >> gpurun_out/mfma_valu.txt 2>&1
for c in c2 c3; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o run -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > $R/gpurun_out/prof_$c.json 2> $R/gpurun_out/prof_$c.err)
  db=$(find gpurun_out/prof_$c -name "*.db" | head -1)
  python profiles/summarize.py $db 40 > gpurun_out/r02_${c}_kernel_stats.txt
  rm -f $db
  python tools/pmc_traffic.py collect --config $c --out gpurun_out/r02_traffic_$c.json > /dev/null 2> gpurun_out/pmc_$c.err
done
python bench.py --config c3 --steps 3 --warmup 1 > gpurun_out/r02_bench_c3.json 2> gpurun_out/b_c3.err
python bench.py --config c5 --steps 2 --warmup 1 --no-pmc > gpurun_out/r02_bench_c5.json 2> gpurun_out/b_c5.err
python bench.py --config c5cfg32 --steps 1 --warmup 1 --no-pmc > gpurun_out/r02_bench_c5cfg32.json 2> gpurun_out/b_c5cfg32.err
python bench.py --config c6 --steps 2 --warmup 1 --no-pmc > gpurun_out/r02_bench_c6.json 2> gpurun_out/b_c6.err
head -8 gpurun_out/r02_c2_kernel_stats.txt
