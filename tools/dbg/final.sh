set -x
R=$PWD
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/final_pytest.txt
python bench.py > gpurun_out/r02_bench_c2.json 2> gpurun_out/b_c2.err
for c in c2 c3; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o run -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-pmc --no-cpu-baseline > $R/gpurun_out/prof_$c.json 2> $R/gpurun_out/prof_$c.err)
  db=$(find gpurun_out/prof_$c -name "*.db" | head -1)
  python profiles/summarize.py $db 40 > gpurun_out/r02_${c}_kernel_stats.txt
  rm -f $db
done
cp profiles/r02_traffic_c2.json gpurun_out/r02_traffic_c2_prev.json
python bench.py --config c3 --steps 3 --warmup 1 > gpurun_out/r02_bench_c3.json 2> gpurun_out/b_c3.err
python bench.py --config c5 --steps 2 --warmup 1 --no-pmc > gpurun_out/r02_bench_c5.json 2> gpurun_out/b_c5.err
python bench.py --config c5cfg32 --steps 1 --warmup 1 --no-pmc > gpurun_out/r02_bench_c5cfg32.json 2> gpurun_out/b_c5cfg32.err
python bench.py --config c6 --steps 2 --warmup 1 --no-pmc > gpurun_out/r02_bench_c6.json 2> gpurun_out/b_c6.err
python bench.py --config c4 --denoise-steps 4 --steps 1 --warmup 0 --no-pmc > gpurun_out/r02_bench_c4_4steps.json 2> gpurun_out/b_c4.err
cat gpurun_out/final_pytest.txt
head -8 gpurun_out/r02_c2_kernel_stats.txt
