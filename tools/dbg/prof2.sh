R=$PWD
export TMPDIR=/tmp
for c in c5 c6; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$c -o run -- python $R/bench.py --config $c --steps 2 --warmup 1 --no-pmc > $R/gpurun_out/prof_$c.json 2> $R/gpurun_out/prof_$c.err)
  db=$(find gpurun_out/prof_$c -name "*.db" | head -1)
  python profiles/summarize.py $db 30 > gpurun_out/r02_${c}_kernel_stats.txt
  rm -f $db
done
head -12 gpurun_out/r02_c5_kernel_stats.txt | cut -c1-150
