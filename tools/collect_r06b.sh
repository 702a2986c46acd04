#!/bin/bash
# round-6 final collection with f16x2 as the default mode: tools/collect_profiles.sh (all bench lines, kernel-stats tables, PMC traffic)
# + the counters and the power / clock of the f16x2 kernels themselves.
TAG=r06b
bash tools/collect_profiles.sh $TAG c4full > gpurun_out/${TAG}_collect.log 2>&1




bash tools/pmc_f16x2.sh $TAG > /dev/null 2>&1
python tools/power_probe.py 3 f16x2 > gpurun_out/${TAG}_f16x2_power.txt 2>&1
tail -20 gpurun_out/${TAG}_collect.log
