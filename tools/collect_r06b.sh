#!/bin/bash
# round-6 final collection with f16x2 as the default mode: tools/collect_profiles.sh (all bench lines, kernel-stats tables, PMC traffic)
# + the counters and the power / clock of the f16x2 kernels themselves.
TAG=r06b
bash tools/collect_profiles.sh $TAG c4full > gpurun_out/${TAG}_collect.log 2>&1
AZ_WINO=wh2 bash tools/pmc_kernel.sh conv_winograd_x3 gpurun_out/${TAG}_wh2_pmc.txt python tools/conv_micro.py 4 256 256 256 256 3 1 5 > /dev/null 2>&1
AZ_WINO=wx3 bash tools/pmc_kernel.sh conv_winograd_x3 gpurun_out/${TAG}_wx3_pmc.txt python tools/conv_micro.py 4 256 256 256 256 3 1 5 > /dev/null 2>&1
AZ_WINO=h2 AZ_ACT=0 bash tools/pmc_kernel.sh conv_gemm_x3_big gpurun_out/${TAG}_h2gemm_pmc.txt python tools/conv_micro.py 64 256 1 768 3072 1 1 5 > /dev/null 2>&1
AZ_WINO=x3 AZ_ACT=0 bash tools/pmc_kernel.sh conv_gemm_x3_big gpurun_out/${TAG}_x3gemm_pmc.txt python tools/conv_micro.py 64 256 1 768 3072 1 1 5 > /dev/null 2>&1
python tools/power_probe.py 3 f16x2 > gpurun_out/${TAG}_f16x2_power.txt 2>&1
tail -20 gpurun_out/${TAG}_collect.log
