#!/bin/bash
# hardware counters of the f16x2 kernels beside their bf16x3 forms (one counter pair per pass: tools/pmc_kernel.sh)
R=$PWD
TAG=${1:-r06b}
AZ_WINO=wh2 bash tools/pmc_kernel.sh conv_winograd_x3 gpurun_out/${TAG}_wh2_pmc.txt python $R/tools/conv_micro.py 4 256 256 256 256 3 1 5 > /dev/null 2>&1
AZ_WINO=wx3 bash tools/pmc_kernel.sh conv_winograd_x3 gpurun_out/${TAG}_wx3_pmc.txt python $R/tools/conv_micro.py 4 256 256 256 256 3 1 5 > /dev/null 2>&1
AZ_WINO=h2 AZ_ACT=0 bash tools/pmc_kernel.sh conv_gemm_x3_big gpurun_out/${TAG}_h2gemm_pmc.txt python $R/tools/conv_micro.py 64 256 1 768 3072 1 1 5 > /dev/null 2>&1
AZ_WINO=x3 AZ_ACT=0 bash tools/pmc_kernel.sh conv_gemm_x3_big gpurun_out/${TAG}_x3gemm_pmc.txt python $R/tools/conv_micro.py 64 256 1 768 3072 1 1 5 > /dev/null 2>&1
for f in wh2 wx3 h2gemm x3gemm; do echo "== $f"; cat gpurun_out/${TAG}_${f}_pmc.txt; done
