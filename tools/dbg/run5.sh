cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/full_tests.txt
bash tools/pmc_kernel.sh attention_x3 gpurun_out/att_pmc.txt python $PWD/tools/attn_micro.py 64 12 256 64 20 > /dev/null 2>&1
for shp in "64 12 256 64" "32 12 288 64" "4 8 1024 64"; do
  echo -n "bf16 head: "; AZULA_AMD_LIB=azula_amd/csrc/_ab/libazula_amd_head.so AZ_ATTN_HALF=bf16 python tools/attn_micro.py $shp 100 2>&1 | grep -v amdgpu.ids
  echo -n "bf16 new:  "; AZ_ATTN_HALF=bf16 python tools/attn_micro.py $shp 100 2>&1 | grep -v amdgpu.ids
done > gpurun_out/att_half_ab.txt 2>&1
