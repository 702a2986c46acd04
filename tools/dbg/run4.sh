cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
{
for r in 1 2; do
for shp in "64 12 256 64" "32 12 288 64" "4 8 1024 64"; do
  for v in main att_noexp att_nopsplit att_nokvsplit att_novalu att_1mfma att_all; do
    lib=azula_amd/csrc/_ab/libazula_amd_$v.so; [ "$v" = main ] && lib=azula_amd/csrc/libazula_amd.so
    echo -n "$v: "; AZULA_AMD_LIB=$lib python tools/attn_micro.py $shp 100 2>&1 | grep -v amdgpu.ids
  done
done; done
} > gpurun_out/att_ablate.txt 2>&1
bash tools/pmc_kernel.sh attention_x3 gpurun_out/att_pmc.txt python tools/attn_micro.py 64 12 256 64 20 > /dev/null 2>&1
