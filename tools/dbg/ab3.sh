mkdir -p gpurun_out; AB=azula_amd/csrc/_ab
run() { # variant config extra
  if [ $1 = base ]; then unset AZULA_AMD_LIB; else export AZULA_AMD_LIB=$PWD/$AB/libazula_amd_$1.so; fi
  echo -n "$1 $2: "; timeout 600 python bench.py --config $2 $3 --no-pmc --no-cpu-baseline --no-native-line 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_denoise_step'], d['roofline']['avg_us'])"
}
{
for v in base tr_plain; do if [ $v = base ]; then unset AZULA_AMD_LIB; else export AZULA_AMD_LIB=$PWD/$AB/libazula_amd_$v.so; fi; echo "== $v"; timeout 300 python tools/stream_ab.py 2>&1 | grep "transition"; done
for r in 1 2; do for v in base epi_ntst; do run $v c2 "--steps 2 --warmup 1"; done; done
for r in 1 2; do for v in base epi_ntst aff_ntst; do run $v c5 "--steps 2 --warmup 1"; done; done
for v in base aff_ntst epi_ntst base aff_ntst; do run $v c4 "--denoise-steps 4 --steps 1 --warmup 0"; done
} > gpurun_out/s7_stream_ab3.txt 2>&1
cat gpurun_out/s7_stream_ab3.txt
