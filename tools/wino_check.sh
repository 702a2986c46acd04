#!/bin/bash
for asm in 0 1; do
for c in "1 8 8 8 64" "1 16 16 16 64" "2 16 16 32 64" "1 16 16 64 128" "1 32 32 256 256" "1 8 8 8 64 2" "2 9 7 40 24" "1 16 16 24 64 3" "4 64 64 512 512"; do
  AZ_DEBUG_AB=1 AZ_WINOGRAD_ASM=$asm timeout 120 python tools/wino_check.py $c 2>&1 | grep -v amdgpu.ids | tail -4
done
done
