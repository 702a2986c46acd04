r"""Micro-benchmark of az_attention_f32 (fused qkv layout "nHC", q/k RMS norm on):  python tools/attn_micro.py B H T D [reps]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from azula_amd.engine import Act, Builder

B, H, T, D = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 50
dev = torch.device("cuda")
torch.manual_seed(0)
qkv = torch.randn(B * T * 3 * H * D, device=dev)
half = {"bf16": torch.bfloat16, "f16": torch.float16}.get(os.environ.get("AZ_ATTN_HALF", ""))  # the half-precision-operand kernel
bld = Builder(dev, half=half)
# AZ_ATTN_OPTS: comma list of  norms (q/k RMS norm, default on), gains (learned q/k gains), rope (rotary tables), order=3HC
opts = set(os.environ.get("AZ_ATTN_OPTS", "norms").split(","))
rope = (torch.randn(T, H * D // 2, device=dev), torch.randn(T, H * D // 2, device=dev)) if "rope" in opts else None
gains = (torch.rand(D, device=dev) + 0.5, torch.rand(D, device=dev) + 0.5) if "gains" in opts else None
qkv_act = Act(qkv, B, T, 1, 3 * H * D, 3 * H * D, True)
qkv_act.bounded = True  # (a projection of normalised tokens: what every attention layer of the backbones reads -- the f16x2 form in the default mode)
out = bld.attention(qkv_act, H, "3HC" if "order=3HC" in opts else "nHC", "norms" in opts,
                    1.0 / math.sqrt(D), rope=rope, qk_weight=gains)
for _ in range(5):
    bld.tape.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    bld.tape.run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
tag = ",".join(sorted(opts))
print(f"{bld.tape.ops[-1][2][3:-4]} {B}x{H}x{T}x{D} [{tag}]: {ms * 1e3:.1f} us  {4 * B * H * T * T * D / ms / 1e9:.1f} TF/s")
