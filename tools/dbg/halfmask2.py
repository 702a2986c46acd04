import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from azula_amd.engine import Act, Builder
torch.set_grad_enabled(False)
dev = torch.device("cuda")
B, H, T, D = 2, 4, 9, 16
torch.manual_seed(0)
qkv = torch.randn(B * T * 3 * H * D, device=dev)
mask = torch.ones(T, T, dtype=torch.bool).tril().cuda()
outs = {}
for half in (None, torch.bfloat16):
    for m in (None, mask):
        bld = Builder(dev, half=half)
        out = bld.attention(Act(qkv, B, T, 1, 3 * H * D, 3 * H * D, True), H, "nHC", True, 1.0 / math.sqrt(D), mask=m)
        desc = [k for k in bld.tape.keep if hasattr(k, "mask")][-1]
        print(half, m is not None, bld.tape.ops[-1][2], "a.mask =", desc.mask)
        bld.tape.run()
        outs[(half, m is not None)] = out.buf.clone()
print("fp32 mask vs nomask", (outs[(None, True)] - outs[(None, False)]).abs().max().item())
print("bf16 mask vs nomask", (outs[(torch.bfloat16, True)] - outs[(torch.bfloat16, False)]).abs().max().item())
print("bf16 mask vs fp32 mask", (outs[(torch.bfloat16, True)] - outs[(None, True)]).abs().max().item())
