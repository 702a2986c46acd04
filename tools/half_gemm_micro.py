r"""Micro-benchmark of the half-precision GEMM / convolution kernels with half activations in HBM (typed launches).

    python tools/half_gemm_micro.py B H W Cin Cout [ks] [stride] [reps]      (AZ_HALF=f16: IEEE half; AZ_TYPED=0: fp32 activations)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from azula_amd.engine import Builder

B, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
ks = int(sys.argv[6]) if len(sys.argv) > 6 else 1
stride = int(sys.argv[7]) if len(sys.argv) > 7 else 1
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
dev = torch.device("cuda")
torch.manual_seed(0)
half = torch.float16 if os.environ.get("AZ_HALF") == "f16" else torch.bfloat16
bld = Builder(dev, half=half, half_act=os.environ.get("AZ_TYPED", "1") != "0")
x = bld.new_act(B, H, W, Cin, pinned=True)
x.buf.copy_(torch.randn(x.buf.numel(), device=dev))
w = torch.randn(Cout, Cin, ks, ks, device=dev) / (Cin * ks * ks) ** 0.5
y = bld.conv(x, bld.pack_conv(w, torch.randn(Cout, device=dev)), Cout, stride=stride, act=int(os.environ.get("AZ_ACT", "0")))
bld.finish()
desc = [k for k in bld.tape.keep if hasattr(k, "_flops")][-1]
for _ in range(3):
    bld.tape.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    bld.tape.run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"half conv[{desc._algo[10:]}] typed={int(bld.half_act)} {B}x{H}x{W} {Cin}->{Cout} k{ks} s{stride} splitk={desc.splitk}: {ms * 1e3:.1f} us  "
      f"{desc._flops / ms / 1e9:.1f} TF/s = {desc._flops / ms / 1e9 / 2516.8:.3f} of the bf16 peak")
