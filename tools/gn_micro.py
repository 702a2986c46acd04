r"""Micro-benchmark of the GroupNorm kernels: python tools/gn_micro.py B H W C groups"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from azula_amd.engine import Act, Builder

B, H, W, Cc, G = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda")
bld = Builder(dev)
x = Act(torch.randn(B * H * W * Cc, device=dev), B, H, W, Cc, Cc, True)
y = bld.group_norm(x, G, act=int(os.environ.get("ACT", "0")))
names = [n for _, _, n in bld.tape.ops]
for _ in range(3):
    bld.tape.run()
s = torch.cuda.current_stream().cuda_stream
for i, (fn, args, name) in enumerate(bld.tape.ops):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn(*args, s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nbytes = B * H * W * Cc * 4 * (1 if "stats" in name else (2 if "affine" in name else 0))
    print(f"{name:28s} {us:8.1f} us  {nbytes / us / 1e6 if nbytes else 0:7.2f} TB/s")
