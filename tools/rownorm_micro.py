r"""Micro-benchmark of az_rownorm_mod_f32 (layer norm + AdaLN modulation of token rows):  python tools/rownorm_micro.py ROWS C [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from azula_amd.engine import Act, Builder

rows, Cc = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda")
torch.manual_seed(0)
B = 64
bld = Builder(dev)
x = Act(torch.randn(rows * Cc, device=dev), B, rows // B, 1, Cc, Cc, True)
mod = torch.randn(B, 2 * Cc, device=dev)
y = bld.row_norm(x, 0, scale=mod, shift=mod, scale_off=0, shift_off=Cc, bstride=2 * Cc)
for _ in range(5):
    bld.tape.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    bld.tape.run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"rownorm {rows} x {Cc}: {ms * 1e3:.1f} us  {rows * Cc * 8 / ms / 1e9:.2f} TB/s")
