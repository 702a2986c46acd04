r"""Phase timeline of az_conv2d_winograd_x3_f32 (experiment library: `python tools/ablate.py wx3_tl`):

    AZULA_AMD_LIB=azula_amd/csrc/_ab/libazula_amd_wx3_tl.so python tools/wx3_timeline.py B H W Cin Cout

Per workgroup (waves 0 and 4 of the first 512): s_memtime cycles summed over the K walk for the four sections of each phase --
slots 0-11 | slots 12-23 | last filter loads + wait for the LDS-DMA | barrier + the next phase's first fragment reads -- plus
the cycles from the first stamp (before the prologue's staging) to the end of the K loop and from there to the end of the stores."""
import ctypes as C
import os
import statistics as st
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from azula_amd import _lib
from azula_amd.engine import Act, Builder

B, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda")
torch.manual_seed(0)
bld = Builder(dev)
x = Act(torch.randn(B * H * W * Cin, device=dev), B, H, W, Cin, Cin, True)
w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
y = bld.conv(x, bld.pack_conv(w, torch.randn(Cout, device=dev)), Cout, act=1, winograd="wx3")
bld.finish()
for _ in range(3):
    bld.tape.run()
torch.cuda.synchronize()
nwg = ((B * ((H + 1) // 2) * ((W + 1) // 2) + 63) // 64) * ((Cout + 63) // 64)
n = min(nwg, 512)
buf = (C.c_uint * (n * 24))()
fn = _lib.lib().az_debug_wx3_timeline
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, n * 24) == 0
steps = (Cin + 15) // 16
names = ["ph0 slots 0-11", "ph0 slots 12-23", "ph0 loads+dma wait", "ph0 barrier+frags", "ph1 slots 0-11", "ph1 slots 12-23", "ph1 loads", "ph1 barrier+frags"]
for role in (0, 1):
    rows = [[buf[(i * 2 + role) * 12 + k] for k in range(12)] for i in range(n)]
    print(f"wave {4 * role}: medians over {n} workgroups, {steps} steps")
    tot = 0
    for k, nm in enumerate(names):
        v = st.median(r[k] for r in rows) / steps
        tot += v
        print(f"   {nm:22s} {v:8.0f} cycles per step")
    print(f"   {'sum':22s} {tot:8.0f} cycles per step;  entry -> end of K loop {st.median(r[8] for r in rows):9.0f};  epilogue {st.median(r[9] for r in rows):8.0f}")
