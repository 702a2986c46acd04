r"""Does an HBM-bound pass hide under a matrix-bound convolution on this chip?  One Winograd layer and one GroupNorm-apply + SiLU
pass (ADM shapes, batch 32), back to back on one stream vs concurrently on two:  python tools/overlap_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from azula_amd.engine import Act, Builder

dev = torch.device("cuda")
torch.manual_seed(0)
B, H, W, Cc = 32, 128, 128, 256
bc = Builder(dev)
x = Act(torch.randn(B * H * W * Cc, device=dev), B, H, W, Cc, Cc, True)
w = torch.randn(Cc, Cc, 3, 3, device=dev) / (Cc * 9) ** 0.5
yc = bc.conv(x, bc.pack_conv(w, torch.randn(Cc, device=dev)), Cc, winograd=True)
bc.finish()
bp = Builder(dev)
x2 = Act(torch.randn(B * H * W * Cc, device=dev), B, H, W, Cc, Cc, True)
yp = bp.group_norm(x2, 32, weight=torch.ones(Cc, device=dev), bias=torch.zeros(Cc, device=dev), act=1)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 20


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N


def seq():
    for _ in range(N):
        bc.tape.run()
        bp.tape.run()


def conc():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    for _ in range(N):
        bc.tape.run(s1.cuda_stream)
        bp.tape.run(s2.cuda_stream)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


for _ in range(2):
    seq(); conc()
tc = timed(lambda: [bc.tape.run() for _ in range(N)])
tp = timed(lambda: [bp.tape.run() for _ in range(N)])
print(f"conv alone {tc * 1e3:.1f} us, pass alone {tp * 1e3:.1f} us, back to back {timed(seq) * 1e3:.1f} us, two streams {timed(conc) * 1e3:.1f} us")
