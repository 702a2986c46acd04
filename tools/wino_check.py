r"""Debug aid: az_conv2d_winograd_f32 against F.conv2d on one shape (one process per case: a memory fault must not hide the rest).
    python tools/wino_check.py B H W Cin Cout [splitk]        (AZ_WINOGRAD_ASM=0: the C++ loop)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from azula_amd.engine import Act, Builder

B, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
splitk = int(sys.argv[6]) if len(sys.argv) > 6 else 0
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Cin, H, W, generator=g)
w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
b = torch.randn(Cout, generator=g)
ref = F.conv2d(x, w, b, padding=1)
dev = torch.device("cuda")
cs = (Cin + 3) // 4 * 4
xn = torch.zeros(B, H, W, cs)
xn[..., :Cin] = x.permute(0, 2, 3, 1)
bld = Builder(dev)
xa = Act(xn.to(dev).reshape(-1), B, H, W, Cin, cs, True)
y = bld.conv(xa, bld.pack_conv(w.to(dev), b.to(dev)), Cout, winograd=True)
if splitk:
    a = bld.tape.keep[-1]
    a.splitk = splitk
    bld._ws_need = max(bld._ws_need, splitk * B * y.H * y.W * y.cs)
    bld._ws_users.append(a)
bld.finish()
bld.tape.run()
torch.cuda.synchronize()
out = y.buf.reshape(B, H, W, y.cs)[..., :Cout].permute(0, 3, 1, 2).cpu()
err = (out - ref).abs()
print(f"{B}x{H}x{W} {Cin}->{Cout} splitk={splitk} asm={os.environ.get('AZ_WINOGRAD_ASM', '1')}: max|d| {err.max().item():.3e}  (scale {ref.abs().max().item():.2f})"
      f"  worst at {tuple(int(v) for v in (err == err.max()).nonzero()[0])}")
if err.max() > 1e-3:
    bad = (err > 1e-3)
    print("   bad fraction", bad.float().mean().item(), "per cout (first 8 blocks of 8):", [round(bad[:, c:c + 8].float().mean().item(), 2) for c in range(0, min(Cout, 64), 8)],
          "per row:", [round(bad[:, :, r].float().mean().item(), 2) for r in range(min(H, 8))], "per col:", [round(bad[:, :, :, c].float().mean().item(), 2) for c in range(min(W, 8))])
