r"""Micro-benchmark of az_conv2d_f32 on one shape (for rocprofv3 PMC passes and A/B tuning).

    python tools/conv_micro.py B H W Cin Cout [ks] [stride] [reps]
"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from azula_amd.engine import Act, Builder

B, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
ks = int(sys.argv[6]) if len(sys.argv) > 6 else 3
stride = int(sys.argv[7]) if len(sys.argv) > 7 else 1
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 20
dev = torch.device("cuda")
torch.manual_seed(0)
bld = Builder(dev)
scale = 0.0 if os.environ.get("AZ_ZERO") else 1.0
x = Act(torch.randn(B * H * W * Cin, device=dev) * scale, B, H, W, Cin, Cin, True)
w = torch.randn(Cout, Cin, ks, ks, device=dev) / (Cin * ks * ks) ** 0.5 * scale
b = torch.randn(Cout, device=dev)
if os.environ.get("AZ_AFFINE"):  # the input carries a pending normalisation (AzConvArgs.in_affine): AZ_AFFINE=1 plain, 2 with SiLU
    x.affine = (torch.randn(2 * B * Cin, device=dev) * scale, int(os.environ["AZ_AFFINE"]) - 1)
wino = {"1": True, "0": False, "4": 4, "x3": "x3", "wx3": "wx3", "h2": "h2", "wh2": "wh2"}.get(os.environ.get("AZ_WINO", ""), None)
y = bld.conv(x, bld.pack_conv(w, b), Cout, stride=stride, act=int(os.environ.get("AZ_ACT", "1")), winograd=wino, gn_stats=bool(os.environ.get("AZ_GN")))
if os.environ.get("AZ_SPLITK"):  # override the suggested split-K (A/B)
    _d = [k for k in bld.tape.keep if hasattr(k, "_flops")][-1]
    _d.splitk = int(os.environ["AZ_SPLITK"])
    bld._ws_need = max(bld._ws_need, _d.splitk * B * ((H + stride - 1) // stride) * ((W + stride - 1) // stride) * _d.cout_s)
    if _d not in bld._ws_users:
        bld._ws_users.append(_d)
bld.finish()
desc = bld.tape.keep[-1] if hasattr(bld.tape.keep[-1], "_flops") else [k for k in bld.tape.keep if hasattr(k, "_flops")][-1]
for _ in range(3):
    bld.tape.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    bld.tape.run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
if len(bld.tape.ops) > 1:  # per-op times (e.g. the split pass of AZ_X3_PLANES=1, split-K combine is inside its entry)
    import ctypes
    from azula_amd import _lib
    st = torch.cuda.current_stream().cuda_stream
    per = []
    for fn, args, name in bld.tape.ops:
        e0.record()
        for _ in range(reps):
            fn(*args, st)
        e1.record()
        torch.cuda.synchronize()
        per.append(f"{name} {e0.elapsed_time(e1) / reps * 1e3:.1f} us")
    print("   ", "; ".join(per))
print(f"conv[{desc._algo[10:-4]}] {B}x{H}x{W} {Cin}->{Cout} k{ks} s{stride} splitk={desc.splitk}: {ms * 1e3:.1f} us  {desc._flops / ms / 1e9:.1f} TF/s")
