import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
torch.set_grad_enabled(False)
dev = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "c5cfg32"
cfg = dict(bench.CONFIGS[which])
den = bench.build_denoiser(cfg, dev)
plain = getattr(den, "denoiser", den)
bb = plain.backbone
torch.manual_seed(1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
x = torch.randn(N, 3, 256, 256, device=dev)
idx = torch.full((N,), 500, device=dev)
y = (torch.arange(N, device=dev) * 31) % 1000 if bb.num_classes is not None else None
def run(sl):
    return bb(x[sl], idx[sl], y=None if y is None else y[sl])
def md(a, b):
    return [f"{v:.1e}" for v in (a - b).abs().flatten(1).max(1).values.tolist()]
one = {i: run(slice(i, i + 1)) for i in (0, 1, N // 2 - 1, N // 2, N - 1)}
full = run(slice(0, N))
full2 = run(slice(0, N))
print("full deterministic:", torch.equal(full, full2))
for i, o in one.items():
    print("full vs batch-1, sample", i, md(full[i:i+1], o))
h = N // 2
pa = run(slice(0, h)); pa2 = run(slice(0, h)); pb = run(slice(h, N))
print("half-plan first-run == second-run:", torch.equal(pa, pa2))
for i, o in one.items():
    src = pa if i < h else pb
    print("half vs batch-1, sample", i, md(src[i % h:i % h + 1], o))
