import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
torch.set_grad_enabled(False)
dev = torch.device("cuda")

def poison(gb=8):
    t = torch.full((gb << 28,), float("nan"), device=dev)  # gb GiB of NaN
    del t  # back to the caching allocator: the next allocations are carved out of it

which = sys.argv[1]
cfg = dict(bench.CONFIGS[which])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
den = bench.build_denoiser(cfg, dev)
poison(int(sys.argv[3]) if len(sys.argv) > 3 else 8)
from azula_amd.sample import DDIMSampler
smp = DDIMSampler(den, steps=2, silent=True)
torch.manual_seed(1)
x1 = smp.init((B, *cfg["shape"]), device=dev)
kw = bench.sampler_kwargs(dict(cfg, batch=B), dev)
# eager denoiser call first
plain = getattr(den, "denoiser", den)
if which.startswith("c5") or which == "c4":
    out = plain.backbone(x1, torch.full((B,), 500, device=dev), y=kw.get("label", kw.get("positive", {}).get("label")))
    print(which, "backbone NaNs:", torch.isnan(out).sum().item(), "of", out.numel())
poison(2)
x0 = smp(x1, **kw)
print(which, "sampler NaNs:", torch.isnan(x0).sum().item(), "of", x0.numel())
