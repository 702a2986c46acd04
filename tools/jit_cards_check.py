r"""Builds the larger JiT cards (random init) and runs a short fused DDIM sampling on the GPU: a functional check that
JiT-L (head_dim 64) and JiT-H (head_dim 80) compile and run at full size.   python tools/jit_cards_check.py"""
import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from azula_amd.plugins import jit
from azula_amd.sample import DDIMSampler
from bench import rerandomise_zero_tensors
torch.set_grad_enabled(False)
for name in ("JiT-L/16", "JiT-H/32", "JiT-H/16"):
    torch.manual_seed(0)
    den = jit.make_model(name)
    rerandomise_zero_tensors(den.backbone)
    den = den.cuda().eval()
    smp = DDIMSampler(den, steps=10, silent=True)
    x1 = smp.init((8, 3, 256, 256), device="cuda")
    lab = torch.arange(8, device="cuda")
    x0 = smp(x1, label=lab); torch.cuda.synchronize()
    t0 = time.perf_counter(); x0 = smp(x1, label=lab); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    n = sum(p.numel() for p in den.parameters())
    print(f"{name}: {n/1e6:.0f}M params, batch 8, DDIM-10: {dt/10*1e3:.1f} ms/step, finite={bool(torch.isfinite(x0).all())}, |x0|max={x0.abs().max().item():.2f}")
    del den, smp; torch.cuda.empty_cache()
