r"""Headline benchmark: images/s of DDIMSampler(steps=64) on the ADM-shaped 256x256 UNet
(BASELINE.json configs[1]), one process per GPU, batch sharded (weak scaling, 4 images/GPU).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" of this contract is one pass of the hot path over one batch: a full 64-step DDIM
sampling of the per-GPU batch (64 hipGraph replays), followed for N > 1 by the RCCL all-gather
of x0.  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3  # MI355X fp32 MFMA == fp32 VALU peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0  # HBM3E spec (6.3 TB/s measured achievable)

CONFIGS = {
    # BASELINE.json configs[1]: azula.nn.unet ADM-shaped UNet, 3x256x256, DDIM-64, batch 4 per GPU
    "c2": dict(
        kind="unet", batch=4, shape=(3, 256, 256), steps=64,
        net=dict(in_channels=3, out_channels=3, hid_channels=(256, 256, 512, 512, 1024, 1024),
                 hid_blocks=(2, 2, 2, 2, 2, 2), norm="group", groups=32, mod_features=1024),
        name="azula.nn.unet ADM-shaped UNet 3x256x256 (320.5M params), KarrasDenoiser+VPSchedule, DDIMSampler(steps=64, eta=0)",
    ),
    # BASELINE.json configs[2]: azula.nn.vit ViT as DiT-B/2 (4x32x32 latent), DDIM-50, batch 64 per GPU
    "c3": dict(
        kind="vit", batch=64, shape=(4, 32, 32), steps=50,
        net=dict(in_channels=4, out_channels=4, hid_channels=768, hid_blocks=12, attention_heads=12, patch_size=2,
                 mod_features=768),
        name="azula.nn.vit ViT DiT-B/2 4x32x32 (115M params), KarrasDenoiser+VPSchedule, DDIMSampler(steps=50, eta=0)",
    ),
    # BASELINE.json configs[4]: ADM imagenet_256x256 architecture (random init, zero-init layers re-randomised),
    # DDIM-64; "guidance" adds the second (negative) backbone evaluation of CFG on the class-conditional card
    "c5": dict(
        kind="adm", card="imagenet_256x256", batch=4, shape=(3, 256, 256), steps=64,
        name="azula.plugins.adm imagenet_256x256 UNetModel (552.8M params, random init), AblatedDenoiser, DDIMSampler(steps=64)",
    ),
    # BASELINE.json configs[3]: ADM 256x256, DDPMSampler(steps=1000), batch 256 sharded 8-way = 32 per GPU
    "c4": dict(
        kind="adm", card="imagenet_256x256", batch=32, shape=(3, 256, 256), steps=1000, sampler="ddpm",
        name="azula.plugins.adm imagenet_256x256 UNetModel (random init), DDPMSampler(steps=1000), 32 images per GPU",
    ),
    "c5cfg": dict(
        kind="adm", card="imagenet_256x256_cond", batch=4, shape=(3, 256, 256), steps=64, cfg=2.0,
        name="azula.plugins.adm imagenet_256x256_cond (random init) + CFGDenoiser(g=2), DDIMSampler(steps=64)",
    ),
    # BASELINE.json configs[4] at its stated batch: 32 images on one GPU, two backbone evaluations per step
    "c5cfg32": dict(
        kind="adm", card="imagenet_256x256_cond", batch=32, shape=(3, 256, 256), steps=64, cfg=2.0,
        name="azula.plugins.adm imagenet_256x256_cond (random init) + CFGDenoiser(g=2), DDIMSampler(steps=64), batch 32",
    ),
    # JiT-B/16 pixel-space transformer at 256x256 (131M params, 256 + 32 tokens), class labels
    "c6": dict(
        kind="jit", model="JiT-B/16", batch=32, shape=(3, 256, 256), steps=50, labels=True,
        name="azula.plugins.jit JiT-B/16 3x256x256 (random init), JITDenoiser+RectifiedSchedule, DDIMSampler(steps=50)",
    ),
    # small variant for quick functional checks of the harness
    "tiny": dict(
        kind="unet", batch=2, shape=(3, 64, 64), steps=8,
        net=dict(in_channels=3, out_channels=3, hid_channels=(32, 64), hid_blocks=(1, 1), norm="group", groups=8,
                 mod_features=64),
        name="tiny UNet 3x64x64 (harness check only)",
    ),
}


def rerandomise_zero_tensors(module, seed=123):
    r"""every all-zero weight tensor with ndim > 1 (ADM zero_module layers) is refilled
    N(0, 1/fan_in) from a fixed generator, otherwise random-init ADM outputs exactly 0."""
    import math

    g = torch.Generator().manual_seed(seed)
    for _, v in sorted(module.state_dict().items()):
        if torch.is_floating_point(v) and v.ndim > 1 and not torch.any(v != 0):
            v.copy_(torch.randn(v.shape, generator=g) / math.sqrt(v[0].numel()))


def build_denoiser(cfg, device):
    from azula_amd.denoise import KarrasDenoiser
    from azula_amd.nn import TimeModulated, UNet, ViT
    from azula_amd.noise import VPSchedule

    torch.manual_seed(0)  # weights = module default init under seed 0
    if cfg["kind"] == "adm":
        from azula_amd.guidance import CFGDenoiser
        from azula_amd.plugins import adm

        den = adm.make_model(**adm.load_cards(adm)[cfg["card"]].config)
        rerandomise_zero_tensors(den.backbone)
        den = den.to(device).eval()
        return CFGDenoiser(den) if cfg.get("cfg") else den
    if cfg["kind"] == "jit":
        from azula_amd.plugins import jit

        den = jit.make_model(cfg["model"], input_size=cfg["shape"][-1])
        rerandomise_zero_tensors(den.backbone)
        return den.to(device).eval()
    net = UNet(**cfg["net"]) if cfg["kind"] == "unet" else ViT(**cfg["net"])
    wrapped = TimeModulated(net, cfg["net"]["mod_features"], name=cfg["kind"])
    return KarrasDenoiser(wrapped, VPSchedule()).to(device).eval()


CONV_OPS = ("az_conv2d_f32", "az_conv2d_winograd_f32", "az_conv2d_winograd4_f32", "az_conv2d_bf16_f32", "az_conv2d_f16_f32",
            "az_conv2d_x3_f32")


def conv_roofline(sampler, device):
    r"""Per-launch HIP-event timing of the dominant kernel (conv_igemm, fp32 MFMA) over one
    backbone forward run eagerly on the launch stream; achieved = sum(flops) / sum(time)."""
    loop = next(iter(sampler._fused_cache.values()))
    tape = loop.tape
    stream = torch.cuda.current_stream(device)
    sptr = stream.cuda_stream
    recs = []
    loop.counter.zero_()
    for rep in range(2):  # first repetition warms caches / clocks
        recs = []
        loop.counter.zero_()
        for fn, args, name in tape.ops:
            if name in CONV_OPS:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                rc = fn(*args, sptr)
                e1.record(stream)
                desc = args[0]._obj
                recs.append((e0, e1, desc._flops, desc.splitk))
            else:
                rc = fn(*args, sptr)
            assert rc == 0, (name, rc)
        torch.cuda.synchronize(device)
    convs = [a[0]._obj for _, a, n in tape.ops if n in CONV_OPS]
    if os.environ.get("AZ_BENCH_DETAIL"):
        for (e0, e1, fl, sk), d in zip(recs, convs):
            t = e0.elapsed_time(e1)
            print(
                f"conv {d.batch}x{d.hin}x{d.win} cin={d.c0s}+{d.c1s} cout={d.cout_s} k={d.ksize} s={d.stride} "
                f"splitk={sk} {d._algo[10:-4]}: {t * 1e3:8.1f} us {fl / t / 1e9:7.1f} TF/s",
                file=sys.stderr,
            )
    out = {}
    for algo in ("az_conv2d_winograd_f32", "az_conv2d_f32"):
        same = (algo,) if algo != "az_conv2d_f32" else ("az_conv2d_f32", "az_conv2d_bf16_f32", "az_conv2d_f16_f32", "az_conv2d_x3_f32")
        sel = [(r, d) for r, d in zip(recs, convs) if d._algo in same]
        out[algo] = dict(
            flops=sum(r[2] for r, _ in sel), ms=sum(r[0].elapsed_time(r[1]) for r, _ in sel), launches=len(sel)
        )
    out["all"] = dict(flops=sum(r[2] for r in recs), ms=sum(r[0].elapsed_time(r[1]) for r in recs), launches=len(recs))
    return out


def transition_roofline(device, n=1 << 26):
    r"""K1 at a size that defeats the 256 MiB Infinity Cache (64 Mi elements = 256 MiB per tensor):
    DDIM eta=0 form, 12 B/element algorithmic (read x_t, read F, write x_s)."""
    import ctypes as C
    from azula_amd import _lib

    x = torch.randn(n, device=device)
    F = torch.randn(n, device=device)
    out = torch.empty(n, device=device)
    row = torch.zeros(16, device=device)
    row[1], row[2], row[4], row[5], row[6] = 0.4, 0.6, 0.5, 0.7, 0.9
    row[9], row[10] = -float("inf"), float("inf")
    a = _lib.AzTransitionArgs(x_t=x.data_ptr(), F=F.data_ptr(), x_s=out.data_ptr(), batch=1, channels=1, inner=n,
                              f_channels=1, coef=row.data_ptr())
    stream = torch.cuda.current_stream(device)
    for _ in range(3):
        _lib.call("az_transition_f32", C.byref(a), stream.cuda_stream)
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        _lib.call("az_transition_f32", C.byref(a), stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / reps
    gbs = 12.0 * n / (ms * 1e-3) / 1e9
    # HBM bytes per launch from rocprofv3 PMC passes of this same kernel and size (FETCH_SIZE doubled per the
    # gfx950 note + WRITE_SIZE; profiles/r01_hbm_traffic_pmc.txt): equal to the algorithmic bytes, no re-reads
    pmc_traffic = 805306368 if n == 1 << 26 else None
    return dict(bound="hbm", achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4),
                traffic=pmc_traffic, traffic_source="profiles/r01_hbm_traffic_pmc.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)",
                algorithmic_bytes=12 * n, elements=n, bytes_per_element=12, avg_us=round(ms * 1e3, 2),
                kernel="transition_flat_kernel (DDIM eta=0)")


def cpu_baseline(denoiser, cfg, budget_s=25.0):
    r"""The oracle (CPU restatement of azula's op sequence, bit-checked against the reference in the
    build container) timed on this host: DDIM steps of the same network at batch 1."""
    from oracle import nets, sampling

    sd = {k: v.detach().cpu() for k, v in denoiser.backbone.state_dict().items()}
    ncfg = dict(cfg["net"])
    # Use the thread count that is fastest on this host for the dominant op (a 256->256 3x3 conv):
    # os.cpu_count() may exceed the cores this process may run on, and oversubscription is slow.
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    probe_x, probe_w = torch.randn(1, 256, 128, 128), torch.randn(256, 256, 3, 3)
    best, threads = None, 1
    for nt in sorted({min(avail, c) for c in (8, 16, 32, 64, 128, avail)}):
        torch.set_num_threads(nt)
        torch.nn.functional.conv2d(probe_x, probe_w, padding=1)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.conv2d(probe_x, probe_w, padding=1)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, threads = dt, nt
    torch.set_num_threads(threads)
    mean = lambda x, t: sampling.karras_mean(lambda a, c: nets.time_wrapped_unet(sd, ncfg, a, c), x, t)  # noqa: E731
    torch.manual_seed(1)
    x = torch.randn(1, *cfg["shape"])
    pairs = sampling.time_pairs(steps=cfg["steps"])
    a_t, s_t = sampling.vp_schedule(pairs[0, 0])
    a_s, s_s = sampling.vp_schedule(pairs[0, 1])

    def one_step(x):
        m = mean(x, pairs[0, 0])
        return sampling.transition(x, m, torch.zeros_like(x), a_t, s_t, a_s, s_s, 0.0)

    t0 = time.perf_counter()
    one_step(x)  # warm-up
    warm = time.perf_counter() - t0
    n, t0 = 0, time.perf_counter()
    while True:
        one_step(x)
        n += 1
        el = time.perf_counter() - t0
        if n >= 2 and el + warm > budget_s or n >= 8:
            break
    s_per_step = el / n
    return dict(
        value=round(1.0 / (cfg["steps"] * s_per_step), 6), unit="images/s", cores=threads, kind="port",
        sample=f"{n} DDIM steps of the same UNet at batch 1 ({s_per_step:.2f} s/step, 1 warm-up), extrapolated x{cfg['steps']} steps",
    )


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--denoise-steps", type=int, default=0, help="override the config's sampler steps (checks only)")
    ap.add_argument("--half", choices=["bf16", "f16"], default=None,
                    help="cast the backbone to half precision (mixed-precision mode; NOT the headline fp32 number)")
    ap.add_argument("--fp32-mfma", choices=["native", "bf16x3"], default=None,
                    help="how fp32 convs / GEMMs use the matrix pipe (default: env AZ_FP32_MFMA or native fp32 MFMA); "
                         "bf16x3 = exact 3-piece bf16 split, 6 partial products, fp32 accumulate (opt-in, NOT the headline)")
    args = ap.parse_args()
    if args.fp32_mfma:
        os.environ["AZ_FP32_MFMA"] = args.fp32_mfma  # read by azula_amd.engine at import

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    device = torch.device("cuda", local % torch.cuda.device_count())  # (modulo: lets a 1-GPU box rehearse N > 1 over gloo)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist

        backend = os.environ.get("AZ_DIST_BACKEND", "nccl")  # "nccl" IS RCCL over xGMI on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    torch.set_grad_enabled(False)

    from azula_amd.sample import DDIMSampler, DDPMSampler

    cfg = dict(CONFIGS[args.config])
    if args.denoise_steps:
        cfg["steps"] = args.denoise_steps
        cfg["name"] += f" [steps overridden to {args.denoise_steps}: not a headline number]"
    den = build_denoiser(cfg, device)
    if args.half:
        inner = den.denoiser if hasattr(den, "denoiser") else den
        inner.backbone.to(torch.bfloat16 if args.half == "bf16" else torch.float16)
        cfg["name"] += f" [backbone cast to {args.half}: MFMA operands {args.half}, fp32 accumulate -- not a headline number]"
    from azula_amd import engine as _engine

    if _engine.FP32_MFMA != "native" and not args.half:
        cfg["name"] += (f" [AZ_FP32_MFMA={_engine.FP32_MFMA}: fp32 operands split into 3 bf16 pieces, 6 partial products on "
                        "the bf16 MFMA, fp32 accumulate -- opt-in mode, not the headline number]")
    Smp = DDPMSampler if cfg.get("sampler") == "ddpm" else DDIMSampler
    sampler = Smp(den, steps=cfg["steps"], silent=True)
    B = cfg["batch"]
    from azula_amd.parallel import init_sharded, sample_sharded

    torch.manual_seed(1)  # same seed on every rank: the full batch is drawn and sliced (parity with 1 GPU)
    x1 = init_sharded(sampler, (world * B, *cfg["shape"]), device=device)  # resident in HBM before timing

    kwargs = {}
    if cfg.get("cfg"):
        lab = torch.arange(B, device=device) % 1000
        kwargs = dict(positive={"label": lab}, negative={"label": torch.zeros_like(lab)}, guidance=cfg["cfg"])

    if cfg.get("labels"):
        kwargs = dict(label=torch.arange(B, device=device) % 1000)

    def one_pass():
        # 64 graph replays on this rank's shard, then the only collective: all-gather of x0
        return sample_sharded(sampler, x1, **kwargs)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        one_pass()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x0 = one_pass()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    assert torch.isfinite(x0).all()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        images_per_s = world * B * args.steps / elapsed
        out = {
            "metric": "images/sec (whole node), DDIM-64 256x256 UNet" if args.config == "c2"
            else f"images/sec (whole node), {args.config}",
            "value": round(images_per_s, 4),
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "ms_per_denoise_step": round(ms_per_step / cfg["steps"], 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": (f"{args.half} operands / f32 accumulate" if args.half
                      else ("f32" if _engine.FP32_MFMA == "native" else f"f32 ({_engine.FP32_MFMA} split on the bf16 MFMA)")),
            "data": "synthetic (random-init weights under seed 0, x1 ~ sampler.init under seed 1)",
            "config": {"workload": cfg["name"], "per_gpu_batch": B, "global_batch": world * B,
                       "denoise_steps": cfg["steps"], "parallelism": f"batch-sharded x{world}, all-gather of x0"},
        }
        conv = conv_roofline(sampler, device)
        wino, direct, allc = conv["az_conv2d_winograd_f32"], conv["az_conv2d_f32"], conv["all"]
        dom, dom_name = (wino, "conv_winograd_kernel") if wino["ms"] >= direct["ms"] else (direct, "conv_igemm_kernel")
        tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        executed = tf / 2.25 if dom is wino else tf
        out["roofline"] = {
            "bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_FP32_TFLOPS, 4), "traffic": None,
            "traffic_note": "MFMA-bound kernel over 55 launches of different shapes; rocprofv3 --pmc on its largest layer "
                            "(4x256x256, 256->256): FETCH_SIZE 388 MB + WRITE_SIZE 240 MB per launch = 1.4x the compulsory "
                            "541 MB, L2 hit rate 92 % (8-byte gathers: counters uncalibrated, hence null) -- "
                            "profiles/r01_hbm_traffic_pmc.txt",
            "kernel": dom_name + " (fp32 v_mfma_f32_32x32x2_f32), all its launches in one backbone forward",
            "launches": dom["launches"], "avg_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
            "note": "achieved = ALGORITHMIC direct-conv FLOP (2*pixels*Cout*Cin*9) / HIP-event time; the Winograd "
                    "F(2x2,3x3) kernel executes 2.25x fewer multiplies in exact fp32, hence frac can exceed 1",
            "executed_mfma_tflops": round(executed, 2), "executed_frac": round(executed / PEAK_FP32_TFLOPS, 4),
            "all_convs": {"launches": allc["launches"], "ms_per_forward": round(allc["ms"], 3),
                          "algorithmic_tflops": round(allc["flops"] / (allc["ms"] * 1e-3) / 1e12, 2),
                          "flops_per_forward": allc["flops"]},
        }
        out["roofline_transition"] = transition_roofline(device)
        if world == 1 and not args.no_cpu_baseline and cfg["kind"] == "unet":
            out["cpu_baseline"] = cpu_baseline(den, cfg)
            out["speedup_vs_cpu"] = round(images_per_s / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
