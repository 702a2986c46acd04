r"""HBM traffic of the hot kernels from the rocprofv3 PMC counters, calibrated on known-traffic kernels.

    python tools/pmc_traffic.py collect [--config c2] [--out profiles/r02_traffic.json] [--keep DIR]
    python tools/pmc_traffic.py workload [--config c2] --manifest FILE      (what runs under rocprofv3)

``collect`` runs ``workload`` twice under ``rocprofv3 --pmc <counter> --kernel-trace`` -- FETCH_SIZE and WRITE_SIZE in
SEPARATE passes (they do not fit one pass: MI355X_MICROARCH.md, "rocprofv3 PMC slots"), never combined with any other
trace domain -- and reduces the per-dispatch CSV to bytes per launch for every kernel of interest.

The workload is one process with three parts:

1. calibration: ``az_calib_read_f32`` / ``az_calib_write_f32`` touch every byte of a 2 GiB buffer exactly once in the
   access shapes of the product kernels, so ``true bytes / counter bytes`` is the correction factor of that shape
   (the guide's x2 for 16-byte-per-lane streams is re-measured here, the 8-byte patch gathers and the stores are new);
2. the transition kernels at a size that defeats the 256 MiB Infinity Cache (96 Mi elements per tensor);
3. ``--forwards`` eager backbone forwards of the bench configuration (the same tape the hipGraph replays).

Counter units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in units of 1024 B (with that unit the calibration reproduces
the guide's factor 2.00 for 16-byte-per-lane streaming reads).
"""

from __future__ import annotations

import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CALIB_BYTES = 2 << 30
# (label, width, group_bytes, row_bytes): see az_calib_read_f32
CALIB_READS = [
    ("read16_stream", 16, 1024, 1024),       # 16 B/lane, 1 KB contiguous per wave instruction (transition, GN, epilogue reads)
    ("read8_stream", 8, 512, 512),           # 8 B/lane contiguous
    ("read8_gather32_row1k", 8, 32, 1024),   # Winograd patch gather: 4 lanes x 8 B = one 32-byte sector per pixel, 256 channels
    ("read8_gather32_row4k", 8, 32, 4096),   # the same at 1024 channels
    ("read16_gather64_row1k", 16, 64, 1024),  # direct-conv gather: 4 lanes x 16 B per pixel
]
TRANSITION_SHAPE = (32, 3, 1 << 20)  # (B, C, H*W): 96 Mi elements = 384 MiB per tensor


def transition_cases(device):
    r"""(label, AzTransitionArgs, algorithmic bytes, keep-alive) for the kernels the captured loops launch."""
    import torch

    from azula_amd import _lib
    from azula_amd.engine import transition_args

    B, Cc, inner = TRANSITION_SHAPE
    n = B * Cc * inner
    row = torch.zeros(_lib.COEF_WORDS, device=device)
    col = {k: i for i, k in enumerate(_lib.COEF_FIELDS)}
    for k, v in dict(c_skip=0.4, c_out=0.6, alpha_t=0.5, alpha_s=0.7, k_x=0.9, k_eps=0.3, c_in_next=1.1,
                     clip_lo=-float("inf"), clip_hi=float("inf")).items():
        row[col[k]] = v
    x = torch.randn(n, device=device)
    F3 = torch.randn(n, device=device)
    F6 = torch.randn(2 * n, device=device)
    eps = torch.randn(n, device=device)
    xin = torch.empty(B * inner * 4, device=device)
    xs = torch.empty(n, device=device)
    cases = []
    # C2 (azula UNet, DDIM eta=0): planar F with 3 channels, x stepped in place, NHWC (stride 4) pre-scaled input written
    a = transition_args(x_t=x.data_ptr(), F=F3.data_ptr(), x_s=x.data_ptr(), xin_next=xin.data_ptr(), batch=B, channels=Cc,
                        inner=inner, f_channels=3, nhwc_pad=4, coef=row.data_ptr())
    cases.append(("image_ddim", "transition_image4_kernel<false, false, false>", a, 16 * n,
                  "read x_t, F; write x_s, c_in' x_s: 16 B/element"))
    # C4 (ADM, DDPM): F = first 3 of 6 planar channels, eps read
    a = transition_args(x_t=x.data_ptr(), F=F6.data_ptr(), eps=eps.data_ptr(), x_s=x.data_ptr(), xin_next=xin.data_ptr(), batch=B,
                        channels=Cc, inner=inner, f_channels=6, nhwc_pad=4, coef=row.data_ptr())
    cases.append(("image_ddpm", "transition_image4_kernel<false, true, false>", a, 20 * n,
                  "read x_t, F, eps; write x_s, c_in' x_s: 20 B/element"))
    # generic-loop / toy path: every tensor flat, no second output
    a = transition_args(x_t=x.data_ptr(), F=F3.data_ptr(), x_s=xs.data_ptr(), batch=1, channels=1, inner=n, f_channels=1,
                        coef=row.data_ptr())
    cases.append(("flat_ddim", "transition_flat_kernel<false, false, false, false>", a, 12 * n,
                  "read x_t, F; write x_s: 12 B/element"))
    # ViT / JiT latents: every tensor flat, the pre-scaled backbone input is a second flat output
    a = transition_args(x_t=x.data_ptr(), F=F3.data_ptr(), x_s=x.data_ptr(), xin_next=xs.data_ptr(), batch=1, channels=1, inner=n,
                        f_channels=1, coef=row.data_ptr())
    cases.append(("flat_ddim_xin", "transition_flat_kernel<false, false, true, false>", a, 16 * n,
                  "read x_t, F; write x_s, c_in' x_s: 16 B/element"))
    # ADM with the planar stem (C4 / C5, DDPM): F = first 3 of 6 planar channels per image, eps read, planar second output
    a = transition_args(x_t=x.data_ptr(), F=F6.data_ptr(), eps=eps.data_ptr(), x_s=x.data_ptr(), xin_next=xs.data_ptr(), batch=B,
                        channels=Cc, inner=inner, f_channels=6, coef=row.data_ptr())
    cases.append(("flat_ddpm_xin", "transition_flat_kernel<false, true, true, false>", a, 20 * n,
                  "read x_t, F (3 of 6 planar channels), eps; write x_s, c_in' x_s: 20 B/element"))
    return cases, (row, x, F3, F6, eps, xin, xs)


def build_loop(config: str, device):
    r"""The fused loop of the bench configuration, NOT captured: its tape is run eagerly."""
    import torch

    import bench
    from azula_amd._lib import COEF_WORDS
    from azula_amd.parallel import init_sharded
    from azula_amd.sample import DDIMSampler, DDPMSampler, _FusedLoop

    cfg = dict(bench.CONFIGS[config])
    den = bench.build_denoiser(cfg, device)
    Smp = DDPMSampler if cfg.get("sampler") == "ddpm" else DDIMSampler
    sampler = Smp(den, steps=cfg["steps"], silent=True)
    torch.manual_seed(1)
    x1 = init_sharded(sampler, (cfg["batch"], *cfg["shape"]), device=device)
    kwargs = bench.sampler_kwargs(cfg, device)
    cur = torch.zeros(COEF_WORDS, dtype=torch.float32, device=device)
    fused = den._az_fused(x1, kwargs, cur)
    loop = _FusedLoop(sampler, fused, x1, cur)
    loop._upload_table(kwargs)
    for p in fused.programs:
        if p.prepare is not None:
            p.prepare(kwargs)
    loop.x.copy_(x1)
    if loop.eps is not None:
        loop.eps.normal_()
    return loop, cfg


def workload(args) -> None:
    import torch

    from azula_amd import _lib

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)
    stream = torch.cuda.current_stream(dev).cuda_stream
    manifest = {"calib_bytes": CALIB_BYTES, "calib_reads": [c[0] for c in CALIB_READS], "transition": [], "forwards": args.forwards,
                "config": args.config}
    # ---- 1. calibration (each byte exactly once; the data is written by torch first, a different kernel name)
    buf = torch.empty(CALIB_BYTES // 4, dtype=torch.float32, device=dev)
    buf.fill_(1.0)
    sink = torch.zeros(16, device=dev)
    for _, width, group, rowb in CALIB_READS:
        _lib.call("az_calib_read_f32", buf.data_ptr(), sink.data_ptr(), CALIB_BYTES, width, group, rowb, stream)
    _lib.call("az_calib_write_f32", buf.data_ptr(), CALIB_BYTES, 2.0, stream)
    torch.cuda.synchronize(dev)
    del buf
    # ---- 2. transition kernels, 2 launches each
    cases, keep = transition_cases(dev)
    for label, kname, a, alg, note in cases:
        for _ in range(2):
            _lib.call("az_transition_f32", C.byref(a), stream)
        manifest["transition"].append({"label": label, "kernel": kname, "algorithmic_bytes": alg, "note": note,
                                       "elements": TRANSITION_SHAPE[0] * TRANSITION_SHAPE[1] * TRANSITION_SHAPE[2]})
    torch.cuda.synchronize(dev)
    del cases, keep
    torch.cuda.empty_cache()
    # ---- 3. eager backbone forwards of the bench configuration
    if args.forwards > 0:
        loop, cfg = build_loop(args.config, dev)
        # marker dispatch: everything after the SECOND calib_write_kernel launch belongs to the timed forwards (plan
        # building launches weight-packing kernels and, for the ViT, one GEMM of its own)
        mark = torch.zeros(4, device=dev)
        _lib.call("az_calib_write_f32", mark.data_ptr(), 16, 0.0, stream)
        for _ in range(args.forwards):
            loop.counter.zero_()
            loop.tape.run(stream)
        torch.cuda.synchronize(dev)
        convs = [a[0]._obj for _, a, n in loop.tape.ops if n.startswith("az_conv2d")]
        manifest["tape_ops"] = len(loop.tape)
        manifest["conv_launches"] = {}
        for d in convs:
            manifest["conv_launches"][d._algo] = manifest["conv_launches"].get(d._algo, 0) + 1
        # compulsory bytes of the Winograd launches: every input, filter, residual and output element once
        comp = 0
        for d in convs:
            if d._algo not in ("az_conv2d_winograd_f32", "az_conv2d_winograd_x3_f32", "az_conv2d_winograd_f16x2_f32"):
                continue
            npix_in = d.batch * d.h0 * d.w0 * d.c0s + (d.batch * d.h1 * d.w1 * d.c1s if d.src1 else 0)
            npix_out = d.batch * d.hout * d.wout * d.cout_s
            filt = 16 * d.cout_s * (d.c0s + d.c1s)  # (x3: three bf16 pieces per value = 6 bytes; f16x2: two half pieces are READ, the third is derived = 4 bytes)
            comp += 4 * (npix_in + npix_out + (npix_out if d.res else 0)) + (6 if d._algo.endswith("_x3_f32") else 4) * filt
        manifest["winograd_compulsory_bytes_per_forward"] = comp
    with open(args.manifest, "w") as f:
        json.dump(manifest, f)


def parse_counter_csv(directory: str, counter: str):
    r"""[(dispatch id, kernel name, value)] in dispatch order from rocprofv3's counter_collection CSV."""
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise RuntimeError(f"no counter_collection.csv under {directory}")
    rows = []
    for fn in files:
        with open(fn, newline="") as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


def collect(args) -> dict:
    keep = args.keep or tempfile.mkdtemp(prefix="az_pmc_")
    os.makedirs(keep, exist_ok=True)
    env = dict(os.environ)
    env.setdefault("TMPDIR", "/tmp")
    per = {}
    manifest = None
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = os.path.join(keep, counter.lower())
        shutil.rmtree(out, ignore_errors=True)
        man = os.path.join(keep, f"manifest_{counter.lower()}.json")
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "workload", "--config", args.config, "--forwards", str(args.forwards),
               "--manifest", man]
        res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=args.timeout)
        if res.returncode != 0:
            raise RuntimeError(f"rocprofv3 --pmc {counter} failed ({res.returncode}): {res.stderr[-2000:]}")
        per[counter] = parse_counter_csv(out, counter)
        manifest = json.load(open(man))
    KB = 1024.0
    result = {"tool": "tools/pmc_traffic.py", "counters": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, --kernel-trace only)",
              "counter_unit_bytes": KB, "config": manifest["config"], "forwards": manifest["forwards"]}

    def by_kernel(counter, substr):
        return [v for _, k, v in per[counter] if substr in k]

    # ---- calibration factors
    reads = by_kernel("FETCH_SIZE", "calib_read_kernel")
    assert len(reads) == len(CALIB_READS), (len(reads), [k for _, k, _ in per["FETCH_SIZE"]][:20])
    calib = {}
    for (label, width, group, rowb), v in zip(CALIB_READS, reads):
        calib[label] = {"true_bytes": CALIB_BYTES, "fetch_size_raw_bytes": v * KB, "factor": CALIB_BYTES / (v * KB),
                        "width": width, "group_bytes": group, "row_bytes": rowb}
    wr = by_kernel("WRITE_SIZE", "calib_write_kernel")[:1]  # (a second, 16-byte launch marks the start of the forwards)
    calib["write16_stream"] = {"true_bytes": CALIB_BYTES, "write_size_raw_bytes": wr[0] * KB, "factor": CALIB_BYTES / (wr[0] * KB)}
    # reads made by the write kernel / writes made by the read kernels (sanity: ~0)
    calib["write16_stream"]["fetch_size_raw_bytes"] = by_kernel("FETCH_SIZE", "calib_write_kernel")[0] * KB
    result["calibration"] = calib
    f16, f8g, fw = calib["read16_stream"]["factor"], calib["read8_gather32_row1k"]["factor"], calib["write16_stream"]["factor"]
    result["factors_used"] = {"stream_read_16B": f16, "gather_read_8B": f8g, "write_16B": fw}

    # ---- transition kernels
    result["transition"] = {}
    for t in manifest["transition"]:
        fe = by_kernel("FETCH_SIZE", t["kernel"])[:2]  # the two large launches come first; the forwards' own
        wv = by_kernel("WRITE_SIZE", t["kernel"])[:2]  # transition launches (786 k elements) follow
        if not fe:
            continue
        fb, wb = sum(fe) / len(fe) * KB, sum(wv) / len(wv) * KB
        traffic = fb * f16 + wb * fw
        result["transition"][t["label"]] = {
            "kernel": t["kernel"], "launches": len(fe), "elements": t["elements"], "algorithmic_bytes": t["algorithmic_bytes"],
            "note": t["note"], "fetch_size_raw_bytes": fb, "write_size_raw_bytes": wb, "traffic_bytes": traffic,
            "traffic_over_algorithmic": traffic / t["algorithmic_bytes"],
        }
    # ---- backbone forward: the dispatches after the marker (second calib_write_kernel launch)
    if manifest["forwards"] > 0:
        def after_marker(counter):
            rows, seen = per[counter], 0
            for i, (_, k, _) in enumerate(rows):
                if "calib_write_kernel" in k:
                    seen += 1
                    if seen == 2:
                        return rows[i + 1:]
            raise RuntimeError("marker dispatch not found")

        tail = {c: after_marker(c) for c in ("FETCH_SIZE", "WRITE_SIZE")}
        names = sorted({k for _, k, _ in tail["FETCH_SIZE"]})
        fwd = {}
        for name in names:
            fe = [v for _, k, v in tail["FETCH_SIZE"] if k == name]
            wv = [v for _, k, v in tail["WRITE_SIZE"] if k == name]
            if not fe or len(fe) % manifest["forwards"] or len(fe) != len(wv):
                continue
            gather = "winograd" in name and "winograd_x3" not in name  # (8-byte gather loads; the x3 kernel reads 16 bytes per lane)
            fr = f8g if gather else f16
            fwd[name] = {
                "launches_per_forward": len(fe) // manifest["forwards"],
                "fetch_size_raw_bytes_per_forward": sum(fe) * KB / manifest["forwards"],
                "write_size_raw_bytes_per_forward": sum(wv) * KB / manifest["forwards"],
                "read_factor": fr, "write_factor": fw,
                "traffic_bytes_per_forward": (sum(fe) * fr + sum(wv) * fw) * KB / manifest["forwards"],
            }
            fwd[name]["traffic_bytes_per_launch"] = fwd[name]["traffic_bytes_per_forward"] / fwd[name]["launches_per_forward"]
        result["forward"] = fwd
        result["winograd_compulsory_bytes_per_forward"] = manifest.get("winograd_compulsory_bytes_per_forward")
        result["conv_launches"] = manifest.get("conv_launches")
    if not args.keep:
        shutil.rmtree(keep, ignore_errors=True)
    return result


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["collect", "workload"])
    ap.add_argument("--config", default="c2")
    ap.add_argument("--forwards", type=int, default=2)
    ap.add_argument("--manifest", default="/tmp/az_pmc_manifest.json")
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep", default=None, help="directory for the raw rocprofv3 output (default: temporary)")
    ap.add_argument("--timeout", type=int, default=600)
    args = ap.parse_args()
    if args.mode == "workload":
        workload(args)
        return
    res = collect(args)
    text = json.dumps(res, indent=1)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
