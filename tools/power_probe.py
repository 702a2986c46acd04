r"""What limits the clock?  Samples `rocm-smi` (socket power, sclk, power cap) in a background thread while one workload runs
back to back for a few seconds: the fp32 MFMA calibration kernel (registers only), the Winograd kernel on the 256^2 layer, its
MFMA-only / no-staging ablations when A/B libraries are given.

    python tools/power_probe.py [seconds]          # -> stdout table; raw samples in gpurun_out/power_probe_raw.txt
    python tools/power_probe.py [seconds] wx3      # only: az_conv2d_winograd_f32 against az_conv2d_winograd_x3_f32 on the gate layers
    python tools/power_probe.py [seconds] f16x2    # only: the bf16x3 against the f16x2 forms (Winograd layers, token GEMMs), random data and zeros
"""
import os
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from azula_amd import _lib
from azula_amd.engine import Act, Builder

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
MODE = sys.argv[2] if len(sys.argv) > 2 else "all"
dev = torch.device("cuda")
raw = []


def smi():
    try:
        return subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showtemp"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e!r}"


def sample_while(fn, label):
    stop = threading.Event()
    samples = []

    def loop():
        while not stop.is_set():
            samples.append((time.time(), smi()))

    th = threading.Thread(target=loop)
    t0 = time.time()
    th.start()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < SECS:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1) / n
    pw, clk = [], []
    for ts, txt in samples[1:]:
        raw.append(f"--- {label} t={ts - t0:.2f}\n{txt}")
        m = re.search(r"Current Socket Graphics Package Power \(W\):\s*([\d.]+)", txt)
        if m:
            pw.append(float(m.group(1)))
        m = re.search(r"sclk clock level:.*\((\d+)Mhz\)", txt)
        if m:
            clk.append(float(m.group(1)))
    cap = re.search(r"Max Graphics Package Power \(W\):\s*([\d.]+)", samples[-1][1] if samples else "")
    f = lambda v: f"{sum(v) / len(v):7.1f} (max {max(v):7.1f}, n={len(v)})" if v else "   n/a"  # noqa: E731
    print(f"{label:34s} {ms * 1e3:9.1f} us/launch   power W {f(pw)}   sclk MHz {f(clk)}   cap {cap.group(1) if cap else '?'}", flush=True)
    return ms


if MODE == "all":
    print("idle:", re.sub(r"\s+", " ", smi())[:600])
    sink = torch.zeros(4, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ms = sample_while(lambda: _lib.call("az_calib_mfma_f32", sink.data_ptr(), 512, 3000, 1.0, 0.5, st), "calib: fp32 MFMA, constant operands")
    print(f"    -> {512 * 4 * 3000 * 8 * 4096 / ms / 1e9:.1f} TF/s")
    ms = sample_while(lambda: _lib.call("az_calib_mfma_random_f32", sink.data_ptr(), 512, 3000, 1.0, 0.5, st), "calib: fp32 MFMA, random operands")
    print(f"    -> {512 * 4 * 3000 * 8 * 4096 / ms / 1e9:.1f} TF/s")

    ms = sample_while(lambda: _lib.call("az_calib_mfma_random_bf16", sink.data_ptr(), 512, 6000, 1.0, 0.5, st), "calib: bf16 MFMA, random operands")
    print(f"    -> {512 * 4 * 6000 * 8 * 32768 / ms / 1e9:.1f} TF/s (nominal 2516.8)")

    torch.manual_seed(0)
    # the bf16x3 GEMM on a token-linear shape (DiT-B MLP: 16384 tokens, 768 -> 3072)
    for (T, Cin, Cout) in ((16384, 768, 3072), (16384, 3072, 768)):
        bld = Builder(dev)
        x = Act(torch.randn(T * Cin, device=dev), 1, T, 1, Cin, Cin, True)
        w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5
        y = bld.conv(x, bld.pack_conv(w, torch.randn(Cout, device=dev)), Cout, winograd="x3")
        bld.finish()
        ms = sample_while(bld.tape.run, f"bf16x3 GEMM {T} x {Cin} -> {Cout}")
        fl = 2 * T * Cin * Cout
        print(f"    -> {fl / ms / 1e9:.1f} TF/s algorithmic, {fl * 6 / ms / 1e9 / 2516.8:.3f} of the bf16 MFMA peak executed")
    for (B, H, W, Cin, Cout) in ((4, 256, 256, 256, 256), (4, 64, 64, 512, 512)):
        for zero in (False, True):
            bld = Builder(dev)
            sc = 0.0 if zero else 1.0
            x = Act(torch.randn(B * H * W * Cin, device=dev) * sc, B, H, W, Cin, Cin, True)
            w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5 * sc
            y = bld.conv(x, bld.pack_conv(w, torch.randn(Cout, device=dev)), Cout, act=1, winograd=True)
            bld.finish()
            ms = sample_while(bld.tape.run, f"winograd {B}x{H}x{W} {Cin}->{Cout}{' zeros' if zero else ''}")
            fl = 2 * B * H * W * Cin * Cout * 9
            print(f"    -> {fl / ms / 1e9:.1f} TF/s algorithmic, {fl / ms / 1e9 / 2.25 / 157.3:.3f} of the fp32 MFMA peak executed")

if MODE == "wx3":
    print("idle:", re.sub(r"\s+", " ", smi())[:600])
    torch.manual_seed(0)
    for (B, H, W, Cin, Cout) in ((4, 256, 256, 256, 256), (4, 64, 64, 512, 512), (32, 128, 128, 256, 256)):
        for mode, name in ((True, "winograd f32"), ("wx3", "winograd x3 ")):
            for zero in (False, True):
                bld = Builder(dev)
                sc = 0.0 if zero else 1.0
                x = Act(torch.randn(B * H * W * Cin, device=dev) * sc, B, H, W, Cin, Cin, True)
                w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5 * sc
                y = bld.conv(x, bld.pack_conv(w, torch.randn(Cout, device=dev)), Cout, act=1, winograd=mode)
                bld.finish()
                ms = sample_while(bld.tape.run, f"{name} {B}x{H}x{W} {Cin}->{Cout}{' zeros' if zero else ''}")
                fl = 2 * B * H * W * Cin * Cout * 9
                print(f"    -> {fl / ms / 1e9:.1f} TF/s algorithmic")
if MODE == "f16x2":  # bf16x3 against f16x2: Winograd layers and token GEMMs, random data and zeros (is the kernel still at the cap?)
    print("idle:", re.sub(r"\s+", " ", smi())[:600])
    torch.manual_seed(0)
    for (B, H, W, Cin, Cout) in ((4, 256, 256, 256, 256), (4, 64, 64, 512, 512)):
        for mode, name in (("wx3", "winograd bf16x3"), ("wh2", "winograd f16x2 ")):
            for zero in (False, True):
                bld = Builder(dev)
                sc = 0.0 if zero else 1.0
                x = Act(torch.randn(B * H * W * Cin, device=dev) * sc, B, H, W, Cin, Cin, True)
                w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5 * (sc if zero else 1.0)
                y = bld.conv(x, bld.pack_conv(w, torch.randn(Cout, device=dev)), Cout, act=1, winograd=mode)
                bld.finish()
                ms = sample_while(bld.tape.run, f"{name} {B}x{H}x{W} {Cin}->{Cout}{' zeros' if zero else ''}")
                print(f"    -> {2 * B * H * W * Cin * Cout * 9 / ms / 1e9:.1f} TF/s algorithmic")
    for (T, Cin, Cout) in ((16384, 768, 3072), (16384, 3072, 768)):
        for mode, name in (("x3", "GEMM bf16x3"), ("h2", "GEMM f16x2 ")):
            for zero in (False, True):
                bld = Builder(dev)
                sc = 0.0 if zero else 1.0
                x = Act(torch.randn(T * Cin, device=dev) * sc, 1, T, 1, Cin, Cin, True)
                w = torch.randn(Cout, Cin, device=dev) / Cin ** 0.5 * (sc if zero else 1.0)
                y = bld.conv(x, bld.pack_conv(w, torch.randn(Cout, device=dev)), Cout, winograd=mode)
                bld.finish()
                ms = sample_while(bld.tape.run, f"{name} {T} x {Cin} -> {Cout}{' zeros' if zero else ''}")
                print(f"    -> {2 * T * Cin * Cout / ms / 1e9:.1f} TF/s algorithmic")
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/power_probe_raw.txt", "w").write("\n".join(raw[:400]))
