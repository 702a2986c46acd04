r"""Timing gate for F(4x4,3x3) on the split operands (VERDICT r05 "next round" #1).

    python tools/f4_gate.py [seconds]                                              # the shipped library: full maps + 0.75-scaled maps
    AZULA_AMD_LIB=azula_amd/csrc/_ab/libazula_amd_wx3_f4proxy.so python tools/f4_gate.py [seconds] proxy   # the proxy on the scaled maps

The proxy.  F(4x4) on an H x W map runs 36 frequencies on (H / 4)(W / 4) tiles; the shipped F(2x2) stream on a (0.75 H) x (0.75 W)
map runs 16 frequencies on (0.375 H)(0.375 W) tiles: the SAME number of (tile, frequency) pairs = 0.5625 of the full map's, and
with them the same matrix instructions, patch reads, V stores, fragment reads and splits.  The `wx3_f4proxy` variant of
tools/ablate.py adds what F(4x4) pays on top per V value and per matrix instruction: the 6 x 6 transforms' packed operations
(4.0 per value against 2.0) and twice the filter fragment bytes per matrix instruction (a 64-cout x 32-tile block is what the
295 KB of accumulators of 36 frequencies leave room for).  Not in the proxy, added by hand in the summary: the output stores of
the full map (the scaled map stores 0.5625 of them), priced at the epilogue's measured store rate.

Every line: microseconds per launch over `seconds` of back-to-back launches, socket power and sclk from rocm-smi meanwhile.
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

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
MODE = sys.argv[2] if len(sys.argv) > 2 else "shipped"
dev = torch.device("cuda")
LAYERS = ((4, 256, 256, 256, 256), (4, 64, 64, 512, 512), (32, 128, 128, 256, 256), (4, 128, 128, 512, 512))


def smi():
    try:
        return subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e!r}"


def timed(fn, label):
    stop = threading.Event()
    samples = []

    def loop():
        while not stop.is_set():
            samples.append(smi())

    for _ in range(10):
        fn()
    torch.cuda.synchronize()
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
    us = e0.elapsed_time(e1) / n * 1e3
    pw, clk = [], []
    for txt in samples[1:]:
        m = re.search(r"Current Socket Graphics Package Power \(W\):\s*([\d.]+)", txt)
        if m:
            pw.append(float(m.group(1)))
        m = re.search(r"sclk clock level:.*\((\d+)Mhz\)", txt)
        if m:
            clk.append(float(m.group(1)))
    f = lambda v: f"{sum(v) / len(v):6.0f}" if v else "   n/a"  # noqa: E731
    print(f"{label:58s} {us:9.1f} us   {f(pw)} W   {f(clk)} MHz", flush=True)
    return us


def layer(B, H, W, Cin, Cout, mode):
    bld = Builder(dev)
    x = Act(torch.randn(B * H * W * Cin, device=dev), B, H, W, Cin, Cin, True)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
    bld.conv(x, bld.pack_conv(w, torch.randn(Cout, device=dev)), Cout, act=1, winograd=mode)
    bld.finish()
    return bld


print(f"library: {_lib.LIB_PATH}  mode: {MODE}")
torch.manual_seed(0)
res = {}
for (B, H, W, Cin, Cout) in LAYERS:
    hs, ws = H * 3 // 4, W * 3 // 4
    if MODE == "shipped":
        bld = layer(B, H, W, Cin, Cout, "wx3")
        res[(H, "full")] = timed(bld.tape.run, f"F(2x2) x3, shipped, {B}x{H}x{W} {Cin}->{Cout}")
        del bld
        bld = layer(B, H, W, Cin, Cout, 4)
        timed(bld.tape.run, f"F(4x4) fp32 kernel of round 1, {B}x{H}x{W} {Cin}->{Cout}")
        del bld
    bld = layer(B, hs, ws, Cin, Cout, "wx3")
    timed(bld.tape.run, f"F(2x2) x3, {MODE}, scaled map {B}x{hs}x{ws} {Cin}->{Cout}")
    del bld
    torch.cuda.empty_cache()
