r"""A/B of the streaming (HBM-bound) kernels under another build of the library (AZULA_AMD_LIB=<tools/ablate.py variant>):

    python tools/stream_ab.py            ->  one line per case: achieved GB/s and fraction of the 8 TB/s HBM peak

  * az_transition_f32 in the forms the captured loops launch (bench.transition_roofline: 96 Mi elements, Infinity-Cache defeating);
  * az_affine_act_f32 (GroupNorm apply + SiLU of ADM) on 4 x 256^2 x 256 (C5, batch 4), 32 x 128^2 x 256 and 32 x 256^2 x 256 (the C4 shard),
    8 B per element, median of 9 HIP-event timed launches each.
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from azula_amd import _lib  # noqa: E402


def affine_cases(device):
    stream = torch.cuda.current_stream(device)
    for B, H, Cc in ((4, 256, 256), (32, 128, 256), (32, 256, 256), (4, 64, 512)):
        n = B * H * H * Cc
        x = torch.empty(n, dtype=torch.float32, device=device).normal_()
        y = torch.empty_like(x)
        S = torch.rand(B * Cc, device=device) + 0.5
        T = torch.randn(B * Cc, device=device)
        args = (y.data_ptr(), x.data_ptr(), None, 0, S.data_ptr(), T.data_ptr(), B, H, H, Cc, 1, 0, stream.cuda_stream)
        for _ in range(3):
            _lib.call("az_affine_act_f32", *args)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(9)]
        for e0, e1 in evs:
            e0.record(stream)
            _lib.call("az_affine_act_f32", *args)
            e1.record(stream)
        torch.cuda.synchronize(device)
        ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)[4]
        gbs = 8 * n / (ms * 1e-3) / 1e9
        print(f"affine_act+silu {B}x{H}x{H}x{Cc}: {ms * 1e3:8.1f} us  {gbs:7.1f} GB/s  {gbs / bench.PEAK_HBM_GBS:.3f}", flush=True)
        del x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    print("library:", os.environ.get("AZULA_AMD_LIB", "(in-tree)"), flush=True)
    for label, r in bench.transition_roofline(dev).items():
        print(f"transition {label:16s}: {r['avg_us']:8.1f} us  {r['achieved']:7.1f} GB/s  {r['frac']:.3f}  {r.get('frac_min_median_max', '')}", flush=True)
    affine_cases(dev)
