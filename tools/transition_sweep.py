r"""Transition-kernel roofline (the numbers bench.py prints as roofline_transition) on its own:
    [AZULA_AMD_LIB=...] python tools/transition_sweep.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
out = bench.transition_roofline(torch.device("cuda"))
for k, v in out.items():
    print(f"{k:14s} {v['avg_us']:8.1f} us  {v['achieved']:7.1f} GB/s  frac min/median/max {v['frac_min_median_max']}")
