r"""Micro-benchmark of az_transition_f32 (flat DDIM eta=0 form) at a MALL-defeating size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

print(bench.transition_roofline(torch.device("cuda"), n=int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26))
