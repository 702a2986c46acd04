r"""Micro-benchmark of az_transition_f32 in the forms the captured loops launch, at a MALL-defeating size."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

for k, v in bench.transition_roofline(torch.device("cuda")).items():
    print(k, json.dumps({a: v[a] for a in ("kernel", "achieved", "frac", "avg_us")}))
