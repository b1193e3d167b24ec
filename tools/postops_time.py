#!/usr/bin/env python3
"""Time normalize_and_pad on the training batch shape [12,40,5,128,128] for dense and sparse (voxel-like) content."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import postops  # noqa: E402

torch.manual_seed(0)
dense = torch.round(torch.randn((12, 40, 5, 128, 128), device="cuda") * 3)
sparse = dense * (torch.rand_like(dense) < 0.1)
for name, v in (("dense  (round(3*randn))", dense), ("sparse (90 % zeros)", sparse)):
    for _ in range(3):
        postops.normalize_and_pad(v, True, 16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        postops.normalize_and_pad(v, True, 16)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:26s} {e0.elapsed_time(e1) / 10:.3f} ms per batch ({v.numel() * 4 / 1e6:.0f} MB)")
