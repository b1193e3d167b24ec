#!/usr/bin/env python3
"""Time the v2e kernel (config 3 shape) with model features switched off one at a time.

Run on the GPU box: `python tools/v2e_breakdown.py [u8|f32]`.  Prints ms per 256-clip launch (kernel + pre-pass)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import esim, v2e  # noqa: E402

dtype = torch.uint8 if (len(sys.argv) > 1 and sys.argv[1] == "u8") else torch.float32
B, N, H, W = 256, 32, 256, 256
frames = esim.synth_clips(B, N, H, W, dtype=dtype, seed=20240001)
out = torch.empty((B, 5, H, W), dtype=torch.float32, device="cuda")
base = dict(FPS=24, threshold_model="pn_related", thres_mean_mean=0.5, thres_mean_std=0.1, thres_diff_mean=0.0,
            thres_diff_std=0.1, cutoff_hz=30, leak_rate_hz=0.1, refractory_period_s=0, shot_noise_rate_hz=5.0,
            leak_jitter_fraction=0.1, noise_rate_cov_decades=0.1)
variants = {
    "all features (cfg 3)": {},
    "no shot noise": dict(shot_noise_rate_hz=0.0),
    "no leak": dict(leak_rate_hz=0.0),
    "no low-pass": dict(cutoff_hz=0),
    "no shot, no leak": dict(shot_noise_rate_hz=0.0, leak_rate_hz=0.0),
    "no shot, no leak, no low-pass": dict(shot_noise_rate_hz=0.0, leak_rate_hz=0.0, cutoff_hz=0),
    "shot rate 50 Hz": dict(shot_noise_rate_hz=50.0),
    "refractory 1/240 s": dict(refractory_period_s=1 / 240),
    "spatial_temporal_independent": dict(threshold_model="spatial_temporal_independent"),
}
only = sys.argv[2] if len(sys.argv) > 2 else None
for name, kw in variants.items():
    if only is not None and only != name:
        continue
    p = v2e.make_params(**{**base, **kw})
    for _ in range(3):
        v2e.v2e_voxel_batch(frames, p, bin_mode="bilinear", num_bins=5, seed=7, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        v2e.v2e_voxel_batch(frames, p, bin_mode="bilinear", num_bins=5, seed=7, out=out)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:34s} {e0.elapsed_time(e1) / 10:.3f} ms")
