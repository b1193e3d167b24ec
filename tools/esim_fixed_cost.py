"""Fixed (per-launch / per-workgroup) cost of the headline ESIM launch: kernel time for 256 clips x N frames x 256 x 256 float32,
reference-default parameters, bilinear 5 bins, N = 2 .. 33 -> slope (ms per frame pair) and intercept (prologue + epilogue + stores)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import esim  # noqa: E402


def time_ms(fn, reps=30):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2]


full = esim.synth_clips(256, 33, 256, 256, dtype=torch.float32, seed=1, clip_id0=0)
o = torch.empty((256, 5, 256, 256), dtype=torch.float32, device="cuda")
for name, P in (("reference defaults", [0.2, 0.2, 0.1, 1e-3, 0.1]), ("base noise only", [0.2, 0.2, 0.1, 0.0, 0.0]), ("noise-free", [0.2, 0.2, 0.0, 0.0, 0.0])):
    pt = torch.tensor(P, dtype=torch.float64, device="cuda")
    xs, ys = [], []
    for n in (3, 5, 9, 17, 33):
        frames = full[:, :n].contiguous()
        ms = time_ms(lambda: esim.esim_voxel_batch(frames, pt, bin_mode="bilinear", num_bins=5, seed=1, out=o, validate=False))
        xs.append(n - 1); ys.append(ms)
    a, b = np.polyfit(xs, ys, 1)
    print(f"{name:20s} ms {[round(y, 4) for y in ys]}  slope {a:.5f} ms per frame pair, intercept {b:.4f} ms", flush=True)
print(f"output stores alone: {o.numel() * 4 / 5.5e9:.4f} ms at 5.5 TB/s")

# the same question for the v2e launch (config 3's parameters): pre-pass + simulation per call
from v2v_amd import v2e  # noqa: E402
vp = v2e.make_params(24, "pn_related", 0.5, 0.1, 0.0, 0.1, 30, 0.1, 0, 5.0, 0.1, 0.1)          # bench.py's V2E_NOISY
xs, ys = [], []
for n in (3, 5, 9, 17, 33):
    frames = full[:, :n].contiguous()
    ms = time_ms(lambda: v2e.v2e_voxel_batch(frames, vp, bin_mode="bilinear", num_bins=5, seed=1, out=o))
    xs.append(n - 1); ys.append(ms)
a, b = np.polyfit(xs, ys, 1)
print(f"{'v2e (config 3)':20s} ms {[round(y, 4) for y in ys]}  slope {a:.5f} ms per frame pair, intercept {b:.4f} ms", flush=True)
