"""Kernel time of the ESIM launch at config 4's per-GPU shapes (B x 41 x 256 x 256 uint8 -> 8 x 5 SUM bins) for several batch sizes
and every work-item mapping (4 / 2 / 1 pixels), same box, interleaved.  Run on the GPU box: python tools/cfg4_shape_time.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import esim  # noqa: E402

P = [0.2, 0.3, 0.05, 5e-4, 1.0]


def time_ms(fn, reps=30):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2]


for dt in (torch.uint8, torch.float32):
    for b in (8, 16, 24, 32, 48, 64):
        frames = esim.synth_clips(b, 41, 256, 256, dtype=dt, seed=1, clip_id0=0)
        pt = torch.tensor(P, dtype=torch.float64, device="cuda")
        o = torch.empty((b, 8, 5, 256, 256), dtype=torch.float32, device="cuda")
        row = {}
        for rnd in range(2):
            for m in ("4px", "2px", "1px", "auto"):
                ms = time_ms(lambda: esim.esim_voxel_batch(frames, pt, bin_mode="sum", num_bins=5, seed=1, out=o, validate=False, no_noise=False, mapping=m))
                row[m] = min(row.get(m, 1e9), ms)
        print(str(dt).split(".")[-1], b, {k: round(v, 4) for k, v in row.items()}, flush=True)
