"""Kernel time of the ESIM launch at the reference's training shape (N x 201 x 128 x 128 uint8 -> 40 x 5 SUM bins) for 6..96
clips and every work-item mapping (4 / 2 / 1 pixels), same box, interleaved.  Run on the GPU box: python tools/train_shape_time.py"""
import json
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


out = {}
for dt in (torch.uint8, torch.float32):
    for b in (6, 12, 24, 48, 96):
        frames = esim.synth_clips(b, 201, 128, 128, dtype=dt, seed=1, clip_id0=0)
        pt = torch.tensor(P, dtype=torch.float64, device="cuda")
        o = torch.empty((b, 40, 5, 128, 128), dtype=torch.float32, device="cuda")
        row = {}
        for rnd in range(2):
            for m in ("4px", "2px", "1px", "auto"):
                ms = time_ms(lambda: esim.esim_voxel_batch(frames, pt, bin_mode="sum", num_bins=5, seed=1, out=o, validate=False, no_noise=False, mapping=m))
                row[m] = min(row.get(m, 1e9), ms)
        out[f"{str(dt).split('.')[-1]}_b{b}"] = row
        print(str(dt).split(".")[-1], b, {k: round(v, 4) for k, v in row.items()}, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/train_shape_time.json", "w"), indent=1)
