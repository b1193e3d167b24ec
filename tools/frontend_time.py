"""Kernel time of the decode-side front-end at config 4's geometry (24 clips x 41 frames of 720p BGR -> 256x256) for both colour modes,
with and without the resized frames as an output.  Run on the GPU box; with V2V_HIP_LIB pointing at a -DV2V_FRONTEND_FORCE_GATHER
build it times the per-pixel gather kernel instead of the LDS-tiled one."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import esim, frontend  # noqa: E402

b, n, sh, sw, crop = 24, int(os.environ.get("V2V_FT_N", "41")), 720, 1280, 256
gray_video = esim.synth_clips(b, n, sh, sw, dtype=torch.uint8, seed=20240001, clip_id0=0)
raw = torch.stack([gray_video, gray_video.flip(-1), 255 - gray_video], dim=-1).contiguous()
del gray_video
g = np.random.default_rng(20240001)
keep_h = int(sh * 0.54)
scale = g.uniform(crop / keep_h, 1.3, size=b)
cb = (crop / scale).astype(np.int64)
table = torch.as_tensor(np.stack([[g.integers(0, keep_h - c + 1), g.integers(0, sw - c + 1), c, int(g.random() > 0.5)] for c in cb]).astype(np.int32), device="cuda")
idx = torch.as_tensor(np.tile(np.arange(n, dtype=np.int32), (b, 1)), device="cuda")


def t(fn, reps=20):
    for _ in range(3):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    x = sorted(s.elapsed_time(e) for s, e in ev)
    return x[len(x) // 2]


for mode, imgs in (("gray", False), ("gray_in_bgr_out", False), ("gray_in_bgr_out", True)):
    ms = t(lambda: frontend.prepare_clips_batch(raw, table, idx, crop, mode, want_imgs=imgs, validate=False, max_crop_before=int(cb.max())))
    print(f"{os.environ.get('V2V_HIP_LIB', 'main')[-24:]:24s} {mode:16s} want_imgs={imgs!s:5s} {ms:.4f} ms", flush=True)
