"""EXPERIMENT: config 4's pipeline (front-end -> simulator, 24 clips) as ONE chain of two launches against TWO half-batch chains on
two streams (front-end of one half beside the simulator of the other), both as hipGraph replays.  Run on the GPU box."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import esim, frontend  # noqa: E402

b, n, sh, sw, h = 24, 41, 720, 1280, 256
P = [0.2, 0.3, 0.05, 5e-4, 1.0]
dev = torch.device("cuda")
gray_video = esim.synth_clips(b, n, sh, sw, dtype=torch.uint8, seed=20240001, clip_id0=0)
raw = gray_video.unsqueeze(-1).expand(b, n, sh, sw, 3).contiguous()
del gray_video
g = np.random.default_rng(20240001)
keep_h = int(sh * 0.54)
min_scale = max(0, h / keep_h, h / sw)
scale = g.uniform(min_scale, max(1.3, min_scale), size=b)
cb = (h / scale).astype(np.int64)
table = np.stack([[g.integers(0, keep_h - c + 1), g.integers(0, sw - c + 1), c, int(g.random() > 0.5)] for c in cb]).astype(np.int32)
idx = np.tile(np.arange(n, dtype=np.int32), (b, 1))
table_d, idx_d = torch.as_tensor(table, device=dev), torch.as_tensor(idx, device=dev)
cb_max = int(cb.max())
pt = torch.tensor(P, dtype=torch.float64, device=dev)
out = torch.empty((b, 8, 5, h, h), dtype=torch.float32, device=dev)


def chain(lo, hi):
    gray = frontend.prepare_clips_batch(raw[lo:hi], table_d[lo:hi], idx_d[lo:hi], h, "gray", validate=False, max_crop_before=cb_max)[1]
    esim.esim_voxel_batch(gray, pt, bin_mode="sum", num_bins=5, frames_per_bin=1, seed=20240001, clip_id0=lo, out=out[lo:hi], validate=False, no_noise=False)


def one():
    chain(0, b)


s2 = torch.cuda.Stream()


def split(parts):
    def run():
        cur = torch.cuda.current_stream()
        ev0 = torch.cuda.Event(); ev0.record(cur)
        bounds = [b * i // parts for i in range(parts + 1)]
        evs = []
        for i in range(parts):
            st = cur if i % 2 == 0 else s2
            if st is not cur:
                st.wait_event(ev0)
            with torch.cuda.stream(st):
                chain(bounds[i], bounds[i + 1])
                if st is not cur:
                    e = torch.cuda.Event(); e.record(st); evs.append(e)
        for e in evs:
            cur.wait_event(e)
    return run


def graphed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn()
    return gr.replay


def time_ms(fn, reps=40):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2]


ref = None
for name, fn in (("one chain", one), ("2 halves / 2 streams", split(2)), ("4 quarters / 2 streams", split(4))):
    rep = graphed(fn)
    ms = time_ms(rep)
    torch.cuda.synchronize()
    cur = out.clone()
    if ref is None:
        ref = cur
    print(f"{name:24s} {ms:.4f} ms  identical to one chain: {bool(torch.equal(cur, ref))}", flush=True)
