"""Time the voxel post-ops (normalize_batch_voxel + padding, v2v_amd/postops.py) at the training shape and at config 4's shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import postops  # noqa: E402


def time_ms(fn, reps=30):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    t = sorted(s.elapsed_time(e) for s, e in ev)
    return t[len(t) // 2]


for shape in ((12, 40, 5, 128, 128), (24, 8, 5, 256, 256), (256, 1, 5, 256, 256)):
    g = torch.Generator(device="cuda").manual_seed(1)
    # voxel-like content: mostly zeros, a few +-1 / +-2, rare larger counts
    u = torch.rand(shape, generator=g, device="cuda")
    v = torch.where(u < 0.8, torch.zeros_like(u), torch.round((u - 0.9) * 40.0))
    mb = v.numel() * 4 / 1e6
    for name, kw, passes in (("count select + normalise, in place", dict(normalize=True, method="count", inplace=True, valid_hw=shape[-2:]), 3),
                             ("count select + normalise", dict(normalize=True, method="count"), 3),
                             ("radix select + normalise", dict(normalize=True, method="radix"), 5)):
        if kw.get("inplace"):                                        # in place destroys the integers: restore them before every call
            w = v.clone()
            def run():
                w.copy_(v)
                postops.normalize_and_pad(w, PAD=16, **kw)
            ms = time_ms(run) - time_ms(lambda: w.copy_(v))
        else:
            ms = time_ms(lambda: postops.normalize_and_pad(v, PAD=16, **kw))
        print(shape, f"{mb:.0f} MB", f"{name:36s} {ms:.4f} ms = {passes * mb / ms / 1e3:.2f} TB/s over {passes} passes of the tensor", flush=True)
