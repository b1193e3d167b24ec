"""E2VIDRecurrent.forward_sequence: step-by-step loop vs the two-stream time pipeline, both replayed from a hipGraph, at the training
shape (12 x 40 x 5 x 128 x 128) and at config 5's (8 x 8 x 5 x 256 x 256).  Prints ms per sequence, ms per time step, and checks
that the two give identical images.  usage: python tools/e2vid_pipeline_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd.unet import E2VIDRecurrent  # noqa: E402


def timed_graph(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        res = fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, res


for (b, t, h, w) in ((12, 40, 128, 128), (8, 8, 256, 256)):
    torch.manual_seed(0)
    net = E2VIDRecurrent(dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
                              num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)).cuda().eval()
    ev = torch.round(torch.randn((b, t, 5, h, w), device="cuda") * 2)
    sc = torch.ones((b, 2), device="cuda") * 3
    res = {}
    with torch.no_grad():
        for name, overlap in (("loop", False), ("one_side", 1), ("two_sides", 2), ("three_sides", 3), ("loop", False), ("one_side", 1), ("two_sides", 2), ("three_sides", 3)):
            def run():
                net.reset_states()
                return net.forward_sequence(ev, sc, overlap=overlap)
            ms, img = timed_graph(run)
            res[name] = img.clone()
            print(f"{b}x{t}x5x{h}x{w}  {name:12s} {ms:8.3f} ms / sequence   {ms / t:6.3f} ms / step   {b / (ms * 1e-3):8.0f} samples/s")
    print("identical:", [torch.equal(res["loop"], res[k]) for k in ("one_side", "two_sides", "three_sides")], float(res["loop"].float().abs().mean()))
