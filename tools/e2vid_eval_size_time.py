"""Run ON THE GPU BOX: ms per reconstructed frame of the package network at the sizes the reference EVALUATES on, batch 1 (HQF / IJRR
180 x 240 and MVSEC 260 x 346, padded to multiples of 16 as model/train_utils.py:322-326 does), as the per-step call `net(voxel)` that
test_e2vid.py's loop makes and as forward_sequence(graph=True) over 40 steps, beside the all-stock network (tools/e2vid_consumer.py)
in float32 and under bf16 autocast."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from e2vid_consumer import E2VIDShapedConsumer  # noqa: E402
from v2v_amd.unet import E2VIDRecurrent  # noqa: E402


def per_step(fn, ev, reset):
    with torch.no_grad():
        for rep in range(3):
            reset()
            if rep == 2:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            for t in range(ev.shape[1]):
                fn(ev[:, t])
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / ev.shape[1] * 1e3


torch.manual_seed(0)
net = E2VIDRecurrent(dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
                          num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)).cuda().eval()
stock = E2VIDShapedConsumer(num_bins=5).cuda().eval()
for name, h, w in (("HQF / IJRR 180x240 -> 192x240", 192, 240), ("MVSEC 260x346 -> 272x352", 272, 352)):
    ev = torch.round(torch.randn((1, 40, 5, h, w), device="cuda") * 2)
    a = per_step(lambda x: net(x), ev, net.reset_states)
    with torch.no_grad():
        for _ in range(3):
            net.forward_sequence(ev, graph=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            net.forward_sequence(ev, graph=True)
        torch.cuda.synchronize()
        b = (time.perf_counter() - t0) / 5 / 40 * 1e3
    c = per_step(lambda x: stock(x), ev, stock.reset_states)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        d = per_step(lambda x: stock(x), ev, stock.reset_states)
    print(f"{name}: package per-step call {a:.3f} ms/frame, forward_sequence(graph) {b:.3f}; stock float32 {c:.3f}, stock bf16 autocast {d:.3f}", flush=True)
