"""Time v2v_upsample2x_nhwc_hip at the decoder levels of the E2VID-shaped network (8 clips, 256x256 input), per rows-per-work-item
setting (V2V_UP_RS knob of a tuning build: make -C v2v_amd/csrc EXTRA=-DV2V_TUNING_KNOBS; unset = the launcher's choice) and check every setting against rs = 1 bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import convlstm as CL


def t(fn, reps=40):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    x = sorted(s.elapsed_time(e) for s, e in ev)
    return x[len(x) // 2] * 1e3


for shape in ((8, 32, 32, 256), (8, 64, 64, 128), (8, 128, 128, 64)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(shape, generator=g).to(torch.bfloat16).cuda()
    s = torch.randn(shape, generator=g).to(torch.bfloat16).cuda()
    mb = (x.numel() * 2 * 2 + x.numel() * 4 * 2) / 1e6
    os.environ["V2V_UP_RS"] = "1"
    ref = CL.upsample2x_nhwc(x, s)
    row = {}
    for rs in ("1", "2", "4", "8", "16", "32", ""):
        if rs:
            os.environ["V2V_UP_RS"] = rs
        else:
            os.environ.pop("V2V_UP_RS", None)
        assert torch.equal(CL.upsample2x_nhwc(x, s), ref), rs
        us = t(lambda: CL.upsample2x_nhwc(x, s))
        row[rs or "auto"] = f"{us:.1f}us/{mb / us:.2f}TB/s"
    print(shape, f"{mb:.0f} MB", row, flush=True)
