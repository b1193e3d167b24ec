"""Time the fused ConvLSTM step (v2v_amd.convlstm) against the stock PyTorch graph of the same module
(model/submodules.py:179-235: cat -> Conv2d -> chunk -> 3 sigmoid + 2 tanh -> cell/hidden) at the E2VID encoder shapes."""
import json
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import convlstm as CL  # noqa: E402


class Stock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.Gates = nn.Conv2d(2 * c, 4 * c, 3, padding=1)

    def forward(self, x, state):
        h, c = state
        i, r, o, g = self.Gates(torch.cat((x, h), 1)).chunk(4, 1)
        c = torch.sigmoid(r) * c + torch.sigmoid(i) * torch.tanh(g)
        return torch.sigmoid(o) * torch.tanh(c), c


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2]


def main():
    shapes = [(8, 64, 128, 128), (8, 128, 64, 64), (8, 256, 32, 32), (8, 64, 64, 64), (8, 128, 32, 32), (8, 256, 16, 16)]
    rows = []
    for b, c, h, w in shapes:
        flops = 2.0 * b * h * w * (2 * c * 9) * 4 * c
        torch.manual_seed(0)
        stock = Stock(c).cuda().eval()
        fused = CL.ConvLSTM(c, c, 3).cuda().eval()
        fused.load_state_dict(stock.state_dict())
        x = torch.relu(torch.randn((b, c, h, w), device="cuda"))
        row = {"shape": [b, c, h, w], "gflop": flops / 1e9}
        with torch.no_grad():
            st = (torch.zeros_like(x), torch.zeros_like(x))
            row["stock_fp32_ms"] = timeit(lambda: stock(x, st))
            xb = x.to(torch.bfloat16)
            stb = (torch.zeros_like(xb), torch.zeros_like(xb))
            with torch.autocast("cuda", dtype=torch.bfloat16):
                row["stock_bf16_autocast_ms"] = timeit(lambda: stock(xb, stb))
            state = fused(x, None)
            row["fused_module_fp32io_ms"] = timeit(lambda: fused(x, state))
            stateb = fused(xb, None)
            row["fused_module_bf16io_ms"] = timeit(lambda: fused(xb, stateb))
            xn = CL.nchw_to_nhwc_bf16(x)
            hs, cs, _ = CL.convlstm_step(xn, None, None, fused._weights(), fused.Gates.bias)
            for tr in (0, 64, 128, 256):
                row[f"fused_step_only_t{tr}_ms"] = timeit(lambda: CL.convlstm_step(xn, hs, cs, fused._weights(), fused.Gates.bias, nchw_dtype=torch.bfloat16,
                                                                               tile_rows=tr))
            best = min(v for k, v in row.items() if k.startswith("fused_step_only_t"))
            row["fused_step_tflops"] = flops / best / 1e9
            row["conv_to_nhwc_ms"] = timeit(lambda: CL.nchw_to_nhwc_bf16(x))
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
