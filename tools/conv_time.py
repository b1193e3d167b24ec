"""Time v2v_amd.convlstm.conv_nhwc (the matrix-core convolution, EPI = 1) against torch's channels-last bf16 nn.Conv2d at the
encoder / decoder shapes of the recurrent UNet (model/unet.py with config/train_v2v_e2vid_10k.yaml: 5x5, 8 clips of 256^2)."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import convlstm as CL  # noqa: E402
from tools.convlstm_time import timeit  # noqa: E402


def main():
    # (name, B, Cin, Hin, Win, Cout, ks, stride)
    if len(sys.argv) > 1 and sys.argv[1] == "train":      # the reference's training shape: 12 clips of 128^2 per time step
        shapes = [("enc1", 12, 32, 128, 128, 64, 5, 2), ("enc2", 12, 64, 64, 64, 128, 5, 2), ("enc3", 12, 128, 32, 32, 256, 5, 2), ("dec1", 12, 256, 32, 32, 128, 5, 1),
                  ("dec2", 12, 128, 64, 64, 64, 5, 1), ("dec3", 12, 64, 128, 128, 32, 5, 1), ("res", 12, 256, 16, 16, 256, 3, 1)]
    else:
      shapes = [("enc1", 8, 32, 256, 256, 64, 5, 2), ("enc2", 8, 64, 128, 128, 128, 5, 2), ("enc3", 8, 128, 64, 64, 256, 5, 2), ("dec1", 8, 256, 64, 64, 128, 5, 1),
                ("dec2", 8, 128, 128, 128, 64, 5, 1), ("dec3", 8, 64, 256, 256, 32, 5, 1), ("res", 8, 256, 32, 32, 256, 3, 1)]
    for name, b, cin, h, w, cout, ks, stride in shapes:
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        flops = 2.0 * b * ho * wo * cin * ks * ks * cout
        torch.manual_seed(0)
        weight = torch.randn((cout, cin, ks, ks), device="cuda") * 0.02
        bias = torch.randn((cout,), device="cuda")
        x = torch.randn((b, cin, h, w), device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wb = weight.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        bb = bias.to(torch.bfloat16)
        row = {"layer": name, "shape": [b, cin, h, w, cout, ks, stride], "gflop": flops / 1e9}
        with torch.no_grad():
            row["stock_bf16_channels_last_ms"] = timeit(lambda: F.relu(F.conv2d(x, wb, bb, stride=stride, padding=ks // 2)))
            xn = x.permute(0, 2, 3, 1)
            packed = CL.pack_conv_weights(weight)
            trs = (0, 32, 64, 128, 256) if cout % 256 == 0 else (0, 16, 128, 256)
            for tr in trs:
                try:
                    row[f"fused_t{tr}_ms"] = timeit(lambda: CL.conv_nhwc(xn, packed, bias, ks, stride, relu=True, tile_rows=tr))
                except Exception:                                     # a tile this layer does not take (halo tiles: stride 1, <= 64 columns)
                    pass
            best = min(v for k, v in row.items() if k.startswith("fused_t"))
            row["fused_tflops"] = flops / best / 1e9
            row["stock_tflops"] = flops / row["stock_bf16_channels_last_ms"] / 1e9
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
