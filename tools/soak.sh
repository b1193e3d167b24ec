#!/bin/bash
# Run ON THE GPU BOX: soak of the parity suites -- the fuzz tests at V2V_FUZZ_SCALE (default 20) and the convolution / ConvLSTM
# tests repeated, then a race screen of every convolution tile against the float64 reference under memory load.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
V2V_FUZZ_SCALE=${1:-20} timeout 1500 python -m pytest tests/test_hip_fuzz.py -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do timeout 600 python -m pytest tests/test_convlstm.py -q -m gpu -p no:randomly 2>&1 | tail -1; done
timeout 600 python - <<'PY'
import sys, torch, numpy as np
sys.path.insert(0, ".")
from v2v_amd import convlstm as CL
import torch.nn.functional as F
torch.manual_seed(0)
noise = torch.empty(512 << 20, dtype=torch.uint8, device="cuda"); side = torch.cuda.Stream()
bad = 0
for (b, cin, h, w, cout, ks, stride, tiles) in [(8, 64, 64, 64, 32, 5, 1, (0, 16, 128, 256)), (8, 128, 32, 32, 64, 5, 1, (0, 16, 128, 256)), (4, 256, 32, 32, 128, 5, 1, (0, 128, 256)),
                                                (4, 256, 16, 16, 256, 3, 1, (0, 32, 64, 128, 256)), (4, 128, 32, 32, 256, 5, 2, (0, 32, 64, 128, 256)), (4, 64, 64, 64, 128, 5, 2, (0, 128, 256))]:
    x = torch.randn((b, h, w, cin), device="cuda").to(torch.bfloat16)
    wgt = torch.randn((cout, cin, ks, ks), device="cuda") * (2.0 / (cin * ks * ks) ** 0.5)
    bias = torch.randn((cout,), device="cuda") * 0.1
    packed = CL.pack_conv_weights(wgt)
    want = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), wgt.to(torch.bfloat16).double(), bias.double(), stride=stride, padding=ks // 2)).permute(0, 2, 3, 1)
    first = {}
    for r in range(40):
        tile = tiles[r % len(tiles)]
        if r % 2:
            with torch.cuda.stream(side):
                noise.add_(1)
        out = CL.conv_nhwc(x, packed, bias, ks, stride, relu=True, tile_rows=tile)
        torch.cuda.synchronize()
        err = float(((out.double() - want).abs() / (want.abs() + 1.0)).max())
        if err >= 2.0 ** -8 or (tile in first and not torch.equal(first[tile], out)):
            bad += 1; print("BAD", (b, cin, h, w, cout, ks, stride), tile, r, err, flush=True)
        first.setdefault(tile, out.clone())
print("conv race screen:", "ok" if bad == 0 else f"{bad} bad launches")
PY
