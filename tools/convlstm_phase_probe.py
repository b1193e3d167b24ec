"""EXPERIMENT (needs a library built with -DV2V_CL_TIMING, passed through V2V_HIP_LIB): where a wave of the fused ConvLSTM step
spends its cycles -- vmcnt(0) wait, barrier, everything else (LDS-DMA issue + ds_read + MFMA)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import _lib, convlstm as CL  # noqa: E402

L = _lib.lib()
for (b, c, h, w) in [(8, 64, 128, 128), (8, 128, 64, 64), (8, 256, 32, 32)]:
    torch.manual_seed(0)
    x = CL.nchw_to_nhwc_bf16(torch.relu(torch.randn((b, c, h, w), device="cuda")))
    wgt = torch.randn((4 * c, 2 * c, 3, 3), device="cuda") * 0.02
    packed, bias = CL.pack_gate_weights(wgt), torch.zeros(4 * c, device="cuda")
    hs, cs, _ = CL.convlstm_step(x, None, None, packed, bias)
    for tr in (64, 128, 256):
        out = (C.c_ulonglong * 4)()
        torch.cuda.synchronize()
        L.v2v_convlstm_debug_read(out, 1)
        for _ in range(5):
            CL.convlstm_step(x, hs, cs, packed, bias, nchw_dtype=torch.bfloat16, tile_rows=tr)
        torch.cuda.synchronize()
        L.v2v_convlstm_debug_read(out, 1)
        wait, bar, rest, n = [float(v) for v in out]
        tot = wait + bar + rest
        print(f"{(b, c, h, w)} tile {tr}: per wave {tot / n:9.0f} cycles  vmcnt wait {wait / tot:5.1%}  barrier {bar / tot:5.1%}  rest {rest / tot:5.1%}")
