import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from v2v_amd import convlstm as CL
def t(fn, reps=30):
    for _ in range(5): fn()
    ev=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s,e in ev: s.record(); fn(); e.record()
    torch.cuda.synchronize(); x=sorted(s.elapsed_time(e) for s,e in ev); return x[len(x)//2]
for (b,c,h,w) in ((8,64,128,128),(8,128,64,64),(8,256,32,32),(8,64,64,64),(8,128,32,32),(8,256,16,16)):
    g = torch.Generator().manual_seed(c)
    x = torch.randn((b,c,h,w), generator=g).cuda(); hp = torch.tanh(torch.randn((b,c,h,w), generator=g)).cuda(); cp = torch.randn((b,c,h,w), generator=g).cuda()
    wgt = ((torch.rand((4*c,2*c,3,3), generator=g)*2-1)*(3.0/(18*c)**0.5)).cuda(); bias = torch.zeros(4*c).cuda()
    packed = CL.pack_gate_weights(wgt); xn, hn = CL.nchw_to_nhwc_bf16(x), CL.nchw_to_nhwc_bf16(hp); cn = cp.permute(0,2,3,1).contiguous()
    flops = 2.0*b*h*w*(18*c)*(4*c)
    row = {}
    ref = None
    for tr in (0, 64, 128, 256):
        try:
            out = CL.convlstm_step(xn, hn, cn, packed, bias, nchw_dtype=None, tile_rows=tr)
            if ref is None: ref = out
            same = torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
            ms = t(lambda: CL.convlstm_step(xn, hn, cn, packed, bias, nchw_dtype=None, tile_rows=tr))
            row[tr] = f"{ms:.4f}ms/{flops/ms/1e9:.0f}TF" + ("" if same else "/DIFF")
        except Exception as e:
            row[tr] = "n/a"
    ms0 = t(lambda: CL.convlstm_step(xn, None, None, packed, bias, nchw_dtype=None, tile_rows=256)) if (b*h*w) % 256 == 0 else None
    print((b,c,h,w), row, "zero state (half the chunks), 256-px tile:", None if ms0 is None else round(ms0, 4), flush=True)
