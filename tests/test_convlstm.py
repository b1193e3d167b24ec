"""Fused ConvLSTM step against the reference's module semantics (model/submodules.py:179-235), restated with
stock PyTorch ops: Conv2d(2C -> 4C, 3x3, pad 1) over cat(x, h), chunk(4) -> in / remember / out / cell gates, sigmoid x3 + tanh,
cell = remember * c + in * cell_gate, hidden = out * tanh(cell).

Tolerances (floating point; the kernel multiplies in bf16 and accumulates in fp32):
  * against the fp64 evaluation of the SAME bf16-rounded operands: 2e-5 absolute (summation order + hardware exp/rcp)
  * against the plain fp32 module on unrounded operands: 2e-2 absolute (bf16 has 8 bits of mantissa; gates are in [-1, 1])
"""
import numpy as np
import pytest

TOL_SAME_OPERANDS = 2e-5
TOL_FP32_MODULE = 2e-2


def _ref_step(x, h, c, weight, bias, dtype):
    """The reference's forward (:211-230) in `dtype` on CPU; x, h, c [B,C,H,W]."""
    import torch
    import torch.nn.functional as F
    gates = F.conv2d(torch.cat([x, h], 1).to(dtype), weight.to(dtype), bias.to(dtype), padding=1)
    i, r, o, g = gates.chunk(4, 1)
    cell = torch.sigmoid(r) * c.to(dtype) + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(cell), cell


def _bf16_round(t):
    import torch
    return t.to(torch.bfloat16).to(torch.float32)


def _case(b, c, h, w, seed, scale=1.0):
    import torch
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((b, c, h, w), generator=g) * scale
    hp = torch.tanh(torch.randn((b, c, h, w), generator=g))
    cp = torch.randn((b, c, h, w), generator=g)
    k = 1.0 / np.sqrt(2 * c * 9)
    weight = (torch.rand((4 * c, 2 * c, 3, 3), generator=g) * 2 - 1) * k * 3
    bias = (torch.rand((4 * c,), generator=g) * 2 - 1) * 0.5
    return x, hp, cp, weight, bias


@pytest.mark.gpu
@pytest.mark.parametrize("shape,tile_rows", [((2, 64, 16, 16), 0), ((2, 64, 16, 16), 128), ((2, 64, 16, 16), 64), ((1, 128, 8, 24), 0),
                                               ((3, 64, 12, 16), 64), ((1, 256, 8, 8), 0), ((2, 64, 64, 64), 0), ((2, 64, 16, 16), 256), ((1, 128, 16, 32), 256), ((4, 128, 32, 32), 128), ((4, 128, 32, 32), 0),
                                               ((8, 64, 64, 64), 0),
                                               # pixel counts that are not multiples of the tile (a partial last tile): 180 x 240 real-data frames at 1/8
                                               # (720 pixels), 36 pixels on every tile size, 3 x 10 x 14 on the 256-pixel tile
                                               ((1, 256, 24, 30), 0), ((1, 64, 6, 6), 0), ((1, 64, 6, 6), 64), ((1, 128, 6, 6), 128), ((3, 64, 10, 14), 256), ((5, 128, 12, 10), 0)])
def test_step_matches_reference_semantics(shape, tile_rows):
    import torch
    from v2v_amd import convlstm as CL
    b, c, h, w = shape
    x, hp, cp, weight, bias = _case(b, c, h, w, seed=sum(shape) + tile_rows)
    dev = "cuda"
    packed = CL.pack_gate_weights(weight.to(dev))
    xn, hn = CL._to_nhwc_bf16(x.to(dev)).contiguous(), CL._to_nhwc_bf16(hp.to(dev)).contiguous()      # the layout kernel where it applies (H*W % 64), a torch copy otherwise
    cn = cp.to(dev).permute(0, 2, 3, 1).contiguous()
    h_state, c_state, h_nchw = CL.convlstm_step(xn, hn, cn, packed, bias.to(dev), tile_rows=tile_rows)
    torch.cuda.synchronize()
    want_h, want_c = _ref_step(_bf16_round(x), _bf16_round(hp), cp, _bf16_round(weight), bias, torch.float64)
    got_h, got_c = h_nchw.cpu().double(), c_state.permute(0, 3, 1, 2).cpu().double()
    assert float((got_c - want_c).abs().max()) < TOL_SAME_OPERANDS
    assert float((got_h - want_h).abs().max()) < TOL_SAME_OPERANDS
    # the bf16 state the next step reads is the round-to-nearest-even of the fp32 hidden output, in NHWC
    assert torch.equal(h_state, h_nchw.permute(0, 2, 3, 1).to(torch.bfloat16))
    f32_h, f32_c = _ref_step(x, hp, cp, weight, bias, torch.float32)
    assert float((got_h.float() - f32_h).abs().max()) < TOL_FP32_MODULE and float((got_c.float() - f32_c).abs().max()) < TOL_FP32_MODULE


@pytest.mark.gpu
def test_zero_state_and_in_place_cell():
    import torch
    from v2v_amd import convlstm as CL
    x, _, _, weight, bias = _case(2, 64, 8, 16, seed=5)
    dev = "cuda"
    packed, xn = CL.pack_gate_weights(weight.to(dev)), CL.nchw_to_nhwc_bf16(x.to(dev))
    zh, zc = torch.zeros_like(xn), torch.zeros(xn.shape, dtype=torch.float32, device=dev)
    a = CL.convlstm_step(xn, None, None, packed, bias.to(dev))                      # prev_state=None (:196-209): K over x only
    bb = CL.convlstm_step(xn, zh, zc, packed, bias.to(dev))
    # same sums in another order (on small inputs the workgroup is two K groups walking alternate chunks): fp32 rounding apart --
    # 2e-6 on the fp32 outputs, one bf16 ulp of a value in [-1, 1] on the bf16 hidden state
    assert float((a[1] - bb[1]).abs().max()) < 2e-6 and float((a[2] - bb[2]).abs().max()) < 2e-6
    assert float((a[0].float() - bb[0].float()).abs().max()) <= 2.0 ** -8
    c_buf = torch.randn(xn.shape, dtype=torch.float32, device=dev)
    keep = c_buf.clone()
    out_of_place = CL.convlstm_step(xn, a[0], keep, packed, bias.to(dev))
    in_place = CL.convlstm_step(xn, a[0], c_buf, packed, bias.to(dev), c_out=c_buf)
    assert in_place[1].data_ptr() == c_buf.data_ptr() and torch.equal(in_place[1], out_of_place[1]) and torch.equal(in_place[0], out_of_place[0])


@pytest.mark.gpu
def test_layout_and_packing_kernels_are_exact():
    import torch
    from v2v_amd import convlstm as CL
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 128, 8, 16), generator=g).cuda()
    x[0, 0, 0, 0], x[0, 1, 0, 0] = float("nan"), float("inf")
    got = CL.nchw_to_nhwc_bf16(x)
    want = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    nan = torch.isnan(want)
    assert int(nan.sum()) == 1 and torch.equal(torch.isnan(got), nan)
    assert torch.equal(got.view(torch.int16)[~nan], want.contiguous().view(torch.int16)[~nan])       # inf, signs, ties-to-even
    assert torch.equal(CL.nchw_to_nhwc_bf16(x[1:], relu=True), torch.relu(x[1:]).permute(0, 2, 3, 1).to(torch.bfloat16).contiguous())
    xb = x[1:].to(torch.bfloat16)
    assert torch.equal(CL.nchw_to_nhwc_bf16(xb), xb.permute(0, 2, 3, 1).contiguous())                  # bf16 source: a pure transpose
    c = 128
    w = torch.randn((4 * c, 2 * c, 3, 3), generator=g).cuda()
    packed = CL.pack_gate_weights(w)
    # layout of v2v_convlstm.hpp: [col tile][tap][chunk][wn][gate][c32][k] <- weight[gate*C + t*64 + wn*32 + c32, cc*64 + k, ky, kx]
    v = w.view(4, c // 64, 2, 32, 2 * c // 64, 64, 3, 3).permute(1, 6, 7, 4, 2, 0, 3, 5).contiguous().to(torch.bfloat16).reshape(-1)
    assert torch.equal(packed, v)


@pytest.mark.gpu
def test_module_is_a_drop_in_over_a_sequence():
    """Same constructor / parameter names / forward contract as the reference's ConvLSTM; 6 recurrent steps against the stock
    fp32 module with the SAME weights (error does not blow up through the recurrence); the cached bf16 state and a cloned
    float32 prev_state give identical bits."""
    import torch
    import torch.nn as nn
    from v2v_amd import convlstm as CL

    class StockConvLSTM(nn.Module):                                              # restatement of model/submodules.py:179-235
        def __init__(self, input_size, hidden_size, kernel_size):
            super().__init__()
            self.Gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size, padding=kernel_size // 2)

        def forward(self, input_, prev_state=None):
            if prev_state is None:
                prev_state = (torch.zeros_like(input_), torch.zeros_like(input_))
            i, r, o, g = self.Gates(torch.cat((input_, prev_state[0]), 1)).chunk(4, 1)
            cell = torch.sigmoid(r) * prev_state[1] + torch.sigmoid(i) * torch.tanh(g)
            return torch.sigmoid(o) * torch.tanh(cell), cell

    torch.manual_seed(11)
    stock = StockConvLSTM(64, 64, 3).cuda().eval()
    fused = CL.ConvLSTM(64, 64, 3).cuda().eval()
    fused.load_state_dict(stock.state_dict())                                   # same parameter names
    xs = torch.relu(torch.randn((6, 2, 64, 16, 32), device="cuda"))
    with torch.no_grad():
        s_ref = s_fused = s_clone = None
        for t in range(6):
            h_ref, c_ref = stock(xs[t], s_ref)
            s_ref = (h_ref, c_ref)
            h, cell = fused(xs[t], s_fused)
            s_fused = (h, cell)
            assert h.shape == h_ref.shape and cell.shape == c_ref.shape and h.dtype == torch.float32 and h.is_contiguous()
            assert float((h - h_ref).abs().max()) < TOL_FP32_MODULE and float((cell - c_ref).abs().max()) < 2 * TOL_FP32_MODULE
            h2, cell2 = fused(xs[t], s_clone)                                      # prev_state rebuilt from float32 clones
            assert torch.equal(h2, h) and torch.equal(cell2, cell)
            s_clone = (h.clone(), cell.clone())
        # under autocast the stock layers hand over bfloat16: hidden comes back in bfloat16 (the RNE of the float32 result),
        # the cell state stays float32
        hb, cb = fused(xs[1].to(torch.bfloat16), (s_fused[0].to(torch.bfloat16), s_fused[1]))
        hf, cf = fused(xs[1].to(torch.bfloat16).float(), (s_fused[0].to(torch.bfloat16).float(), s_fused[1]))
        assert hb.dtype == torch.bfloat16 and cb.dtype == torch.float32 and torch.equal(hb, hf.to(torch.bfloat16)) and torch.equal(cb, cf)
    with pytest.raises(RuntimeError):
        fused(xs[0])                                                             # grad mode: loud, no silent graph break
    with pytest.raises(ValueError), torch.no_grad():
        CL.ConvLSTM(32, 32, 3).cuda().eval()(xs[0][:, :32].contiguous())          # hidden_size % 64 != 0: refused, no fallback
    with pytest.raises(ValueError):
        CL.ConvLSTM(64, 64, 5)


@pytest.mark.gpu
def test_consumer_with_fused_blocks_tracks_the_stock_consumer():
    """tools/e2vid_consumer.py with every layer on the device kernels (bf16 operands, fp32 accumulation) against the all-stock fp32
    network with the same weights, 4 recurrent time steps: the prediction differs by bf16 operand rounding only -- within 3 % of
    its spread, and by less than the stock network itself moves under torch's bf16 autocast (the precision the reference trains in)."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from e2vid_consumer import E2VIDShapedConsumer, forward_sequence
    torch.manual_seed(2)
    stock = E2VIDShapedConsumer().cuda().eval()
    fused = E2VIDShapedConsumer(fused_convlstm=True).cuda().eval()
    fused.load_stock_state_dict(stock.state_dict())
    events = torch.round(torch.randn((2, 4, 5, 64, 64), device="cuda") * 2)
    with torch.no_grad():
        want, got = forward_sequence(stock, events), forward_sequence(fused, events)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            auto = forward_sequence(stock, events)
    for a, b, c in zip(want, got, auto):
        err = float((a - b.float()).abs().max())
        assert a.shape == b.shape and err < 0.03 * float(a.std()) + 1e-3 and err < float((a - c.float()).abs().max())


def test_shape_errors_are_reported_without_a_gpu():
    """Argument checks of the C ABI run before any HIP call."""
    import ctypes as C
    from v2v_amd import _lib
    L = _lib.lib()
    n = C.c_uint64(0)
    assert L.v2v_convlstm_packed_bytes(64, C.byref(n)) == 0 and n.value == 4 * 64 * 2 * 64 * 9 * 2
    assert L.v2v_convlstm_packed_bytes(48, C.byref(n)) == _lib.ERR_SHAPE
    buf = (C.c_char * 4096)()
    p = C.cast(buf, C.c_void_p)
    assert L.v2v_convlstm_step_hip(p, None, None, p, p, 1, 8, 8, 32, p, p, None, 1, 0, None) == _lib.ERR_SHAPE          # C % 64
    assert L.v2v_convlstm_step_hip(p, None, None, p, p, 1, 5, 5, 64, p, p, None, 1, 0, None) == _lib.ERR_SHAPE          # H*W % 4
    assert L.v2v_convlstm_step_hip(p, None, None, p, p, 1, 8, 8, 64, p, p, None, 1, 32, None) == _lib.ERR_PARAM
    assert L.v2v_convlstm_step_hip(p, p, None, p, p, 1, 8, 8, 64, p, p, None, 1, 0, None) == _lib.ERR_PARAM             # h_state aliases
    assert L.v2v_convlstm_step_hip(None, None, None, p, p, 1, 8, 8, 64, p, p, None, 1, 0, None) == _lib.ERR_NULL
    assert b"x/packed" in L.v2v_last_error()
    assert L.v2v_convlstm_step_hip(p, None, None, p, p, 1, 8, 8, 64, p, p, p, _lib.U8, 0, None) == _lib.ERR_DTYPE
    assert L.v2v_nchw_to_nhwc_bf16_hip(p, _lib.F64, 1, 64, 8, 8, 0, p, None) == _lib.ERR_DTYPE
    assert L.v2v_nchw_to_nhwc_bf16_hip(p, _lib.F32, 1, 32, 8, 8, 0, p, None) == _lib.ERR_SHAPE
    assert L.v2v_conv3x3_nhwc_hip(p, p, p, None, 1, 1, 5, 5, 64, 128, C.cast((C.c_char * 64)(), C.c_void_p), 0, None) == _lib.ERR_SHAPE    # H*W % 4
    assert L.v2v_conv3x3_nhwc_hip(p, p, p, None, 1, 1, 8, 8, 64, 96, C.cast((C.c_char * 64)(), C.c_void_p), 0, None) == _lib.ERR_SHAPE     # Cout not 32 / 64 / 128 / 256k
    assert L.v2v_conv3x3_nhwc_hip(p, p, p, None, 1, 1, 8, 8, 64, 256, p, 0, None) == _lib.ERR_PARAM                                       # out aliases x
    assert L.v2v_conv3x3_pack_weights_hip(p, 48, 256, p, None) == _lib.ERR_SHAPE


@pytest.mark.gpu
def test_recurrence_captures_into_a_hip_graph():
    """A three-step recurrence of the fused module (layout change, step kernel, all three tile sizes via the shapes) captured in
    a hipGraph replays to the same bytes as the eager run: the entry points only enqueue kernels on the caller's stream."""
    import torch
    from v2v_amd import convlstm as CL
    torch.manual_seed(4)
    for c, h, w in ((64, 16, 32), (64, 64, 64), (128, 64, 64)):          # 64-, 128- and 256-pixel tiles at batch 8
        cell = CL.ConvLSTM(c, c, 3).cuda().eval()
        xs = torch.relu(torch.randn((3, 8, c, h, w), device="cuda"))
        outs = [torch.empty((8, c, h, w), device="cuda") for _ in range(2)]

        def run():
            state = None
            with torch.no_grad():
                for t in range(3):
                    state = cell(xs[t], state)
            outs[0].copy_(state[0])
            outs[1].copy_(state[1])

        run()
        torch.cuda.synchronize()
        want = [o.clone() for o in outs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run()
        for o in outs:
            o.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(outs[0], want[0]) and torch.equal(outs[1], want[1]) and float(want[0].abs().sum()) > 0


@pytest.mark.gpu
def test_channels_last_bf16_path_is_in_place_and_equal():
    """A bfloat16 channels-last input (torch.channels_last network under autocast) is consumed as the kernel's NHWC layout without a
    layout-change kernel and the hidden state comes back as a channels-last view of the kernel's own buffer: same values as the
    NCHW path, over a 4-step recurrence, including a prev_state rebuilt from clones."""
    import torch
    from v2v_amd import convlstm as CL
    torch.manual_seed(21)
    cell = CL.ConvLSTM(64, 64, 3).cuda().eval()
    xs = torch.relu(torch.randn((4, 2, 64, 16, 32), device="cuda")).to(torch.bfloat16)
    with torch.no_grad():
        s_a = s_b = s_c = None
        for t in range(4):
            h_a, c_a = cell(xs[t], s_a)                                                        # NCHW bf16
            x_cl = xs[t].contiguous(memory_format=torch.channels_last)
            h_b, c_b = cell(x_cl, s_b)                                                         # channels-last bf16: in place
            assert h_b.dtype == torch.bfloat16 and h_b.is_contiguous(memory_format=torch.channels_last) and not h_b.is_contiguous()
            assert torch.equal(h_a, h_b.contiguous()) and torch.equal(c_a, c_b)
            h_c, c_c = cell(x_cl, s_c)
            assert torch.equal(h_c, h_b) and torch.equal(c_c, c_b)
            s_a, s_b = (h_a, c_a), (h_b, c_b)
            s_c = (h_b.clone(memory_format=torch.preserve_format), c_b.clone())                # not the cached object: re-read from its values
        h_r, _ = cell(xs[0].contiguous(memory_format=torch.channels_last), None, input_relu=True)
        h_n, _ = cell(xs[0], None, input_relu=True)
        assert torch.equal(h_r.contiguous(), h_n)


# ---- residual blocks on the same kernel (model/submodules.py:143-177) -----------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape,tile_rows,res,relu", [((2, 256, 8, 16), 0, True, True), ((2, 64, 16, 16), 64, False, True), ((1, 128, 16, 16), 128, True, False),
                                                        ((8, 256, 32, 32), 0, True, True), ((4, 256, 16, 16), 256, False, False)])
def test_conv3x3_matches_reference_semantics(shape, tile_rows, res, relu):
    """out = [relu](conv3x3(x) + bias [+ residual]) against the float64 evaluation of the same bf16-rounded operands (2e-5 before
    the output's own bf16 rounding: compared at 1 bf16 ulp = 2^-8 relative) -- Cin from the shape, Cout = 256."""
    import torch
    import torch.nn.functional as F
    from v2v_amd import convlstm as CL
    b, cin, h, w = shape
    cout = 256
    g = torch.Generator().manual_seed(sum(shape) + tile_rows)
    x = torch.randn((b, cin, h, w), generator=g)
    r = torch.randn((b, cout, h, w), generator=g)
    weight = (torch.rand((cout, cin, 3, 3), generator=g) * 2 - 1) * (3.0 / np.sqrt(cin * 9))
    bias = (torch.rand((cout,), generator=g) * 2 - 1) * 0.5
    xn = CL.nchw_to_nhwc_bf16(x.cuda())
    rn = CL.nchw_to_nhwc_bf16(r.cuda()) if res else None
    out = CL.conv3x3_nhwc(xn, CL.pack_conv3x3_weights(weight.cuda()), bias.cuda(), residual=rn, relu=relu, tile_rows=tile_rows)
    want = F.conv2d(_bf16_round(x).double(), _bf16_round(weight).double(), bias.double(), padding=1)
    if res:
        want = want + _bf16_round(r).double()
    if relu:
        want = torch.relu(want)
    got = out.permute(0, 3, 1, 2).float().cpu().double()
    assert float(((got - want).abs() / (want.abs() + 1.0)).max()) < 2.0 ** -8


@pytest.mark.gpu
def test_residual_block_is_a_drop_in():
    """Same constructor / parameter names / forward contract as the reference's ResidualBlock; against the stock fp32 block with the
    same weights: 2e-2 absolute on unit-scale activations (bf16 operands, two convolutions); channels-last bf16 in place."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from v2v_amd import convlstm as CL

    class StockRes(nn.Module):                                                      # restatement of model/submodules.py:143-177, norm=None
        def __init__(self, c):
            super().__init__()
            self.conv1, self.conv2 = nn.Conv2d(c, c, 3, padding=1), nn.Conv2d(c, c, 3, padding=1)

        def forward(self, x):
            return F.relu(self.conv2(F.relu(self.conv1(x))) + x)

    torch.manual_seed(8)
    stock, fused = StockRes(256).cuda().eval(), CL.ResidualBlock(256, 256).cuda().eval()
    fused.load_state_dict(stock.state_dict())
    x = torch.relu(torch.randn((4, 256, 16, 32), device="cuda"))
    with torch.no_grad():
        want = stock(x)
        got = fused(x)
        assert got.dtype == torch.float32 and got.is_contiguous() and float((got - want).abs().max()) < TOL_FP32_MODULE * float(want.abs().max())
        x_cl = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        got_cl = fused(x_cl)
        assert got_cl.dtype == torch.bfloat16 and got_cl.is_contiguous(memory_format=torch.channels_last)
        assert float((got_cl.float() - want).abs().max()) < 2 * TOL_FP32_MODULE * float(want.abs().max())
    with pytest.raises(RuntimeError):
        fused(x)
    with pytest.raises(ValueError):
        CL.ResidualBlock(256, 128)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout,ks,stride,tile_rows,res,relu", [
    ((2, 64, 32, 32), 128, 5, 2, 0, False, True),       # enc2 of the recurrent UNet: 64 -> 128, 5x5 stride 2
    ((1, 128, 32, 32), 256, 5, 2, 0, False, True),      # enc3: 128 -> 256
    ((1, 128, 16, 32), 256, 5, 2, 64, False, False),
    ((1, 256, 16, 16), 128, 5, 1, 0, False, True),      # dec1 after upsampling: 256 -> 128
    ((1, 128, 16, 16), 64, 5, 1, 0, True, True),        # dec2: 128 -> 64
    ((1, 64, 16, 32), 32, 5, 1, 0, False, True),        # dec3: 64 -> 32
    ((1, 64, 16, 16), 32, 3, 1, 0, True, False),
    ((2, 64, 24, 40), 64, 3, 2, 0, False, True),        # Wout = 20, 480 pixels: 3.75 tiles of 128 (a partial last tile)
    ((2, 64, 32, 24), 64, 3, 2, 0, True, True),         # 2 x 16 x 12 = 384 pixels: three 128-pixel tiles
    ((1, 128, 16, 16), 32, 5, 1, 256, False, True),     # pinned 256-pixel tile
    ((1, 64, 32, 32), 128, 5, 1, 128, True, False),     # pinned 128-pixel tile
    ((2, 64, 31, 33), 256, 5, 2, 64, False, True),      # odd input size: Hout x Wout = 16 x 17 -> 544 pixels = 8.5 tiles of 64
    ((4, 64, 15, 31), 256, 5, 2, 128, True, True),      # odd input size, 4 x 8 x 16 = 512 output pixels
    ((2, 64, 31, 33), 256, 5, 2, 0, True, True),        # 544 = 17 x 32 output pixels: the 32-pixel tile
    ((1, 256, 16, 16), 256, 3, 1, 32, True, True),      # residual-block shape on the 32-pixel tile
    ((1, 128, 32, 32), 512, 5, 2, 0, False, True),      # two packed column tiles, four 64 px x 128 column workgroup tiles each
    ((2, 256, 16, 16), 256, 3, 1, 0, True, True),       # small layer: 64 px x 128 columns (half a packed column tile per workgroup)
    ((2, 32, 32, 32), 64, 5, 2, 0, False, True),        # 32 input channels (enc1): two taps per K chunk, 25 taps -> the last half chunk is zero
    ((1, 32, 16, 16), 128, 3, 1, 256, True, False),     # 32 input channels, 3x3, 128 columns, pinned 256-pixel tile
    ((1, 32, 33, 31), 64, 5, 2, 128, True, True),       # 32 input channels, odd input size, 17 x 16 = 272 pixels = 2.125 tiles of 128
    ((1, 64, 16, 16), 32, 3, 1, 16, True, False),       # halo tiles (16 x 16 patch + halo staged once per channel chunk), pinned
    ((1, 128, 32, 16), 128, 3, 1, 16, False, True),     # halo tiles, 3x3, 128 columns, two channel chunks
    ((2, 192, 16, 32), 64, 5, 1, 16, True, True),       # halo tiles, 5x5, three channel chunks, one tap per weight group
    ((1, 64, 48, 16), 32, 5, 1, 0, False, True),        # halo tiles picked by the launcher (5x5, 32 columns)
    ((1, 64, 24, 16), 32, 5, 1, 128, False, True),      # same layer on the 128-pixel tile when H is not a multiple of 16
    ((1, 64, 18, 10), 64, 3, 2, 0, False, True),        # 9 x 5 = 45 output pixels per image: not whole groups of 4 -> rejected
    ((3, 64, 6, 6), 256, 3, 1, 0, True, True),          # 108 pixels: less than one tile of any size
    ((1, 128, 20, 36), 128, 5, 1, 256, False, True),    # 720 pixels on the pinned 256-pixel tile: 2.8 tiles
    ((5, 64, 12, 20), 32, 3, 1, 0, True, False),        # 1,200 pixels, 32 columns
])
def test_conv_nhwc_matches_reference_semantics(shape, cout, ks, stride, tile_rows, res, relu):
    """out = [relu](conv_ks(x, stride, pad ks // 2) + bias [+ residual]) (ConvLayer.forward, model/submodules.py:25-33) against
    the float64 evaluation of the same bf16-rounded operands, at 1 bf16 ulp (2^-8 relative).  Any B*H*W runs (the last pixel tile may be partial: its rows past the end read zeros and are not stored); only an output image that is not whole groups of 4 pixels is refused, loudly."""
    import torch
    import torch.nn.functional as F
    from v2v_amd import convlstm as CL
    b, cin, h, w = shape
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    g = torch.Generator().manual_seed(sum(shape) + cout + ks + stride)
    x = torch.randn((b, cin, h, w), generator=g)
    r = torch.randn((b, cout, ho, wo), generator=g)
    weight = (torch.rand((cout, cin, ks, ks), generator=g) * 2 - 1) * (3.0 / np.sqrt(cin * ks * ks))
    bias = (torch.rand((cout,), generator=g) * 2 - 1) * 0.5
    xn = _bf16_round(x).cuda().to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()
    rn = _bf16_round(r).cuda().to(torch.bfloat16).permute(0, 2, 3, 1).contiguous() if res else None
    packed = CL.pack_conv_weights(weight.cuda())
    if (ho * wo) % 4:                                          # the only tiling constraint left: whole groups of 4 output pixels per image
        with pytest.raises(ValueError):
            CL.conv_nhwc(xn, packed, bias.cuda(), ks, stride, residual=rn, relu=relu, tile_rows=tile_rows)
        return
    out = CL.conv_nhwc(xn, packed, bias.cuda(), ks, stride, residual=rn, relu=relu, tile_rows=tile_rows)
    assert tuple(out.shape) == (b, ho, wo, cout)
    want = F.conv2d(_bf16_round(x).double(), _bf16_round(weight).double(), bias.double(), stride=stride, padding=ks // 2)
    if res:
        want = want + _bf16_round(r).double()
    if relu:
        want = torch.relu(want)
    got = out.permute(0, 3, 1, 2).float().cpu().double()
    assert float(((got - want).abs() / (want.abs() + 1.0)).max()) < 2.0 ** -8


@pytest.mark.gpu
@pytest.mark.parametrize("shape,with_skip", [((2, 5, 7, 64), True), ((1, 1, 1, 8), False), ((3, 16, 8, 32), False), ((1, 9, 1, 256), True),
                                             ((2, 32, 32, 128), True)])
def test_upsample2x_matches_interpolate(shape, with_skip):
    """up2(x [+ skip]) against f.interpolate(scale_factor=2, mode='bilinear', align_corners=False) of the bf16-rounded sum in
    float32 (model/submodules.py:86-87 behind model/unet.py:304): one bf16 ulp (2^-8 relative) -- the weights are exact in
    binary, the only difference is the summation / contraction order before the output's rounding."""
    import torch
    import torch.nn.functional as F
    from v2v_amd import convlstm as CL
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(shape, generator=g).to(torch.bfloat16).cuda()
    s = torch.randn(shape, generator=g).to(torch.bfloat16).cuda() if with_skip else None
    out = CL.upsample2x_nhwc(x, s)
    b, h, w, c = shape
    assert tuple(out.shape) == (b, 2 * h, 2 * w, c) and out.dtype == torch.bfloat16
    src = (x + s) if with_skip else x
    want = F.interpolate(src.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
    err = ((out.float() - want).abs() / (want.abs() + 2.0 ** -6)).max()
    assert float(err) < 2.0 ** -8
    # rows per work-item (the launcher's choice depends on the size) never changes a bit, ragged last segment included
    import os
    try:
        for rs in ("1", "3", "4", "64"):
            os.environ["V2V_UP_RS"] = rs
            assert torch.equal(CL.upsample2x_nhwc(x, s), out), rs
    finally:
        os.environ.pop("V2V_UP_RS", None)
    with pytest.raises(ValueError):
        CL.upsample2x_nhwc(x.float())
    with pytest.raises(ValueError):
        CL.upsample2x_nhwc(x, torch.cat((x, x), 0))


@pytest.mark.gpu
def test_conv_layer_is_a_drop_in():
    """v2v_amd.convlstm.ConvLayer against nn.Conv2d + ReLU with the same `conv2d` parameters (model/submodules.py:6-33) and, with
    upsample=True, against UpsampleConvLayer (:68-96): 2e-2 absolute on unit-scale activations; channels-last bf16 stays in place."""
    import torch
    import torch.nn.functional as F
    from v2v_amd import convlstm as CL
    torch.manual_seed(7)
    for cin, cout, stride, up, act in ((64, 128, 2, False, "relu"), (128, 64, 1, True, "relu"), (64, 32, 1, True, None)):
        layer = CL.ConvLayer(cin, cout, 5, stride=stride, padding=2, activation=act, upsample=up).cuda().eval()
        assert set(layer.state_dict()) == {"conv2d.weight", "conv2d.bias"}
        x = torch.randn(2, cin, 32, 32, device="cuda")
        with torch.no_grad():
            xi = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False) if up else x
            want = layer.conv2d(xi)
            want = torch.relu(want) if act else want
            got = layer(x)
            assert got.dtype == torch.float32 and got.shape == want.shape and got.is_contiguous()
            assert float((got - want).abs().max()) < 2e-2 * max(1.0, float(want.abs().max()))
            xc = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            got_cl = layer(xc)
            assert got_cl.dtype == torch.bfloat16 and got_cl.is_contiguous(memory_format=torch.channels_last)
            assert float((got_cl.float() - want).abs().max()) < 4e-2 * max(1.0, float(want.abs().max()))
            if up:                                                                  # the decoder's sum skip folded into the upsampling
                half = (0.5 * xc).contiguous(memory_format=torch.channels_last)
                assert torch.equal(layer(half, half), layer(half + half))
                assert float((layer(0.5 * x, 0.5 * x) - want).abs().max()) < 2e-2 * max(1.0, float(want.abs().max()))
            else:
                with pytest.raises(ValueError):
                    layer(x, x)
        with pytest.raises(RuntimeError):
            layer(x.requires_grad_())
    with pytest.raises(ValueError):
        CL.ConvLayer(64, 64, 5, padding=2, norm="BN")
    with pytest.raises(ValueError):
        CL.ConvLayer(64, 64, 7, padding=3)


@pytest.mark.gpu
def test_conv_nhwc_random_shapes():
    """Seeded sweep over (batch, input size, Cin, Cout tile, kernel size, stride, pixel tile, residual, relu): every shape the
    entry point takes -- any pixel count, the last tile partial more often than not -- is within 1 bf16 ulp of the float64 convolution
    of the same bf16 operands; the shapes it does not take (an output image that is not whole groups of 4 pixels) raise ValueError."""
    import torch
    import torch.nn.functional as F
    from v2v_amd import convlstm as CL
    rng = np.random.default_rng(20261003)
    ran = rejected = 0
    for case in range(60):
        cin = int(rng.choice([64, 128, 192, 256]))
        cout = int(rng.choice([32, 64, 128, 256, 512]))
        ks, stride = int(rng.choice([3, 5])), int(rng.choice([1, 2]))
        b, h, w = int(rng.integers(1, 4)), int(rng.integers(4, 41)), int(rng.integers(4, 41))
        if rng.random() < 0.6:                                                     # steer most cases onto a shape the tiles take
            h, w = int(rng.choice([8, 16, 24, 32])) * stride, int(rng.choice([8, 16, 32])) * stride
        tiles = [0, 32, 64, 128, 256] if cout % 256 == 0 else [0, 128, 256]
        if cout % 256 and stride == 1 and h % 16 == 0 and w % 16 == 0 and (ks == 3 or cout <= 64):
            tiles.append(16)                                                       # halo tiles
        tile = int(rng.choice(tiles))
        res, relu = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        g = torch.Generator().manual_seed(case)
        x = torch.randn((b, cin, h, w), generator=g)
        r = torch.randn((b, cout, ho, wo), generator=g)
        weight = (torch.rand((cout, cin, ks, ks), generator=g) * 2 - 1) * (3.0 / np.sqrt(cin * ks * ks))
        bias = (torch.rand((cout,), generator=g) * 2 - 1) * 0.5
        xn = _bf16_round(x).cuda().to(torch.bfloat16).permute(0, 2, 3, 1).contiguous()
        rn = _bf16_round(r).cuda().to(torch.bfloat16).permute(0, 2, 3, 1).contiguous() if res else None
        packed = CL.pack_conv_weights(weight.cuda())
        if (ho * wo) % 4:                                      # whole groups of 4 output pixels per image: the one tiling constraint
            with pytest.raises(ValueError):
                CL.conv_nhwc(xn, packed, bias.cuda(), ks, stride, residual=rn, relu=relu, tile_rows=tile)
            rejected += 1
            continue
        out = CL.conv_nhwc(xn, packed, bias.cuda(), ks, stride, residual=rn, relu=relu, tile_rows=tile)
        want = F.conv2d(_bf16_round(x).double(), _bf16_round(weight).double(), bias.double(), stride=stride, padding=ks // 2)
        want = want + _bf16_round(r).double() if res else want
        want = torch.relu(want) if relu else want
        got = out.permute(0, 3, 1, 2).float().cpu().double()
        err = float(((got - want).abs() / (want.abs() + 1.0)).max())
        assert err < 2.0 ** -8, (case, (b, cin, h, w), cout, ks, stride, tile, res, relu, err)
        ran += 1
    assert ran >= 40 and rejected >= 3, (ran, rejected)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,cout,with_skip,out_dtype", [((2, 16, 16, 32), 1, True, "bfloat16"), ((1, 5, 7, 64), 3, False, "float32"),
                                                            ((1, 1, 3, 8), 2, True, "float32"), ((2, 32, 32, 512), 1, True, "bfloat16")])
def test_conv1x1_prediction_layer(shape, cout, with_skip, out_dtype):
    """pred(skip_sum(x, head)) (model/unet.py:58-64, :307) against the float64 evaluation of the same bf16 operands (sum and
    weights rounded to bf16 as bf16 autocast does): 1e-5 relative for float32 output, 1 bf16 ulp for bfloat16."""
    import torch
    from v2v_amd import convlstm as CL
    g = torch.Generator().manual_seed(sum(shape) + cout)
    x = torch.randn(shape, generator=g).to(torch.bfloat16).cuda()
    s = torch.randn(shape, generator=g).to(torch.bfloat16).cuda() if with_skip else None
    w = (torch.rand((cout, shape[-1], 1, 1), generator=g) - 0.5).cuda()
    b = torch.randn((cout,), generator=g).cuda()
    out = CL.conv1x1_nhwc(x, w, b, s, out_dtype=getattr(torch, out_dtype))
    src = (x + s) if with_skip else x
    want = src.double() @ w.reshape(cout, -1).to(torch.bfloat16).double().t() + b.double()
    assert out.shape == shape[:-1] + (cout,) and out.dtype == getattr(torch, out_dtype)
    tol = 2.0 ** -8 if out_dtype == "bfloat16" else 1e-5
    assert float(((out.double() - want).abs() / (want.abs() + 1.0)).max()) < tol
    layer = CL.ConvLayer(shape[-1], cout, 1, activation=None).cuda().eval()            # the drop-in form, NCHW float32 in / out
    xn = x.permute(0, 3, 1, 2).float().contiguous()
    if (shape[1] * shape[2]) % 64 == 0 and shape[-1] % 64 == 0:
        with torch.no_grad():
            got = layer(xn)
            ref = layer.conv2d(xn)
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()))
    with pytest.raises(ValueError):
        CL.ConvLayer(32, 8, 1, activation=None)


@pytest.mark.gpu
@pytest.mark.parametrize("b,cin,h,w,ks,relu", [(2, 5, 32, 48, 5, True), (1, 8, 16, 16, 3, False), (1, 1, 64, 16, 5, True), (3, 3, 16, 32, 3, True)])
def test_head_convolution(b, cin, h, w, ks, relu):
    """The UNet's head ConvLayer(num_bins, 32, ks, stride 1, padding ks // 2, relu) (model/unet.py:77-78): taps packed along K, the
    input as bf16 NHWC padded to 8 channels.  1 bf16 ulp against the float64 convolution of the same bf16 operands; the layout
    kernel is exact for any input strides; the module form tracks nn.Conv2d + ReLU in float32."""
    import torch
    import torch.nn.functional as F
    from v2v_amd import convlstm as CL
    g = torch.Generator().manual_seed(b + cin + h + w + ks)
    big = torch.randn((b, 3, cin, h, w), generator=g).cuda()
    x = big[:, 1]                                                                   # a strided view, as forward_sequence slices events[:, t]
    x8 = CL.to_nhwc8_bf16(x)
    assert torch.equal(x8[..., :cin], x.permute(0, 2, 3, 1).to(torch.bfloat16)) and not x8[..., cin:].any()
    assert torch.equal(CL.to_nhwc8_bf16(x.contiguous(memory_format=torch.channels_last)), x8)
    weight = ((torch.rand((32, cin, ks, ks), generator=g) * 2 - 1) * (3.0 / np.sqrt(cin * ks * ks))).cuda()
    bias = ((torch.rand((32,), generator=g) * 2 - 1) * 0.5).cuda()
    out = CL.conv_head_nhwc(x8, CL.pack_head_weights(weight), bias, ks, relu=relu)
    want = F.conv2d(x.to(torch.bfloat16).double(), weight.to(torch.bfloat16).double(), bias.double(), padding=ks // 2)
    want = torch.relu(want) if relu else want
    assert float(((out.permute(0, 3, 1, 2).double() - want).abs() / (want.abs() + 1.0)).max()) < 2.0 ** -8
    layer = CL.ConvLayer(cin, 32, ks, stride=1, padding=ks // 2, activation="relu" if relu else None).cuda().eval()
    with torch.no_grad():
        ref = layer.conv2d(x)
        ref = torch.relu(ref) if relu else ref
        got = layer(x)
    assert got.dtype == torch.float32 and got.shape == ref.shape and float((got - ref).abs().max()) < 2e-2 * max(1.0, float(ref.abs().max()))
    with pytest.raises(ValueError):
        CL.ConvLayer(5, 64, 5, padding=2)
    with pytest.raises(ValueError):
        CL.conv_head_nhwc(CL.to_nhwc8_bf16(x[:, :, :15]), CL.pack_head_weights(weight), bias, ks)


@pytest.mark.gpu
def test_whole_consumer_captures_into_a_hip_graph():
    """Every layer of the E2VID-shaped consumer on the device kernels, channels-last, bf16 autocast: a 3-step forward captured
    into one hipGraph (after a warm-up that packs the weights and raises the kernels' LDS limits) replays to the eager result
    bit for bit, also on new input written into the captured buffer."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from e2vid_consumer import E2VIDShapedConsumer, forward_sequence
    torch.manual_seed(5)
    model = E2VIDShapedConsumer(fused_convlstm=True).cuda().eval().to(memory_format=torch.channels_last)
    events = torch.round(torch.randn((2, 3, 5, 64, 64), device="cuda") * 2)

    def run():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return torch.stack([o.float() for o in forward_sequence(model, events, channels_last=True)])
    eager = run()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = run()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    events.copy_(torch.round(torch.randn(events.shape, device="cuda") * 2))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, run()) and not torch.equal(out, eager)
