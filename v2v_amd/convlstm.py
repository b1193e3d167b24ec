"""The recurrent UNet's layers on the matrix cores: host side of v2v_convlstm_step_hip, v2v_conv_nhwc_hip and
v2v_upsample2x_nhwc_hip.

    ConvLSTM(input_size, hidden_size, kernel_size)        model/submodules.py:179-235 -- same constructor, same `Gates`
                                                          parameter names (state_dicts load unchanged), same
                                                          forward(input_, prev_state=None) -> (hidden, cell)
    ResidualBlock(in_channels, out_channels)              model/submodules.py:143-177 (norm=None): `conv1` / `conv2`
    ConvLayer(in, out, kernel_size, stride, padding, activation, norm=None, upsample=False)
                                                          model/submodules.py:6-33, and :68-96 (UpsampleConvLayer) with
                                                          upsample=True; `conv2d`; forward(x, skip=None) folds the sum skip
    convlstm_step / conv_nhwc / conv3x3_nhwc / upsample2x_nhwc     the raw NHWC bfloat16 operators
    conv1x1_nhwc                                           the 1x1 prediction layer on skip_sum(x, head) (ConvLayer with kernel_size 1)
    conv_head_nhwc / to_nhwc8_bf16 / pack_head_weights     the head (voxel bins -> 32 channels; ConvLayer with <= 8 input channels)
    pack_gate_weights / pack_conv_weights                  one-off weight packing
    nchw_to_nhwc_bf16(x, relu=False)                      layout change in front of them (not needed for channels-last bf16 input)

The 3x3 gate convolution runs as an implicit GEMM on the bf16 matrix cores with fp32 accumulation and the gate / cell / hidden
update fused on the accumulators (v2v_amd/csrc/v2v_convlstm.hpp).  Inference only (no autograd through the kernel: a call
that would need a gradient raises).  No fallback: shapes the kernel does not take (hidden_size % 64, H*W % 4, kernel_size
!= 3, input_size != hidden_size) raise ValueError.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib


_DTYPES = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def nchw_to_nhwc_bf16(x: torch.Tensor, relu: bool = False) -> torch.Tensor:
    """float32 or bfloat16 [B,C,H,W] -> bfloat16 [B,H,W,C] (optionally through ReLU) in one HIP kernel."""
    _lib.require_gpu()
    if not x.is_cuda or x.dtype not in _DTYPES or x.dim() != 4:
        raise ValueError("x must be a float32 or bfloat16 CUDA tensor [B,C,H,W]")
    x = x.contiguous()
    b, c, h, w = x.shape
    out = torch.empty((b, h, w, c), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().v2v_nchw_to_nhwc_bf16_hip(_ptr(x), _DTYPES[x.dtype], b, c, h, w, int(bool(relu)), _ptr(out), _lib.stream_ptr()))
    return out



def _to_nhwc_bf16(x: torch.Tensor, relu: bool = False) -> torch.Tensor:
    """[B,C,H,W] float32 / bfloat16 -> a contiguous bfloat16 [B,H,W,C] tensor: the layout kernel where it applies (64-channel x
    64-pixel tiles: C % 64 == 0 and H*W % 64 == 0), one torch copy otherwise -- the same guard for every layer, so an odd
    spatial size does not surface as an opaque V2V_ERR_SHAPE from the kernel."""
    if x.shape[1] % 64 == 0 and (x.shape[2] * x.shape[3]) % 64 == 0:
        return nchw_to_nhwc_bf16(x, relu=relu)
    if relu:
        x = torch.relu(x)
    return x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)


def pack_gate_weights(weight: torch.Tensor) -> torch.Tensor:
    """Gates.weight float32 [4C, 2C, 3, 3] -> the packed bfloat16 stream the kernel reads (flat tensor)."""
    _lib.require_gpu()
    if not weight.is_cuda or weight.dtype != torch.float32 or weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3) \
            or weight.shape[0] != 2 * weight.shape[1]:
        raise ValueError("weight must be a float32 CUDA tensor [4C, 2C, 3, 3]")
    c = weight.shape[0] // 4
    n = C.c_uint64(0)
    _lib.check(_lib.lib().v2v_convlstm_packed_bytes(c, C.byref(n)))
    packed = torch.empty((n.value // 2,), dtype=torch.bfloat16, device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.check(_lib.lib().v2v_convlstm_pack_weights_hip(_ptr(weight.detach().contiguous()), c, _ptr(packed), _lib.stream_ptr()))
    return packed


def convlstm_step(x, h_prev, c_prev, packed, bias, nchw_dtype=torch.float32, tile_rows: int = 0, c_out=None):
    """One step on NHWC state.  x, h_prev: bfloat16 [B,H,W,C]; c_prev: float32 [B,H,W,C]; h_prev / c_prev None = zero state.
    Returns (h_state bf16 NHWC, c_state fp32 NHWC, h as [B,C,H,W] in nchw_dtype -- float32 / bfloat16 -- or None when
    nchw_dtype is None).  c_out may be c_prev (updated in place)."""
    _lib.require_gpu()
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 4 or not x.is_contiguous():
        raise ValueError("x must be a contiguous bfloat16 CUDA tensor [B,H,W,C]")
    b, h, w, c = x.shape
    for name, t, dt in (("h_prev", h_prev, torch.bfloat16), ("c_prev", c_prev, torch.float32)):
        if t is not None and (t.dtype != dt or tuple(t.shape) != (b, h, w, c) or not t.is_contiguous() or t.device != x.device):
            raise ValueError(f"{name} must be a contiguous {dt} tensor [B,H,W,C] on x's device")
    if bias.dtype != torch.float32 or bias.numel() != 4 * c or packed.dtype != torch.bfloat16 or packed.numel() != 4 * c * 2 * c * 9:
        raise ValueError("bias must be float32 [4C] and packed the output of pack_gate_weights for the same C")
    h_state = torch.empty_like(x)
    c_state = c_out if c_out is not None else torch.empty((b, h, w, c), dtype=torch.float32, device=x.device)
    if nchw_dtype is not None and nchw_dtype not in _DTYPES:
        raise ValueError("nchw_dtype must be torch.float32, torch.bfloat16 or None")
    h_nchw = torch.empty((b, c, h, w), dtype=nchw_dtype, device=x.device) if nchw_dtype is not None else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().v2v_convlstm_step_hip(_ptr(x), _ptr(h_prev), _ptr(c_prev), _ptr(packed), _ptr(bias.detach().contiguous()),
                                                    b, h, w, c, _ptr(h_state), _ptr(c_state), _ptr(h_nchw), _DTYPES.get(nchw_dtype, _lib.F32), tile_rows,
                                                    _lib.stream_ptr()))
    return h_state, c_state, h_nchw


class ConvLSTM(nn.Module):
    """Drop-in for model/submodules.py:ConvLSTM (:179-235) on the fused kernel.

    forward(input_, prev_state=None) -> (hidden, cell), both logically [B,C,H,W] as in the reference: `hidden` is a
    contiguous NCHW tensor in the input's dtype (float32, or bfloat16 under autocast -- the reference's state takes the
    input's dtype too, :202-203); `cell` is the kernel's float32 NHWC cell buffer seen through permute(0,3,1,2) (a
    channels-last tensor; float32 even under autocast: the cell state is never rounded to bf16).  A bfloat16 input that is
    channels-last (torch.channels_last networks under autocast) is consumed and produced in place: `hidden` is then a
    channels-last view of the kernel's NHWC buffer and no layout-change kernel runs.  The bf16 NHWC copy of `hidden` that the next step's matrix-core GEMM reads is
    kept beside it and reused when the (hidden, cell) pair comes back untouched (UNetRecurrent.forward, model/unet.py:293-296);
    any other prev_state (cloned, loaded, edited) is converted from its float32 values, which gives the same bits."""

    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        if kernel_size != 3 or input_size != hidden_size:
            raise ValueError("the fused ConvLSTM covers the configuration the reference instantiates "
                             "(model/submodules.py:112: input_size == hidden_size, kernel_size=3)")
        self.input_size, self.hidden_size = input_size, hidden_size
        self.Gates = nn.Conv2d(input_size + hidden_size, 4 * hidden_size, kernel_size, padding=kernel_size // 2)
        self._packed, self._packed_key = None, None
        self._h_cache = None                                           # (hidden tensor, its version, bf16 NHWC twin)

    def _weights(self):
        w = self.Gates.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._packed_key != key:
            self._packed, self._packed_key = pack_gate_weights(w.detach()), key
        return self._packed

    def forward(self, input_, prev_state=None, input_relu: bool = False):
        """input_relu=True takes the PRE-activation output of the convolution in front (RecurrentConvLayer.conv,
        model/submodules.py:110-116) and applies its ReLU inside the layout-change kernel."""
        if torch.is_grad_enabled() and (input_.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("v2v_amd.convlstm.ConvLSTM is inference-only (no autograd through the fused kernel): "
                               "call it under torch.no_grad() / in eval mode")
        # channels-last bfloat16 input (a network run in torch.channels_last under autocast): its memory IS the kernel's NHWC
        # layout -- no layout-change kernel on the way in, and the hidden state goes out as a channels-last view of the
        # kernel's own NHWC buffer (no NCHW copy either)
        nhwc_io = (input_.dtype == torch.bfloat16 and input_.dim() == 4 and input_.is_contiguous(memory_format=torch.channels_last)
                   and not input_.is_contiguous())
        if nhwc_io:
            x = input_.permute(0, 2, 3, 1)
            if input_relu:
                x = torch.relu(x)
        else:
            x = _to_nhwc_bf16(input_, relu=input_relu)
        h_prev = c_prev = None
        if prev_state is not None:
            hidden, cell = prev_state
            cache = self._h_cache
            if cache is not None and cache[0] is hidden and cache[1] == hidden._version:
                h_prev = cache[2]
            elif hidden.dtype == torch.bfloat16 and hidden.is_contiguous(memory_format=torch.channels_last) and not hidden.is_contiguous():
                h_prev = hidden.permute(0, 2, 3, 1)
            else:
                h_prev = _to_nhwc_bf16(hidden)
            c_prev = cell.permute(0, 2, 3, 1)
            if c_prev.dtype != torch.float32 or not c_prev.is_contiguous():
                c_prev = c_prev.float().contiguous()
        h_state, c_state, h_nchw = convlstm_step(x, h_prev, c_prev, self._weights(), self.Gates.bias.detach().float(),
                                                 nchw_dtype=None if nhwc_io else input_.dtype)
        hidden_out = h_state.permute(0, 3, 1, 2) if nhwc_io else h_nchw
        self._h_cache = (hidden_out, hidden_out._version, h_state)
        return hidden_out, c_state.permute(0, 3, 1, 2)


# ---- the residual blocks of the same encoder (model/submodules.py:143-177) on the same matrix-core kernel ---------------------
def pack_conv3x3_weights(weight: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d(Cin, Cout, 3, padding=1).weight float32 [Cout, Cin, 3, 3] -> the packed bfloat16 stream of v2v_conv3x3_nhwc_hip."""
    _lib.require_gpu()
    if not weight.is_cuda or weight.dtype != torch.float32 or weight.dim() != 4 or tuple(weight.shape[2:]) != (3, 3):
        raise ValueError("weight must be a float32 CUDA tensor [Cout, Cin, 3, 3]")
    cout, cin = weight.shape[:2]
    packed = torch.empty((cout * cin * 9,), dtype=torch.bfloat16, device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.check(_lib.lib().v2v_conv3x3_pack_weights_hip(_ptr(weight.detach().contiguous()), cin, cout, _ptr(packed), _lib.stream_ptr()))
    return packed


def conv3x3_nhwc(x, packed, bias, residual=None, relu=False, tile_rows: int = 0):
    """out = [relu](conv3x3(x) + bias [+ residual]) on NHWC bfloat16: x [B,H,W,Cin], residual / out [B,H,W,Cout]."""
    _lib.require_gpu()
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 4 or not x.is_contiguous():
        raise ValueError("x must be a contiguous bfloat16 CUDA tensor [B,H,W,Cin]")
    b, h, w, cin = x.shape
    cout = bias.numel()
    if bias.dtype != torch.float32 or packed.dtype != torch.bfloat16 or packed.numel() != cout * cin * 9:
        raise ValueError("bias must be float32 [Cout] and packed the output of pack_conv3x3_weights for the same Cin, Cout")
    if residual is not None and (residual.dtype != torch.bfloat16 or tuple(residual.shape) != (b, h, w, cout) or not residual.is_contiguous()
                                 or residual.device != x.device):
        raise ValueError("residual must be a contiguous bfloat16 tensor [B,H,W,Cout] on x's device")
    out = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().v2v_conv3x3_nhwc_hip(_ptr(x), _ptr(packed), _ptr(bias.detach().contiguous()), _ptr(residual), int(bool(relu)),
                                                   b, h, w, cin, cout, _ptr(out), tile_rows, _lib.stream_ptr()))
    return out


class ResidualBlock(nn.Module):
    """Drop-in for model/submodules.py:ResidualBlock (:143-177) as E2VID instantiates it (model/unet.py:48: in == out channels,
    stride 1, no downsample, norm=None): relu(conv2(relu(conv1(x))) + x), both convolutions on the matrix-core kernel with the
    bias / residual / ReLU in its epilogue.  Same parameter names (conv1, conv2).  Inference only; bfloat16 operands with fp32
    accumulation; a channels-last bfloat16 input is consumed and produced in place, anything else goes through the
    layout-change kernel and comes back NCHW in the input's dtype."""

    def __init__(self, in_channels, out_channels, stride=1, downsample=None, norm=None, BN_momentum=0.1):
        super().__init__()
        if stride != 1 or downsample is not None or norm is not None or in_channels != out_channels:
            raise ValueError("the fused ResidualBlock covers the configuration E2VID instantiates "
                             "(model/unet.py:48: in_channels == out_channels, stride 1, no downsample, norm=None)")
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)
        self._packed = {}

    def _weights(self, conv, name):
        w = conv.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._packed.get(name, (None, None))[0] != key:
            self._packed[name] = (key, pack_conv3x3_weights(w.detach()))
        return self._packed[name][1]

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("v2v_amd.convlstm.ResidualBlock is inference-only (no autograd through the fused kernel)")
        nhwc_io = x.dtype == torch.bfloat16 and x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        xn = x.permute(0, 2, 3, 1) if nhwc_io else _to_nhwc_bf16(x)
        mid = conv3x3_nhwc(xn, self._weights(self.conv1, "conv1"), self.conv1.bias.detach().float(), relu=True)
        out = conv3x3_nhwc(mid, self._weights(self.conv2, "conv2"), self.conv2.bias.detach().float(), residual=xn, relu=True)
        out = out.permute(0, 3, 1, 2)
        return out if nhwc_io else out.contiguous().to(x.dtype)


# ---- the encoder / decoder convolutions around those blocks (ConvLayer / UpsampleConvLayer, model/submodules.py:6-96) -----------
def pack_conv_weights(weight: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d(Cin, Cout, ks, padding=ks//2).weight float32 [Cout, Cin, ks, ks] (ks 3 or 5) -> the packed bfloat16 stream."""
    _lib.require_gpu()
    if not weight.is_cuda or weight.dtype != torch.float32 or weight.dim() != 4 or weight.shape[2] != weight.shape[3]:
        raise ValueError("weight must be a float32 CUDA tensor [Cout, Cin, ks, ks]")
    cout, cin, ks = weight.shape[0], weight.shape[1], weight.shape[2]
    n = _lib.lib().v2v_conv_packed_elems(cin, cout, ks)
    if n < 0:
        raise ValueError(f"the convolution kernel does not take {cin} -> {cout} channels, {ks}x{ks}")
    packed = torch.empty((n,), dtype=torch.bfloat16, device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.check(_lib.lib().v2v_conv_pack_weights_hip(_ptr(weight.detach().contiguous()), cin, cout, ks, _ptr(packed), _lib.stream_ptr()))
    return packed


def conv_nhwc(x, packed, bias, ks: int, stride: int = 1, residual=None, relu=False, tile_rows: int = 0):
    """out = [relu](conv_ks(x, stride, pad ks//2) + bias [+ residual]) on NHWC bfloat16: x [B,Hin,Win,Cin] -> [B,Hout,Wout,Cout]."""
    _lib.require_gpu()
    if not x.is_cuda or x.dtype != torch.bfloat16 or x.dim() != 4 or not x.is_contiguous():
        raise ValueError("x must be a contiguous bfloat16 CUDA tensor [B,H,W,Cin]")
    b, hin, win, cin = x.shape
    cout = bias.numel()
    if bias.dtype != torch.float32 or packed.dtype != torch.bfloat16 or packed.numel() != _lib.lib().v2v_conv_packed_elems(cin, cout, ks):
        raise ValueError("bias must be float32 [Cout] and packed the output of pack_conv_weights for the same Cin, Cout, ks")
    h, w = (hin - 1) // stride + 1, (win - 1) // stride + 1
    if residual is not None and (residual.dtype != torch.bfloat16 or tuple(residual.shape) != (b, h, w, cout) or not residual.is_contiguous()
                                 or residual.device != x.device):
        raise ValueError("residual must be a contiguous bfloat16 tensor [B,Hout,Wout,Cout] on x's device")
    out = torch.empty((b, h, w, cout), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().v2v_conv_nhwc_hip(_ptr(x), _ptr(packed), _ptr(bias.detach().contiguous()), _ptr(residual), int(bool(relu)),
                                                b, hin, win, cin, cout, ks, stride, _ptr(out), tile_rows, _lib.stream_ptr()))
    return out


def upsample2x_nhwc(x, skip=None):
    """out = bilinear_x2(x [+ skip]) on NHWC bfloat16 ([B,H,W,C] -> [B,2H,2W,C]): f.interpolate(scale_factor=2, mode='bilinear',
    align_corners=False) of UpsampleConvLayer.forward (model/submodules.py:86-87) behind the sum skip (model/unet.py:304)."""
    _lib.require_gpu()
    for name, v in (("x", x), ("skip", skip)):
        if v is not None and (not v.is_cuda or v.dtype != torch.bfloat16 or v.dim() != 4 or not v.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous bfloat16 CUDA tensor [B,H,W,C]")
    if skip is not None and (skip.shape != x.shape or skip.device != x.device):
        raise ValueError("skip must have x's shape and device")
    b, h, w, c = x.shape
    out = torch.empty((b, 2 * h, 2 * w, c), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().v2v_upsample2x_nhwc_hip(_ptr(x), _ptr(skip), b, h, w, c, _ptr(out), _lib.stream_ptr()))
    return out


def conv1x1_nhwc(x, weight, bias, skip=None, out_dtype=torch.bfloat16):
    """out[..., o] = bias[o] + sum_c weight[o, c] * (x[..., c] + skip[..., c]) on NHWC bfloat16 ([B,H,W,C] -> [B,H,W,Cout], Cout <= 3):
    the prediction layer ConvLayer(base, out, 1, activation=None) on skip_sum(x, head) (model/unet.py:58-64, :307)."""
    _lib.require_gpu()
    for name, v in (("x", x), ("skip", skip)):
        if v is not None and (not v.is_cuda or v.dtype != torch.bfloat16 or v.dim() != 4 or not v.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous bfloat16 CUDA tensor [B,H,W,C]")
    if skip is not None and (skip.shape != x.shape or skip.device != x.device):
        raise ValueError("skip must have x's shape and device")
    b, h, w, c = x.shape
    weight = weight.detach().reshape(weight.shape[0], -1).float().contiguous()
    if weight.shape[1] != c or bias.numel() != weight.shape[0] or out_dtype not in _DTYPES:
        raise ValueError("weight must be [Cout, C(,1,1)], bias [Cout], out_dtype float32 or bfloat16")
    out = torch.empty((b, h, w, weight.shape[0]), dtype=out_dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().v2v_conv1x1_nhwc_hip(_ptr(x), _ptr(skip), _ptr(weight), _ptr(bias.detach().float().contiguous()), b * h * w, c,
                                                   weight.shape[0], _ptr(out), _DTYPES[out_dtype], _lib.stream_ptr()))
    return out


def to_nhwc8_bf16(x, scales=None):
    """float32 [B, C <= 8, H, W] of any strides -> bfloat16 [B, H, W, 8] with the channels zero-padded to 8: the head's input layout.
    scales: optional float32 [B,2] = (neg_max, pos_max) per sample (v2v_amd.postops.scales_from_stats): normalize_batch_voxel's
    where(x > 0, x / pos_max, x / neg_max) (model/train_utils.py:162-166) applied while the voxels are read."""
    _lib.require_gpu()
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] > 8:
        raise ValueError("x must be a float32 CUDA tensor [B, C <= 8, H, W]")
    b, c, h, w = x.shape
    if scales is not None and (scales.dtype != torch.float32 or tuple(scales.shape) != (b, 2) or not scales.is_contiguous() or scales.device != x.device):
        raise ValueError(f"scales must be a contiguous float32 [{b},2] tensor on x's device")
    out = torch.empty((b, h, w, 8), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().v2v_to_nhwc8_bf16_scaled_hip(_ptr(x), *x.stride(), b, c, h, w, _ptr(scales), _ptr(out), _lib.stream_ptr()))
    return out


def pack_head_weights(weight):
    """nn.Conv2d(Cin <= 8, 32, ks, padding=ks//2).weight float32 -> the head kernel's packed bfloat16 stream (taps along K)."""
    _lib.require_gpu()
    if not weight.is_cuda or weight.dtype != torch.float32 or weight.dim() != 4 or weight.shape[0] != 32 or weight.shape[1] > 8 \
            or weight.shape[2] != weight.shape[3] or weight.shape[2] not in (3, 5):
        raise ValueError("weight must be a float32 CUDA tensor [32, Cin <= 8, ks, ks], ks 3 or 5")
    ks = weight.shape[2]
    packed = torch.empty((_lib.lib().v2v_conv_head_packed_elems(ks),), dtype=torch.bfloat16, device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.check(_lib.lib().v2v_conv_head_pack_weights_hip(_ptr(weight.detach().contiguous()), weight.shape[1], ks, _ptr(packed), _lib.stream_ptr()))
    return packed


def conv_head_nhwc(x8, packed, bias, ks: int, relu=True):
    """out = [relu](conv_ks(x, stride 1, pad ks//2) + bias): x8 [B,H,W,8] bfloat16 (to_nhwc8_bf16) -> [B,H,W,32] bfloat16; the UNet's
    head ConvLayer(num_bins, 32, 5, stride 1, padding 2) (model/unet.py:77-78).  H and W multiples of 16."""
    _lib.require_gpu()
    if not x8.is_cuda or x8.dtype != torch.bfloat16 or x8.dim() != 4 or x8.shape[3] != 8 or not x8.is_contiguous():
        raise ValueError("x8 must be a contiguous bfloat16 CUDA tensor [B,H,W,8]")
    if bias.numel() != 32 or packed.dtype != torch.bfloat16 or packed.numel() != _lib.lib().v2v_conv_head_packed_elems(ks):
        raise ValueError("bias must be [32] and packed the output of pack_head_weights for the same ks")
    b, h, w, _ = x8.shape
    out = torch.empty((b, h, w, 32), dtype=torch.bfloat16, device=x8.device)
    with torch.cuda.device(x8.device):
        _lib.check(_lib.lib().v2v_conv_head_nhwc_hip(_ptr(x8), _ptr(packed), _ptr(bias.detach().float().contiguous()), int(bool(relu)), b, h, w, ks,
                                                     _ptr(out), _lib.stream_ptr()))
    return out


class ConvLayer(nn.Module):
    """Drop-in for model/submodules.py:ConvLayer (:6-33) as the recurrent UNet builds its encoder / decoder convolutions
    (model/unet.py: kernel_size 5, padding 2, stride 2 or 1, activation 'relu' or None, norm=None): same constructor, same
    `conv2d` parameter, convolution + bias + ReLU in one matrix-core kernel.  upsample=True puts the bilinear x2 upsampling
    (f.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False), its own bfloat16 NHWC kernel) in front, i.e.
    UpsampleConvLayer (:68-96).  Inference only; bfloat16 operands, fp32 accumulation; channels-last bfloat16
    inputs are consumed and produced in place, anything else goes through the layout-change kernel and comes back NCHW in the
    input's dtype.  in_channels % 64 == 0 and out_channels in {32, 64, 128, 256k}; in_channels 32 with 64 / 128 outputs (the first
    encoder); <= 8 input channels with 32 outputs = the head; kernel_size 1 = the prediction layer; else ValueError (no fallback)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, activation="relu", norm=None, BN_momentum=0.1,
                 upsample=False):
        super().__init__()
        if norm is not None or activation not in ("relu", None) or kernel_size not in (1, 3, 5) or padding != kernel_size // 2 or stride not in (1, 2) \
                or (kernel_size == 1 and (stride != 1 or activation is not None or upsample or out_channels > 3)):
            raise ValueError("the fused ConvLayer covers norm=None, activation 'relu' or None, kernel_size 3 or 5 with padding "
                             "kernel_size // 2, stride 1 or 2, and the 1x1 prediction layer (stride 1, no activation, <= 3 outputs)")
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True)
        self.relu, self.upsample = activation == "relu", upsample
        self.head = in_channels <= 8 and kernel_size in (3, 5)                  # the UNet's head: voxel bins -> 32 channels
        if self.head and (out_channels != 32 or stride != 1 or upsample):
            raise ValueError("with <= 8 input channels the fused ConvLayer is the UNet's head: 32 output channels, stride 1")
        self.force_channels_last = False          # head only: hand out the kernel's NHWC buffer as a channels-last view whatever came in
        self._packed = (None, None)

    def _weights(self):
        w = self.conv2d.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._packed[0] != key:
            self._packed = (key, (pack_head_weights if self.head else pack_conv_weights)(w.detach()))
        return self._packed[1]

    def forward(self, x, skip=None, scales=None):
        """skip (upsample=True only): the sum skip connection model/unet.py:304 adds in front of the decoder, folded into the
        upsampling kernel -- layer(x, skip) == layer(x + skip)."""
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise RuntimeError("v2v_amd.convlstm.ConvLayer is inference-only (no autograd through the fused kernel)")
        if scales is not None and not self.head:
            # only the head kernel (<= 8 input channels, 3x3 / 5x5) divides by normalize_batch_voxel's scales while it reads; anything else
            # would silently run on raw, un-normalised events (RingLoader(normalize='scales') hands out raw voxels)
            raise ValueError("`scales` is applied by the head kernel only (in_channels <= 8, kernel 3 or 5): normalise the events first "
                             "(v2v_amd.postops.apply_scales / RingLoader(normalize=True)) for this layer")
        if self.conv2d.kernel_size[0] == 1:                                           # prediction layer: pred(skip_sum(x, head)), model/unet.py:307
            nhwc = all(v is None or (v.dtype == torch.bfloat16 and v.dim() == 4 and v.is_contiguous(memory_format=torch.channels_last)
                                     and not v.is_contiguous()) for v in (x, skip))
            if not nhwc:
                x, skip = (x if skip is None else x + skip), None
            if nhwc:
                xn = x.permute(0, 2, 3, 1)
            else:
                xn = _to_nhwc_bf16(x)                    # layout kernel, or one torch copy below its 64-channel x 64-pixel tile
            out = conv1x1_nhwc(xn, self.conv2d.weight, self.conv2d.bias, None if skip is None else skip.permute(0, 2, 3, 1),
                               out_dtype=torch.bfloat16 if nhwc else x.dtype).permute(0, 3, 1, 2)
            return out if nhwc else out.contiguous()
        if skip is not None and not self.upsample:
            raise ValueError("skip is the decoder's (upsample=True) sum skip connection")
        if self.head:                                                                  # model/unet.py:77-78: any float layout in, bf16 out
            low = x.dtype == torch.bfloat16 or (x.is_cuda and torch.is_autocast_enabled())
            # channels-last out when the input or (as torch's own convolution decides) the weight is channels-last
            w = self.conv2d.weight
            cl = self.force_channels_last or any(v.is_contiguous(memory_format=torch.channels_last) and not v.is_contiguous() for v in (x, w))
            out = conv_head_nhwc(to_nhwc8_bf16(x.float(), scales), self._weights(), self.conv2d.bias, self.conv2d.kernel_size[0], relu=self.relu).permute(0, 3, 1, 2)
            out = out if cl else out.contiguous()
            return out if low else out.to(x.dtype)

        def is_nhwc(v):
            return v.dtype == torch.bfloat16 and v.dim() == 4 and v.is_contiguous(memory_format=torch.channels_last) and not v.is_contiguous()
        nhwc_io = is_nhwc(x)
        if skip is not None and not (nhwc_io and is_nhwc(skip)):
            x, skip = x + skip, None
            nhwc_io = is_nhwc(x)
        if nhwc_io:
            xn = x.permute(0, 2, 3, 1)
        else:
            xn = _to_nhwc_bf16(x)                        # layout kernel, or one torch copy below its 64-channel x 64-pixel tile
        if self.upsample:
            xn = upsample2x_nhwc(xn, None if skip is None else skip.permute(0, 2, 3, 1))
        out = conv_nhwc(xn, self._weights(), self.conv2d.bias.detach().float(), self.conv2d.kernel_size[0], self.conv2d.stride[0],
                        relu=self.relu).permute(0, 3, 1, 2)
        return out if nhwc_io else out.contiguous().to(x.dtype)
