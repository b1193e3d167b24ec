"""Voxel post-ops of the consumer side: host side of v2v_normalize_pad_hip.

    normalize_batch_voxel(voxel)          model/train_utils.py:147-166  (same name, same [B,T,C,H,W] contract)
    pad_events(voxel, PAD=16)             model/train_utils.py:322-326  (zero padding of H,W to multiples of PAD)
    normalize_and_pad(voxel, ...)         both in one pass over the data
Everything runs in the HIP kernels (exact k-th values by radix select); CUDA float32 tensors in and out.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def normalize_and_pad(voxel: torch.Tensor, normalize: bool = True, PAD: int = 16, method: str = "radix", valid_hw=None,
                      inplace: bool = False) -> torch.Tensor:
    """normalize_batch_voxel (+ zero padding of H, W to multiples of PAD) in the HIP kernels.

    method    "radix": exact 3-pass radix select, any float32 content.  "count": exact counting select for integer-valued
              voxels with |v| <= 255 (SUM-mode grids without external noise): two passes fewer; a sample with any other
              value comes out as NaN.
    valid_hw  (H, W) when `voxel` is ALREADY padded (esim_voxel_batch(pad_to=PAD)): only that interior enters the k-th values
              (method "count"); with inplace=True the result overwrites `voxel`."""
    _lib.require_gpu()
    assert len(voxel.shape) == 5                                                # train_utils.py:149
    if not voxel.is_cuda or voxel.dtype != torch.float32:
        raise ValueError("voxel must be a float32 CUDA tensor [B,T,C,H,W]")
    if method not in ("radix", "count"):
        raise ValueError("method must be 'radix' or 'count'")
    voxel = voxel.contiguous()
    b, t, c, h_in, w_in = voxel.shape
    h, w = valid_hw if valid_hw is not None else (h_in, w_in)
    if valid_hw is not None and normalize and method != "count":
        raise ValueError("padded input needs method='count' (the radix select reads unpadded input)")
    hp, wp = (h + PAD - 1) // PAD * PAD, (w + PAD - 1) // PAD * PAD
    if inplace:
        if (hp, wp) != (h_in, w_in):
            raise ValueError("inplace needs an input that already has the padded layout")
        out = voxel
    else:
        out = torch.empty((b, t, c, hp, wp), dtype=torch.float32, device=voxel.device)
    ws = None
    if normalize:
        ws = torch.empty((_lib.lib().v2v_postops_workspace_bytes(b) // 8 + 1,), dtype=torch.int64, device=voxel.device)
    m = _lib.NORM_NONE if not normalize else (_lib.NORM_COUNT if method == "count" else _lib.NORM_RADIX)
    with torch.cuda.device(voxel.device):
        rc = _lib.lib().v2v_normalize_pad_ex_hip(C.c_void_p(voxel.data_ptr()), b, t * c, h, w, h_in, w_in, m, PAD,
                                                 C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()) if ws is not None else None,
                                                 _lib.stream_ptr())
    _lib.check(rc)
    return out


def normalize_batch_voxel(voxel: torch.Tensor, method: str = "radix") -> torch.Tensor:
    """Drop-in for model/train_utils.py:normalize_batch_voxel (no padding: PAD=1)."""
    return normalize_and_pad(voxel, normalize=True, PAD=1, method=method)


def pad_events(voxel: torch.Tensor, PAD: int = 16) -> torch.Tensor:
    return normalize_and_pad(voxel, normalize=False, PAD=PAD)


def scales_from_stats(stats: torch.Tensor, elems_per_sample: int) -> torch.Tensor:
    """The writer's statistics (esim_voxel_batch(stats=...)) -> float32 [B,2] = (neg_max, pos_max) per sample: exactly
    clamp(-kthvalue(1 %), min=1) and clamp(kthvalue(99 %), min=1) of normalize_batch_voxel (model/train_utils.py:153-160) over the sample's
    `elems_per_sample` = L*Tb*H*W voxels (padding excluded).  NaN for a sample whose rank falls among the overflow counts."""
    _lib.require_gpu()
    if stats.dtype != torch.int32 or stats.ndim != 2 or stats.shape[1] != _lib.VOXEL_STATS_WORDS or not stats.is_cuda or not stats.is_contiguous():
        raise ValueError(f"stats must be a contiguous int32 CUDA tensor [B,{_lib.VOXEL_STATS_WORDS}]")
    scales = torch.empty((stats.shape[0], 2), dtype=torch.float32, device=stats.device)
    with torch.cuda.device(stats.device):
        rc = _lib.lib().v2v_voxel_scales_hip(C.c_void_p(stats.data_ptr()), stats.shape[0], int(elems_per_sample), C.c_void_p(scales.data_ptr()),
                                             _lib.stream_ptr())
    _lib.check(rc)
    return scales


def apply_scales(voxel: torch.Tensor, scales, PAD: int = 16, valid_hw=None, inplace: bool = False) -> torch.Tensor:
    """where(voxel > 0, voxel / pos_max, voxel / neg_max) with the given per-sample (neg_max, pos_max) [B,2] (+ zero padding of H, W to
    multiples of PAD): the last step of normalize_batch_voxel (model/train_utils.py:162-166) as ONE pass.  scales=None pads only."""
    _lib.require_gpu()
    if not voxel.is_cuda or voxel.dtype != torch.float32 or voxel.ndim != 5:
        raise ValueError("voxel must be a float32 CUDA tensor [B,T,C,H,W]")
    voxel = voxel.contiguous()
    b, t, c, h_in, w_in = voxel.shape
    h, w = valid_hw if valid_hw is not None else (h_in, w_in)
    hp, wp = (h + PAD - 1) // PAD * PAD, (w + PAD - 1) // PAD * PAD
    if scales is not None and (scales.dtype != torch.float32 or tuple(scales.shape) != (b, 2) or not scales.is_contiguous() or scales.device != voxel.device):
        raise ValueError(f"scales must be a contiguous float32 [{b},2] tensor on the voxels' device")
    if inplace:
        if (hp, wp) != (h_in, w_in):
            raise ValueError("inplace needs an input that already has the padded layout")
        out = voxel
    else:
        out = torch.empty((b, t, c, hp, wp), dtype=torch.float32, device=voxel.device)
    with torch.cuda.device(voxel.device):
        rc = _lib.lib().v2v_voxel_apply_scales_hip(C.c_void_p(voxel.data_ptr()), b, t * c, h, w, h_in, w_in, PAD,
                                                   C.c_void_p(scales.data_ptr()) if scales is not None else None, C.c_void_p(out.data_ptr()),
                                                   _lib.stream_ptr())
    _lib.check(rc)
    return out


def voxel_scales_radix(voxel: torch.Tensor, method: str = "radix", valid_hw=None) -> torch.Tensor:
    """(neg_max, pos_max) [B,2] of normalize_batch_voxel by selection over the finished tensor (3-pass radix select: any float32 content;
    "count": integer-valued content, padded input allowed through valid_hw) -- for grids the writer's statistics cannot cover."""
    _lib.require_gpu()
    if not voxel.is_cuda or voxel.dtype != torch.float32 or voxel.ndim != 5:
        raise ValueError("voxel must be a float32 CUDA tensor [B,T,C,H,W]")
    voxel = voxel.contiguous()
    b, t, c, h_in, w_in = voxel.shape
    h, w = valid_hw if valid_hw is not None else (h_in, w_in)
    ws = torch.empty((_lib.lib().v2v_postops_workspace_bytes(b) // 8 + 1,), dtype=torch.int64, device=voxel.device)
    scales = torch.empty((b, 2), dtype=torch.float32, device=voxel.device)
    with torch.cuda.device(voxel.device):
        rc = _lib.lib().v2v_voxel_scales_select_hip(C.c_void_p(voxel.data_ptr()), b, t * c, h, w, h_in, w_in,
                                                    _lib.NORM_COUNT if method == "count" else _lib.NORM_RADIX, C.c_void_p(scales.data_ptr()),
                                                    C.c_void_p(ws.data_ptr()), _lib.stream_ptr())
    _lib.check(rc)
    return scales
