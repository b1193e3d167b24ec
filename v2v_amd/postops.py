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
