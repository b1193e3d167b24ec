"""Decode-side front-end on the GPU: host side of v2v_frontend_hip.

`prepare_clip` takes decoded frames that are already on the device and does what WebvidDatasetV2.read_video +
the gather in __getitem__ do on the host (data/v2v_datasets.py:191-224, :311-316): [BGR->gray] -> crop ->
bilinear resize -> [flip] -> shake crop -> pause-index gather -> gray.  Output feeds esim_voxel_batch directly.
PARITY UNPINNED against OpenCV (not available here); bit-exact against oracle/frontend_oracle.py's restatement of the
OpenCV 4.x 8-bit algorithms.  `cv_version`: "cv4" (default; cvtColor BGR2GRAY with the 15-bit weights 3735/19235/9798 every
OpenCV >= 4.0 uses -- what the reference's unpinned `opencv-python` installs) or "cv3" (the 14-bit weights of OpenCV 2.x/3.x).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

GRAY_VERSIONS = {"cv4": 1, "cv3": 2}       # C ABI gray_first: BGR2GRAY as OpenCV >= 4.0 (15-bit weights) / as 2.x-3.x (14-bit)


def prepare_clip(raw: torch.Tensor, crop_before: int, min_i: int, min_j: int, flip: bool, crop_size: int, img_idxes,
                 all_di=None, all_dj=None, color_mode: str = "gray", want_imgs: bool = True, cv_version: str = "cv4"):
    """raw [T,Hs,Ws,C] uint8 CUDA (C = 3 BGR or 1).  Returns (all_imgs [N,crop,crop,Cout] uint8 or None,
    gray [N,crop,crop] uint8), both CUDA tensors."""
    _lib.require_gpu()
    if raw.ndim != 4 or raw.dtype != torch.uint8 or not raw.is_cuda:
        raise ValueError("raw must be a [T,Hs,Ws,C] uint8 CUDA tensor")
    assert color_mode in ["gray", "gray_in_bgr_out"]                          # v2v_datasets.py:76
    raw = raw.contiguous()
    t, hs, ws, cs = raw.shape
    dev = raw.device
    idx = torch.as_tensor(np.asarray(img_idxes, dtype=np.int32), device=dev)
    n = idx.numel()
    if n and (int(idx.min()) < 0 or int(idx.max()) >= t):
        raise IndexError("img_idxes outside the decoded frames")
    di = dj = None
    need_h = need_w = crop_size
    if all_di is not None:
        di_h = np.asarray(all_di, dtype=np.int64) - int(np.min(all_di))         # :217-218
        dj_h = np.asarray(all_dj, dtype=np.int64) - int(np.min(all_dj))
        need_h, need_w = crop_size + int(di_h.max()), crop_size + int(dj_h.max())
        di = torch.as_tensor(di_h.astype(np.int32), device=dev)
        dj = torch.as_tensor(dj_h.astype(np.int32), device=dev)
    gray_first = GRAY_VERSIONS[cv_version] if color_mode == "gray" else 0
    cout = 1 if (gray_first or cs == 1) else 3
    imgs = torch.empty((n, crop_size, crop_size, cout), dtype=torch.uint8, device=dev) if want_imgs else None
    gray = torch.empty((n, crop_size, crop_size), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().v2v_frontend_hip(
            C.c_void_p(raw.data_ptr()), t, hs, ws, cs, min_i, min_j, crop_before, need_h, need_w, crop_size, int(bool(flip)),
            int(gray_first), C.c_void_p(idx.data_ptr()), n, C.c_void_p(di.data_ptr()) if di is not None else None,
            C.c_void_p(dj.data_ptr()) if dj is not None else None, C.c_void_p(imgs.data_ptr()) if imgs is not None else None,
            C.c_void_p(gray.data_ptr()), _lib.stream_ptr())
    _lib.check(rc)
    return imgs, gray


def prepare_clips_batch(raw: torch.Tensor, clip_table, img_idxes, crop_size: int, color_mode: str = "gray",
                        want_imgs: bool = False, validate: bool = True, max_crop_before: int = 0, cv_version: str = "cv4", device=None):
    """Batch form of prepare_clip (no shake): raw [B,T,Hs,Ws,C] uint8 CUDA; clip_table [B,4] int = {min_i, min_j,
    crop_before, flip}; img_idxes [B,N].  Returns (imgs [B,N,crop,crop,Cout] or None, gray [B,N,crop,crop]) -- ONE launch.
    max_crop_before: optional upper bound of crop_before for device-resident tables (taken from the table when it is on
    the host); it only sizes the kernel's LDS tile.
    ZERO-COPY HOST INPUT (round 6): `raw` may be a contiguous PAGE-LOCKED host tensor (`raw.is_pinned()`); the kernel then stages each crop
    rectangle straight out of host memory over PCIe -- only the cb x cb x C bytes the resize reads cross the link, no staging copy of the
    whole 720p frames, no host-side slicing (BASELINE config 4's stream: 11x fewer bytes than shipping the frames).  Results are placed on
    `device` (default: the current CUDA device).  The host buffer must stay untouched until the launch has finished (stream-ordered, like
    any asynchronous copy out of page-locked memory)."""
    _lib.require_gpu()
    host_input = (not raw.is_cuda) and raw.is_pinned()
    if raw.ndim != 5 or raw.dtype != torch.uint8 or not (raw.is_cuda or host_input):
        raise ValueError("raw must be a [B,T,Hs,Ws,C] uint8 CUDA tensor (or a page-locked host tensor: zero-copy input)")
    assert color_mode in ["gray", "gray_in_bgr_out"]
    if host_input and not raw.is_contiguous():
        raise ValueError("a page-locked host `raw` must be contiguous (a .contiguous() copy would not be page-locked)")
    raw = raw.contiguous()
    b, t, hs, ws, cs = raw.shape
    dev = raw.device if raw.is_cuda else (torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device()))
    def on_device(x):
        return isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.int32 and x.is_contiguous()
    if not validate and on_device(clip_table) and on_device(img_idxes):
        # device-resident tables are used as they are: no host round trip (and no synchronisation) per batch
        tab, idx = clip_table.reshape(b, 4), img_idxes.reshape(b, -1)
    else:
        tab_h = np.asarray(clip_table.cpu() if isinstance(clip_table, torch.Tensor) else clip_table, dtype=np.int32).reshape(b, 4)
        idx_h = np.asarray(img_idxes.cpu() if isinstance(img_idxes, torch.Tensor) else img_idxes, dtype=np.int32).reshape(b, -1)
        if validate:
            if (tab_h[:, 0] < 0).any() or (tab_h[:, 1] < 0).any() or (tab_h[:, 2] < 1).any() or \
                    (tab_h[:, 0] + tab_h[:, 2] > hs).any() or (tab_h[:, 1] + tab_h[:, 2] > ws).any():
                raise ValueError("crop rectangle outside the frame")
            if idx_h.size and (idx_h.min() < 0 or idx_h.max() >= t):
                raise IndexError("img_idxes outside the decoded frames")
        tab = torch.as_tensor(tab_h, device=dev)
        idx = torch.as_tensor(idx_h, device=dev)
        if b:
            max_crop_before = int(tab_h[:, 2].max())
    n = idx.shape[1]
    gray_first = GRAY_VERSIONS[cv_version] if color_mode == "gray" else 0
    cout = 1 if (gray_first or cs == 1) else 3
    imgs = torch.empty((b, n, crop_size, crop_size, cout), dtype=torch.uint8, device=dev) if want_imgs else None
    gray = torch.empty((b, n, crop_size, crop_size), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().v2v_frontend_batch_hip(C.c_void_p(raw.data_ptr()), b, t, hs, ws, cs, C.c_void_p(tab.data_ptr()), int(max_crop_before), crop_size,
                                               int(gray_first), C.c_void_p(idx.data_ptr()), n,
                                               C.c_void_p(imgs.data_ptr()) if imgs is not None else None,
                                               C.c_void_p(gray.data_ptr()), _lib.stream_ptr())
    _lib.check(rc)
    return imgs, gray
