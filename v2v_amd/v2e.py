"""v2e-derived DVS model -- host side of v2v_v2e_voxel_hip (BASELINE config 3).

Mirrors data/v2v_core_v2e.py:video_to_voxel (:556-581): same argument names and order, `[N,H,W]` in, `[N-1,H,W]`
signed event counts out (float64 ndarray for NumPy input; float32 CUDA tensor for CUDA input), plus the batched
form `v2e_voxel_batch`.  All arithmetic runs in the HIP kernel (v2v_amd/csrc/v2v_v2e.hpp).

rng='numpy'  -> the reference's behaviour: `np.random.seed(seed)`, then the fields are drawn on the host from the
                global stream in the reference's order and replayed on the GPU (bit-exact with the reference for
                integer-valued video).  The Poisson rates of the shot noise are host-side NumPy expressions here
                (they must exist before np.random.poisson can be called in stream order).
rng='philox' -> device-native fields keyed by (seed, clip id); no host RNG work; statistically equivalent.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np
import torch

from . import _lib
from .esim import BIN_MODES, _OUT, _TORCH_IN


def make_params(FPS, threshold_model, thres_mean_mean, thres_mean_std, thres_diff_mean, thres_diff_std, cutoff_hz,
                leak_rate_hz, refractory_period_s, shot_noise_rate_hz, leak_jitter_fraction, noise_rate_cov_decades,
                uint8_wrap=True) -> _lib.V2EParams:
    if threshold_model not in _lib.V2E_MODELS:
        # 'spatial_independent_temporal_changing' crashes in the reference on the first frame (v2v_core_v2e.py:423-426)
        raise ValueError(f"unsupported threshold_model {threshold_model!r}; one of {list(_lib.V2E_MODELS)}")
    return _lib.V2EParams(float(FPS), _lib.V2E_MODELS[threshold_model], thres_mean_mean, thres_mean_std, thres_diff_mean,
                          thres_diff_std, cutoff_hz, leak_rate_hz, refractory_period_s, shot_noise_rate_hz,
                          leak_jitter_fraction, noise_rate_cov_decades, int(bool(uint8_wrap)))


def v2e_voxel_batch(frames: torch.Tensor, params: _lib.V2EParams, *, bin_mode: str = "sum", num_bins: int = 5,
                    frames_per_bin: int = 1, rng_mode: str = "philox", seed: int = 0, clip_id0: int = 0,
                    out_dtype: torch.dtype = torch.float32, out: Optional[torch.Tensor] = None,
                    counts: Optional[torch.Tensor] = None, replay: Optional[dict] = None) -> torch.Tensor:
    """frames [B,N,H,W] uint8/float32 CUDA -> [B,L,Tb,H,W] ('sum') or [B,Tb,H,W] ('bilinear').

    replay (rng_mode='replay'): dict of tensors pos_thres, neg_thres ([B,H,W] or [B,N-1,H,W] for the temporal
    model), noise_rate [B,H,W] float32, leak_randn [B,N-1,H,W] (if leak), shot_pos/shot_neg [B,N-1,H,W] int64 (if shot)."""
    _lib.require_gpu()
    if frames.ndim != 4 or not frames.is_cuda or frames.dtype not in _TORCH_IN:
        raise ValueError("frames must be a [B,N,H,W] uint8/float32 CUDA tensor")
    b, n, h, w = frames.shape
    if frames.stride(3) != 1 or frames.stride(2) != w:
        frames = frames.contiguous()
    k = n - 1
    if bin_mode == "sum":
        assert k % (num_bins * frames_per_bin) == 0, "(N-1) % (num_bins*frames_per_bin) != 0"
        shape = (b, k // (num_bins * frames_per_bin), num_bins, h, w)
    elif bin_mode == "bilinear":
        shape = (b, num_bins, h, w)
    else:
        raise ValueError(f"bin_mode must be one of {list(BIN_MODES)}")
    if out is None:
        out = torch.empty(shape, dtype=out_dtype, device=frames.device)
        if b == 0:
            return out
    elif tuple(out.shape) != shape or not out.is_contiguous() or out.dtype not in _OUT or out.device != frames.device:
        raise ValueError(f"out must be a contiguous {shape} float32/float64 tensor on {frames.device}")
    dev = frames.device
    rp, keep, ws = None, [], None
    if rng_mode == "replay":
        if replay is None:
            raise ValueError("rng_mode='replay' needs the replay dict")
        temporal = params.threshold_model == _lib.V2E_MODELS["spatial_temporal_independent"]

        def put(name, dtype, shape_):
            t = torch.as_tensor(replay[name]).to(device=dev, dtype=dtype).contiguous()
            if tuple(t.shape) != shape_:
                raise ValueError(f"replay[{name!r}] shape {tuple(t.shape)} != {shape_}")
            keep.append(t)
            return t.data_ptr()
        tshape = (b, k, h, w) if temporal else (b, h, w)
        rp = _lib.V2EReplay(put("pos_thres", torch.float64, tshape), put("neg_thres", torch.float64, tshape),
                            h * w if temporal else 0, put("noise_rate", torch.float32, (b, h, w)),
                            put("leak_randn", torch.float64, (b, k, h, w)) if params.leak_rate_hz > 0 else None,
                            put("shot_pos", torch.int64, (b, k, h, w)) if params.shot_noise_rate_hz > 0 else None,
                            put("shot_neg", torch.int64, (b, k, h, w)) if params.shot_noise_rate_hz > 0 else None)
        mode = _lib.RNG_REPLAY
    elif rng_mode == "philox":
        mode = _lib.RNG_PHILOX
        if params.shot_noise_rate_hz > 0:
            ws = torch.empty((_lib.lib().v2v_v2e_workspace_bytes(b, n) // 8,), dtype=torch.int64, device=dev)
    else:
        raise ValueError("rng_mode must be 'philox' or 'replay'")
    if counts is not None and (counts.dtype != torch.int64 or tuple(counts.shape) != (b, 2) or not counts.is_contiguous()
                               or counts.device != dev):
        raise ValueError("counts must be a contiguous int64 [B,2] tensor on the frames' device")
    with torch.cuda.device(dev):
        rc = _lib.lib().v2v_v2e_voxel_hip(
            C.c_void_p(frames.data_ptr()), _TORCH_IN[frames.dtype], b, n, h, w,
            frames.stride(0) if b > 1 else n * frames.stride(1), frames.stride(1), C.byref(params), mode,
            C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_uint64(clip_id0), C.byref(rp) if rp is not None else None,
            BIN_MODES[bin_mode], num_bins, frames_per_bin, C.c_void_p(out.data_ptr()), _OUT[out.dtype],
            C.c_void_p(counts.data_ptr()) if counts is not None else None,
            C.c_void_p(ws.data_ptr()) if ws is not None else None, _lib.stream_ptr())
    _lib.check(rc)
    return out


def draw_numpy_v2e_fields(video: np.ndarray, params: _lib.V2EParams):
    """Draw the v2e model's random fields for ONE clip from the global np.random stream in the reference's order
    (v2v_core_v2e.py: frame 0 -> [temporal: 2 discarded normals :417-421] _init normals :328-342, randn :348;
    later frames -> [temporal: 2 normals] [leak: randn :201] [shot: poisson(pos), poisson(neg) :102-103]) and
    shape them for replay.  Thresholds / Poisson rates use the reference's own float64 expressions."""
    n, h, w = video.shape
    k = n - 1
    P = params
    temporal = P.threshold_model == _lib.V2E_MODELS["spatial_temporal_independent"]
    pn = P.threshold_model == _lib.V2E_MODELS["pn_related"]
    pos_nominal = P.thres_mean_mean + P.thres_diff_mean / 2
    neg_nominal = P.thres_mean_mean - P.thres_diff_mean / 2

    def thres_pair(first_frame=False):
        if pn and (first_frame or not temporal):
            mean = np.random.normal(loc=P.thres_mean_mean, scale=P.thres_mean_std, size=(h, w))
            diff = np.random.normal(loc=P.thres_diff_mean, scale=P.thres_diff_std, size=(h, w))
            pt, nt = mean + (diff / 2), mean - (diff / 2)
        else:
            pt = np.random.normal(loc=P.thres_mean_mean, scale=P.thres_mean_std, size=(h, w))
            nt = np.random.normal(loc=P.thres_mean_mean, scale=P.thres_mean_std, size=(h, w))
        return np.clip(pt, a_min=0.01, a_max=None), np.clip(nt, a_min=0.01, a_max=None)

    if temporal:
        thres_pair()                                   # frame 0 pre-draw, overwritten by _init
    pt, nt = thres_pair(first_frame=True)
    noise_rate = np.exp(math.log(10) * P.noise_rate_cov_decades * np.random.randn(h, w).astype(np.float32))
    pts, nts, leaks, sps, sns = [], [], [], [], []
    for i in range(1, n):
        dt = i / P.fps - (i - 1) / P.fps
        if temporal:
            pt, nt = thres_pair()
            pts.append(pt)
            nts.append(nt)
        if P.leak_rate_hz > 0:
            leaks.append(np.random.randn(h, w))
        if P.shot_noise_rate_hz > 0:
            frame = video[i]
            if frame.dtype == np.uint8 and not P.uint8_wrap:
                frame = frame.astype(np.float64)
            inten01 = (frame + 20) / 275.
            fac = 1 - (1 - 0.25) * inten01
            pf = fac * np.divide(pos_nominal, pt)
            nf = fac * np.divide(neg_nominal, nt)
            f = (P.shot_noise_rate_hz / 2) * dt
            sps.append(np.random.poisson(pf / np.mean(pf) * f))
            sns.append(np.random.poisson(nf / np.mean(nf) * f))
    out = {"pos_thres": np.stack(pts) if temporal else pt, "neg_thres": np.stack(nts) if temporal else nt,
           "noise_rate": noise_rate}
    if leaks:
        out["leak_randn"] = np.stack(leaks)
    if sps:
        out["shot_pos"], out["shot_neg"] = np.stack(sps), np.stack(sns)
    return out


def video_to_voxel(video, FPS, threshold_model, thres_mean_mean, thres_mean_std, thres_diff_mean, thres_diff_std,
                   cutoff_hz, leak_rate_hz, refractory_period_s, shot_noise_rate_hz, leak_jitter_fraction,
                   noise_rate_cov_decades, seed, rng="numpy", clip_id=0, uint8_compat=True, device="cuda"):
    """Drop-in for data/v2v_core_v2e.py:video_to_voxel (same positional arguments).  video: [N,H,W]."""
    is_np = isinstance(video, np.ndarray)
    params = make_params(FPS, threshold_model, thres_mean_mean, thres_mean_std, thres_diff_mean, thres_diff_std, cutoff_hz,
                         leak_rate_hz, refractory_period_s, shot_noise_rate_hz, leak_jitter_fraction,
                         noise_rate_cov_decades, uint8_wrap=uint8_compat)
    if is_np:
        v = video
        if v.dtype != np.uint8 and v.dtype != np.float32:
            v = v.astype(np.float32)
        frames = torch.from_numpy(np.ascontiguousarray(v)).to(device)
    else:
        frames = video if video.is_cuda else video.to(device)
        if frames.dtype not in _TORCH_IN:
            frames = frames.to(torch.float32)
        v = None
    n = frames.shape[0]
    kw = dict(bin_mode="sum", num_bins=n - 1, frames_per_bin=1, out_dtype=torch.float64 if is_np else torch.float32)
    if rng == "numpy":
        if seed is not None:
            np.random.seed(seed)                                     # v2v_core_v2e.py:312-314
        host = v if v is not None else frames.cpu().numpy()
        fields = draw_numpy_v2e_fields(host, params)
        out = v2e_voxel_batch(frames[None], params, rng_mode="replay",
                              replay={k_: torch.from_numpy(np.ascontiguousarray(a))[None] for k_, a in fields.items()}, **kw)
    elif rng == "philox":
        out = v2e_voxel_batch(frames[None], params, rng_mode="philox", seed=0 if seed is None else int(seed),
                              clip_id0=clip_id, **kw)
    else:
        raise ValueError("rng must be 'numpy' or 'philox'")
    out = out[0, 0]
    return out.cpu().numpy() if is_np else out
