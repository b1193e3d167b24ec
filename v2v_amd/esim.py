"""ESIM frame-pair simulator fused with voxel binning -- host side of v2v_esim_voxel_hip.

Mirrors data/v2v_core_esim.py (EventEmulator.video_to_voxel, :6-69) and adds the batched form that
the throughput numbers are measured on.  All arithmetic happens in the HIP kernel
(v2v_amd/csrc/v2v_esim.hpp); this file only validates arguments, owns tensors and picks the RNG mode.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib

_TORCH_IN = {torch.uint8: _lib.U8, torch.float32: _lib.F32}
_OUT = {torch.float32: _lib.F32, torch.float64: _lib.F64}
DEFAULT_MAPPING = "auto"        # work-item mapping of esim_voxel_batch when the caller does not pin it
BIN_MODES = {"sum": _lib.BIN_SUM, "bilinear": _lib.BIN_BILINEAR}
RNG_MODES = {"none": _lib.RNG_NONE, "philox": _lib.RNG_PHILOX, "replay": _lib.RNG_REPLAY, "philox_fast": _lib.RNG_PHILOX_FAST}


def _params_tensor(params, batch: int, device):
    """-> (float64 device tensor, stride).  Accepts one 5-sequence (broadcast) or [B,5]."""
    if isinstance(params, torch.Tensor):
        p = params.to(device=device, dtype=torch.float64).contiguous()
    else:
        p = torch.as_tensor(np.asarray(params, dtype=np.float64), device=device)
    if p.ndim == 1:
        if p.numel() != 5:
            raise ValueError("params must have 5 entries: pos_thres, neg_thres, base_noise_std, "
                             "hot_pixel_fraction, hot_pixel_std")
        return p, 0
    if tuple(p.shape) != (batch, 5):
        raise ValueError(f"params must be [5] or [{batch},5], got {tuple(p.shape)}")
    return p, 5


def esim_voxel_batch(frames: torch.Tensor, params, *, bin_mode: str = "sum", num_bins: int = 5,
                     frames_per_bin: int = 1, rng_mode: str = "philox", seed: int = 0, clip_id0: int = 0,
                     put_noise_external: bool = False, out_dtype: torch.dtype = torch.float32,
                     out: Optional[torch.Tensor] = None, counts: Optional[torch.Tensor] = None,
                     replay: Optional[Sequence[torch.Tensor]] = None, validate: bool = True,
                     no_noise: Optional[bool] = None, clip_keys: Optional[torch.Tensor] = None, pad_to: int = 1,
                     symmetric: Optional[bool] = None, mapping: Optional[str] = None, stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Simulate a batch of clips and bin the events, in one kernel launch on the current stream.

    frames  [B,N,H,W] uint8 or float32 CUDA tensor (grayscale; dims 2,3 contiguous).
    params  [5] or [B,5]: pos_thres, neg_thres, base_noise_std, hot_pixel_fraction, hot_pixel_std
            (EventEmulator's constructor arguments, data/v2v_core_esim.py:8-16).
    Returns [B,L,Tb,H,W] ("sum", data/v2v_datasets.py:399-400) or [B,Tb,H,W] ("bilinear").
    counts  optional int64 [B,2] tensor; ON/OFF event totals per clip are ADDED into it.
    clip_keys  optional int64 [B,2] tensor of per-clip {seed, clip id} (overrides seed / clip_id0 + b): clip b then
            gets exactly the result of simulating it alone with that seed and clip id.
    pad_to    > 1: the voxel planes are written straight into a buffer whose H, W are padded with zeros to multiples of
            pad_to (forward_sequence's PAD = 16, model/train_utils.py:322-326); the returned tensor has the padded shape.
    no_noise  True asserts base_noise_std == 0 and hot_pixel_fraction == 0 for every clip, which selects the
            kernel variant without the noise adds (identical results).  Default: detected from `params` when
            they are host values, False when `params` is already a device tensor.
    symmetric True asserts pos_thres == neg_thres for every clip (EventEmulator's own defaults): instances compiled without
            the asymmetric loop, 4 waves per SIMD (identical results; a clip that breaks the promise comes out as NaN).
            Default: detected from host `params`, False for a device tensor.
    stats     optional int32 [B, VOXEL_STATS_WORDS] CUDA tensor: the writer's per-clip value histogram for normalize_batch_voxel
            (v2v_esim_voxel_stats_hip; zeroed and filled by the launch; SUM mode, float32 grid, no external noise) -> feed it to
            v2v_amd.postops.scales_from_stats.
    mapping   "auto" (4 pixels per work-item for aligned layouts unless the batch is small, then 2 or 1), "4px", "2px" or "1px" to pin it;
            results do not depend on it.  None = the module default DEFAULT_MAPPING ("auto"; the test-suite sweeps it).
    """
    _lib.require_gpu()
    if frames.ndim != 4:
        raise ValueError("frames must be [B,N,H,W]")
    if not frames.is_cuda:
        raise ValueError("frames must be a CUDA (ROCm) tensor")
    if frames.dtype not in _TORCH_IN:
        raise ValueError(f"unsupported input dtype {frames.dtype}: use uint8 or float32")
    b, n, h, w = frames.shape
    if frames.stride(3) != 1 or frames.stride(2) != w:
        frames = frames.contiguous()
    p, pstride = _params_tensor(params, b, frames.device)
    if not isinstance(params, torch.Tensor):
        pa = np.asarray(params, dtype=np.float64).reshape(-1, 5)
        if validate and not (np.all(pa[:, 0] > 0) and np.all(pa[:, 1] > 0)):
            raise ValueError("pos_thres and neg_thres must be > 0")
        if validate and (np.any(pa[:, 0] < 1e-9) or np.any(pa[:, 1] < 1e-9)):
            raise ValueError("thresholds below 1e-9 are outside the exact floor-divide domain (|potential|/C must stay < 2^40)")
        if no_noise is None:
            no_noise = bool(np.all(pa[:, 2] == 0) and np.all(pa[:, 3] <= 0))
        if symmetric is None:
            symmetric = bool(np.all(pa[:, 0] == pa[:, 1]))
    no_noise = bool(no_noise) and not put_noise_external and rng_mode != "replay"
    k = n - 1
    if bin_mode not in BIN_MODES:
        raise ValueError(f"bin_mode must be one of {list(BIN_MODES)}")
    hp, wp = (h + pad_to - 1) // pad_to * pad_to, (w + pad_to - 1) // pad_to * pad_to
    if bin_mode == "sum":
        assert k % (num_bins * frames_per_bin) == 0, "(N-1) % (num_bins*frames_per_bin) != 0"   # v2v_datasets.py:365
        shape = (b, k // (num_bins * frames_per_bin), num_bins, hp, wp)
    else:
        shape = (b, num_bins, hp, wp)
    if out is None:
        out = torch.empty(shape, dtype=out_dtype, device=frames.device)
        if hp != h:
            out[..., h:, :] = 0                                 # only the padding is zeroed; the kernel writes every interior element
        if wp != w:
            out[..., :h, w:] = 0
        if b == 0:
            return out
    elif tuple(out.shape) != shape or not out.is_contiguous() or out.dtype not in _OUT or out.device != frames.device:
        raise ValueError(f"out must be a contiguous {shape} float32/float64 tensor on {frames.device}")
    rp = None
    keep = None
    if rng_mode == "replay":
        if replay is None or len(replay) != 4:
            raise ValueError("rng_mode='replay' needs replay=(u_init[B,H,W], u_hot[B,H,W], g_hot[B,H,W], "
                             "g_base[B,N-1,H,W])")
        keep = [torch.as_tensor(x).to(device=frames.device, dtype=torch.float64).contiguous() for x in replay]
        want = [(b, h, w)] * 3 + [(b, k, h, w)]
        for t, s in zip(keep, want):
            if tuple(t.shape) != s:
                raise ValueError(f"replay field shape {tuple(t.shape)} != {s}")
        rp = _lib.EsimReplay(*[t.data_ptr() for t in keep])
    elif rng_mode not in RNG_MODES:
        raise ValueError(f"rng_mode must be one of {list(RNG_MODES)}")
    if counts is not None:
        if counts.dtype != torch.int64 or tuple(counts.shape) != (b, 2) or not counts.is_contiguous() \
                or counts.device != frames.device:
            raise ValueError("counts must be a contiguous int64 [B,2] tensor on the frames' device")
    if clip_keys is not None:
        clip_keys = torch.as_tensor(clip_keys).to(device=frames.device, dtype=torch.int64).contiguous()
        if tuple(clip_keys.shape) != (b, 2):
            raise ValueError("clip_keys must be [B,2] (seed, clip id)")
    if stats is not None:
        if stats.dtype != torch.int32 or tuple(stats.shape) != (b, _lib.VOXEL_STATS_WORDS) or not stats.is_contiguous() or stats.device != frames.device:
            raise ValueError(f"stats must be a contiguous int32 [{b},{_lib.VOXEL_STATS_WORDS}] tensor on the frames' device")
    fn = _lib.lib().v2v_esim_voxel_padded_hip if stats is None else _lib.lib().v2v_esim_voxel_stats_hip
    extra = () if stats is None else (C.c_void_p(stats.data_ptr()),)
    with torch.cuda.device(frames.device):
        rc = fn(
            C.c_void_p(frames.data_ptr()), _TORCH_IN[frames.dtype], b, n, h, w,
            frames.stride(0) if b > 1 else n * frames.stride(1), frames.stride(1),
            C.c_void_p(p.data_ptr()), pstride,
            (_lib.FLAG_NOISE_EXTERNAL if put_noise_external else 0) | (_lib.FLAG_NO_NOISE if no_noise else 0) | (_lib.FLAG_SYMMETRIC if symmetric else 0) | {"auto": 0, "4px": _lib.FLAG_MAP_4PX, "2px": _lib.FLAG_MAP_2PX, "1px": _lib.FLAG_MAP_1PX}[mapping or DEFAULT_MAPPING],
            RNG_MODES[rng_mode], C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), C.c_uint64(clip_id0),
            C.c_void_p(clip_keys.data_ptr()) if clip_keys is not None else None,
            C.byref(rp) if rp is not None else None, BIN_MODES[bin_mode], num_bins, frames_per_bin,
            C.c_void_p(out.data_ptr()), _OUT[out.dtype], wp, hp * wp,
            C.c_void_p(counts.data_ptr()) if counts is not None else None, *extra, _lib.stream_ptr())
    _lib.check(rc)
    # The launch is asynchronous: PyTorch's caching allocator keeps freed blocks stream-ordered, so dropping
    # p / keep here is safe for work queued on the same stream.
    return out


def esim_voxel_packed(frames: torch.Tensor, clip_offsets: torch.Tensor, frame_index: torch.Tensor, h: int, w: int, params: torch.Tensor,
                      clip_keys: torch.Tensor, *, num_bins: int = 5, frames_per_bin: int = 1, pad_to: int = 1, stats: Optional[torch.Tensor] = None,
                      align: int = 16, mapping: Optional[str] = None, stored_frames: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The batch launch over PACKED clips (v2v_esim_voxel_ex_hip with v2v_esim_extras): `frames` is one flat uint8 CUDA buffer, clip b starts
    at element clip_offsets[b] (int64 [B], multiples of `align`) and holds its DECODED frames once each; frame_index int32 [B,N] names the
    stored frame every simulator frame shows -- the reference's pause-index gather (data/v2v_datasets.py:286-311) done by the kernel's
    loads.  SUM bins, device-native noise, float32 grid; params float64 [B,5] and clip_keys int64 [B,2] on the device.  Same results as
    esim_voxel_batch on the gathered clips (tests/test_loader.py).  stored_frames int32 [B] on the device = frames each clip holds: the
    launch then checks every index row (and that the clip fits `frames`) while it stages it -- a clip that fails comes out as NaN planes with
    its `stats` flag set, nothing is read out of bounds; None = the caller vouches for the rows.  -> [B, L, num_bins, Hp, Wp]"""
    _lib.require_gpu()
    dev = frames.device
    if frames.dtype != torch.uint8 or frames.ndim != 1 or not frames.is_cuda:
        raise ValueError("frames must be a flat uint8 CUDA buffer")
    b, n = frame_index.shape
    for t, dt, shape in ((clip_offsets, torch.int64, (b,)), (frame_index, torch.int32, (b, n)), (params, torch.float64, (b, 5)), (clip_keys, torch.int64, (b, 2))):
        if t.dtype != dt or tuple(t.shape) != shape or t.device != dev or not t.is_contiguous():
            raise ValueError(f"expected a contiguous {dt} {shape} tensor on {dev}")
    k = n - 1
    assert k % (num_bins * frames_per_bin) == 0, "(N-1) % (num_bins*frames_per_bin) != 0"   # v2v_datasets.py:365
    hp, wp = (h + pad_to - 1) // pad_to * pad_to, (w + pad_to - 1) // pad_to * pad_to
    out = torch.empty((b, k // (num_bins * frames_per_bin), num_bins, hp, wp), dtype=torch.float32, device=dev)
    if hp != h:
        out[..., h:, :] = 0
    if wp != w:
        out[..., :h, w:] = 0
    if b == 0:
        return out
    if stats is not None and (stats.dtype != torch.int32 or tuple(stats.shape) != (b, _lib.VOXEL_STATS_WORDS) or not stats.is_contiguous() or stats.device != dev):
        raise ValueError(f"stats must be a contiguous int32 [{b},{_lib.VOXEL_STATS_WORDS}] tensor on the frames' device")
    if stored_frames is not None and (stored_frames.dtype != torch.int32 or tuple(stored_frames.shape) != (b,) or stored_frames.device != dev or not stored_frames.is_contiguous()):
        raise ValueError(f"stored_frames must be a contiguous int32 [{b}] tensor on {dev}")
    ex = _lib.EsimExtras(stats.data_ptr() if stats is not None else None, frame_index.data_ptr(), clip_offsets.data_ptr(),
                         stored_frames.data_ptr() if stored_frames is not None else None, frames.numel() if stored_frames is not None else 0)
    with torch.cuda.device(dev):
        rc = _lib.lib().v2v_esim_voxel_ex_hip(
            C.c_void_p(frames.data_ptr()), _lib.U8, b, n, h, w, align, h * w, C.c_void_p(params.data_ptr()), 5,
            {"auto": 0, "4px": _lib.FLAG_MAP_4PX, "2px": _lib.FLAG_MAP_2PX, "1px": _lib.FLAG_MAP_1PX}[mapping or DEFAULT_MAPPING],
            _lib.RNG_PHILOX, C.c_uint64(0), C.c_uint64(0), C.c_void_p(clip_keys.data_ptr()), None, _lib.BIN_SUM, num_bins, frames_per_bin,
            C.c_void_p(out.data_ptr()), _lib.F32, wp, hp * wp, None, C.byref(ex), _lib.stream_ptr())
    _lib.check(rc)
    return out


def algorithmic_bytes(frames_dtype: torch.dtype, b: int, n: int, h: int, w: int, bin_mode: str, num_bins: int,
                      frames_per_bin: int = 1, out_dtype: torch.dtype = torch.float32) -> int:
    """HBM bytes one launch must move: every input byte read once + every voxel byte written once."""
    v = _lib.lib().v2v_esim_voxel_bytes(_TORCH_IN[frames_dtype], b, n, h, w, BIN_MODES[bin_mode], num_bins,
                                        frames_per_bin, _OUT[out_dtype])
    if v < 0:
        _lib.check(int(v))
    return int(v)


def synth_clips(b: int, n: int, h: int, w: int, *, dtype: torch.dtype = torch.float32, seed: int = 20240001,
                clip_id0: int = 0, device="cuda") -> torch.Tensor:
    """Device-generated synthetic clips: integer-valued 0..255, [B,N,H,W]."""
    _lib.require_gpu()
    out = torch.empty((b, n, h, w), dtype=dtype, device=device)
    with torch.cuda.device(out.device):
        rc = _lib.lib().v2v_synth_clips_hip(C.c_void_p(out.data_ptr()), _TORCH_IN[dtype], b, n, h, w,
                                            C.c_uint64(seed), C.c_uint64(clip_id0), _lib.stream_ptr())
    _lib.check(rc)
    return out


def draw_numpy_replay_fields(n: int, h: int, w: int):
    """Draw the simulator's random fields from the GLOBAL np.random stream in exactly the order the
    reference consumes it (data/v2v_core_esim.py:29 rand, :37 rand, :38 randn, :44 one randn per pair),
    so that `np.random.seed(s)` followed by EventEmulator(rng='numpy') reproduces the reference bit for bit."""
    u_init = np.random.rand(h, w)
    u_hot = np.random.rand(h, w)
    g_hot = np.random.randn(h, w)
    g_base = np.empty((n - 1, h, w))
    for i in range(n - 1):
        g_base[i] = np.random.randn(h, w)
    return u_init, u_hot, g_hot, g_base


class EventEmulator(object):
    """Drop-in for data/v2v_core_esim.py:EventEmulator (same constructor, same video_to_voxel).

    rng: 'numpy'  - the reference's behaviour: fields come from the global np.random stream (drawn on the
                    host in the reference's order, replayed on the GPU); bit-exact with the reference for
                    integer-valued input.  The reference ignores `seed`; so does this mode.
         'philox' - device-native counter RNG keyed by `seed` (drawn from np.random when None) and
                    `clip_id`; no host RNG work, results independent of batching.  Statistically
                    equivalent to the reference, pinned to it by golden G11.
    """

    def __init__(self, pos_thres: float = 0.2, neg_thres: float = 0.2, base_noise_std: float = 0.1,
                 hot_pixel_fraction: float = 0.001, hot_pixel_std: float = 0.1, put_noise_external: bool = False,
                 seed: int = None, rng: str = "numpy", clip_id: int = 0, device="cuda"):
        self.pos_threshold = pos_thres
        self.neg_threshold = neg_thres
        self.base_noise_std = base_noise_std
        self.hot_pixel_fraction = hot_pixel_fraction
        self.hot_pixel_std = hot_pixel_std
        self.put_noise_external = put_noise_external
        self.seed = seed
        if rng not in ("numpy", "philox"):
            raise ValueError("rng must be 'numpy' or 'philox'")
        self.rng = rng
        self.clip_id = clip_id
        self.device = device

    def _params(self):
        return [self.pos_threshold, self.neg_threshold, self.base_noise_std, self.hot_pixel_fraction,
                self.hot_pixel_std]

    def video_to_voxel(self, video):
        """video [N,H,W] (NumPy array or torch tensor) -> signed event counts per frame pair [N-1,H,W]
        (float64 ndarray for NumPy input, like the reference; float32 CUDA tensor for CUDA input)."""
        is_np = isinstance(video, np.ndarray)
        if is_np:
            v = video
            if v.dtype != np.uint8 and v.dtype != np.float32:
                iv = v.astype(np.int64)
                if np.array_equal(iv, v) and iv.min(initial=0) >= 0 and iv.max(initial=0) <= 255:
                    v = iv.astype(np.uint8)          # integer-valued: the float64 table path is exact
                else:
                    # The reference would run its pow/log in float64 here (v2v_core_esim.py:33-34).  The fused kernel's generic
                    # (non-table) path is float32: same algorithm, log intensities accurate to ~1e-7 instead of ~1e-16, so a
                    # potential that ends within that distance of a threshold multiple can yield a count that differs by one.
                    import warnings
                    warnings.warn("EventEmulator.video_to_voxel: non-integer %s video is simulated through the float32 log path; "
                                  "counts can differ from the reference's float64 path on ~1e-5 of the pixel-steps" % v.dtype,
                                  RuntimeWarning, stacklevel=2)
                    v = v.astype(np.float32)
            frames = torch.from_numpy(np.ascontiguousarray(v)).to(self.device)
        else:
            frames = video if video.is_cuda else video.to(self.device)
            if frames.dtype not in _TORCH_IN:
                frames = frames.to(torch.float32)
        if frames.ndim != 3:
            raise ValueError("video must be [N,H,W]")
        n, h, w = frames.shape
        kw = dict(bin_mode="sum", num_bins=n - 1, frames_per_bin=1, put_noise_external=self.put_noise_external,
                  out_dtype=torch.float64 if is_np else torch.float32)
        if self.rng == "numpy":
            fields = draw_numpy_replay_fields(n, h, w)
            out = esim_voxel_batch(frames[None], self._params(), rng_mode="replay",
                                   replay=[torch.from_numpy(f)[None] for f in fields], **kw)
        else:
            seed = self.seed if self.seed is not None else int(np.random.randint(0, 2**31 - 1))
            out = esim_voxel_batch(frames[None], self._params(), rng_mode="philox", seed=seed,
                                   clip_id0=self.clip_id, **kw)
        out = out[0, 0]
        return out.cpu().numpy() if is_np else out
