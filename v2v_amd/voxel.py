"""Event-list -> voxel-grid: host side of v2v_events_to_voxel_hip.

Mirrors (same names, argument order, shapes and dtypes):
    make_voxel(evs, H, W, num_bins=5, interpolate_bins=True)     scripts/visualize_esim_sample.py:113-135
    MakeVoxelMixin.make_voxel(self, evs)                         data/testh5.py:60-90  (TestH5Dataset.make_voxel)
    events_to_voxel(xs, ys, ts, ps, B, sensor_size, temporal_bilinear)   utils/event_utils.py:692-728
    events_to_voxel_torch / events_to_voxel_timesync_torch / voxel_grids_fixed_n_torch / voxel_grids_fixed_t_torch   utils/event_utils.py:378-507
NumPy in -> NumPy float64 out; CUDA tensors in -> CUDA float64 tensor out.  The scatter runs in the HIP kernel
(v2v_amd/csrc/v2v_events.hpp); nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


def _dev(x, dtype, device):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype).contiguous().reshape(-1)
    return torch.as_tensor(np.ascontiguousarray(np.asarray(x).reshape(-1)), device=device).to(dtype)


def _scatter(ts, xs, ys, ps, mode, num_bins, h, w, device, check_bounds=True):
    _lib.require_gpu()
    ts_d = _dev(ts, torch.float64, device)
    xs_d = _dev(xs, torch.int64, device)
    ys_d = _dev(ys, torch.int64, device)
    ps_d = _dev(ps, torch.float64, device)
    n = ts_d.numel()
    if not (xs_d.numel() == n and ys_d.numel() == n and ps_d.numel() == n):
        raise AssertionError("len(xs)==len(ys)==len(ts)==len(ps) violated")      # event_utils.py:710
    out = torch.empty((num_bins, h, w), dtype=torch.float64, device=device)
    dropped = torch.empty((1,), dtype=torch.int64, device=device)
    with torch.cuda.device(out.device):
        rc = _lib.lib().v2v_events_to_voxel_hip(
            C.c_void_p(ts_d.data_ptr()) if n else None, C.c_void_p(xs_d.data_ptr()) if n else None,
            C.c_void_p(ys_d.data_ptr()) if n else None, C.c_void_p(ps_d.data_ptr()) if n else None, n, mode, num_bins,
            h, w, C.c_void_p(out.data_ptr()), C.c_void_p(dropped.data_ptr()), _lib.stream_ptr())
    _lib.check(rc)
    if check_bounds and n and int(dropped.item()) != 0:
        raise IndexError(f"{int(dropped.item())} event(s) outside the {h}x{w} sensor / {num_bins} bins")   # np.add.at would raise
    return out


def make_voxel(evs, H, W, num_bins=5, interpolate_bins=True, device="cuda"):
    """evs = [ts, xs, ys, ps] (ps in {0,1}) -> [num_bins,H,W] float64.  scripts/visualize_esim_sample.py:113-135."""
    ts, xs, ys, ps = evs
    is_np = not isinstance(ts, torch.Tensor)
    if isinstance(ts, torch.Tensor) and ts.is_cuda:
        device = ts.device
    mode = _lib.EV_MAKE_VOXEL_INTERP if interpolate_bins else _lib.EV_MAKE_VOXEL_DISCRETE
    out = _scatter(ts, xs, ys, ps, mode, num_bins, H, W, device)
    return out.cpu().numpy() if is_np else out


class MakeVoxelMixin:
    """`make_voxel(self, evs)` exactly as TestH5Dataset.make_voxel (data/testh5.py:60-90): reads self.num_bins,
    self.H, self.W, self.interpolate_bins."""

    def make_voxel(self, evs):
        return make_voxel(evs, self.H, self.W, self.num_bins, self.interpolate_bins)


def events_to_voxel(xs, ys, ts, ps, B, sensor_size=(180, 240), temporal_bilinear=True, device="cuda"):
    """utils/event_utils.py:692-728.  ts/ps may be 1-D or the [N,1] columns the reference requires.
    temporal_bilinear=False is broken in the reference (undefined `weights`): NotImplementedError here;
    discrete bins are make_voxel(..., interpolate_bins=False)."""
    if not temporal_bilinear:
        raise NotImplementedError("the reference's temporal_bilinear=False branch references an undefined variable")
    is_np = not isinstance(ts, torch.Tensor)
    if isinstance(ts, torch.Tensor) and ts.is_cuda:
        device = ts.device
    out = _scatter(ts, xs, ys, ps, _lib.EV_BILINEAR, B, sensor_size[0], sensor_size[1], device)
    return out.cpu().numpy() if is_np else out


def events_to_voxel_torch(xs, ys, ts, ps, B, device=None, sensor_size=(180, 240), temporal_bilinear=True):
    """utils/event_utils.py:466-507 (float32 torch twin; only caller data/dataset.py:328).  Tensors in, float32 tensor
    [B,H,W] out on `device` (default: a CUDA device -- the scatter runs in the HIP kernel)."""
    _lib.require_gpu()
    assert len(xs) == len(ys) and len(ys) == len(ts) and len(ts) == len(ps)                       # :487
    dev = torch.device(device) if device is not None else (xs.device if getattr(xs, "is_cuda", False) else torch.device("cuda"))
    if dev.type != "cuda":
        raise RuntimeError("events_to_voxel_torch runs on the GPU only (no CPU fallback)")
    ts_d = _dev(ts, torch.float32, dev)
    ps_d = _dev(ps, torch.float32, dev)
    xs_d = _dev(xs, torch.int64, dev)
    ys_d = _dev(ys, torch.int64, dev)
    n = ts_d.numel()
    h, w = sensor_size
    out = torch.empty((B, h, w), dtype=torch.float32, device=dev)
    dropped = torch.empty((1,), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().v2v_events_to_voxel_f32_hip(
            C.c_void_p(ts_d.data_ptr()) if n else None, C.c_void_p(xs_d.data_ptr()) if n else None,
            C.c_void_p(ys_d.data_ptr()) if n else None, C.c_void_p(ps_d.data_ptr()) if n else None, n,
            0 if temporal_bilinear else 1, B, h, w, C.c_void_p(out.data_ptr()), C.c_void_p(dropped.data_ptr()),
            _lib.stream_ptr())
    _lib.check(rc)
    if n and int(dropped.item()) != 0:
        raise IndexError(f"{int(dropped.item())} event(s) outside the sensor / bin range")
    return out


def events_to_neg_pos_voxel(xs, ys, ts, ps, B, sensor_size=(180, 240), temporal_bilinear=True, device="cuda"):
    """utils/event_utils.py:730-759: separate grids for positive (`ps` truthy) and negative events -> (voxel_pos, voxel_neg)."""
    is_t = isinstance(ps, torch.Tensor)
    pos_w = torch.where(ps != 0, 1, 0) if is_t else np.where(ps, 1, 0)
    neg_w = torch.where(ps != 0, 0, 1) if is_t else np.where(ps, 0, 1)
    return (events_to_voxel(xs, ys, ts, pos_w, B, sensor_size, temporal_bilinear, device),
            events_to_voxel(xs, ys, ts, neg_w, B, sensor_size, temporal_bilinear, device))


def events_to_neg_pos_voxel_torch(xs, ys, ts, ps, B, device=None, sensor_size=(180, 240), temporal_bilinear=True):
    """utils/event_utils.py:509-541 (float32 torch twin): weights 1 where ps > 0 / where ps <= 0."""
    one, zero = torch.ones((), dtype=torch.float32, device=ps.device), torch.zeros((), dtype=torch.float32, device=ps.device)
    pos_w, neg_w = torch.where(ps > 0, one, zero), torch.where(ps <= 0, one, zero)
    return (events_to_voxel_torch(xs, ys, ts, pos_w, B, device, sensor_size, temporal_bilinear),
            events_to_voxel_torch(xs, ys, ts, neg_w, B, device, sensor_size, temporal_bilinear))


def _grids_of_ranges(xs, ys, ts, ps, B, ranges, device, sensor_size, temporal_bilinear):
    """float32 grids of the event ranges [a, b): ONE segmented launch when the ranges tile a contiguous stretch (grid f = events
    [seg[f], seg[f+1]), each with its own time normalisation, exactly one events_to_voxel_torch call per range), one call per range
    otherwise."""
    dev = torch.device(device) if device is not None else (xs.device if getattr(xs, "is_cuda", False) else torch.device("cuda"))
    if len(ranges) > 1 and all(ranges[i][1] == ranges[i + 1][0] for i in range(len(ranges) - 1)):
        _lib.require_gpu()
        lo, hi = ranges[0][0], ranges[-1][1]
        ts_d = _dev(ts[lo:hi], torch.float64, dev)                                    # float32 in: ts - ts[seg] is exact in float64, then cast (:194 rule)
        xs_d, ys_d, ps_d = _dev(xs[lo:hi], torch.int64, dev), _dev(ys[lo:hi], torch.int64, dev), _dev(ps[lo:hi], torch.float32, dev)
        seg = torch.tensor([a - lo for a, _ in ranges] + [hi - lo], dtype=torch.int64, device=dev)
        h, w = sensor_size
        out = torch.empty((len(ranges), B, h, w), dtype=torch.float32, device=dev)
        dropped = torch.empty((1,), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().v2v_events_to_voxel_f32_segmented_hip(
                C.c_void_p(ts_d.data_ptr()), C.c_void_p(xs_d.data_ptr()), C.c_void_p(ys_d.data_ptr()), C.c_void_p(ps_d.data_ptr()), hi - lo,
                C.c_void_p(seg.data_ptr()), len(ranges), 0, 0 if temporal_bilinear else 1, B, h, w, C.c_void_p(out.data_ptr()),
                C.c_void_p(dropped.data_ptr()), _lib.stream_ptr())
        _lib.check(rc)
        if int(dropped.item()) != 0:
            raise IndexError(f"{int(dropped.item())} event(s) outside the sensor / bin range")
        return list(out.unbind(0))
    return [events_to_voxel_torch(xs[a:b], ys[a:b], ts[a:b], ps[a:b], B, dev, sensor_size, temporal_bilinear) for a, b in ranges]


def voxel_grids_fixed_n_torch(xs, ys, ts, ps, B, n, sensor_size=(180, 240), temporal_bilinear=True):
    """utils/event_utils.py:378-402: a list of voxel grids of `n` consecutive events each (the last, incomplete group is dropped, and
    so is a complete one that ends exactly at the last event: `range(0, len - n, n)`).  One launch for the list."""
    return _grids_of_ranges(xs, ys, ts, ps, B, [(i, i + n) for i in range(0, len(xs) - n, n)], None, sensor_size, temporal_bilinear)


def events_to_voxel_timesync_torch(xs, ys, ts, ps, B, t0, t1, device=None, np_ts=None, sensor_size=(180, 240), temporal_bilinear=True):
    """utils/event_utils.py:440-464: the grid of the events with t0 <= t < t1 (np.searchsorted on the timestamps; both asserts kept)."""
    assert t1 > t0
    if np_ts is None:
        np_ts = ts.cpu().numpy() if isinstance(ts, torch.Tensor) else np.asarray(ts)
    a, b = int(np.searchsorted(np_ts, t0)), int(np.searchsorted(np_ts, t1))
    assert a < b
    return events_to_voxel_torch(xs[a:b], ys[a:b], ts[a:b], ps[a:b], B, device, sensor_size, temporal_bilinear)


def voxel_grids_fixed_t_torch(xs, ys, ts, ps, B, t, sensor_size=(180, 240), temporal_bilinear=True):
    """utils/event_utils.py:404-438: a list of voxel grids of temporal width `t`, starting at np.arange(ts[0], ts[-1] - t, t); every window's
    borders are looked up on their own (t_start + t is not bit-for-bit the next t_start), so the windows go out as one segmented launch only
    when the lookups happen to tile the events."""
    np_ts = ts.cpu().numpy() if isinstance(ts, torch.Tensor) else np.asarray(ts)
    ranges = []
    for t_start in np.arange(float(ts[0]), float(ts[-1]) - t, t):
        a, b = int(np.searchsorted(np_ts, t_start)), int(np.searchsorted(np_ts, t_start + t))
        assert a < b
        ranges.append((a, b))
    return _grids_of_ranges(xs, ys, ts, ps, B, ranges, None, sensor_size, temporal_bilinear)


def make_voxels_segmented(evs, event_idx, H, W, num_bins=5, interpolate_bins=False, device="cuda"):
    """All voxel grids of a sequence in ONE launch: grid f = make_voxel(events[event_idx[f]:event_idx[f+1]]), exactly
    what TestH5Dataset.__getitem__ (data/testh5.py:111-119) builds with one make_voxel call per image.
    evs = [ts, xs, ys, ps] for the whole range, event_idx = ascending offsets [F+1].  Returns [F,num_bins,H,W] float64."""
    _lib.require_gpu()
    ts, xs, ys, ps = evs
    is_np = not isinstance(ts, torch.Tensor)
    if isinstance(ts, torch.Tensor) and ts.is_cuda:
        device = ts.device
    ts_d, xs_d, ys_d, ps_d = _dev(ts, torch.float64, device), _dev(xs, torch.int64, device), _dev(ys, torch.int64, device), \
        _dev(ps, torch.float64, device)
    n = ts_d.numel()
    seg_h = np.asarray(event_idx.cpu() if isinstance(event_idx, torch.Tensor) else event_idx, dtype=np.int64).reshape(-1)
    if seg_h.size < 2 or np.any(np.diff(seg_h) < 0) or seg_h[0] < 0 or seg_h[-1] > n:
        raise ValueError("event_idx must be ascending offsets within the event arrays")
    f = seg_h.size - 1
    seg = torch.as_tensor(seg_h, device=ts_d.device)
    out = torch.empty((f, num_bins, H, W), dtype=torch.float64, device=ts_d.device)
    dropped = torch.empty((1,), dtype=torch.int64, device=ts_d.device)
    mode = _lib.EV_MAKE_VOXEL_INTERP if interpolate_bins else _lib.EV_MAKE_VOXEL_DISCRETE
    with torch.cuda.device(out.device):
        rc = _lib.lib().v2v_events_to_voxel_segmented_hip(
            C.c_void_p(ts_d.data_ptr()) if n else None, C.c_void_p(xs_d.data_ptr()) if n else None,
            C.c_void_p(ys_d.data_ptr()) if n else None, C.c_void_p(ps_d.data_ptr()) if n else None, n,
            C.c_void_p(seg.data_ptr()), f, mode, num_bins, H, W, C.c_void_p(out.data_ptr()), C.c_void_p(dropped.data_ptr()),
            _lib.stream_ptr())
    _lib.check(rc)
    if n and int(dropped.item()) != 0:
        raise IndexError(f"{int(dropped.item())} event(s) outside the {H}x{W} sensor / {num_bins} bins")
    return out.cpu().numpy() if is_np else out
