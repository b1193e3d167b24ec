"""Event-list -> voxel-grid: host side of v2v_events_to_voxel_hip.

Mirrors (same names, argument order, shapes and dtypes):
    make_voxel(evs, H, W, num_bins=5, interpolate_bins=True)     scripts/visualize_esim_sample.py:113-135
    MakeVoxelMixin.make_voxel(self, evs)                         data/testh5.py:60-90  (TestH5Dataset.make_voxel)
    events_to_voxel(xs, ys, ts, ps, B, sensor_size, temporal_bilinear)   utils/event_utils.py:692-728
    events_to_voxel_torch / events_to_voxel_timesync_torch / voxel_grids_fixed_n_torch / voxel_grids_fixed_t_torch   utils/event_utils.py:378-507
    events_to_image / events_to_image_torch / interpolate_to_image                                              utils/event_utils.py:155-184, 330-376
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


def events_to_image(xs, ys, ps, sensor_size=(180, 240), interpolation=None, padding=False, device="cuda"):
    """utils/event_utils.py:155-174: img[y, x] = sum of the weights `ps` of the events at (x, y), float64 [H, W] (np.bincount over the
    raveled coordinates).  It is the one-bin case of the temporal-bilinear voxel grid -- with a single bin every event's weight is 1 -- so it
    runs on the same scatter kernel; an event outside the sensor raises ValueError as np.ravel_multi_index does.
    interpolation='bilinear' goes through events_to_image_torch for EVERY NumPy dtype, integer coordinates included, as in the reference
    (:160 tests `xs.dtype is not torch.long` on a NumPy array, which is always true): events at x >= W-1 or y >= H-1 are masked out
    (clip_out_of_range), not counted."""
    if interpolation == "bilinear":
        img = events_to_image_torch(torch.from_numpy(np.asarray(xs)).float(), torch.from_numpy(np.asarray(ys)).float(), torch.from_numpy(np.asarray(ps)).float(),
                                    clip_out_of_range=True, interpolation="bilinear", padding=padding, sensor_size=sensor_size)
        return img.cpu().numpy().reshape(sensor_size)
    xs_a, ys_a, ps_a = (np.asarray(v).reshape(-1) for v in (xs, ys, ps))
    n = xs_a.size
    # two zero-weight sentinel events at t = 0 and t = 1 (the real ones at t = 0) give the kernel a time span whatever n is; with one bin
    # they change no weight
    ts_a = np.concatenate([np.zeros(n + 1), [1.0]])
    try:
        out = _scatter(ts_a, np.concatenate([[0], xs_a, [0]]), np.concatenate([[0], ys_a, [0]]), np.concatenate([[0.0], ps_a.astype(np.float64), [0.0]]),
                       _lib.EV_BILINEAR, 1, sensor_size[0], sensor_size[1], device)
    except IndexError as e:
        raise ValueError(str(e)) from None                                                # np.ravel_multi_index: "invalid entry in coordinates array"
    return out[0].cpu().numpy()


def _scatter_f32(xs_d, ys_d, w_d, h, w, dev):
    """sum of float32 weights at integer coordinates -> [h, w] float32: the one-bin discrete case of the float32 voxel kernel."""
    n = xs_d.numel()
    # (negative indices wrap ONCE inside the kernel, like index_put_ -- the reference's accumulator, utils/event_utils.py:180-183, :375;
    # below -size the event is counted in `dropped` and raises here, as torch does)
    out = torch.empty((1, h, w), dtype=torch.float32, device=dev)
    dropped = torch.empty((1,), dtype=torch.int64, device=dev)
    ts_d = torch.zeros((max(n, 1),), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().v2v_events_to_voxel_f32_hip(C.c_void_p(ts_d.data_ptr()) if n else None, C.c_void_p(xs_d.data_ptr()) if n else None,
                                                    C.c_void_p(ys_d.data_ptr()) if n else None, C.c_void_p(w_d.data_ptr()) if n else None, n, 1, 1, h, w,
                                                    C.c_void_p(out.data_ptr()), C.c_void_p(dropped.data_ptr()), _lib.stream_ptr())
    _lib.check(rc)
    if n and int(dropped.item()) != 0:
        raise IndexError(f"{int(dropped.item())} event(s) outside the {h}x{w} image")   # index_put_ would raise
    return out[0]


def interpolate_to_image(pxs, pys, dxs, dys, weights, img):
    """utils/event_utils.py:176-184: bilinear splatting -- every event adds weights * {(1-dx)(1-dy), dx(1-dy), (1-dx)dy, dx dy} to the four
    pixels around it, accumulated into `img` (float32 CUDA [H, W]) in place.  The four weight products are formed in float32 exactly as the
    reference writes them; the sums run in the scatter kernel."""
    if not (isinstance(img, torch.Tensor) and img.is_cuda and img.dtype == torch.float32 and img.dim() == 2):
        raise ValueError("img must be a float32 CUDA tensor [H, W] (the scatter runs in the HIP kernel; there is no CPU fallback)")
    dev = img.device
    h, w = img.shape
    px, py = _dev(pxs, torch.int64, dev), _dev(pys, torch.int64, dev)
    dx, dy, wt = _dev(dxs, torch.float32, dev), _dev(dys, torch.float32, dev), _dev(weights, torch.float32, dev)
    for ox, oy, wk in ((0, 0, wt * (1.0 - dx) * (1.0 - dy)), (1, 0, wt * dx * (1.0 - dy)), (0, 1, wt * (1.0 - dx) * dy), (1, 1, wt * dx * dy)):
        img += _scatter_f32((px + ox).contiguous(), (py + oy).contiguous(), wk.contiguous(), h, w, dev)
    return img


def events_to_image_torch(xs, ys, ps, device=None, sensor_size=(180, 240), clip_out_of_range=True, interpolation=None, padding=True):
    """utils/event_utils.py:330-376: float32 image of the event weights; interpolation='bilinear' splats fractional coordinates over the
    four neighbours (the image is one pixel larger with `padding`; events at or beyond the last row / column are masked out: moved to (0,0)
    with weight 0, as the reference does).  Returns a CUDA tensor (the accumulation runs in the HIP kernel)."""
    _lib.require_gpu()
    dev = torch.device(device) if device is not None else (xs.device if getattr(xs, "is_cuda", False) else torch.device("cuda"))
    if dev.type != "cuda":
        raise RuntimeError("events_to_image_torch runs on the GPU only (no CPU fallback)")
    bilinear = interpolation == "bilinear"
    h, w = (sensor_size[0] + 1, sensor_size[1] + 1) if (bilinear and padding) else (sensor_size[0], sensor_size[1])
    xs_d, ys_d = xs.to(dev), ys.to(dev)
    if bilinear and xs.dtype is not torch.long:
        mask = torch.ones(xs_d.shape, device=dev)
        if clip_out_of_range:                                                            # :349-354
            mask = torch.where(xs_d >= w - 1, 0.0, 1.0) * torch.where(ys_d >= h - 1, 0.0, 1.0)
        pxs, pys = xs_d.floor().float(), ys_d.floor().float()
        dxs, dys = (xs_d - pxs).float(), (ys_d - pys).float()
        img = torch.zeros((h, w), dtype=torch.float32, device=dev)
        return interpolate_to_image((pxs * mask).long(), (pys * mask).long(), dxs, dys, ps.to(dev).squeeze() * mask, img)
    return _scatter_f32(_dev(xs_d, torch.int64, dev), _dev(ys_d, torch.int64, dev), _dev(ps, torch.float32, dev), h, w, dev)


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
