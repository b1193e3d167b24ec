"""Cached-voxel writer: scripts/esim_to_voxel.py:17-56 over DynamicH5Dataset (data/dataset.py:176-231, 375-427).

The reference walks a Monash-format sequence index by index (`between_frames`: events from the previous image's `event_idx`
to this one's), builds each grid with events_to_voxel_torch on the host and stores five stacked datasets.  Here every grid of
the sequence is built in ONE segmented launch of the float32 scatter kernel (v2v_events_to_voxel_f32_segmented_hip); the
per-index rules are kept: ts - ts_0 cast to float32 per interval (:194), ps*2-1 (:385), fewer than 3 events -> empty grid
(:189-190), frame / 255 (:342), zero flow when the file has none (:215), timestamp = the image's, dt = ts_k - ts_0 in float64.
Output: the reference's dataset names (frames, flow, events, timestamps, dt + attrs sensor_resolution, source), written with
h5py when it is installed, else as an .npz with the same names.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, monash


def sequence_voxels(path, num_bins=5, temporal_bilinear=False, device="cuda"):
    """All `between_frames` items of DynamicH5Dataset(path, temporal_bilinear=...) as stacked arrays (len = num_imgs - 1)."""
    _lib.require_gpu()
    with monash.open_sequence(path) as f:
        if f.has_flow():
            raise NotImplementedError("sequences with optic flow (data/dataset.py:209-213) are outside the accelerated path")
        keys = list(f.image_keys)                            # insertion order of the file (dataset.py:402-403, 422-425)
        n_img = int(f.attr("num_imgs", len(keys)))
        length = n_img - 1                                   # :291
        ends = [int(f.image_attr(k, "event_idx")) for k in keys[:length]]                # compute_frame_indices (:419-427)
        seg = np.asarray([0] + ends, dtype=np.int64)
        n_ev = int(seg[-1])
        ts = np.asarray(f.events("ts", 0, n_ev), dtype=np.float64)
        xs = np.asarray(f.events("xs", 0, n_ev)).astype(np.int64)
        ys = np.asarray(f.events("ys", 0, n_ev)).astype(np.int64)
        ps = (np.asarray(f.events("ps", 0, n_ev)) * 2.0 - 1.0).astype(np.float32)       # :385, :196
        res = f.attr("sensor_resolution")
        h, w = (int(res[0]), int(res[1])) if res is not None else f.image(keys[0]).shape[:2]
        frames = np.stack([np.asarray(f.image("image{:09d}".format(i)), dtype=np.float32)[None] / 255 for i in range(length)])   # :375-376, :342
        timestamps = np.asarray([f.image_attr(k, "timestamp") for k in keys], dtype=np.float64)[:length]                      # :217
        source = f.attr("source", "unknown")
    if np.any(np.diff(seg) < 0):
        raise ValueError("event_idx attributes must be non-decreasing")
    dev = torch.device(device)
    out = torch.empty((length, num_bins, h, w), dtype=torch.float32, device=dev)
    dropped = torch.empty((1,), dtype=torch.int64, device=dev)
    t_d, x_d, y_d, p_d, s_d = (torch.as_tensor(a, device=dev) for a in (ts, xs, ys, ps, seg))
    with torch.cuda.device(dev):
        rc = _lib.lib().v2v_events_to_voxel_f32_segmented_hip(
            C.c_void_p(t_d.data_ptr()) if n_ev else None, C.c_void_p(x_d.data_ptr()) if n_ev else None,
            C.c_void_p(y_d.data_ptr()) if n_ev else None, C.c_void_p(p_d.data_ptr()) if n_ev else None, n_ev,
            C.c_void_p(s_d.data_ptr()), length, 3, 0 if temporal_bilinear else 1, num_bins, h, w,
            C.c_void_p(out.data_ptr()), C.c_void_p(dropped.data_ptr()), _lib.stream_ptr())
    _lib.check(rc)
    if n_ev and int(dropped.item()) != 0:
        raise IndexError(f"{int(dropped.item())} event(s) outside the sensor / bin range")
    lo, hi = seg[:-1], seg[1:]
    nonempty = hi > lo
    dt = np.where(nonempty, ts[np.maximum(hi - 1, 0)] - ts[np.minimum(lo, max(n_ev - 1, 0))], 0.0) if n_ev else np.zeros(length)   # :184-188,202-204
    return {"frames": frames.astype(np.float32), "flow": np.zeros((length, 2, h, w), dtype=np.float32), "events": out.cpu().numpy(),
            "timestamps": timestamps, "dt": np.asarray(dt, dtype=np.float64), "sensor_resolution": np.asarray([h, w]), "source": source}


def convert(in_path, out_path, temporal_bilinear, num_bins=5, device="cuda"):
    """scripts/esim_to_voxel.py:17-56 for one file: float32 datasets frames / flow / events / timestamps / dt."""
    d = sequence_voxels(in_path, num_bins, temporal_bilinear, device)
    data = {k: np.asarray(d[k], dtype=np.float32) for k in ("frames", "flow", "events", "timestamps", "dt")}       # dtype=np.float32, :46-50
    if str(out_path).endswith(".npz"):
        np.savez_compressed(out_path, **data, **{"attrs/sensor_resolution": d["sensor_resolution"], "attrs/source": np.array("esim")})
    else:
        import h5py
        with h5py.File(out_path, "w") as f:
            f.attrs["sensor_resolution"] = d["sensor_resolution"]
            f.attrs["source"] = "esim"
            for k, v in data.items():
                f.create_dataset(k, data=v, dtype=np.float32)
    return data


def testh5_to_cache(in_path, out_path, configs):
    """Write the voxel cache TestH5CacheDataset reads (data/testh5.py:383-446: datasets `frames` [n,H,W] float32 and `events`
    [n,Tb,H,W] float32, attributes num_bins / interpolate_bins) from a Monash-format sequence: every item of
    v2v_amd.testh5.TestH5Dataset(in_path, configs) -- one segmented launch of the scatter kernel for the whole sequence.  The reference's
    docstring names scripts/testh5_to_voxel_cache.py for this; the script is not in the repository, the reader defines the format."""
    from .testh5 import TestH5Dataset
    cfg = dict(configs, sequence_length=1 << 30, warm_up_length=0, max_samples=None, output_additional_frame=False, output_additional_evs=False)
    ds = TestH5Dataset(in_path, cfg)
    s = ds[0]
    frames = s["frame"].numpy()[:, 0].astype(np.float32)
    events = s["events"].numpy().astype(np.float32)
    if str(out_path).endswith(".npz"):
        np.savez_compressed(out_path, frames=frames, events=events, **{"attrs/num_bins": np.array(ds.num_bins), "attrs/interpolate_bins": np.array(bool(ds.interpolate_bins))})
    else:
        import h5py
        with h5py.File(out_path, "w") as f:
            f.attrs["num_bins"] = ds.num_bins
            f.attrs["interpolate_bins"] = bool(ds.interpolate_bins)
            f.create_dataset("frames", data=frames)
            f.create_dataset("events", data=events)
    return {"frames": frames, "events": events}
