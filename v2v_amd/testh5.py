"""`TestH5Dataset`, `TestH5FlowDataset`, `TestH5EventDataset`, `TestH5CacheDataset`, `FPS_H5Dataset` -- drop-ins for data/testh5.py:14-173, :175-303, :305-381, :383-446, :448-520 (the real-data validation loaders around `make_voxel`).

Same constructor `(h5_path, configs)`, config keys and defaults (:17-58), sample table (:43-52), `make_voxel(evs)` (:60-90)
and `__getitem__` dict (:96-173: frame [L(+1),1,H,W] float32, events [L(+1),Tb,H,W] float32, data_source_idx, sequence_name,
real_begin_idx, frame_idx).  The reference voxelises one image interval at a time with np.add.at on the host; here ALL
intervals of a sample go through ONE segmented launch of the HIP scatter kernel (v2v_events_to_voxel_segmented_hip):
the events of [event_idx[begin], event_idx[end]) are uploaded once and the per-image `event_idx` attrs are the segment
offsets.  File access: v2v_amd/monash.py (.h5 through h5py, or the .npz form of the same layout).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import monash, voxel
from .datasets import data_sources


class TestH5Dataset(torch.utils.data.Dataset, voxel.MakeVoxelMixin):
    __test__ = False                                          # not a pytest class

    def __init__(self, h5_path, configs):
        self.h5_path = h5_path
        self.sequence_name = os.path.basename(h5_path).split(".")[0]                     # :20
        self.configs = configs
        self.dataset_name = configs.get("dataset_name", "hqf")
        self.sequence_length = configs.get("sequence_length", 40)
        self.warm_up_length = configs.get("warm_up_length", 0)
        self.max_samples = configs.get("max_samples", None)
        self.num_bins = configs.get("num_bins", 5)
        self.interpolate_bins = configs.get("interpolate_bins", False)
        self.image_range = configs.get("image_range", 255)
        assert self.image_range in [255, 1], "image_range should be 255 or 1."
        self.device = configs.get("sim_device", "cuda")                                  # this implementation only
        with monash.open_sequence(h5_path) as f:
            self.img_keys = sorted(f.image_keys)
            self.total_frame_cnt = len(self.img_keys)
            img_shape = f.image(self.img_keys[0]).shape
            self.H, self.W = img_shape[0], img_shape[1]
            self.samples = []                                                            # (begin, real_begin, end), :43-52
            for i in range(0, self.total_frame_cnt - 1, self.sequence_length - self.warm_up_length):
                begin = max(0, i - self.warm_up_length)
                end_idx = min(self.total_frame_cnt - 1, begin + self.sequence_length)
                self.samples.append((begin, i - begin, end_idx))
        if self.max_samples is not None:
            self.samples = self.samples[:self.max_samples]
        self.output_additional_frame = configs.get("output_additional_frame", False)
        self.output_additional_evs = configs.get("output_additional_evs", False)

    def __len__(self):
        return len(self.samples)

    def get_img(self, f, idx):
        return f.image(self.img_keys[idx])

    def __getitem__(self, idx):
        begin, real_begin, end = self.samples[idx]
        with monash.open_sequence(self.h5_path) as f:
            frames = [torch.tensor(self.get_img(f, i + 1), dtype=torch.float32).unsqueeze(0) for i in range(begin, end)]   # :105-106
            ev_idx = [int(f.image_attr(self.img_keys[i], "event_idx")) for i in range(begin, end + 1)]                    # :108-109
            if self.output_additional_evs:                                               # :133-143: the interval before `begin`
                pre_idx = max(0, begin - 1)
                ev_idx = [int(f.image_attr(self.img_keys[pre_idx], "event_idx"))] + ev_idx
            lo, hi = ev_idx[0], ev_idx[-1]
            evs = [f.events(k, lo, hi) for k in ("ts", "xs", "ys", "ps")]
            first_frame = self.get_img(f, begin) if self.output_additional_frame else None
        seg = np.asarray(ev_idx, dtype=np.int64) - lo
        if np.any(np.diff(seg) < 0):                          # pre_idx == begin (begin == 0): an empty leading interval
            seg = np.maximum.accumulate(seg)
        # ps of hqf h5 files are in {0,1} (:69); one segmented launch = one make_voxel per image interval (:111-119)
        grids = voxel.make_voxels_segmented([np.asarray(evs[0], dtype=np.float64), evs[1], evs[2], evs[3]], seg, self.H, self.W, self.num_bins,
                                            self.interpolate_bins, device=self.device)
        all_events = torch.as_tensor(grids).to(torch.float32).cpu()                      # torch.tensor(voxel, dtype=float32), :120
        all_frames = torch.stack(frames, dim=0)
        if self.output_additional_frame:                                                 # :129-131,149-150
            ff = torch.tensor(first_frame, dtype=torch.float32).unsqueeze(0).unsqueeze(0)
            all_frames = torch.cat([ff, all_frames], dim=0)
        if self.image_range == 1:
            all_frames = all_frames / 255.0
        n = end - begin
        return {
            "frame": all_frames,
            "events": all_events,
            "data_source_idx": torch.tensor(data_sources.index(self.dataset_name.lower()), dtype=torch.int64),
            "sequence_name": [self.sequence_name] * n,
            "real_begin_idx": torch.tensor([real_begin] * n, dtype=torch.int64),
            "frame_idx": torch.tensor(list(range(begin, end)), dtype=torch.int64),
        }


class TestH5EventDataset(TestH5Dataset):
    """data/testh5.py:305-381: the same samples, but `events` is the list of RAW event rows per image interval -- float64 [n,5] =
    [x, y, t, p in {-1,+1}, 0] (one [1,5] row of zeros for an empty interval) -- instead of voxel grids.  No voxelisation, so no device
    work: host IO only.  (`output_additional_evs` is not read by the reference's method either.)"""
    __test__ = False

    def __getitem__(self, idx):
        begin, real_begin, end = self.samples[idx]
        with monash.open_sequence(self.h5_path) as f:
            frames = [torch.tensor(self.get_img(f, i + 1), dtype=torch.float32).unsqueeze(0) for i in range(begin, end)]
            ev_idx = [int(f.image_attr(self.img_keys[i], "event_idx")) for i in range(begin, end + 1)]
            lo, hi = ev_idx[0], max(ev_idx[-1], ev_idx[0])
            ts, xs, ys, ps = (np.asarray(f.events(k, lo, hi)).astype(np.float64) for k in ("ts", "xs", "ys", "ps"))   # float64: :329-333
            first_frame = self.get_img(f, begin) if self.output_additional_frame else None
        ps = ps * 2 - 1                                                                  # :334
        rows = np.stack([xs, ys, ts, ps, np.zeros_like(ps)], axis=1)
        all_events = []
        for a, b in zip(ev_idx[:-1], ev_idx[1:]):
            ev = torch.tensor(rows[a - lo:max(b, a) - lo].copy(), dtype=torch.float64)
            all_events.append(ev if ev.shape[0] else torch.zeros((1, 5), dtype=torch.float64))   # :342-343
        all_frames = torch.stack(frames, dim=0)
        if self.output_additional_frame:
            ff = torch.tensor(first_frame, dtype=torch.float32).unsqueeze(0).unsqueeze(0)
            all_frames = torch.cat([ff, all_frames], dim=0)
        if self.image_range == 1:
            all_frames = all_frames / 255.0
        n = end - begin
        return {
            "frame": all_frames,
            "events": all_events,
            "data_source_idx": torch.tensor(data_sources.index(self.dataset_name.lower()), dtype=torch.int64),
            "sequence_name": [self.sequence_name] * n,
            "real_begin_idx": torch.tensor([real_begin] * n, dtype=torch.int64),
            "frame_idx": torch.tensor(list(range(begin, end)), dtype=torch.int64),
        }


class FPS_H5Dataset(TestH5Dataset):
    """data/testh5.py:448-520: a frame-less event stream cut at a fixed rate -- `FPS` cuts per second between the first and the last
    timestamp (np.linspace borders, np.searchsorted into events/ts), `sequence_length` consecutive cuts per sample, one voxel grid per cut
    on an H x W sensor given by the configuration.  Returns {events [L,Tb,H,W] float32, data_source_idx, sequence_name}.  All grids of a
    sample go through ONE segmented launch of the scatter kernel."""
    __test__ = False

    def __init__(self, h5_path, configs):  # noqa: D107 - the reference does not call its parent's constructor either
        self.h5_path = h5_path
        self.sequence_name = os.path.basename(h5_path).split(".")[0]
        self.configs = configs
        self.dataset_name = configs.get("dataset_name", "hqf")
        self.sequence_length = configs.get("sequence_length", 40)
        self.warm_up_length = configs.get("warm_up_length", 0)
        self.num_bins = configs.get("num_bins", 5)
        self.interpolate_bins = configs.get("interpolate_bins", False)
        self.FPS = configs.get("FPS", 100)
        self.H = configs.get("H", 260)
        self.W = configs.get("W", 346)
        self.device = configs.get("sim_device", "cuda")                                  # this implementation only
        with monash.open_sequence(h5_path) as f:
            ts = np.asarray(f.events("ts"))
            min_t, max_t = ts[0], ts[-1]
            self.total_frame_cnt = int((max_t - min_t) * self.FPS)                       # :470
            border_timestamps = np.linspace(min_t, max_t, self.total_frame_cnt + 1)
            self.event_idx = np.searchsorted(ts, border_timestamps)                      # :473
        self.samples = []                                                                # (begin, end), :475-480
        for i in range(0, self.total_frame_cnt - 1, self.sequence_length):
            self.samples.append((i, min(self.total_frame_cnt - 1, i + self.sequence_length)))

    def __getitem__(self, idx):
        begin, end = self.samples[idx]
        seg = np.asarray(self.event_idx[begin:end + 1], dtype=np.int64)
        lo, hi = int(seg[0]), int(seg[-1])
        with monash.open_sequence(self.h5_path) as f:
            evs = [f.events(k, lo, hi) for k in ("ts", "xs", "ys", "ps")]
        grids = voxel.make_voxels_segmented([np.asarray(evs[0], dtype=np.float64), evs[1], evs[2], evs[3]], seg - lo, self.H, self.W, self.num_bins,
                                            self.interpolate_bins, device=self.device)
        return {
            "events": torch.as_tensor(grids).to(torch.float32).cpu(),                   # torch.tensor(voxel, dtype=float32), :503
            "data_source_idx": torch.tensor(data_sources.index(self.dataset_name.lower()), dtype=torch.int64),
            "sequence_name": [self.sequence_name] * (end - begin),
        }


class TestH5FlowDataset(TestH5Dataset):
    """data/testh5.py:175-303: MVSEC-style sequences.  A sample item per optic-flow map k+1: the events between map k's and map k+1's
    `event_idx`, the image named by map k+1's `image_idx` (clamped to the last image, :236), the map itself.  Returns {frame, events,
    flow [L,2,H,W], data_source_idx, sequence_name, frame_idx}.  All voxel grids of a sample: ONE segmented launch."""
    __test__ = False

    def __init__(self, h5_path, configs):  # noqa: D107 - own constructor, as in the reference
        self.h5_path = h5_path
        self.sequence_name = os.path.basename(h5_path).split(".")[0]
        self.configs = configs
        self.dataset_name = configs.get("dataset_name", "mvsec")
        self.sequence_length = configs.get("sequence_length", 40)
        self.max_samples = configs.get("max_samples", None)
        self.num_bins = configs.get("num_bins", 5)
        self.interpolate_bins = configs.get("interpolate_bins", False)
        self.image_range = configs.get("image_range", 255)
        assert self.image_range in [255, 1], "image_range should be 255 or 1."
        self.device = configs.get("sim_device", "cuda")                                  # this implementation only
        with monash.open_sequence(h5_path) as f:
            self.img_keys = sorted(f.image_keys)
            self.flow_keys = sorted(f.flow_keys)
            self.total_frame_cnt = len(self.flow_keys)
            img_shape = f.image(self.img_keys[0]).shape
            self.H, self.W = img_shape[0], img_shape[1]
        self.samples = []                                                                # (begin, end), :202-207
        for i in range(0, self.total_frame_cnt - 1, self.sequence_length):
            self.samples.append((i, min(self.total_frame_cnt - 1, i + self.sequence_length)))
        if self.max_samples is not None:
            self.samples = self.samples[:self.max_samples]
        self.output_additional_frame = configs.get("output_additional_frame", False)
        self.output_additional_evs = configs.get("output_additional_evs", False)

    def __getitem__(self, idx):
        begin, end = self.samples[idx]
        last_img = len(self.img_keys) - 1
        with monash.open_sequence(self.h5_path) as f:
            ev_idx = [int(f.flow_attr(self.flow_keys[k], "event_idx")) for k in range(begin, end + 1)]
            img_idx = [min(int(f.flow_attr(self.flow_keys[k + 1], "image_idx")), last_img) for k in range(begin, end)]          # :235-236
            frames = [torch.tensor(self.get_img(f, i), dtype=torch.float32).unsqueeze(0) for i in img_idx]
            flows = [torch.tensor(np.asarray(f.flow(self.flow_keys[k + 1]))) for k in range(begin, end)]
            if self.output_additional_evs:                                               # :262-272: the interval in front of map `begin`
                ev_idx = [int(f.flow_attr(self.flow_keys[max(0, begin - 1)], "event_idx"))] + ev_idx
            lo, hi = ev_idx[0], max(ev_idx)
            evs = [f.events(k, lo, hi) for k in ("ts", "xs", "ys", "ps")]
            first_frame = self.get_img(f, int(f.flow_attr(self.flow_keys[begin], "image_idx"))) if self.output_additional_frame else None   # :257-260 (not clamped there)
        seg = np.maximum.accumulate(np.asarray(ev_idx, dtype=np.int64) - lo)
        grids = voxel.make_voxels_segmented([np.asarray(evs[0], dtype=np.float64), evs[1], evs[2], evs[3]], seg, self.H, self.W, self.num_bins,
                                            self.interpolate_bins, device=self.device)
        all_frames = torch.stack(frames, dim=0)
        if self.output_additional_frame:
            all_frames = torch.cat([torch.tensor(first_frame, dtype=torch.float32).unsqueeze(0).unsqueeze(0), all_frames], dim=0)
        if self.image_range == 1:
            all_frames = all_frames / 255.0
        return {
            "frame": all_frames,
            "events": torch.as_tensor(grids).to(torch.float32).cpu(),
            "flow": torch.stack(flows, dim=0),
            "data_source_idx": torch.tensor(data_sources.index(self.dataset_name.lower()), dtype=torch.int64),
            "sequence_name": [self.sequence_name] * (end - begin),
            "frame_idx": torch.tensor(img_idx, dtype=torch.int64),
        }


class TestH5CacheDataset(torch.utils.data.Dataset):
    """data/testh5.py:383-446: pre-built voxel caches -- datasets `frames` [n,H,W(,..)] and `events` [n,Tb,H,W] with the attributes
    num_bins / interpolate_bins (checked against the configuration, :401-402), cut into samples of `sequence_length`.  Host IO only;
    v2v_amd.voxel_cache.testh5_to_cache writes the format from a TestH5Dataset (the reference names a scripts/testh5_to_voxel_cache.py it
    does not ship)."""
    __test__ = False

    def __init__(self, h5_path, configs):
        self.h5_path = h5_path
        self.sequence_name = os.path.basename(h5_path).split(".")[0]
        self.configs = configs
        self.dataset_name = configs.get("dataset_name", "hqf")
        self.sequence_length = configs.get("sequence_length", 40)
        self.num_bins = configs.get("num_bins", 5)
        self.interpolate_bins = configs.get("interpolate_bins", False)
        with monash.open_sequence(h5_path) as f:
            assert self.num_bins == int(f.attr("num_bins"))
            assert bool(self.interpolate_bins) == bool(f.attr("interpolate_bins"))
            self.total_frame_cnt = f.dataset_len("frames")
            img_shape = f.dataset("frames", 0, 1).shape[1:]
            self.H, self.W = img_shape[0], img_shape[1]
        self.samples = [(i, min(self.total_frame_cnt, i + self.sequence_length)) for i in range(0, self.total_frame_cnt, self.sequence_length)]

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        begin, end = self.samples[idx]
        with monash.open_sequence(self.h5_path) as f:
            all_frames = torch.tensor(np.asarray(f.dataset("frames", begin, end)))
            all_events = torch.tensor(np.asarray(f.dataset("events", begin, end)))
        src = data_sources.index(self.dataset_name.lower())
        return {"frame": all_frames, "events": all_events, "data_source_idx": torch.tensor([src] * (end - begin), dtype=torch.int64),
                "sequence_name": [self.sequence_name] * (end - begin)}
