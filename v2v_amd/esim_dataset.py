"""`ESIMH5Dataset` -- drop-in for data/esim_dataset.py:49-152, the TRAINING loader over cached voxels (the files
v2v_amd.voxel_cache.convert / scripts/esim_to_voxel.py write: datasets frames [n,1,H,W], flow [n,2,H,W], events [n,Tb,H,W], attribute
sensor_resolution), with the reference's augmentation: random crop, horizontal flip, a pause schedule (paused steps repeat the last
frame and carry no events), noise on every step's voxels, hot pixels over the whole sample.

Host-side only (the voxels are already cached; nothing here is on the simulator's hot path).  The draws come from Python's `random` and
from `np.random` in the reference's order, so after the same two seeds a sample is the reference's bit for bit (golden G21).  Files are
opened through v2v_amd.monash (.h5 with h5py, or the .npz form of the same datasets) per item, so the dataset pickles into workers.
"""
from __future__ import annotations

import random

import numpy as np
import torch

from . import monash
from .datasets import data_sources


def _poisson_lambda(std):
    # N = k * s with k ~ Poisson(lam), s = +-1: var N = lam^2 + lam, set equal to std^2 (data/esim_dataset.py:16-20)
    return (-1 + np.sqrt(1 + 4 * std ** 2)) / 2


def _signed_counts(lam, size):
    k = np.random.poisson(lam=lam, size=size)
    return k * (2 * np.random.randint(0, 2, size=size) - 1)


def add_noise_to_voxel(voxel, noise_std=1.0, noise_fraction=0.1, integer_noise=False):
    """data/esim_dataset.py:35-47: zero-mean noise of standard deviation `noise_std` on a `noise_fraction` of the cells (signed Poisson
    counts with integer_noise).  Draw order: the noise field, then (fraction < 1) the uniform field of the mask."""
    noise = _signed_counts(_poisson_lambda(noise_std), voxel.shape) if integer_noise else noise_std * np.random.randn(*voxel.shape)
    if noise_fraction < 1.0:
        keep = np.random.rand(*voxel.shape) < noise_fraction
        noise = np.where(keep, noise, 0)
    return voxel + noise


def add_hot_pixels_to_voxels(voxels, hot_pixel_std=1.0, max_hot_pixel_fraction=0.001, integer_noise=False):
    """data/esim_dataset.py:7-32: one offset per hot pixel added to EVERY step and bin of the [T,C,H,W] sample, in place.  Draw order:
    random.uniform (the fraction), x then y positions, the values (repeated positions accumulate)."""
    h, w = voxels.shape[-2:]
    count = int(random.uniform(0, max_hot_pixel_fraction) * h * w)
    cols = np.random.randint(0, w, count)
    rows = np.random.randint(0, h, count)
    if integer_noise:
        # the reference stores the Poisson counts in the variable that held the row positions (data/esim_dataset.py:21 `y = ...poisson`), so
        # with integer noise the offsets land in row k = the count itself, not in the drawn row: reproduced, because the golden is the
        # reference's behaviour (an IndexError there for a count >= H is an IndexError here)
        k = np.random.poisson(lam=_poisson_lambda(hot_pixel_std), size=count)
        vals = k * (2 * np.random.randint(0, 2, size=count) - 1)
        rows = k
    else:
        vals = np.random.randn(count) * hot_pixel_std
    offset = np.zeros((h, w))
    np.add.at(offset, (rows, cols), vals)
    voxels += offset[None, None]
    return voxels


class ESIMH5Dataset(torch.utils.data.Dataset):
    def __init__(self, h5_path, configs):
        self.h5_path = h5_path
        get = configs.get
        self.sequence_length = get("sequence_length", 40)
        self.step_size = get("step_size", self.sequence_length)
        self.proba_pause_when_running = get("proba_pause_when_running", 0.05)
        self.proba_pause_when_paused = get("proba_pause_when_paused", 0.9)
        self.noise_std = get("noise_std", 0.1)
        self.noise_fraction = get("noise_fraction", 1.0)
        self.hot_pixel_std = get("hot_pixel_std", 0.1)
        self.max_hot_pixel_fraction = get("max_hot_pixel_fraction", 0.001)
        self.random_crop_size = get("random_crop_size", 112)
        self.random_flip = get("random_flip", True)
        self.integer_noise = get("integer_noise", False)
        with monash.open_sequence(h5_path) as f:
            self.sensor_resolution = np.asarray(f.attr("sensor_resolution"))[0:2]
            self.num_frames = f.dataset_len("frames")
        self.data_source_name = "esim"
        self.data_source_idx = data_sources.index(self.data_source_name)
        self.samples = [(i, i + self.sequence_length) for i in range(0, self.num_frames - self.sequence_length, self.step_size)]

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, index):
        lo, hi = self.samples[index]
        with monash.open_sequence(self.h5_path) as f:
            src = {k: np.asarray(f.dataset(k, lo, hi)) for k in ("frames", "flow", "events")}
        h, w = src["frames"].shape[-2:]
        if self.random_crop_size is not None:                 # :96-103: top row, then left column
            size = self.random_crop_size
            top = random.randint(0, h - size)
            left = random.randint(0, w - size)
            src = {k: v[:, :, top:top + size, left:left + size] for k, v in src.items()}
        if self.random_flip and random.random() > 0.5:        # :106-109
            src = {k: v[..., ::-1] for k, v in src.items()}
        out = {k: np.zeros_like(v) for k, v in src.items()}
        paused, taken = False, 0
        for t in range(self.sequence_length):                 # :117-141: one uniform per step decides the pause, then the step's noise
            p = self.proba_pause_when_paused if paused else self.proba_pause_when_running
            paused = np.random.rand() < p
            if paused and t > 0:
                out["frames"][t] = out["frames"][t - 1]       # the last frame again; flow and events of the step stay zero
            else:
                for k in out:
                    out[k][t] = src[k][taken]
                taken += 1
            out["events"][t] = add_noise_to_voxel(out["events"][t], self.noise_std, self.noise_fraction, integer_noise=self.integer_noise)
        add_hot_pixels_to_voxels(out["events"], self.hot_pixel_std, self.max_hot_pixel_fraction, integer_noise=self.integer_noise)
        return {"frame": torch.Tensor(out["frames"]), "flow": torch.Tensor(out["flow"]), "events": torch.Tensor(out["events"]),
                "data_source_idx": torch.tensor(self.data_source_idx)}
