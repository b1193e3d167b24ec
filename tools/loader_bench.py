"""Delivered throughput of the drop-in loader at the reference's training shape (BASELINE config 5's "end-to-end dataloader
throughput" seen from train.py: B = 12 samples of 201 x 128 x 128 uint8 -> events [12,40,5,128,128] + frame [12,40,1,128,128],
config/train_v2v_e2vid_10k.yaml:50-76, train.py:52-65,76-82, model/train_utils.py:318-326).

    python tools/loader_bench.py [--batches 200] [--workers 9] [--mode ring|simulating|both] [--cpu-port]

Three things are timed on the SAME pre-generated clips (a pool of uint8 videos served by `PooledFrameSource`, i.e. decode is
NOT part of the figure -- no OpenCV and no video files on this box):
  ring        v2v_amd.loader.RingLoader: workers write clips straight into page-locked shared slots, one H2D copy and one
              launch sequence per batch in the process that owns the GPU (the fast path)
  simulating  SimulatingLoader(DataLoader(WebvidDatasetV2(defer_sim=True)), SimulatingCollator(pad_to=16, normalize=True)):
              the round-2/3 path, default-collated batches through the worker queue
  cpu-port    the reference's deployment: the NumPy port of imgs_to_voxels inside `workers` DataLoader workers
Each reports samples/s, host microseconds per batch by stage, and the GPU-busy fraction (kernel time / wall time)."""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# config/train_v2v_e2vid_10k.yaml:56-76 (the training dataset block); WebVid's 596 x 336 frames
TRAIN_CFG = dict(sequence_length=40, num_bins=5, frames_per_bin=1, crop_size=128, data_source_name="webvid", video_reader="opencv",
                 video_size=(596, 336), proba_pause_when_running=0.0102, proba_pause_when_paused=0.9791, random_flip=True,
                 min_resize_scale=1, max_resize_scale=1, threshold_range=[0.05, 2], max_thres_pos_neg_gap=1.5,
                 base_noise_std_range=[0, 0.1], hot_pixel_std_range=[0, 10], max_samples_per_shot=10)


class PooledFrameSource:
    """`frame_source` serving pre-generated uint8 videos (a module-level class: picklable, and fork()ed workers share the pool
    copy-on-write).  Video v = pool[sample_idx % len(pool)]: a smooth random walk, already at the requested crop size."""

    def __init__(self, n_videos=4, frames=204, h=128, w=128, seed=7):
        g = np.random.default_rng(seed)
        base = g.integers(0, 256, size=(n_videos, 1, h, w), dtype=np.int16)
        steps = g.integers(-6, 7, size=(n_videos, frames, h, w), dtype=np.int16)
        walk = np.cumsum(steps, axis=1, dtype=np.int16)
        walk += base
        self.pool = np.clip(walk, 0, 255, out=walk).astype(np.uint8)[..., None]                          # [V,T,H,W,1]
        self.flipped = np.ascontiguousarray(self.pool[:, :, :, ::-1])                                    # cv2.flip hands out contiguous frames too

    def __call__(self, dataset, sample_idx, start, end, crop_before, min_i, min_j, flip, need_h, need_w):
        v = (self.flipped if flip else self.pool)[int(sample_idx) % len(self.pool)]
        n = end - start
        assert n <= v.shape[0] and need_h == v.shape[1] and need_w == v.shape[2]
        return v[:n]                               # one [T,H,W,1] array: the dataset moves it into the slot with one copy


def make_dataset(tmpdir, n_samples, source, **extra):
    from v2v_amd.datasets import WebvidDatasetV2
    lst = os.path.join(tmpdir, "videos.txt")
    with open(lst, "w") as f:
        for i in range(n_samples):
            f.write(f"vid{i:05d}.mp4 450 0.2 0.2\n")
    cfg = dict(TRAIN_CFG, video_list_file=lst, frame_source=source, **extra)
    return WebvidDatasetV2(tmpdir, cfg)


class _PortDataset(torch.utils.data.Dataset):
    """The reference's deployment on the same clips: host work as WebvidDatasetV2 does it, then the NumPy port of
    EventEmulator.video_to_voxel + the [L,Tb,H,W] sum (oracle/v2v_oracle.py; data/v2v_datasets.py:363-410) in the worker."""

    def __init__(self, base):
        self.base = base

    def __len__(self):
        return len(self.base)

    def __getitem__(self, i):
        from oracle import v2v_oracle as O
        s = self.base[i]
        p = s["sim_params"].numpy()
        counts = O.esim_video_to_voxel(s["sim_frames"].numpy(), *p.tolist(), put_noise_external=False, rng=O.GlobalNumpyRNG, use_lut=False)
        vox = O.bin_sum(counts, 5, 1)
        return {"frame": s["frame"], "events": torch.from_numpy(vox.astype(np.float32)), "data_source_idx": s["data_source_idx"]}


def run_cpu_port(ds, workers, batch, budget_s=60.0):
    """A DataLoader worker builds a WHOLE batch (12 simulations in a row), so the first batches of all workers arrive together:
    timed from the creation of the iterator (worker start-up included, ~0.1 s of fork) until `workers` batches have arrived."""
    from torch.utils.data import DataLoader
    loader = DataLoader(_PortDataset(ds), batch_size=batch, num_workers=workers, drop_last=True, persistent_workers=False, prefetch_factor=1)
    t0 = time.perf_counter()
    it = iter(loader)
    n = 0
    for _ in range(workers):
        next(it)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    del it, loader
    return {"samples_per_s": n * batch / dt, "batches": n, "workers": workers, "seconds": dt,
            "what": "oracle/v2v_oracle.py NumPy port of EventEmulator.video_to_voxel + sum binning inside DataLoader workers (the reference's deployment, "
                    "data/v2v_datasets.py:363-410 under train.py:52-65), same pre-generated clips, default collate; timed from iterator creation "
                    "until every worker has delivered its first batch"}


def _consume(batch, dev):
    """What train.py:79-81 does with a batch: move every tensor to the device."""
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            batch[k] = v.to(dev, non_blocking=True)
    return batch


def run_loader(make_iterable, n_batches, batch, dev, timers=None, gpu_ms_per_batch=None):
    loader = make_iterable()
    it = iter(loader)
    for _ in range(3):                                             # worker start-up, allocator warm-up, first launches
        _consume(next(it), dev)
    torch.cuda.synchronize(dev)
    if timers is not None:
        timers.clear()
    t0 = time.perf_counter()
    t_next = 0.0
    last = None
    for _ in range(n_batches):
        ta = time.perf_counter()
        b = next(it)
        t_next += time.perf_counter() - ta
        last = _consume(b, dev)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    assert last["events"].shape[0] == batch and last["events"].is_cuda
    out = {"samples_per_s": n_batches * batch / dt, "ms_per_batch": dt / n_batches * 1e3, "batches": n_batches,
           "host_us_per_batch_in_next": t_next / n_batches * 1e6}
    if timers:
        out["host_us_per_batch_by_stage"] = {k: v / n_batches * 1e6 for k, v in timers.items()}
    if gpu_ms_per_batch is not None:
        out["gpu_kernel_ms_per_batch"] = gpu_ms_per_batch
        out["gpu_busy_fraction"] = gpu_ms_per_batch / (dt / n_batches * 1e3)
    if getattr(loader, "batches_copied", 0):
        out["h2d_bytes_per_batch"] = loader.bytes_copied / loader.batches_copied
    shape = {k: (tuple(v.shape), str(v.dtype)) for k, v in last.items() if isinstance(v, torch.Tensor)}
    out["batch"] = {k: f"{s} {d}" for k, (s, d) in shape.items()}
    del last, b                    # batches of spawned workers are the PRODUCERS' blocks (CUDA IPC): released before the workers are told to go
    del it, loader
    return out


def run_with_consumer(ds, workers, batch, n_batches, dev):
    """BASELINE config 5's "end-to-end dataloader throughput" at the training shape: RingLoader(normalize='scales') feeding the package's
    E2VIDRecurrent (random init, the reference's module tree, every layer on the device kernels) over the sequence's 40 time steps, the
    head applying normalize_batch_voxel's scales while it reads the raw voxels (model/train_utils.py:318-345, inference: no loss / backward)."""
    from v2v_amd.loader import RingLoader
    from v2v_amd.unet import E2VIDRecurrent
    torch.manual_seed(0)
    net = E2VIDRecurrent(dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
                              num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)).to(dev).eval()
    loader = RingLoader(ds, batch_size=batch, num_workers=workers, drop_last=True, pad_to=16, normalize="scales")
    it = iter(loader)

    first = next(it)

    def consume(b):
        # the whole 40-step sequence as one call: reset_states + time loop, decoder halves on side streams, replayed from the hipGraph the
        # network captured on its first call (v2v_amd/unet.py: forward_sequence(graph=True))
        with torch.no_grad():
            return net.forward_sequence(b["events"], b["event_scales"], graph=True)
    consume(first)
    torch.cuda.synchronize(dev)
    graph = getattr(net, "_sequence_graphs", None)
    for _ in range(2):
        consume(next(it))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n_batches):
        last = consume(next(it))
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    del it
    loader.close()
    return {"samples_per_s": n_batches * batch / dt, "ms_per_batch": dt / n_batches * 1e3, "ms_per_time_step": dt / n_batches * 1e3 / 40, "batches": n_batches, "time_steps_per_sample": 40,
            "frames_reconstructed_per_s": n_batches * batch * 40 / dt, "image": f"{tuple(last.shape)} {last.dtype}", "launch": "hipGraph replay" if graph is not None else "eager",
            "what": "RingLoader(normalize='scales') -> v2v_amd.unet.E2VIDRecurrent.forward_sequence(events, event_scales): 40 time steps per sample (inference), "
                    "replayed from the hipGraph the network captures on its first call, decoder halves on three alternating side streams; the consumer is the bound here, the loader idles"}


def run_yaml_only(tmp, src, n_batches, batch, dev, workers=0, worker_output="cpu", **cfg):
    """The ZERO-EDIT integration level: train.py's own loader -- DataLoader(dataset, batch_size, sampler=RandomSampler,
    num_workers, persistent_workers, pin_memory, drop_last=True), default collate, then `batch[k] = v.to(device)` (train.py:52-65,76-82) --
    over v2v_amd.datasets.WebvidDatasetV2 selected by the YAML's class_name; every sample is simulated inside __getitem__ (one launch per
    sample).  workers = 0: the process that owns the GPU does everything serially (`output_device: cuda`).  workers > 0: the YAML key
    `worker_start_method: spawn` makes train.py's DataLoader start spawned workers, each with its own HIP context; samples return on the
    host (`output_device: cpu`) through the worker queue, pin_memory thread and H2D copy like the reference's."""
    from torch.utils.data import DataLoader, RandomSampler
    spawn = workers > 0
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)          # "start method already fixed to 'fork'": expected here, see below
        ds = make_dataset(tmp, (n_batches + 3) * batch, src, defer_sim=False, output_device=worker_output if spawn else "cuda",
                          **(dict(worker_start_method="spawn") if spawn else {}), **cfg)
    # this process has started fork()ed loaders before (the ring / collator legs), so the YAML key cannot change its default start method any
    # more (the dataset warns and leaves it); the loader is built like train.py's, with the dataset's context passed explicitly
    # the sampler draws exactly the batches run_loader asks for (3 warm-up + n_batches): the epoch is exhausted like train.py's loop exhausts
    # it, so nothing stays prefetched in the workers (as producers' CUDA IPC blocks) when they are shut down
    mk = lambda: DataLoader(ds, batch_size=batch, sampler=RandomSampler(ds, num_samples=(n_batches + 3) * batch), num_workers=workers, persistent_workers=spawn,   # noqa: E731
                            pin_memory=spawn and worker_output == "cpu", drop_last=True, multiprocessing_context=ds.multiprocessing_context if spawn else None)
    t0 = time.perf_counter()
    out = run_loader(mk, n_batches, batch, dev)
    out["seconds_with_startup"] = time.perf_counter() - t0
    out["workers"] = workers
    out["yaml"] = ("class_name: v2v_amd.datasets.WebvidDatasetV2, " + ("worker_start_method: spawn, output_device: %s; num_workers: %d, persistent_workers: true, "
                   "pin_memory: %s" % (worker_output, workers, "true" if worker_output == "cpu" else "false") if spawn else "output_device: cuda; num_workers: 0, persistent_workers: false, pin_memory: false"))
    return out


def gpu_ms_of_batch(batch, dev, pad_to=16):
    """Kernel time of one batch's device work (simulator with the writer's statistics, scales, the scaling pass, the frame tensor) at
    this shape: HIP events around a hipGraph replay of exactly the launches the loader issues, device-resident inputs."""
    from v2v_amd import esim, postops
    clips = esim.synth_clips(batch, 201, 128, 128, dtype=torch.uint8, device=dev)
    params = torch.tensor([[0.3, 0.4, 0.02, 5e-4, 0.5]] * batch, dtype=torch.float64, device=dev)
    keys = torch.stack([torch.arange(batch) + 99, torch.arange(batch)], 1).to(dev)

    from v2v_amd import _lib
    from v2v_amd.loader import clip_frames_f32
    stats = torch.empty((batch, _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device=dev)
    pick = torch.arange(5, 201, 5, dtype=torch.int32, device=dev)

    def step():
        vox = esim.esim_voxel_batch(clips, params, bin_mode="sum", num_bins=5, clip_keys=keys, no_noise=False, pad_to=pad_to, stats=stats)
        sc = postops.scales_from_stats(stats, 40 * 5 * 128 * 128)
        postops.apply_scales(vox, sc, pad_to, valid_hw=(128, 128), inplace=True)
        clip_frames_f32(clips, pick)
    for _ in range(3):
        step()
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(3):
        g.replay()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for s, e in ev:
        s.record()
        g.replay()
        e.record()
    torch.cuda.synchronize(dev)
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    return sum(ms) / len(ms)


def measure(batches=200, workers=9, batch=12, modes=("ring", "simulating"), cpu_port=True, dev=None, cpu_port_budget_s=60.0, simulating_batches=None, ring_kw=None, consumer=True,
            consumer_batches=20, yaml_only_batches=10, yaml_only_spawn_batches=40, yaml_only_host_return_batches=10):
    """simulating_batches = 0 skips the round-2/3 collator leg; yaml_only_* = 0 skips the zero-edit legs (each spawned leg costs ~9 s of worker start-up)."""
    ring_kw = ring_kw or {}
    from torch.utils.data import DataLoader
    from v2v_amd.datasets import SimulatingCollator, SimulatingLoader
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    res = {"shape": f"B={batch}, 201x128x128 uint8 -> events [{batch},40,5,128,128] f32 (x16-padded, normalised) + frame [{batch},40,1,128,128] f32",
           "workers": workers, "host_cores": os.cpu_count(), "source": "pre-generated uint8 clips (PooledFrameSource): video decode is NOT part of these figures"}
    src = PooledFrameSource()
    with tempfile.TemporaryDirectory() as tmp:
        ds = make_dataset(tmp, (batches + 8) * batch, src, defer_sim=True)
        gpu_ms = gpu_ms_of_batch(batch, dev)
        res["gpu_kernel_ms_per_batch"] = gpu_ms
        # the link the clips must cross: page-locked H2D copy rate of one batch's bytes on this box
        nbytes = batch * 201 * 128 * 128
        pin, dst = torch.empty(nbytes, dtype=torch.uint8).pin_memory(), torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dst.copy_(pin, non_blocking=True)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(10):
            dst.copy_(pin, non_blocking=True)
        torch.cuda.synchronize(dev)
        h2d = nbytes * 10 / (time.perf_counter() - t0)
        res["pcie"] = {"h2d_GBps_page_locked": h2d / 1e9, "clip_bytes_per_batch": nbytes, "floor_ms_per_batch": nbytes / h2d * 1e3,
                       "floor_samples_per_s": batch / (nbytes / h2d),
                       "note": "floor_* = the GATHERED uint8 clips (201 frames per sample, what default_collate ships) crossing PCIe once at the measured rate; "
                               "a host-fed loader's GPU-busy fraction is bounded by gpu_kernel_ms_per_batch / (its own bytes / rate)"}
        del pin, dst
        if "simulating" in modes and simulating_batches != 0:
            col = SimulatingCollator.from_configs(TRAIN_CFG, output_device="cuda", pad_to=16, normalize=True)
            col.timers = {}
            mk = lambda: SimulatingLoader(DataLoader(ds, batch_size=batch, num_workers=workers, drop_last=True, persistent_workers=False), col)   # noqa: E731
            res["simulating_loader"] = run_loader(mk, simulating_batches or batches, batch, dev, col.timers, gpu_ms)
        if "ring" in modes:
            from v2v_amd.loader import RingLoader
            timers = {}
            mk = lambda: RingLoader(ds, batch_size=batch, num_workers=workers, drop_last=True, pad_to=16, timers=timers,   # noqa: E731
                                    **dict(dict(normalize=True), **ring_kw))
            res["ring_loader"] = run_loader(mk, batches, batch, dev, timers, gpu_ms)
            r = res["ring_loader"]
            # the ring loader stores every DECODED frame once (the pause schedule repeats frames; the simulator gathers through an index):
            # fewer bytes cross PCIe than the gathered clips hold, so it may beat the gathered-clip floor above
            r["pcie_busy_fraction"] = r["h2d_bytes_per_batch"] / (res["pcie"]["h2d_GBps_page_locked"] * 1e9) / (r["ms_per_batch"] * 1e-3)
            r["vs_gathered_clip_pcie_floor"] = res["pcie"]["floor_ms_per_batch"] / r["ms_per_batch"]
        if consumer:
            try:
                res["ring_loader_feeding_e2vid"] = run_with_consumer(ds, workers, batch, consumer_batches, dev)
            except Exception as exc:  # noqa: BLE001 - a secondary figure
                res["ring_loader_feeding_e2vid"] = {"error": f"{type(exc).__name__}: {exc}"}
        if cpu_port:
            res["cpu_port_in_workers"] = run_cpu_port(ds, workers, batch, budget_s=cpu_port_budget_s)
        for key, nb, wk, wo in (("yaml_only_workers0", yaml_only_batches, 0, "cuda"), ("yaml_only_spawn_workers", yaml_only_spawn_batches, workers, "cuda"),
                                ("yaml_only_spawn_workers_host_return", yaml_only_host_return_batches, workers, "cpu")):
            if nb:
                try:
                    res[key] = run_yaml_only(tmp, src, nb, batch, dev, workers=wk, worker_output=wo)
                except Exception as exc:  # noqa: BLE001 - a secondary figure
                    res[key] = {"error": f"{type(exc).__name__}: {exc}"}
    # one table, three integration levels (samples/s; decode excluded everywhere -- see `source`)
    res["integration_levels_samples_per_s"] = {
        "yaml_only_num_workers_0": res.get("yaml_only_workers0", {}).get("samples_per_s"),
        "yaml_only_spawned_workers": res.get("yaml_only_spawn_workers", {}).get("samples_per_s"),
        "yaml_only_spawned_workers_host_return": res.get("yaml_only_spawn_workers_host_return", {}).get("samples_per_s"),
        "one_line_of_train_py_ring_loader": res.get("ring_loader", {}).get("samples_per_s"),
        "reference_numpy_port_in_workers": res.get("cpu_port_in_workers", {}).get("samples_per_s"),
        "decode_caveat": "video decode (cv2.VideoCapture.read + resize, ~0.4 s per 201-frame sample per worker: ~22 samples/s with 9 workers) is NOT in "
                         "any of these figures; with real decode every level is decode-bound unless decoded clips are cached"}
    best = max((res[k]["samples_per_s"] for k in ("ring_loader", "simulating_loader") if k in res), default=None)
    if best and cpu_port:
        res["speedup_vs_cpu_port"] = best / res["cpu_port_in_workers"]["samples_per_s"]
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=200)
    ap.add_argument("--workers", type=int, default=9)
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--mode", default="both", choices=["ring", "simulating", "both"])
    ap.add_argument("--no-cpu-port", action="store_true")
    ap.add_argument("--no-consumer", action="store_true")
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--normalize", default="True")
    a = ap.parse_args()
    modes = ("ring", "simulating") if a.mode == "both" else (a.mode,)
    kw = dict(depth=a.depth, normalize={"True": True, "False": False}.get(a.normalize, a.normalize))
    print(json.dumps(measure(a.batches, a.workers, a.batch, modes, not a.no_cpu_port, ring_kw=kw, consumer=not a.no_consumer), indent=1))
