"""Run ON THE GPU BOX: soak of the RingLoader's slot ring -- many epochs with many workers, shuffled order, epochs abandoned at random points,
persistent and non-persistent workers -- where EVERY sample of EVERY batch is checked against the per-sample path: `fixed_seed` makes a
sample's np.random draws a function of its index, so the main process recomputes frame + simulator input on the host (`defer_sim`
`__getitem__` + the simulator per sample, once) and compares `frame` and `events` of every batch with what came out of the ring, exactly.  A worker writing into a slot that is still being copied, a slot handed out twice or a stale index row
shows up as a mismatch.  usage: python tools/loader_soak.py [seconds]
`python tools/loader_soak.py [seconds] staged`: the zero-edit path instead -- train.py's own DataLoader(num_workers=0, shuffle) over the dataset
whose __getitem__ goes through three rotating page-locked slots (datasets.py `_getitem_staged`), every batch against the plain per-sample
path (`staged_getitem: false`) of the same indices; a slot overwritten while its copy or its kernels are still reading it is a mismatch."""
import os
import sys
import tempfile
import time

import numpy as np
import torch
from torch.utils.data import default_collate

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.loader_bench import PooledFrameSource  # noqa: E402
from v2v_amd.datasets import SimulatingCollator, WebvidDatasetV2  # noqa: E402
from v2v_amd.loader import RingLoader  # noqa: E402


class _Order(torch.utils.data.Sampler):
    """A shuffled order the main process knows in advance."""

    def __init__(self, n):
        self.n, self.order = n, list(range(n))

    def reshuffle(self, seed):
        self.order = np.random.default_rng(seed).permutation(self.n).tolist()

    def __len__(self):
        return self.n

    def __iter__(self):
        return iter(self.order)


def staged_soak(budget):
    tmp = tempfile.mkdtemp()
    lst = os.path.join(tmp, "videos.txt")
    n_samples = 120
    with open(lst, "w") as f:
        for i in range(n_samples):
            f.write(f"vid{i:05d}.mp4 450 0.2 0.2\n")
    src = PooledFrameSource(n_videos=5, frames=70, h=48, w=48, seed=5)
    cfg = dict(video_list_file=lst, sequence_length=8, crop_size=48, data_source_name="webvid", frame_source=src, video_size=(640, 360),
               video_reader="opencv", min_resize_scale=1, max_resize_scale=1, proba_pause_when_running=0.05, proba_pause_when_paused=0.9,
               fixed_seed=23, sim_rng="philox", output_device="cuda")
    plain = WebvidDatasetV2(tmp, dict(cfg, staged_getitem=False))
    fast = WebvidDatasetV2(tmp, dict(cfg))
    assert fast._staged_ok() and not plain._staged_ok()
    exp = [plain[i] for i in range(n_samples)]
    exp_frame, exp_events = torch.stack([e["frame"] for e in exp]), torch.stack([e["events"] for e in exp])
    order = _Order(n_samples)
    loader = torch.utils.data.DataLoader(fast, batch_size=12, sampler=order, num_workers=0, drop_last=True)
    side = torch.cuda.Stream()
    t_end, epochs, checked = time.time() + budget, 0, 0
    while time.time() < t_end:
        order.reshuffle(1000 + epochs)
        epochs += 1
        # every other epoch under a side stream as the current one, and with a slow kernel queued in front: the host runs ahead of the GPU
        with torch.cuda.stream(side if epochs % 2 else torch.cuda.current_stream()):
            if epochs % 3 == 0:
                torch.cuda._sleep(20_000_000)
            for bi, batch in enumerate(loader):
                idx = order.order[12 * bi:12 * bi + 12]
                if not torch.equal(batch["frame"], exp_frame[idx]) or not torch.equal(batch["events"], exp_events[idx]):
                    raise SystemExit(f"MISMATCH (staged __getitem__): epoch {epochs} batch {bi} samples {idx}")
                checked += len(idx)
        torch.cuda.synchronize()
    print(f"staged __getitem__ soak ok: {epochs} epochs, {checked} samples checked against the plain per-sample path")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    if len(sys.argv) > 2 and sys.argv[2] == "staged":
        return staged_soak(budget)
    tmp = tempfile.mkdtemp()
    lst = os.path.join(tmp, "videos.txt")
    n_samples = 96
    with open(lst, "w") as f:
        for i in range(n_samples):
            f.write(f"vid{i:05d}.mp4 450 0.2 0.2\n")
    src = PooledFrameSource(n_videos=5, frames=70, h=48, w=48, seed=3)
    cfg = dict(video_list_file=lst, sequence_length=8, crop_size=48, data_source_name="webvid", frame_source=src, video_size=(640, 360),
               video_reader="opencv", min_resize_scale=1, max_resize_scale=1, proba_pause_when_running=0.05, proba_pause_when_paused=0.9,
               fixed_seed=11, defer_sim=True)
    ds = WebvidDatasetV2(tmp, cfg)
    col = SimulatingCollator.from_configs({"num_bins": 5}, output_device="cuda")
    order = _Order(len(ds))
    # every sample once through the per-sample path (its draws are a function of its index; the simulator's noise is keyed by the sample):
    # the soak then checks whole batches with two device compares and runs at the loader's own pace
    exp_frame, exp_events = [], []
    for i in range(len(ds)):
        w = ds[i]
        exp_frame.append(w["frame"])
        exp_events.append(col.simulate(default_collate([w]))["events"][0])
    exp_frame, exp_events = torch.stack(exp_frame).cuda(), torch.stack(exp_events)
    t_end = time.time() + budget
    checked = batches = abandoned = epochs = 0
    rng = np.random.default_rng(0)
    for persistent in (False, True):
        loader = RingLoader(ds, batch_size=4, sampler=order, num_workers=6, prefetch_factor=2, persistent_workers=persistent, drop_last=False)
        while time.time() < t_end - (budget / 2 if not persistent else 0):
            order.reshuffle(epochs)
            epochs += 1
            stop_at = int(rng.integers(1, len(loader) + 1)) if rng.random() < 0.4 else None
            for bi, batch in enumerate(loader):
                idx = order.order[4 * bi:4 * bi + 4]
                assert batch["frame"].shape[0] == len(idx)
                if not torch.equal(batch["frame"], exp_frame[idx]) or not torch.equal(batch["events"], exp_events[idx]):
                    raise SystemExit(f"MISMATCH: epoch {epochs} batch {bi} samples {idx} persistent={persistent}")
                checked += len(idx)
                batches += 1
                if stop_at is not None and bi + 1 == stop_at:
                    abandoned += 1
                    break
        loader.close()
    print(f"loader soak ok: {epochs} epochs ({abandoned} abandoned), {batches} batches, {checked} samples checked against the per-sample path")


if __name__ == "__main__":
    main()
