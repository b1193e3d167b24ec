"""RingLoader -- the batch loader of the drop-in path: what `train.py:52-65` builds (DataLoader over the dataset, default collate,
`.to(device)` per tensor, train.py:76-82) re-laid for one-process-per-GPU on MI355X.

    loader = RingLoader(dataset, batch_size=12, sampler=sampler, num_workers=9, drop_last=True, pad_to=16, normalize=True)
    for batch in loader: ...   # the reference's batch dict: frame [B,L,C,H,W] f32, events [B,L,Tb,Hp,Wp] f32 (both on the GPU),
                               # data_source_idx [B] int64, v2e_params {5 x [B] float64}

Why not DataLoader + default_collate + SimulatingCollator (the round-2/3 path, kept in v2v_amd/datasets.py): per batch of 12
training samples that path moves 39.5 MB of uint8 clips and 31 MB of float frames through (worker) np.stack -> torch tensor ->
default_collate copy -> shared-memory copy -> (main) page-locked copy -> H2D, builds the float frames on the host, and crosses
PCIe with both.  Here:
  * fork()ed DataLoader workers (they never touch HIP) write each sample's DECODED frames once each (the reference's pause schedule
    repeats frames -- 27 % of a training clip, data/v2v_datasets.py:286-301; the simulator gathers through a per-clip index instead,
    v2v_esim_extras), the sample's five simulator parameters and its RNG key STRAIGHT into a slot of a ring of shared memory that the GPU process has page-locked (hipHostRegister): no stack,
    no collate copy, no queue payload (a worker returns the slot number), no staging copy in the GPU process;
  * the GPU process issues ONE asynchronous H2D copy per batch (the whole slot: clips + parameters + keys) on a copy stream,
    one batch ahead of the compute stream;
  * the `frame` tensor is produced on the device from the clip that is already there (v2v_clip_frames_f32_hip), so float frames
    never exist on the host or on PCIe (colour frames of 'gray_in_bgr_out' travel as uint8);
  * simulator (+ x16 padding in place, + normalize_batch_voxel) = the fused launches of v2v_amd.esim / v2v_amd.postops on the
    compute stream.
Every sample gets exactly what WebvidDatasetV2.__getitem__ / SimulatingCollator give for the same np.random draws
(tests/test_loader.py).  Samplers, shuffling, worker seeding, drop_last and persistent workers are torch's own DataLoader's.
"""
from __future__ import annotations

import ctypes as C
import mmap
import time

import numpy as np
import torch
from torch.utils.data import BatchSampler, ConcatDataset, DataLoader, RandomSampler, SequentialSampler

from . import _lib, esim

_PARAM_KEYS = ("pos_thres", "neg_thres", "base_noise_std", "hot_pixel_fraction", "hot_pixel_std")


def clip_frames_f32(src: torch.Tensor, pick=None, frames: int | None = None) -> torch.Tensor:
    """uint8 CUDA [B,T,H,W] (gray) or [B,T,H,W,C] -> float32 [B,L,C,H,W] = src[:, pick] / 255 (data/v2v_datasets.py:329-338,352),
    IEEE division on the device = torch's CPU values bit for bit.  pick: int32 CUDA tensor / list of frame indices (None: the first
    `frames` frames, default all)."""
    _lib.require_gpu()
    if not src.is_cuda or src.dtype != torch.uint8 or src.ndim not in (4, 5):
        raise ValueError("src must be a uint8 CUDA tensor [B,T,H,W] or [B,T,H,W,C]")
    if src.ndim == 4:
        src = src.unsqueeze(-1)
    b, t, h, w, c = src.shape
    if src.stride(4) != 1 or src.stride(3) != c or src.stride(2) != w * c:
        src = src.contiguous()
    if pick is not None and not isinstance(pick, torch.Tensor):
        if len(pick) and (min(pick) < 0 or max(pick) >= t):
            raise IndexError("pick outside the clip")
        pick = torch.tensor(list(pick), dtype=torch.int32, device=src.device)
    n_l = int(pick.numel()) if pick is not None else (t if frames is None else int(frames))
    out = torch.empty((b, n_l, c, h, w), dtype=torch.float32, device=src.device)
    if b == 0 or n_l == 0:
        return out
    with torch.cuda.device(src.device):
        rc = _lib.lib().v2v_clip_frames_f32_hip(C.c_void_p(src.data_ptr()), src.stride(0) if b > 1 else t * src.stride(1), src.stride(1),
                                                C.c_void_p(pick.data_ptr()) if pick is not None else None, b, n_l, h, w, c,
                                                C.c_void_p(out.data_ptr()), _lib.stream_ptr())
    _lib.check(rc)
    return out


def clip_frames_packed(clips: torch.Tensor, clip_offsets: torch.Tensor, pick: torch.Tensor, h: int, w: int, align: int = 16,
                       stored_frames: torch.Tensor | None = None) -> torch.Tensor:
    """`frame` of a batch of PACKED gray clips (flat uint8 buffer, clip b at clip_offsets[b], stored frame pick[b, l] for output l):
    float32 [B,L,1,H,W] = frame / 255, as clip_frames_f32.  stored_frames int32 [B] (frames each clip holds): a pick outside its clip, or a
    frame that does not fit `clips`, is not read -- its output frame is NaN (v2v_clip_frames_f32_bounded_hip)."""
    _lib.require_gpu()
    b, n_l = pick.shape
    out = torch.empty((b, n_l, 1, h, w), dtype=torch.float32, device=clips.device)
    if b == 0 or n_l == 0:
        return out
    with torch.cuda.device(clips.device):
        rc = _lib.lib().v2v_clip_frames_f32_bounded_hip(C.c_void_p(clips.data_ptr()), align, C.c_void_p(clip_offsets.data_ptr()), h * w,
                                                        C.c_void_p(pick.data_ptr()), n_l,
                                                        C.c_void_p(stored_frames.data_ptr()) if stored_frames is not None else None,
                                                        clips.numel() if stored_frames is not None else 0,
                                                        b, n_l, h, w, 1, C.c_void_p(out.data_ptr()), _lib.stream_ptr())
    _lib.check(rc)
    return out


def choose_normalize_method(params: np.ndarray, frames_per_bin: int, put_noise_external: bool) -> str:
    """'count' (exact counting select over the integers -255..255 with overflow bins) when the batch's own parameters bound every
    NON-hot pixel's |count| by 255 and hot pixels stay far below the 1 % the quantiles cut off; else 'radix' (any float32 content).
    A SUM bin of a normal pixel holds at most frames_per_bin * (log-intensity range 6.91 + base-noise excursion) / C events (the
    table Gaussians end at 4.009 sigma, both signs); a hot pixel adds hot_pixel_std * |g| / C per frame and can exceed 255 -- those
    land in the overflow bins, which the k-th values never reach while hot pixels are < 0.5 % (hot_pixel_fraction_range ends at
    0.001 in every shipped config; a sample whose rank does reach an overflow bin comes out NaN, never mis-scaled)."""
    if put_noise_external:
        return "radix"
    c_min = max(float(params[:, :2].min()), 1e-12)
    bound = frames_per_bin * (6.91 + 8.1 * float(params[:, 2].max())) / c_min + 1
    return "count" if bound <= 255 and float(params[:, 3].max()) < 0.005 else "radix"


# --------------------------------------------------------------------------------------------------------------------- ring
class _SlotLayout:
    """Byte layout of one slot.  Fixed head (always copied): [clip offsets i64 B][frame index i32 B*N][frame picks i32 B*Lf][params f64 B*5]
    [keys i64 B*2][used bytes i64 1][stored frames i32 B][colour frames u8 B*Lf*H*W*3 (gray_in_bgr_out only)]; then the clips, PACKED: clip b holds its decoded
    frames once each at byte `offsets[b]` of the clip region (a multiple of 16), `used` bytes in all -- the H2D copy ends there."""

    def __init__(self, batch, n, h, w, lf, colour):
        self.batch, self.n, self.h, self.w, self.lf, self.colour = batch, n, h, w, lf, colour
        al = lambda v: (v + 255) // 256 * 256                           # noqa: E731
        self.off_offsets = 0
        self.off_fidx = al(batch * 8)
        self.off_pick = self.off_fidx + al(batch * n * 4)
        self.off_params = self.off_pick + al(batch * lf * 4)
        self.off_keys = self.off_params + al(batch * 5 * 8)
        self.off_used = self.off_keys + al(batch * 2 * 8)
        self.off_stored = self.off_used + 256                           # frames each clip holds: the simulator bounds its gather with them
        self.off_cframes = self.off_stored + al(batch * 4)
        self.off_clips = self.off_cframes + (al(batch * lf * h * w * 3) if colour else 0)
        # every packed clip is rounded up to 16 bytes (see _RingDataset): room for that rounding when n*h*w % 16 != 0 (crop sizes that
        # are not multiples of 4) and no clip of the batch pauses
        self.clip_room = (n * h * w + 15) // 16 * 16
        self.nbytes = self.off_clips + al(batch * self.clip_room)

    @staticmethod
    def _v(buf, off, count, dtype, shape):
        return buf[off:off + count * np.dtype(dtype).itemsize].view(dtype).reshape(shape)

    def views(self, buf: np.ndarray):
        """NumPy views of one slot (a uint8 array of nbytes): offsets, fidx, pick, params, keys, used, colour frames, clip region, stored."""
        b, n, h, w, lf = self.batch, self.n, self.h, self.w, self.lf
        return (self._v(buf, self.off_offsets, b, np.int64, (b,)), self._v(buf, self.off_fidx, b * n, np.int32, (b, n)),
                self._v(buf, self.off_pick, b * lf, np.int32, (b, lf)), self._v(buf, self.off_params, b * 5, np.float64, (b, 5)),
                self._v(buf, self.off_keys, b * 2, np.int64, (b, 2)), self._v(buf, self.off_used, 1, np.int64, (1,)),
                self._v(buf, self.off_cframes, b * lf * h * w * 3, np.uint8, (b, lf, h, w, 3)) if self.colour else None,
                buf[self.off_clips:self.off_clips + b * self.clip_room], self._v(buf, self.off_stored, b, np.int32, (b,)))

    def device_views(self, dbuf: torch.Tensor):
        b, n, h, w, lf = self.batch, self.n, self.h, self.w, self.lf
        v = lambda off, nbytes, dt, shape: dbuf[off:off + nbytes].view(dt).view(shape)       # noqa: E731
        return (v(self.off_offsets, b * 8, torch.int64, (b,)), v(self.off_fidx, b * n * 4, torch.int32, (b, n)),
                v(self.off_pick, b * lf * 4, torch.int32, (b, lf)), v(self.off_params, b * 40, torch.float64, (b, 5)),
                v(self.off_keys, b * 16, torch.int64, (b, 2)),
                dbuf[self.off_cframes:self.off_cframes + b * lf * h * w * 3].view(b, lf, h, w, 3) if self.colour else None,
                dbuf[self.off_clips:self.off_clips + b * self.clip_room], v(self.off_stored, b * 4, torch.int32, (b,)))


def _leaf(dataset, idx):
    """(leaf dataset, local index) through the ConcatDataset nesting of data/data_interface.py:19-27."""
    while isinstance(dataset, ConcatDataset):
        if idx < 0:
            idx += len(dataset)
        import bisect
        di = bisect.bisect_right(dataset.cumulative_sizes, idx)
        idx = idx if di == 0 else idx - dataset.cumulative_sizes[di - 1]
        dataset = dataset.datasets[di]
    return dataset, idx


def _leaves(dataset):
    if isinstance(dataset, ConcatDataset):
        for d in dataset.datasets:
            yield from _leaves(d)
    else:
        yield dataset


class _RingDataset(torch.utils.data.Dataset):
    """What the workers index: item = (sample index, slot, position in the batch).  The sample's host half is written into the
    slot; only the slot number travels back through the worker queue."""

    def __init__(self, base, ring: np.ndarray, layout: _SlotLayout, pick):
        self.base, self.ring, self.layout, self.pick = base, ring, layout, np.asarray(pick, dtype=np.int64)

    def __len__(self):
        return len(self.base)

    def __getitem__(self, item):
        idx, slot, pos = item
        leaf, li = _leaf(self.base, idx)
        lay = self.layout
        offsets, fidx, pick, params, keys, used, cframes, clips, stored = lay.views(self.ring[slot])
        # a DataLoader worker builds a whole batch, sample after sample: the clips are packed back to back as they come
        start = 0 if pos == 0 else int(used[0])
        room = clips[start:start + lay.n * lay.h * lay.w].reshape(lay.n, lay.h, lay.w)
        _, n_stored = leaf.host_sample_into(li, room, params[pos], keys[pos], cframes[pos] if cframes is not None else None, fidx[pos])
        offsets[pos] = start
        stored[pos] = n_stored
        pick[pos] = fidx[pos][self.pick]                                   # the stored frames handed out as `frame`
        used[0] = start + (n_stored * lay.h * lay.w + 15) // 16 * 16
        return slot, leaf.data_source_idx


class _SlotBatchSampler:
    """Wraps a batch sampler: batch number c of the loader's lifetime goes to ring slot c % slots."""

    def __init__(self, batch_sampler, slots, counter):
        self.batch_sampler, self.slots, self.counter = batch_sampler, slots, counter

    def __len__(self):
        return len(self.batch_sampler)

    def __iter__(self):
        for batch in self.batch_sampler:
            slot = self.counter[0] % self.slots
            self.counter[0] += 1
            yield [(int(i), slot, pos) for pos, i in enumerate(batch)]


def _ring_collate(items):
    return items[0][0], [it[1] for it in items]


class RingLoader:
    """See the module docstring.  Arguments as torch's DataLoader where they share a name; the dataset is a
    v2v_amd.datasets.WebvidDatasetV2 (any `defer_sim` setting; not `gpu_frontend`; not `sim_rng: numpy` -- the ring always simulates with the
    device-native Philox noise keyed by {seed drawn per sample, sample index}, the bit-exact np.random replay lives on the per-sample path
    only; `sim_device` / `output_device` are not consulted either: batches are simulated and handed out on `device`) or ConcatDatasets of
    them with one clip shape.

    pad_to / normalize   the consumer-side post-ops done where the voxels are produced (model/train_utils.py:147-166, 322-326): events
                         are written into the x`pad_to`-padded layout and, with normalize=True, normalize_batch_voxel is applied in
                         place -- then run the model with normalize_voxels: false.  The k-th values come from statistics the
                         simulator's writer keeps (no histogram pass over the tensor).  normalize='scales' leaves the events raw and
                         adds batch['event_scales'] float32 [B,2] = (neg_max, pos_max) for a consumer that scales while it reads them
                         (v2v_amd.unet's head): no pass over the tensor at all.
    depth                device-side slots (batches in flight on the GPU): 2 = copy of batch k+1 under the compute of batch k.
    timers               optional dict: accumulates host seconds per stage (wait_batch, h2d_enqueue, sim, postops, frames, assemble)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, sampler=None, num_workers=0, drop_last=True, prefetch_factor=2,
                 persistent_workers=False, worker_init_fn=None, generator=None, pad_to=1, normalize=False, device="cuda", depth=2,
                 timers=None):
        _lib.require_gpu()
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.pad_to, self.normalize = int(pad_to), normalize
        self.timers = timers
        leaves = list(_leaves(dataset))
        if not leaves:
            raise ValueError("empty dataset")
        lf0 = leaves[0]
        shape = lambda d: (d.frames_per_seq + 1, d.crop_size, d.color_mode, len(d.frame_pick()), d.num_bins, d.frames_per_bin,   # noqa: E731
                           d.put_noise_external, d.output_additional_evs, d.sim_rng == "numpy")
        for d in leaves:
            if not hasattr(d, "host_sample_into"):
                raise TypeError("RingLoader needs v2v_amd.datasets.WebvidDatasetV2 leaves")
            if shape(d) != shape(lf0):
                raise ValueError("all datasets of a RingLoader must share clip length, crop size, colour mode and binning")
            if d.gpu_frontend:
                raise TypeError("RingLoader ships host-decoded clips: configure the dataset with gpu_frontend: false")
            if d.sim_rng == "numpy":
                raise TypeError("RingLoader simulates with the device-native Philox noise; `sim_rng: numpy` (bit-exact replay of the reference's "
                                "np.random stream) is served by the per-sample path: use torch's DataLoader (create_dataloader does)")
        self.leaf = lf0
        n, hw = lf0.frames_per_seq + 1, lf0.crop_size
        self.pick = lf0.frame_pick()
        self.layout = _SlotLayout(self.batch_size, n, hw, hw, len(self.pick), lf0.color_mode != "gray")
        if sampler is None:
            sampler = RandomSampler(dataset, generator=generator) if shuffle else SequentialSampler(dataset)
        self.sampler = sampler                                             # train loops call loader.sampler.set_epoch(epoch) under DDP
        self.drop_last = drop_last
        self.num_workers = int(num_workers or 0)
        self.depth = max(2, int(depth))
        in_flight = max(1, self.num_workers) * (prefetch_factor if self.num_workers else 1)
        self.slots = in_flight + self.depth + 1
        # anonymous shared mapping: inherited by fork()ed workers, not limited by the size of /dev/shm; page-locked below
        self._map = mmap.mmap(-1, self.slots * self.layout.nbytes)
        self.ring = np.frombuffer(self._map, dtype=np.uint8).reshape(self.slots, self.layout.nbytes)
        self._registered = False
        self._register()
        self._counter = [0]
        bs = _SlotBatchSampler(BatchSampler(sampler, self.batch_size, drop_last), self.slots, self._counter)
        kw = dict(num_workers=self.num_workers, collate_fn=_ring_collate, worker_init_fn=worker_init_fn)
        if self.num_workers:
            kw.update(prefetch_factor=prefetch_factor, persistent_workers=persistent_workers, multiprocessing_context="fork")
        self.loader = DataLoader(_RingDataset(dataset, self.ring, self.layout, self.pick), batch_sampler=bs, **kw)
        self._it = None
        self._generation = 0                                               # one live iterator: a newer __iter__ retires the older ones
        self.bytes_copied, self.batches_copied = 0, 0                      # H2D bytes / batches so far (heads + packed clips)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._dev = [torch.empty(self.layout.nbytes, dtype=torch.uint8, device=self.device) for _ in range(self.depth)]
        self._dev_free = [torch.cuda.Event() for _ in range(self.depth)]          # the slot's last consumer is done
        self._ring_t = torch.from_numpy(self.ring)

    # ---- page-locking
    def _register(self):
        with torch.cuda.device(self.device):
            rc = torch.cuda.cudart().cudaHostRegister(self.ring.ctypes.data, self.ring.nbytes, 0)
        if int(rc) != 0:
            # not fatal: out of pageable memory the runtime stages every copy itself (slower, and synchronous with the host) -- what the
            # reference's DataLoader(pin_memory=False) lives with; say so once, loudly
            import warnings
            warnings.warn(f"hipHostRegister of the {self.ring.nbytes >> 20} MiB clip ring failed ({rc}): the ring stays pageable and H2D copies run "
                          "at a fraction of the link rate -- raise RLIMIT_MEMLOCK or lower num_workers / prefetch_factor", RuntimeWarning, stacklevel=3)
            return
        self._registered = True

    def close(self):
        if self._registered:
            torch.cuda.synchronize(self.device)
            torch.cuda.cudart().cudaHostUnregister(self.ring.ctypes.data)
            self._registered = False

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def __len__(self):
        return len(self.loader)

    # ---- one batch
    def _t(self, key, t0):
        if self.timers is not None:
            self.timers[key] = self.timers.get(key, 0.0) + (time.perf_counter() - t0)

    def _stage(self, item, k):
        """Enqueue the H2D copy of ring slot `slot` (its head + the packed clips: `used` bytes) into device slot k % depth."""
        slot, src_idx = item
        t0 = time.perf_counter()
        d = k % self.depth
        lay = self.layout
        offsets, _, _, params, _, used, _, _, _ = lay.views(self.ring[slot])
        nbytes = lay.off_clips + int(used[0])
        ev = torch.cuda.Event()
        with torch.cuda.stream(self.copy_stream):
            # the launches that read this device slot `depth` batches ago (a host-side wait instead was measured: no faster)
            self.copy_stream.wait_event(self._dev_free[d])
            self._dev[d][:nbytes].copy_(self._ring_t[slot, :nbytes], non_blocking=True)
            ev.record(self.copy_stream)
        params, offsets = params.copy(), offsets.copy()                  # host-side values of the batch, read before the slot is recycled
        self.bytes_copied += nbytes
        self.batches_copied += 1
        self._t("h2d_enqueue", t0)
        return d, ev, params, src_idx, offsets

    def _finish(self, staged):
        d, ev, params, src_idx, offsets = staged
        lay, leaf = self.layout, self.leaf
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(ev)
        offsets_d, fidx_d, pick_d, params_d, keys_d, cframes, clips, stored_d = lay.device_views(self._dev[d])
        nb = len(src_idx)                                                 # < batch_size only for the last batch of an epoch with drop_last=False
        if nb < lay.batch:
            offsets_d, fidx_d, pick_d, params_d, keys_d, stored_d = offsets_d[:nb], fidx_d[:nb], pick_d[:nb], params_d[:nb], keys_d[:nb], stored_d[:nb]
            cframes = cframes[:nb] if cframes is not None else None
            params = params[:nb]
        t0 = time.perf_counter()
        h, w = lay.h, lay.w
        method = choose_normalize_method(params, leaf.frames_per_bin, leaf.put_noise_external) if self.normalize else None
        stats = torch.empty((nb, _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device=self.device) if method == "count" else None
        if leaf.put_noise_external:
            # external noise has no indexed instance: gather the clips on the device first (an ablation configuration, not a training one)
            gathered = torch.stack([clips[int(o):int(o) + lay.n * h * w].view(lay.n, h, w)[fi.long()] for o, fi in zip(offsets[:nb].tolist(), fidx_d)])
            vox = esim.esim_voxel_batch(gathered, params_d, bin_mode="sum", num_bins=leaf.num_bins, frames_per_bin=leaf.frames_per_bin, rng_mode="philox",
                                        clip_keys=keys_d, put_noise_external=True, pad_to=self.pad_to, validate=False)
        else:
            vox = esim.esim_voxel_packed(clips, offsets_d, fidx_d, h, w, params_d, keys_d, num_bins=leaf.num_bins, frames_per_bin=leaf.frames_per_bin,
                                         pad_to=self.pad_to, stats=stats, stored_frames=stored_d)
        self._t("sim", t0)
        t0 = time.perf_counter()
        batch = {}
        if self.normalize:
            from . import postops
            if method == "count":                                         # the writer's statistics -> exact scales, no pass over the tensor
                scales = postops.scales_from_stats(stats, vox.shape[1] * vox.shape[2] * h * w)
            else:                                                         # counts beyond the counting range / external noise: radix select
                scales = postops.voxel_scales_radix(vox[..., :h, :w])
            if self.normalize == "scales":
                batch["event_scales"] = scales
            else:
                vox = postops.apply_scales(vox, scales, self.pad_to, valid_hw=(h, w), inplace=True)
        self._t("postops", t0)
        t0 = time.perf_counter()
        if cframes is None:
            frame = clip_frames_packed(clips, offsets_d, pick_d, h, w, stored_frames=stored_d)
        else:
            frame = clip_frames_f32(cframes)
        self._dev_free[d].record(cur)                                     # everything that reads the device slot is enqueued
        self._t("frames", t0)
        t0 = time.perf_counter()
        batch["frame"] = frame
        batch["events"] = vox
        batch["data_source_idx"] = torch.tensor(src_idx, dtype=torch.int64)
        batch["v2e_params"] = {k: torch.from_numpy(params[:, i].copy()) for i, k in enumerate(_PARAM_KEYS)}
        self._t("assemble", t0)
        return batch

    def __iter__(self):
        """A plain method, not a generator: the previous epoch is retired HERE, when iter(loader) is called, not at the new iterator's first
        next().  An abandoned epoch's workers may still be writing into ring slots: its DataLoader iterator is shut down explicitly (a
        generator object someone still references would otherwise keep its non-persistent workers alive and writing), persistent workers
        drain their queue when the DataLoader restarts them, and no copy of the old epoch may still read the ring."""
        old, self._it = self._it, None
        self._generation += 1
        if old is not None and not getattr(self.loader, "persistent_workers", False) and hasattr(old, "_shutdown_workers"):
            old._shutdown_workers()
        del old
        self.copy_stream.synchronize()
        it = self._it = iter(self.loader)
        return self._batches(it, self._generation)

    def _batches(self, it, gen):
        k = 0
        copies = []                                                       # H2D events of the batches whose ring slot may still be read

        def fetch():
            nonlocal k
            if gen != self._generation:                                   # the ring and its slot counter serve ONE iterator (torch's DataLoader allows several)
                raise RuntimeError("this RingLoader iterator was retired by a newer iter(loader): the ring's slots serve one epoch at a time")
            # a ring slot is rewritten `slots` batches later, and a worker may start on it once the batch `in_flight` before it
            # has been handed out: make sure the H2D copy of the batch `depth + 1` back has left the ring before asking for more
            while len(copies) > self.depth:
                copies.pop(0).synchronize()
            t0 = time.perf_counter()
            try:
                item = next(it)
            except StopIteration:
                return None
            self._t("wait_batch", t0)
            st = self._stage(item, k)
            copies.append(st[1])
            k += 1
            return st

        nxt = fetch()
        while nxt is not None:
            cur, nxt = nxt, fetch()                                       # one batch of look-ahead: copy of k+1 under the compute of k
            yield self._finish(cur)


def create_dataloader(dataset, configs, batch_size, local_rank):
    """Drop-in for the reference's `create_dataloader` (train.py:52-65; same signature, same sampler choice, drop_last=True):

        from v2v_amd.loader import create_dataloader          # the one line a maintainer changes in train.py

    Training datasets made of v2v_amd.datasets.WebvidDatasetV2 (through any ConcatDataset nesting, data/data_interface.py:19-27) get a
    RingLoader -- batches arrive on the GPU, `batch[k] = v.to(device)` (train.py:79-81) is then a no-op; anything else (the
    TestH5Dataset validation sets) gets the reference's own DataLoader.  Extra keys of the `dataset:` block of the YAML:
        normalize_in_loader   false | true | 'scales'   normalize_batch_voxel where the voxels are written (then normalize_voxels: false
                                                        for the model, model/train_utils.py:200,319-320)
        pad_events_to         1                         16 writes the x16-padded layout forward_sequence builds (:322-326) in place"""
    from torch.utils.data import DistributedSampler
    sampler = DistributedSampler(dataset) if local_rank is not None else RandomSampler(dataset)
    num_workers = configs.get("num_workers")
    persistent_workers = configs.get("persistent_workers", False)
    leaves = list(_leaves(dataset))
    ring_ok = all(hasattr(d, "host_sample_into") and not getattr(d, "gpu_frontend", False) and getattr(d, "sim_rng", "philox") != "numpy"
                  for d in leaves) and torch.cuda.is_available()
    if ring_ok:
        device = torch.device("cuda", local_rank if local_rank is not None else torch.cuda.current_device())
        return RingLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers or 0, drop_last=True,
                          persistent_workers=bool(persistent_workers and num_workers), pad_to=configs.get("pad_events_to", 1),
                          normalize=configs.get("normalize_in_loader", False), device=device)
    # a dataset that asks for spawned workers (`worker_start_method`, v2v_amd/datasets.py) gets them through the loader's own argument here
    ctx = next((c for c in (getattr(d, "multiprocessing_context", None) for d in leaves) if c is not None), None) if num_workers else None
    return DataLoader(dataset, batch_size=batch_size, sampler=sampler, num_workers=num_workers or 0, multiprocessing_context=ctx,
                      persistent_workers=bool(persistent_workers and num_workers), pin_memory=configs.get("pin_memory", False), drop_last=True)
