"""Dataset-side drop-in: the reference's `WebvidDatasetV2` surface over the fused HIP simulator.

Selected exactly like the reference's own dataset, by dotted path in the experiment YAML
(`class_name: v2v_amd.datasets.WebvidDatasetV2`; reference plugin loader data/data_interface.py:7-19,
constructor contract `(dataset_path, configs)`), so `train.py` runs unchanged.

What is mirrored from data/v2v_datasets.py (line numbers of the reference):
    load_configs defaults and asserts                      :26-92
    sample index construction                              :95-141
    __getitem__: RNG draw order (scale, crop, flip, pause  :227-361
      chain), frame gathering, tensor assembly, dict keys
    imgs_to_voxels: 6-draw parameter sampling, assert,     :363-410
      [L,Tb,H,W] sum binning, v2e_params dict
    bgr_to_gray                                            :19-22
The simulator + binning itself (`EventEmulator.video_to_voxel` + reshape/sum) is ONE launch of the HIP kernel.
Decode/crop/resize stay on the host (OpenCV when present); hosts without OpenCV can plug a `frame_source`
callable (used by the tests and the synthetic end-to-end bench).  There is no CPU simulator fallback.

HIP and DataLoader workers: a forked worker cannot use the GPU.  Use `num_workers: 0` (the simulator is no longer
the bottleneck) or a spawn context; see INTEGRATION.md.
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import esim

# utils/data.py:7 of the reference (index of the source name travels with every sample)
data_sources = ('esim', 'ijrr', 'mvsec', 'eccd', 'hqf', 'unknown', 'reds', 'sportsslomo', 'adobe', 'youcook', 'vimeo',
                'webvid', 'evbird', 'evaid', 'hs-ergb', 'openvid')


def bgr_to_gray(img_stack):
    """[N,H,W,3] -> [N,H,W] uint8; weights hit channels 0,1,2 in this order, truncating cast (v2v_datasets.py:19-22)."""
    gray = np.dot(img_stack[..., :3], [0.5870, 0.1140, 0.2989])
    return gray.astype(np.uint8)


def sample_sim_params(threshold_range, max_thres_pos_neg_gap, base_noise_std_range, hot_pixel_fraction_range,
                      hot_pixel_std_range, use_fixed_thresholds=False, pos_thres=None, neg_thres=None,
                      scale_noise_strength=False, put_noise_external=False):
    """The six scalar draws of imgs_to_voxels in the reference's order (v2v_datasets.py:368-386), global np.random."""
    if not use_fixed_thresholds:
        thres_1 = np.random.uniform(*threshold_range)
        gap = np.random.uniform(1, max_thres_pos_neg_gap)
        thres_2 = thres_1 * gap
        if np.random.rand() > 0.5:
            pos_thres, neg_thres = thres_1, thres_2
        else:
            pos_thres, neg_thres = thres_2, thres_1
    base_noise_std = np.random.uniform(*base_noise_std_range)
    hot_pixel_fraction = np.random.uniform(*hot_pixel_fraction_range)
    hot_pixel_std = np.random.uniform(*hot_pixel_std_range)
    if scale_noise_strength and not put_noise_external:
        base_noise_std = base_noise_std * pos_thres
        hot_pixel_std = hot_pixel_std * pos_thres
    return {"pos_thres": pos_thres, "neg_thres": neg_thres, "base_noise_std": base_noise_std,
            "hot_pixel_fraction": hot_pixel_fraction, "hot_pixel_std": hot_pixel_std}


def draw_sim_seed() -> int:
    """Seed half of the device RNG key of one sample: 62 bits from the global np.random stream (a power-of-two range: exactly
    one 64-bit draw, no rejection loop).  With the sample index as the other half of the key, two samples share their noise
    fields only if both halves collide (a 31-bit seed alone collided about twice per 1e5 samples)."""
    return int(np.random.randint(0, 2**62, dtype=np.int64))


def synthetic_frame_source(dataset, sample_idx, start_frame, end_frame, crop_size_before_resize, min_i, min_j, flip, need_h, need_w):
    """A `frame_source` without OpenCV or video files: a smooth random-walk video that is a pure function of the clip's first
    frame index (its own generator; the global np.random stream is untouched).  Module-level, hence picklable: usable with
    spawned DataLoader workers and by the tests / synthetic end-to-end runs."""
    g = np.random.default_rng(1000 + start_frame)
    c = 3 if dataset.color_mode == "gray_in_bgr_out" else 1
    base = g.uniform(0, 255, size=(need_h, need_w, c))
    out = []
    for _ in range(end_frame - start_frame):
        base = np.clip(base + g.normal(0, 6, size=base.shape), 0, 255)
        f = base.astype(np.uint8)
        out.append((f[:, ::-1] if flip else f).copy())
    return out


_worker_exit_hook_installed = False


def _worker_exit(dataset_ref, join_timeout_s=2.0):
    """atexit hook of a SPAWNED DataLoader worker that used the GPU (installed by the first simulating __getitem__ in such a process).

    Why it exists: a spawned child ends through the interpreter's full finalisation (multiprocessing/spawn.py: `sys.exit(exitcode)`;
    a fork()ed child ends with `os._exit`).  torch's worker loop ends with `data_queue.cancel_join_thread(); data_queue.close()`
    (torch/utils/data/_utils/worker.py), so the result queue's feeder thread -- a daemon thread that PICKLES the batches, which for
    `output_device: cuda` means `storage._share_cuda_()` (hipIpcGetMemHandle, GIL released) -- may still be inside that C++ call when
    finalisation starts.  CPython 3.10 ends a daemon thread that asks for the GIL after that point with pthread_exit(); the forced
    unwind crosses a noexcept C++ frame and the worker dies in std::terminate ("terminate called without an active exception",
    SIGABRT), which train.py's DataLoader reports as "worker killed by signal: Aborted".  atexit callbacks run BEFORE finalisation,
    while threads may still take the GIL, so here the worker (1) lets the feeder thread(s) send what is buffered and return (the
    queue is already closed: the sentinel is in its buffer), (2) waits for its own launches, drops the page-locked staging slots and
    device buffers and returns the IPC blocks whose consumers have released them.  If a feeder cannot finish within `join_timeout_s`
    (the training process stopped reading and the pipe is full; the timeout stays below the 5 s torch's DataLoader grants a worker
    before it terminate()s it) the worker ends the way a fork()ed one does, with os._exit(0), rather than abort."""
    import gc
    import multiprocessing.queues as mpq
    stuck = False
    for q in [o for o in gc.get_objects() if type(o) is mpq.Queue]:       # type(), not isinstance(): lazy module attributes must not be woken here
        t = getattr(q, "_thread", None)
        if t is not None and t.is_alive():
            if not getattr(q, "_closed", False):
                q.close()
            t.join(join_timeout_s)
            stuck = stuck or t.is_alive()
    ds = dataset_ref()
    if ds is not None:
        ds.__dict__.pop("_staging", None)
    if torch.cuda.is_initialized():
        try:
            torch.cuda.synchronize()
            torch.cuda.ipc_collect()
        except Exception:  # noqa: BLE001 - the process is ending; nothing useful to do with a device error here
            pass
    if stuck:
        import sys
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


def _install_worker_exit_hook(dataset):
    """Called from __getitem__: only does something inside a DataLoader worker (torch.utils.data.get_worker_info())."""
    global _worker_exit_hook_installed
    if _worker_exit_hook_installed:
        return
    _worker_exit_hook_installed = True
    if torch.utils.data.get_worker_info() is None or os.environ.get("V2V_WORKER_EXIT_HOOK", "1") == "0":
        return
    import atexit
    import weakref
    atexit.register(_worker_exit, weakref.ref(dataset))


class WebvidDatasetV2(torch.utils.data.Dataset):
    """Same constructor, config keys, defaults, `__len__`, `__getitem__` contract as the reference class.

    Extra (optional) config keys understood by this implementation only:
        sim_rng        'philox' (default; device RNG keyed by {a 62-bit seed drawn from np.random per sample, sample_idx} --
                       ONE extra global draw per sample compared with the reference, so the crop/scale/pause draws of
                       later samples differ from a reference run with the same np.random seed; only 'numpy' reproduces the
                       reference's stream), 'philox_fast' (alias of 'philox' since round 2) or 'numpy' (fields drawn on
                       the host from the global stream in the reference's order: bit-exact replay)
        sim_device     device of the simulator launch, default 'cuda'
        output_device  'cpu' (default: what default_collate / pin_memory expect) or 'cuda' (skip the round trip)
        frame_source   callable(dataset, sample_idx, start_frame, end_frame, crop_size_before_resize, min_i, min_j, flip,
                       need_h, need_w) -> list of end_frame - start_frame uint8 frames ALREADY cropped, resized to
                       [need_h, need_w, C] and flipped (need_* = crop_size + the shake margin; C = 1 for color_mode
                       'gray', 3 for 'gray_in_bgr_out'); replaces OpenCV decoding (tests, synthetic benches).  Must be
                       picklable (a module-level function) when DataLoader workers are spawned; see synthetic_frame_source
        video_size     (width, height) reported for every video when frame_source is used
        worker_start_method  'spawn': train.py's own DataLoader then starts spawned workers (each owns a HIP context).  Sets the process-wide
                       default start method ONLY if the program has not fixed one yet (warns otherwise); `dataset.multiprocessing_context`
                       is the explicit form for programs that build the loader themselves -- see load_configs
    """

    def load_configs(self, configs):
        g = configs.get
        self.FPS = g("FPS", 30)
        self.L = g("sequence_length", 40)
        step_size = g("step_size", None)
        self.proba_pause_when_running = g("proba_pause_when_running", 0.01)
        self.proba_pause_when_paused = g("proba_pause_when_paused", 0.98)
        self.fixed_seed = g("fixed_seed", None)
        self.crop_size = g("crop_size", None)
        self.fixed_crop = g("fixed_crop", False)
        self.random_flip = g("random_flip", True)
        self.num_bins = g("num_bins", 5)
        self.frames_per_bin = g("frames_per_bin", 1)
        self.frames_per_img = self.num_bins * self.frames_per_bin
        self.frames_per_seq = self.frames_per_img * self.L
        self.step_size = step_size if step_size is not None else self.frames_per_seq
        self.min_resize_scale = g("min_resize_scale", 0)
        self.max_resize_scale = g("max_resize_scale", 1.3)
        self.max_rotate_degrees = g("max_rotate_degrees", 0)
        self.shake_frames = g("shake_frames", 0)
        self.shake_std = g("shake_std", 0)
        self.threshold_range = g("threshold_range", [0.05, 2])
        self.max_thres_pos_neg_gap = g("max_thres_pos_neg_gap", 1.5)
        self.base_noise_std_range = g("base_noise_std_range", [0, 0.2])
        self.hot_pixel_fraction_range = g("hot_pixel_fraction_range", [0, 0.001])
        self.hot_pixel_std_range = g("hot_pixel_std_range", [0, 0.2])
        self.put_noise_external = g("put_noise_external", False)
        self.scale_noise_strength = g("scale_noise_strength", False)
        self.max_samples_per_shot = g("max_samples_per_shot", 1)
        self.subsample_ratio = g("subsample_ratio", 1)
        self.force_hwaccel = g("force_hwaccel", False)
        self.video_reader = g("video_reader", "ffmpeg")
        assert self.video_reader in ["ffmpeg", "opencv"]
        self.keep_top_percentile = g("keep_top_percentile", 0.54)
        self.use_fixed_thresholds = g("use_fixed_thresholds", False)
        self.data_source_name = g("data_source_name", "reds")
        self.data_source_idx = data_sources.index(self.data_source_name)
        self.color_mode = g("color_mode", "gray")
        assert self.color_mode in ["gray", "gray_in_bgr_out"]
        assert self.L > 0
        assert self.step_size > 0
        self.output_additional_frame = g("output_additional_frame", False)
        self.output_additional_evs = g("output_additional_evs", False)
        if self.output_additional_evs:
            self.frames_per_seq += self.frames_per_img
        self.video_degrade = g("video_degrade", None)
        assert self.video_degrade in [None, "subtitles", "dirtyshotcut", "hdr", "ldr"]
        self.degrade_ratio = g("degrade_ratio", 0)
        # ---- this implementation's own knobs
        self.sim_rng = g("sim_rng", "philox")
        assert self.sim_rng in ["philox", "philox_fast", "numpy"]
        self.sim_device = g("sim_device", "cuda")
        self.output_device = g("output_device", "cpu")
        self.frame_source = g("frame_source", None)
        self.video_size = g("video_size", None)
        # defer_sim: __getitem__ does the host work only (decode, crop, parameter draws) and returns the raw clip;
        # SimulatingCollator then simulates the whole batch in ONE launch in the main process, so DataLoader workers
        # may be fork()ed (they never touch HIP).  Results are identical to the per-sample path.
        self.defer_sim = g("defer_sim", False)
        # gpu_frontend: the host only decodes (and slices the crop rectangle out of each frame); cvtColor / resize /
        # flip / shake crop / pause-index gather run on the GPU (v2v_amd.frontend) and the clip never
        # returns to the host before the simulator.  `raw_frame_source(dataset, sample_idx, start, end)` may supply the
        # decoded [T,Hs,Ws,3] uint8 frames instead of OpenCV.  Parity of the resize itself is unpinned (no cv2 here).
        self.gpu_frontend = g("gpu_frontend", False)
        self.raw_frame_source = g("raw_frame_source", None)
        # worker_start_method: 'spawn' | 'forkserver' -- how train.py's OWN DataLoader (train.py:52-65 passes no multiprocessing_context)
        # starts its workers: a fork()ed worker cannot use HIP, a spawned one owns a HIP context and simulates its samples itself.
        # Set from the YAML, so `num_workers: 9` works with train.py untouched (each worker pays an interpreter start: use
        # `persistent_workers: true` as the shipped YAML does).
        # staged_getitem: false keeps the plain per-sample path (host-side gather, pageable copies) -- for A/B runs and the equality test
        self.staged_getitem = g("staged_getitem", True)
        self.worker_start_method = g("worker_start_method", None)
        if self.worker_start_method is not None:
            assert self.worker_start_method in ["spawn", "forkserver", "fork"]
            # The YAML-only route has no other handle on train.py's DataLoader than the process-wide default, so the key fixes the default
            # -- ONLY while the program has not chosen one (never force=True, nothing restored behind the user's back).  A program that
            # already started fork()ed pools or loaders keeps its choice and gets a warning; a program that builds the loader itself passes
            # `multiprocessing_context=dataset.multiprocessing_context` instead (v2v_amd.loader.create_dataloader does).
            import multiprocessing
            current = multiprocessing.get_start_method(allow_none=True)
            if current is None:
                multiprocessing.set_start_method(self.worker_start_method)
            elif current != self.worker_start_method:
                import warnings
                warnings.warn(f"worker_start_method: {self.worker_start_method} -- this process already fixed its multiprocessing start method to "
                              f"'{current}'; it is left as it is.  Pass multiprocessing_context=dataset.multiprocessing_context to the DataLoader "
                              f"(fork()ed workers cannot use HIP).", RuntimeWarning, stacklevel=3)
        assert not (self.gpu_frontend and self.defer_sim), "gpu_frontend runs in the process that owns the GPU; defer_sim is for fork()ed workers"

    def __init__(self, dataset_path, configs):
        self.load_configs(configs)
        self.dataset_path = dataset_path
        self.video_list_file = configs.get("video_list_file")
        with open(self.video_list_file, "r") as f:
            rows = [line.strip().split(" ") for line in f.readlines()]
        # each line: subpath framecount pos_thres neg_thres (the thresholds matter only with use_fixed_thresholds)
        self.video_list = [r[0] for r in rows]
        self.video_framecounts = [int(r[1]) for r in rows]
        self.video_pos_thres = [float(r[2]) for r in rows]
        self.video_neg_thres = [float(r[3]) for r in rows]

        names, begins, lengths, pts, nts = [], [], [], [], []
        for vi, (vpath, frame_cnt) in enumerate(zip(self.video_list, self.video_framecounts)):
            taken = 0
            for start in range(0, frame_cnt - self.frames_per_seq - 1, self.step_size):
                names.append(vpath)
                begins.append(start)
                lengths.append(self.L)
                pts.append(self.video_pos_thres[vi])
                nts.append(self.video_neg_thres[vi])
                taken += 1
                if taken >= self.max_samples_per_shot:
                    break
        keep = int(len(lengths) * self.subsample_ratio)
        self.sample_video_name = np.array(names)[:keep]
        self.sample_begin_idx = np.array(begins)[:keep]
        self.sample_L = np.array(lengths)[:keep]
        self.sample_pos_thres = pts[:keep]
        self.sample_neg_thres = nts[:keep]

    def __len__(self):
        return len(self.sample_video_name)

    @property
    def multiprocessing_context(self):
        """What a program that builds its own DataLoader passes as `multiprocessing_context=` (None without the YAML key: torch's default)."""
        if self.worker_start_method is None:
            return None
        import multiprocessing
        return multiprocessing.get_context(self.worker_start_method)

    # ------------------------------------------------------------------ decode (host; "next" row of SURVEY §8f)
    def _probe_size(self, video_path):
        if self.frame_source is not None or self.raw_frame_source is not None:
            if self.video_size is None:
                raise ValueError("frame_source / raw_frame_source need video_size=(width, height)")
            return int(self.video_size[0]), int(self.video_size[1])
        import cv2  # the reference's opencv branch, v2v_datasets.py:252-256
        cap = cv2.VideoCapture(video_path)
        size = int(cap.get(cv2.CAP_PROP_FRAME_WIDTH)), int(cap.get(cv2.CAP_PROP_FRAME_HEIGHT))
        cap.release()
        return size

    def degrade_video(self, imgs):
        """The reference's robustness ablations on the decoded clip (data/v2v_datasets.py:413-486): same np.random draw order,
        same arithmetic.  'dirtyshotcut' (cut, swap the parts, mirror one), 'hdr' / 'ldr' (contrast stretch around 127.5 in
        float64, clip, truncate to uint8) are NumPy; 'subtitles' rasterises text with OpenCV's Hershey fonts (cv2.putText) and
        therefore needs cv2 -- without it that mode raises."""
        t = len(imgs)
        if self.video_degrade == "subtitles":
            try:
                import cv2
            except ImportError as exc:
                raise NotImplementedError("video_degrade 'subtitles' draws text with cv2.putText: OpenCV is required") from exc
            fonts = [cv2.FONT_HERSHEY_SIMPLEX, cv2.FONT_HERSHEY_PLAIN, cv2.FONT_HERSHEY_DUPLEX, cv2.FONT_HERSHEY_COMPLEX,
                     cv2.FONT_HERSHEY_TRIPLEX, cv2.FONT_HERSHEY_COMPLEX_SMALL, cv2.FONT_HERSHEY_SCRIPT_SIMPLEX, cv2.FONT_HERSHEY_SCRIPT_COMPLEX]
            font = np.random.choice(fonts)
            font_scale = np.random.uniform(0.5, 1.5)
            color = (np.random.randint(0, 256), np.random.randint(0, 256), np.random.randint(0, 256))
            thickness = np.random.randint(1, 3)
            text_len = np.random.randint(5, 16)
            text = "".join(np.random.choice(list("abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789 "), size=text_len))
            h, w = imgs[0].shape[:2]
            (text_width, text_height), baseline = cv2.getTextSize(text, font, font_scale, thickness)
            org = (np.random.randint(0, max(1, w - text_width)), np.random.randint(text_height, max(text_height + 1, h - baseline)))
            for i in range(t):
                img = imgs[i].copy()
                if img.shape[2] == 1:
                    img = cv2.cvtColor(img, cv2.COLOR_GRAY2BGR)
                cv2.putText(img, text, org, font, font_scale, color, thickness, cv2.LINE_AA)
                imgs[i] = cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)[..., np.newaxis] if imgs[i].shape[2] == 1 else img
            return imgs
        if self.video_degrade == "dirtyshotcut":
            if t < 3:
                return imgs
            cut_idx = np.random.randint(1, t - 1)
            flip_first = np.random.rand() > 0.5
            if flip_first:
                imgs[:cut_idx] = [img[:, ::-1] for img in imgs[:cut_idx]]          # cv2.flip(img, 1): mirror the columns
            else:
                imgs[cut_idx:] = [img[:, ::-1] for img in imgs[cut_idx:]]
            return imgs[cut_idx:] + imgs[:cut_idx]
        if self.video_degrade in ("hdr", "ldr"):
            scale = np.random.uniform(1, 3) if self.video_degrade == "hdr" else np.random.uniform(0.3, 1)
            for i in range(t):
                imgs[i] = np.clip((imgs[i] - 127.5) * scale + 127.5, 0, 255).astype(np.uint8)
            return imgs
        raise NotImplementedError("Video degrade type not supported.")

    def read_video(self, video_path, start_frame, end_frame, crop_size_before_resize, min_i, min_j, flip, sample_idx=None):
        """Decode + crop + resize + flip (+ shake) like v2v_datasets.py:145-225; returns a list of [h,w,C] uint8."""
        n = end_frame - start_frame
        all_di, all_dj = [0] * n, [0] * n
        if self.shake_frames > 0:                       # shake ends at speed 0 (:148-161): drawn back to front
            vi = vj = di = dj = 0
            for i in range(min(self.shake_frames, n) - 1, -1, -1):
                vi += int(np.random.normal(0, self.shake_std))
                vj += int(np.random.normal(0, self.shake_std))
                di += vi
                dj += vj
                all_di[i], all_dj[i] = di, dj
        need_h = self.crop_size + max(all_di) - min(all_di)
        need_w = self.crop_size + max(all_dj) - min(all_dj)
        if self.frame_source is not None:
            imgs = self.frame_source(self, sample_idx, start_frame, end_frame, crop_size_before_resize, min_i, min_j, flip,
                                     need_h, need_w)
        else:
            assert self.video_reader == "opencv", "FFMPEG hasn't been updated to support color."   # :168
            import cv2
            cap = cv2.VideoCapture(video_path)
            cap.set(cv2.CAP_PROP_POS_FRAMES, start_frame)
            imgs = []
            for _ in range(start_frame, end_frame):
                ok, frame = cap.read()
                if not ok:
                    break
                if self.color_mode == "gray":
                    frame = cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY)
                frame = frame[min_i:min_i + crop_size_before_resize, min_j:min_j + crop_size_before_resize, ...]
                frame = cv2.resize(frame, (need_w, need_h), interpolation=cv2.INTER_LINEAR)
                if flip:
                    frame = cv2.flip(frame, 1)
                if self.color_mode == "gray":
                    frame = np.expand_dims(frame, axis=-1)
                imgs.append(frame)
            cap.release()
        off_i = np.array(all_di) - min(all_di)
        off_j = np.array(all_dj) - min(all_dj)
        if isinstance(imgs, np.ndarray) and imgs.ndim == 4 and self.shake_frames <= 0:
            # a frame source that hands out ONE [T,h,w,C] array (pre-decoded stores): no shake means one crop for all frames -- a view,
            # so that host_sample_into moves the clip with one copy instead of one per frame
            return imgs[:, :self.crop_size, :self.crop_size, :]
        return [img[off_i[i]:off_i[i] + self.crop_size, off_j[i]:off_j[i] + self.crop_size, :] for i, img in enumerate(imgs)]

    def read_video_gpu(self, video_path, start_frame, end_frame, crop_size_before_resize, min_i, min_j, flip, img_idxes,
                       sample_idx=None):
        """GPU form of read_video + the pause-index gather (:145-225, :311-316): returns device tensors
        (all_imgs [N,crop,crop,C] uint8, gray [N,crop,crop] uint8).  Same np.random draw order as read_video."""
        from . import frontend
        n = end_frame - start_frame
        all_di, all_dj = [0] * n, [0] * n
        if self.shake_frames > 0:
            vi = vj = di = dj = 0
            for i in range(min(self.shake_frames, n) - 1, -1, -1):
                vi += int(np.random.normal(0, self.shake_std))
                vj += int(np.random.normal(0, self.shake_std))
                di += vi
                dj += vj
                all_di[i], all_dj[i] = di, dj
        if self.raw_frame_source is not None:
            raw = np.asarray(self.raw_frame_source(self, sample_idx, start_frame, end_frame))
        else:
            assert self.video_reader == "opencv", "FFMPEG hasn't been updated to support color."
            import cv2
            cap = cv2.VideoCapture(video_path)
            cap.set(cv2.CAP_PROP_POS_FRAMES, start_frame)
            frames = []
            for _ in range(start_frame, end_frame):
                ok, frame = cap.read()
                if not ok:
                    break
                frames.append(frame)
            cap.release()
            raw = np.stack(frames)
        cb = crop_size_before_resize
        # only the crop rectangle travels over PCIe (cb x cb x 3 per frame instead of the whole frame)
        rect = np.ascontiguousarray(raw[:, min_i:min_i + cb, min_j:min_j + cb, :])
        rect_d = torch.from_numpy(rect).to(self.sim_device)
        shake = self.shake_frames > 0
        return frontend.prepare_clip(rect_d, cb, 0, 0, flip, self.crop_size, img_idxes, all_di if shake else None,
                                     all_dj if shake else None, self.color_mode, want_imgs=self.color_mode != "gray")

    # ------------------------------------------------------------------ the hot path
    def imgs_to_voxels(self, imgs, num_bins, frames_per_bin, FPS, pos_thres=None, neg_thres=None, *, clip_id: int = 0):
        """[N,H,W] uint8 -> (v2e_params dict, [L,num_bins,H,W] voxels).  v2v_datasets.py:363-410.
        clip_id (this implementation only): second half of the device RNG key; __getitem__ passes the sample index.

        Returns a float64 ndarray like the reference when `imgs` is a NumPy array, a float32 CUDA tensor when it is
        a CUDA tensor.  Same AssertionError when (N-1) % (num_bins*frames_per_bin) != 0."""
        n = imgs.shape[0]
        assert (n - 1) % (num_bins * frames_per_bin) == 0
        params = sample_sim_params(self.threshold_range, self.max_thres_pos_neg_gap, self.base_noise_std_range,
                                   self.hot_pixel_fraction_range, self.hot_pixel_std_range, self.use_fixed_thresholds,
                                   pos_thres, neg_thres, self.scale_noise_strength, self.put_noise_external)
        plist = [params[k] for k in ("pos_thres", "neg_thres", "base_noise_std", "hot_pixel_fraction", "hot_pixel_std")]
        is_np = isinstance(imgs, np.ndarray)
        frames = torch.from_numpy(np.ascontiguousarray(imgs)).to(self.sim_device) if is_np else imgs
        if frames.dtype not in (torch.uint8, torch.float32):
            frames = frames.to(torch.float32)
        kw = dict(bin_mode="sum", num_bins=num_bins, frames_per_bin=frames_per_bin,
                  put_noise_external=self.put_noise_external, out_dtype=torch.float64 if is_np else torch.float32)
        if self.sim_rng == "numpy":
            fields = esim.draw_numpy_replay_fields(n, imgs.shape[1], imgs.shape[2])
            vox = esim.esim_voxel_batch(frames[None], plist, rng_mode="replay",
                                        replay=[torch.from_numpy(f)[None] for f in fields], **kw)[0]
        else:
            seed = draw_sim_seed()                               # worker seeding / fixed_seed still govern the noise
            vox = esim.esim_voxel_batch(frames[None], plist, rng_mode=self.sim_rng, seed=seed, clip_id0=int(clip_id), **kw)[0]
        return params, (vox.cpu().numpy() if is_np else vox)

    # ------------------------------------------------------------------ sample assembly
    def _draw_geometry(self, sample_idx):
        """The per-sample draws in front of the decode, in the reference's order (:260-301): resize scale, crop origin, flip, pause
        chain.  Returns (video_path, start_frame, end_frame, crop_before, min_i, min_j, flip, img_idxes, img_cnt)."""
        video_name = self.sample_video_name[sample_idx]
        start_frame = int(self.sample_begin_idx[sample_idx])
        img_cnt = int(self.sample_L[sample_idx])
        video_path = os.path.join(self.dataset_path, video_name)
        vid_width, vid_height = self._probe_size(video_path)
        if self.crop_size is None:
            raise NotImplementedError("crop_size must be provided for WebvidDataset.")
        min_scale = max(self.min_resize_scale, self.crop_size / int(vid_height * self.keep_top_percentile),
                        self.crop_size / vid_width)
        max_scale = max(self.max_resize_scale, min_scale)
        resize_scale = np.random.uniform(min_scale, max_scale)                                         # :272
        crop_before = int(self.crop_size / resize_scale)
        if self.fixed_crop:
            min_i = min_j = 0
        else:
            min_i = np.random.randint(0, int(vid_height * self.keep_top_percentile) - crop_before + 1)  # :279
            min_j = np.random.randint(0, vid_width - crop_before + 1)                                   # :280
        flip = bool(self.random_flip and np.random.rand() > 0.5)                                        # :284

        # pause schedule: a two-state Markov chain deciding which decoded frame each simulator frame shows (:286-301)
        extra = self.frames_per_img if self.output_additional_evs else 0
        # Every step consumes exactly ONE uniform (a paused chain tests `rand() > proba_pause_when_paused`, a running one
        # `rand() < proba_pause_when_running`; the other test short-circuits before its draw), and legacy np.random.rand(n) IS the
        # next n scalar draws -- so the chain's uniforms are drawn in one call (tests/test_host_logic.py checks chain and stream state)
        n_steps = img_cnt * self.frames_per_img + 1 + extra
        img_idxes, idx, paused = [], 0, False
        p_stay, p_pause = self.proba_pause_when_paused, self.proba_pause_when_running
        for u in np.random.rand(n_steps).tolist():
            img_idxes.append(idx)
            if paused:
                if u > p_stay:
                    paused = False
            elif u < p_pause:
                paused = True
            if not paused:
                idx += 1
        return video_path, start_frame, start_frame + idx + 1, crop_before, min_i, min_j, flip, img_idxes, img_cnt

    def frame_pick(self, img_cnt=None):
        """Indices (into the simulator's N frames) of the frames a sample hands out as `frame` (:325-338)."""
        img_cnt = self.L if img_cnt is None else img_cnt
        off = self.frames_per_img if self.output_additional_evs else 0                                  # :325-326
        if not self.output_additional_frame:
            return [off + (i + 1) * self.frames_per_img for i in range(img_cnt)]                       # :329-333
        return [off + i * self.frames_per_img for i in range(img_cnt + 1)]                              # :334-338

    def host_sample_into(self, sample_idx, clip_out, params_out, key_out, frames_out=None, index_out=None):
        """The HOST half of a sample written straight into caller-owned arrays (v2v_amd.loader.RingLoader: slots of page-locked
        shared memory): same np.random draw order as __getitem__ with `defer_sim: true`, no intermediate stack, no float frames.
            clip_out   uint8 [N,H,W]    the simulator's frames (pause-index gather applied; gray)
            params_out float64 [5]      pos_thres, neg_thres, base_noise_std, hot_pixel_fraction, hot_pixel_std
            key_out    int64 [2]        {seed drawn from np.random, sample index}: the device RNG key of this sample
            frames_out uint8 [Lf,H,W,3] only for color_mode 'gray_in_bgr_out': the colour frames handed out as `frame`
            index_out  int32 [N] or None.  Given: PACKED form -- clip_out[:U] receives the U DECODED frames once each (the video
                       pauses, data/v2v_datasets.py:286-301, so U <= N) and index_out[f] the stored frame simulator frame f shows; the
                       simulator gathers through it (v2v_esim_extras.frame_index) and the paused frames never cross PCIe twice
        Returns the v2e_params dict (packed form: (v2e_params, U))."""
        if self.gpu_frontend:
            raise NotImplementedError("host_sample_into serves host-decoded clips; gpu_frontend samples are assembled on the device")
        old_state = None
        if self.fixed_seed is not None:
            old_state = np.random.get_state()
            np.random.seed(self.fixed_seed + int(sample_idx))
        video_path, start_frame, end_frame, crop_before, min_i, min_j, flip, img_idxes, img_cnt = self._draw_geometry(sample_idx)
        raw_imgs = self.read_video(video_path, start_frame, end_frame, crop_before, min_i, min_j, flip, sample_idx)
        if self.video_degrade is not None and np.random.rand() < self.degrade_ratio:                    # :307-308
            raw_imgs = self.degrade_video(list(raw_imgs))
        n = len(img_idxes)
        assert (n - 1) % (self.num_bins * self.frames_per_bin) == 0                                      # :365
        if tuple(clip_out.shape) != (n,) + tuple(raw_imgs[0].shape[:2]):
            raise ValueError(f"clip_out is {tuple(clip_out.shape)}, the clip is {(n,) + tuple(raw_imgs[0].shape[:2])}")
        n_stored = img_idxes[-1] + 1                                     # the chain moves by 0 or 1: every decoded frame up to the last is shown
        if index_out is not None:
            index_out[:] = img_idxes
            if self.color_mode == "gray":
                if isinstance(raw_imgs, np.ndarray):
                    clip_out[:n_stored] = raw_imgs[:n_stored, ..., 0]
                else:
                    for i in range(n_stored):
                        clip_out[i] = raw_imgs[i][..., 0]
            else:
                stored = np.stack(raw_imgs[:n_stored]) if not isinstance(raw_imgs, np.ndarray) else raw_imgs[:n_stored]
                clip_out[:n_stored] = bgr_to_gray(stored)
                frames_out[:] = stored[[img_idxes[p] for p in self.frame_pick(img_cnt)]]
        elif self.color_mode == "gray":
            for j, i in enumerate(img_idxes):
                clip_out[j] = raw_imgs[i][..., 0]
        else:
            all_imgs = np.stack([raw_imgs[i] for i in img_idxes])
            clip_out[:] = bgr_to_gray(all_imgs)
            frames_out[:] = all_imgs[self.frame_pick(img_cnt)]
        pos = self.sample_pos_thres[sample_idx] if self.use_fixed_thresholds else None
        neg = self.sample_neg_thres[sample_idx] if self.use_fixed_thresholds else None
        v2e_params = sample_sim_params(self.threshold_range, self.max_thres_pos_neg_gap, self.base_noise_std_range,
                                       self.hot_pixel_fraction_range, self.hot_pixel_std_range, self.use_fixed_thresholds,
                                       pos, neg, self.scale_noise_strength, self.put_noise_external)
        params_out[:] = [v2e_params[k] for k in ("pos_thres", "neg_thres", "base_noise_std", "hot_pixel_fraction", "hot_pixel_std")]
        key_out[0] = draw_sim_seed()                                  # same draw order as imgs_to_voxels(sim_rng='philox')
        key_out[1] = int(sample_idx)
        if old_state is not None:
            np.random.set_state(old_state)
        return v2e_params if index_out is None else (v2e_params, n_stored)

    # ------------------------------------------------------------------ the per-sample path, staged (zero-edit integration level)
    def _staged_ok(self):
        """The per-sample launch can take the loader's route -- decoded frames written ONCE into a page-locked slot, one asynchronous H2D
        copy, the simulator gathering through the pause index, `frame` built on the device -- whenever the ring loader could serve the
        dataset: device-native noise inside the potential, host-decoded clips, a GPU to simulate on."""
        return (self.staged_getitem and not self.defer_sim and not self.gpu_frontend and self.sim_rng != "numpy" and not self.put_noise_external
                and torch.device(self.sim_device).type == "cuda" and self.crop_size is not None)

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_staging", None)                 # page-locked slots and device buffers belong to the process that made them
        return state

    def _getitem_staged(self, sample_idx):
        """__getitem__ through three rotating page-locked slots (v2v_amd.loader._SlotLayout with a batch of one): same np.random draws,
        same tensors as the plain path below (tests/test_hip_dataset_events.py), without its host-side gather (np.stack over the pause
        index: 201 frames), its pageable synchronous copies and its float `frame` tensor built on the host.  Round 5, measured at the
        training shape under train.py's own DataLoader(num_workers=0): tools/loader_bench.py `yaml_only_workers0`."""
        from . import _lib
        from .loader import _SlotLayout, clip_frames_f32, clip_frames_packed
        _lib.require_gpu()                                                  # no CPU simulator fallback: fail as loudly as the plain path
        dev = torch.device(self.sim_device)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        n, hw = self.frames_per_seq + 1, self.crop_size
        st = self.__dict__.get("_staging")
        if st is None or st["dev"] != dev:
            pick = self.frame_pick()
            lay = _SlotLayout(1, n, hw, hw, len(pick), self.color_mode != "gray")
            slots = 3
            host = torch.empty((slots, lay.nbytes), dtype=torch.uint8).pin_memory()
            st = self._staging = {"dev": dev, "lay": lay, "pick": np.asarray(pick, dtype=np.int64), "host": host, "host_np": host.numpy(),
                                  "dbuf": torch.empty((slots, lay.nbytes), dtype=torch.uint8, device=dev), "k": 0,
                                  "copied": [torch.cuda.Event() for _ in range(slots)], "read": [torch.cuda.Event() for _ in range(slots)]}
        lay, slot = st["lay"], st["k"] % 3
        st["k"] += 1
        st["copied"][slot].synchronize()                                   # the H2D copy that last read this page-locked slot has left it
        offsets, fidx, pick, params, keys, used, cframes, clips, stored = lay.views(st["host_np"][slot])
        v2e_params, n_stored = self.host_sample_into(sample_idx, clips[:n * hw * hw].reshape(n, hw, hw), params[0], keys[0],
                                                     cframes[0] if cframes is not None else None, fidx[0])
        offsets[0], stored[0] = 0, n_stored
        pick[0] = fidx[0][st["pick"]]
        nbytes = lay.off_clips + (n_stored * hw * hw + 15) // 16 * 16
        with torch.cuda.device(dev):
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(st["read"][slot])                               # the launches that read this device slot three samples ago
            dbuf = st["dbuf"][slot]
            dbuf[:nbytes].copy_(st["host"][slot, :nbytes], non_blocking=True)
            st["copied"][slot].record(cur)
            offsets_d, fidx_d, pick_d, params_d, keys_d, cframes_d, clips_d, stored_d = lay.device_views(dbuf)
            vox = esim.esim_voxel_packed(clips_d, offsets_d, fidx_d, hw, hw, params_d, keys_d, num_bins=self.num_bins,
                                         frames_per_bin=self.frames_per_bin, stored_frames=stored_d)[0]          # [L(+1),Tb,H,W] f32
            frame = (clip_frames_packed(clips_d, offsets_d, pick_d, hw, hw, stored_frames=stored_d) if cframes_d is None else clip_frames_f32(cframes_d))[0]
            st["read"][slot].record(cur)
        out_dev = torch.device(self.output_device)
        return {"frame": frame.to(out_dev), "events": vox.to(out_dev), "data_source_idx": torch.tensor(self.data_source_idx), "v2e_params": v2e_params}

    def __getitem__(self, sample_idx):
        if not self.defer_sim and not _worker_exit_hook_installed:
            _install_worker_exit_hook(self)                                 # a DataLoader worker that simulates: clean teardown (see _worker_exit)
        if self._staged_ok():
            return self._getitem_staged(sample_idx)
        old_state = None
        if self.fixed_seed is not None:
            # the reference reads an unbound `idx` here (UnboundLocalError); the intended key is the sample index
            old_state = np.random.get_state()
            np.random.seed(self.fixed_seed + int(sample_idx))
        video_path, start_frame, end_frame, crop_before, min_i, min_j, flip, img_idxes, img_cnt = self._draw_geometry(sample_idx)
        if self.gpu_frontend:
            imgs_d, gray = self.read_video_gpu(video_path, start_frame, end_frame, crop_before, min_i, min_j, flip, img_idxes,
                                               sample_idx)
            all_imgs = gray.unsqueeze(-1) if imgs_d is None else imgs_d                                  # device [N,H,W,C] uint8
        else:
            raw_imgs = self.read_video(video_path, start_frame, end_frame, crop_before, min_i, min_j, flip, sample_idx)
        if self.video_degrade is not None and np.random.rand() < self.degrade_ratio:                    # v2v_datasets.py:307-308
            if self.gpu_frontend:
                raise NotImplementedError("video_degrade acts on the decoded host frames: use gpu_frontend: false with it")
            raw_imgs = self.degrade_video(list(raw_imgs))
        if not self.gpu_frontend:
            all_imgs = np.stack([raw_imgs[i] for i in img_idxes])                                       # [N,H,W,C] uint8
            gray = all_imgs[..., 0] if self.color_mode == "gray" else bgr_to_gray(all_imgs)

        pos = self.sample_pos_thres[sample_idx] if self.use_fixed_thresholds else None
        neg = self.sample_neg_thres[sample_idx] if self.use_fixed_thresholds else None
        if self.defer_sim:
            assert (gray.shape[0] - 1) % (self.num_bins * self.frames_per_bin) == 0
            v2e_params = sample_sim_params(self.threshold_range, self.max_thres_pos_neg_gap, self.base_noise_std_range,
                                           self.hot_pixel_fraction_range, self.hot_pixel_std_range, self.use_fixed_thresholds,
                                           pos, neg, self.scale_noise_strength, self.put_noise_external)
            sim_seed = draw_sim_seed()                            # same draw order as imgs_to_voxels(sim_rng='philox')
            voxels = None
        else:
            dev = torch.device(self.sim_device)
            gray_d = gray if isinstance(gray, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(gray)).to(dev)
            v2e_params, voxels = self.imgs_to_voxels(gray_d, self.num_bins, self.frames_per_bin, 24, pos, neg, clip_id=sample_idx)   # [L(+1),Tb,H,W] f32
        if self.output_additional_evs:
            all_imgs = all_imgs[self.frames_per_img:]
        if not self.output_additional_frame:
            pick = [(i + 1) * self.frames_per_img for i in range(img_cnt)]                              # :329-333
        else:
            pick = [i * self.frames_per_img for i in range(img_cnt + 1)]                                # :334-338
        if isinstance(all_imgs, torch.Tensor):
            # device frames: v / 255 in the library's own kernel (IEEE float32 division = the reference's CPU values bit for bit)
            from .loader import clip_frames_f32
            frames = clip_frames_f32(all_imgs.unsqueeze(0), pick)[0]                                    # [L,C,H,W] in [0,1]
        else:
            frames = torch.from_numpy(all_imgs[pick]).to(torch.float32).permute(0, 3, 1, 2) / 255
        n_ev = img_cnt + 1 if self.output_additional_evs else img_cnt
        out_dev = torch.device(self.output_device)
        if self.defer_sim:
            keys = ("pos_thres", "neg_thres", "base_noise_std", "hot_pixel_fraction", "hot_pixel_std")
            sequence = {
                "frame": frames.contiguous(),
                "sim_frames": torch.from_numpy(np.ascontiguousarray(gray)),                            # uint8 [N,H,W]
                "sim_params": torch.tensor([v2e_params[k] for k in keys], dtype=torch.float64),
                "sim_key": torch.tensor([sim_seed, int(sample_idx)], dtype=torch.int64),
                "data_source_idx": torch.tensor(self.data_source_idx),
                "v2e_params": v2e_params,
            }
        else:
            sequence = {
                "frame": frames.to(out_dev).contiguous(),
                "events": voxels[:n_ev].to(out_dev).contiguous(),
                "data_source_idx": torch.tensor(self.data_source_idx),
                "v2e_params": v2e_params,
            }
        if old_state is not None:
            np.random.set_state(old_state)
        return sequence


class SimulatingCollator:
    """`collate_fn` that replaces default_collate + per-sample simulation with ONE fused launch per batch
    (the north_star's "V2VDataset collate").  Use with a dataset configured `defer_sim: true`:

        loader = DataLoader(dataset, batch_size=B, num_workers=9, collate_fn=SimulatingCollator.from_configs(cfg))

    Workers decode/crop and draw the parameters; this runs in the main process: stack the uint8 clips, one pinned
    H2D copy, one v2v_esim_voxel_keyed_hip launch with per-clip {seed, clip id} keys.  Every sample gets exactly the
    voxels the per-sample path (`defer_sim: false`, sim_rng 'philox') produces for the same draws."""

    def __init__(self, num_bins=5, frames_per_bin=1, put_noise_external=False, output_additional_evs=False,
                 device="cuda", output_device=None, rng_mode="philox", pad_to=1, normalize=False, stager=None):
        self.num_bins, self.frames_per_bin = num_bins, frames_per_bin
        self.put_noise_external = put_noise_external
        self.output_additional_evs = output_additional_evs
        self.device = torch.device(device)
        self.output_device = torch.device(output_device) if output_device is not None else self.device
        self.rng_mode = rng_mode
        # consumer-side post-ops done where the voxels are produced (model/train_utils.py:147-166, 322-326): the simulator writes
        # into the x`pad_to`-padded layout, and `normalize` applies normalize_batch_voxel in place (exact counting select on the
        # integer SUM-mode grids; radix select when the noise is external).  Then run the model with normalize_voxels: false.
        self.pad_to, self.normalize = int(pad_to), bool(normalize)
        self._stager = stager                     # a shared v2v_amd.staging.HostStager; one is created on first use otherwise
        self.timers = None                        # optional dict: host seconds per stage (tools/loader_bench.py)

    @property
    def stager(self):
        if self._stager is None:
            from .staging import HostStager
            self._stager = HostStager(self.device)
        return self._stager

    @classmethod
    def from_configs(cls, configs, **kw):
        args = dict(num_bins=configs.get("num_bins", 5), frames_per_bin=configs.get("frames_per_bin", 1),
                    put_noise_external=configs.get("put_noise_external", False),
                    output_additional_evs=configs.get("output_additional_evs", False),
                    device=configs.get("sim_device", "cuda"), output_device=configs.get("output_device", None))
        args.update(kw)
        return cls(**args)

    def __call__(self, samples):
        """collate_fn form.  NOTE: a DataLoader runs collate_fn INSIDE its workers, so use this only with
        num_workers=0 or spawn-started workers; with fork()ed workers wrap the loader in SimulatingLoader instead."""
        from torch.utils.data import default_collate
        return self.simulate(default_collate(samples))

    def _t(self, key, t0):
        if self.timers is not None:
            import time
            self.timers[key] = self.timers.get(key, 0.0) + (time.perf_counter() - t0)
        import time as _time
        return _time.perf_counter()

    def simulate(self, batch):
        """batch = default-collated deferred samples (sim_frames [B,N,H,W] uint8, sim_params [B,5], sim_key [B,2])."""
        import time
        t0 = time.perf_counter()
        batch = dict(batch)
        clips, params, keys = batch.pop("sim_frames"), batch.pop("sim_params"), batch.pop("sim_key")
        staged = batch.pop("_staged_clips", None)     # SimulatingLoader started this batch's H2D copy one batch ahead
        if staged is not None:
            clips = self.stager.ready(staged)
        elif self.device.type == "cuda" and not clips.is_cuda:
            clips = self.stager.ready(self.stager.stage(clips))   # page-locked double buffers + copy stream (v2v_amd/staging.py)
        pa = params.cpu().numpy()
        no_noise = bool((pa[:, 2] == 0).all() and (pa[:, 3] <= 0).all())
        t0 = self._t("stage_ready", t0)
        # normalize: the k-th values come from statistics the simulator's writer keeps (exact counting over the integers -255..255 +
        # overflow words for the rare hot pixels) when the batch's own parameters bound the normal pixels' counts, else from a radix
        # select over the finished tensor (v2v_amd/loader.py:choose_normalize_method); then ONE scaling pass in place
        from . import _lib
        from .loader import choose_normalize_method
        method = choose_normalize_method(pa, self.frames_per_bin, self.put_noise_external) if self.normalize else None
        stats = torch.empty((clips.shape[0], _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device=self.device) if method == "count" and self.rng_mode != "replay" else None
        vox = esim.esim_voxel_batch(clips, params.to(self.device), bin_mode="sum", num_bins=self.num_bins,
                                    frames_per_bin=self.frames_per_bin, rng_mode=self.rng_mode, clip_keys=keys,
                                    put_noise_external=self.put_noise_external, no_noise=no_noise, pad_to=self.pad_to, stats=stats)   # [B,L(+1),Tb,Hp,Wp]
        t0 = self._t("sim", t0)
        if self.normalize:
            from . import postops
            h, w = clips.shape[-2:]
            if stats is not None:
                scales = postops.scales_from_stats(stats, vox.shape[1] * vox.shape[2] * h * w)
            else:
                scales = postops.voxel_scales_radix(vox[..., :h, :w])
            vox = postops.apply_scales(vox, scales, self.pad_to, valid_hw=(h, w), inplace=True)
        t0 = self._t("postops", t0)
        batch["events"] = vox.to(self.output_device)
        batch["frame"] = batch["frame"].to(self.output_device)
        self._t("to_output_device", t0)
        return batch


class SimulatingLoader:
    """Wraps a DataLoader over a `defer_sim: true` dataset: fork()ed workers decode, crop and default-collate raw
    uint8 clips; the simulation of each batch (ONE fused launch) happens here, in the process that owns the GPU.

        loader = SimulatingLoader(DataLoader(dataset, batch_size=B, num_workers=9), SimulatingCollator.from_configs(cfg))
        for batch in loader: ...        # same dict as the reference's loader: frame, events, data_source_idx, v2e_params
    """

    def __init__(self, loader, collator: SimulatingCollator):
        self.loader, self.collator = loader, collator

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        # one batch of look-ahead: batch k+1's clips cross PCIe (page-locked double buffers, copy stream) while batch k is
        # simulated; the simulator's stream only waits on the copy's event
        it = iter(self.loader)
        def stage(raw):
            if self.collator.device.type != "cuda":
                return raw
            import time
            t0 = time.perf_counter()
            out = dict(raw, _staged_clips=self.collator.stager.stage(raw["sim_frames"]))
            self.collator._t("stage_copy", t0)
            return out
        try:
            nxt = stage(next(it))
        except StopIteration:
            return
        for raw in it:
            cur, nxt = nxt, stage(raw)
            yield self.collator.simulate(cur)
        yield self.collator.simulate(nxt)
