#!/usr/bin/env python3
"""bench.py -- headline benchmark of the fused sim+voxel hot path (BASELINE.json metric:
"voxel grids/sec (B x Tbins x H x W) at 1/2/4/8 GPU; achieved HBM GB/s vs peak").

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (ONE launch of v2v_esim_voxel_hip) over one batch of synthetic clips that is
already resident in HBM.  Workload at N=1: BASELINE configs[1] -- 256 clips of 32x256x256 float32 (integer-valued),
C+ = C- = 0.2, 5 temporal-bilinear voxel bins, simulated with the REFERENCE'S OWN EventEmulator constructor defaults
(data/v2v_core_esim.py:8-16: base_noise_std 0.1, hot_pixel_fraction 1e-3, hot_pixel_std 0.1), i.e. with the noise the
reference always applies.  With N>1 GPUs every rank gets its own 256 clips (global clip ids rank*256..), no data-path
collective: weak scaling.  Rank 0 prints ONE JSON line, last and alone on stdout, at most 4 KB: the contract keys only.  The
default run also times one workload of every other BASELINE config (`also_measured`: own kernel time, algorithmic bytes,
fraction of the HBM peak) and the loader at the training shape, outside the timed region and after the headline; those,
the launch trace and the secondary CPU baselines go to the sidecar file the line names (`extra`: bench_extra.json).
`--full` runs every variant.  They are parity-test cases and secondary measurements, never the headline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Streaming ceilings measured on this pool with tools/ubench/stream_pattern.hip (profiles/stream_pattern_*.txt): context
# for roofline.frac, which is always quoted against the 8 TB/s datasheet peak.
MEASURED_CEILINGS_GBPS = {"read_only_sweep": 6100.0, "float4_copy": 5200.0,
                          "cfg2_access_pattern_without_arithmetic": 6170.0}
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (measured copy ceiling ~6290)

REF_DEFAULTS = [0.2, 0.2, 0.1, 0.001, 0.1]        # EventEmulator() defaults (v2v_core_esim.py:8-16): what "fixed C = 0.2" runs
DATASET_STYLE = [0.2, 0.3, 0.05, 5e-4, 1.0]       # asymmetric thresholds + noise as imgs_to_voxels samples them (v2v_datasets.py:368-386)
NOISE_FREE = [0.2, 0.2, 0.0, 0.0, 0.0]            # the noise-free symmetric fast path (V2V_FLAG_NO_NOISE kernels): NOT reachable from the dataset
V2E_NOISY = [24, "pn_related", 0.5, 0.1, 0.0, 0.1, 30, 0.1, 0, 5.0, 0.1, 0.1]     # SURVEY §8d S3 (v2v_core_v2e.py:365-375,600)
_CFG2 = dict(model="esim", b=256, n=32, h=256, w=256, dtype="float32", bin="bilinear", tb=5, fpb=1)
WORKLOADS = {
    "cfg2_esim_f32_256x32x256x256_bilinear5": dict(_CFG2, params=REF_DEFAULTS),
    "cfg2_noise_free": dict(_CFG2, params=NOISE_FREE),
    "cfg2_dataset_style": dict(_CFG2, params=DATASET_STYLE),
    "cfg2_u8": dict(_CFG2, dtype="uint8", params=REF_DEFAULTS),
    "cfg2_u8_noise_free": dict(_CFG2, dtype="uint8", params=NOISE_FREE),
    "cfg3_v2e_f32_256x32x256x256_bilinear5": dict(_CFG2, model="v2e", params=V2E_NOISY),
    "cfg3_v2e_u8": dict(_CFG2, model="v2e", dtype="uint8", params=V2E_NOISY),
    # the run-time-feature instance of the v2e kernel (per-frame thresholds: no specialised instance): what the rare configurations run
    "cfg3_v2e_f32_per_frame_thresholds": dict(_CFG2, model="v2e", params=[V2E_NOISY[0], "spatial_temporal_independent"] + V2E_NOISY[2:]),
    "cfg4_u8_256x41x256x256_sum5": dict(model="esim", b=256, n=41, h=256, w=256, dtype="uint8", bin="sum", tb=5, fpb=1, params=DATASET_STYLE),
    # BASELINE config 4 (per GPU): decoded 720p BGR frames resident in HBM -> GPU front-end (cvtColor, crop, resize to
    # 256x256, flip) -> fused sim + sum binning.  41 frames so that (N-1) % 5 == 0 as the reference asserts.
    "cfg4_pipeline_720p_to_256_41f_sum5": dict(model="pipeline", b=24, n=41, h=256, w=256, dtype="uint8", bin="sum", tb=5, fpb=1,
                                               params=DATASET_STYLE, src_hw=(720, 1280)),
    # BASELINE config 4 as BASELINE.json words it: 40-frame clips (39 pairs, not divisible by 5 -> the reference's SUM binning would
    # assert) into 5 temporal-bilinear bins
    "cfg4_pipeline_720p_to_256_40f_bilinear5": dict(model="pipeline", b=24, n=40, h=256, w=256, dtype="uint8", bin="bilinear", tb=5, fpb=1,
                                                    params=DATASET_STYLE, src_hw=(720, 1280)),
    # BASELINE config 5 (per GPU): config 4's pipeline feeding a random-init E2VID-shaped recurrent UNet (bf16 autocast,
    # tools/e2vid_consumer.py) -- end-to-end "dataloader -> model forward" throughput.
    "cfg5_pipeline_plus_e2vid_bf16": dict(model="pipeline", b=8, n=41, h=256, w=256, dtype="uint8", bin="sum", tb=5, fpb=1,
                                          params=DATASET_STYLE, src_hw=(720, 1280), consumer=True),
    # the same with the consumer's ConvLSTM blocks, residual blocks, >=64-channel 5x5 convolutions and upsampling on the device
    # kernels
    "cfg5_fused_convlstm": dict(model="pipeline", b=8, n=41, h=256, w=256, dtype="uint8", bin="sum", tb=5, fpb=1,
                                params=DATASET_STYLE, src_hw=(720, 1280), consumer="fused"),
    # both again with the network in torch.channels_last (NHWC): MIOpen's bf16 convolutions are faster there, and the device
    # kernels consume / produce that layout in place (no layout-change kernels)
    "cfg5_channels_last": dict(model="pipeline", b=8, n=41, h=256, w=256, dtype="uint8", bin="sum", tb=5, fpb=1,
                               params=DATASET_STYLE, src_hw=(720, 1280), consumer="stock_cl"),
    "cfg5_fused_convlstm_channels_last": dict(model="pipeline", b=8, n=41, h=256, w=256, dtype="uint8", bin="sum", tb=5, fpb=1,
                                              params=DATASET_STYLE, src_hw=(720, 1280), consumer="fused_cl"),
    "train_u8_12x201x128x128_sum5": dict(model="esim", b=12, n=201, h=128, w=128, dtype="uint8", bin="sum", tb=5, fpb=1, params=DATASET_STYLE),
    # one training batch's device work as the loader issues it (v2v_amd/loader.py): simulator with the writer's statistics (SURVEY 8f-2
    # "in the writer") -> exact k-th values -> [the one scaling pass, in place]; `scales`: the consumer scales while it reads
    "train_batch_stats_scales_12x201x128x128": dict(model="train_batch", b=12, n=201, h=128, w=128, dtype="uint8", bin="sum", tb=5, fpb=1, params=DATASET_STYLE, apply=False),
    "train_batch_normalised_12x201x128x128": dict(model="train_batch", b=12, n=201, h=128, w=128, dtype="uint8", bin="sum", tb=5, fpb=1, params=DATASET_STYLE, apply=True),
    "cfg1_plumbing_u8_1x8x128x128": dict(model="esim", b=1, n=8, h=128, w=128, dtype="uint8", bin="sum", tb=7, fpb=1, params=NOISE_FREE),
}
DEFAULT_WORKLOAD = "cfg2_esim_f32_256x32x256x256_bilinear5"
ALSO_MEASURED = ["cfg2_noise_free", "cfg2_dataset_style", "cfg2_u8", "cfg3_v2e_f32_256x32x256x256_bilinear5", "cfg3_v2e_u8",
                 "cfg3_v2e_f32_per_frame_thresholds",
                 "cfg4_u8_256x41x256x256_sum5", "cfg4_pipeline_720p_to_256_41f_sum5", "cfg4_pipeline_720p_to_256_40f_bilinear5",
                 "train_u8_12x201x128x128_sum5", "train_batch_stats_scales_12x201x128x128", "train_batch_normalised_12x201x128x128",
                 "cfg5_pipeline_plus_e2vid_bf16", "cfg5_fused_convlstm", "cfg5_channels_last", "cfg5_fused_convlstm_channels_last"]


def host_cores():
    """Cores THIS process may run on: the affinity mask, cut by the cgroup CPU quota when that is smaller -- not os.cpu_count(), which counts
    the machine's (round 5 reported 256 "cores" for a figure nobody had checked against the mask).  Returns (usable, detail string)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    detail = f"affinity mask {n} of {os.cpu_count()} logical CPUs"
    try:
        text = open("/sys/fs/cgroup/cpu.max").read()
    except OSError:
        text = ""
    return apply_cpu_quota(n, detail, text)


def apply_cpu_quota(n, detail, cpu_max_text):
    """cgroup v2 `cpu.max` ("<quota> <period>" in microseconds, or "max <period>") applied to an affinity count (pure: unit-tested)."""
    try:
        quota, period = cpu_max_text.split()
        if quota != "max":
            q = float(quota) / float(period)
            detail += f", cgroup quota {q:.1f} CPUs"
            n = max(1, min(n, int(q)))
    except ValueError:
        pass
    return n, detail


def _simulate_port(clip, wl):
    from oracle import v2v_oracle as O
    if wl["model"] in ("esim", "pipeline", "train_batch"):
        counts = O.esim_video_to_voxel(clip, *wl["params"], put_noise_external=False, rng=O.GlobalNumpyRNG, use_lut=False)
    else:
        counts = O.v2e_video_to_voxel(clip, *wl["params"], seed=None)
    return O.bin_bilinear(counts, wl["tb"]) if wl["bin"] == "bilinear" else O.bin_sum(counts, wl["tb"], wl["fpb"])


def cpu_baseline(frames_host, wl, budget_s=10.0):
    """The oracle (a PORT of the reference's NumPy op sequence, data/v2v_core_esim.py:26-69 or data/v2v_core_v2e.py,
    + the binning) timed on ONE of this box's host cores on a bounded sample of the same clips.  Reported, never the target."""
    import numpy as np
    n_done, t0 = 0, time.perf_counter()
    np.random.seed(0)
    for clip in frames_host:
        _simulate_port(clip, wl)
        n_done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    grids = n_done * (1 if wl["bin"] == "bilinear" else (frames_host.shape[1] - 1) // (wl["tb"] * wl["fpb"]))
    return {"value": grids / dt, "unit": "voxel grids/s", "cores": 1, "kind": "port",
            "sample": f"{n_done} of the batch's clips ({'x'.join(map(str, frames_host.shape[1:]))} {frames_host.dtype}), "
                      f"oracle/v2v_oracle.py NumPy port of the reference's op sequence (float64 state, noise on), single thread, {dt:.1f} s"}


_POOL = {}


def _pool_init(path, wl):
    """Initialiser of cpu_baseline_all_cores' workers (top level: picklable under 'spawn').  The clips are a memory-mapped .npy, not
    pickled jobs: a parent feeding 8 MB clips through pipes would be what is timed."""
    import numpy as np
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    _POOL["clips"], _POOL["wl"] = np.load(path, mmap_mode="r"), wl


def _pool_clip(i):
    import numpy as np
    clips = _POOL["clips"]
    np.random.seed(i)
    _simulate_port(np.array(clips[i % clips.shape[0]]), _POOL["wl"])
    return 1


def cpu_baseline_all_cores(frames_host, wl, budget_s=8.0, distinct_clips=32):
    """The same NumPy port in a process pool over EVERY core this process may use (north_star: "the reference CPU path timed
    on the same box's host cores in the same run, core count stated"): how the reference itself scales -- one single-threaded simulator per
    DataLoader worker.  'spawn' start method (the parent owns a HIP context); worker start-up, imports and one clip per worker run before
    the clock; then rounds of one clip per worker for about `budget_s` seconds."""
    import multiprocessing as mp
    import tempfile
    import numpy as np
    cores, detail = host_cores()
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    wl_small = {k: wl[k] for k in ("model", "params", "tb", "bin", "fpb")}
    with tempfile.TemporaryDirectory(dir=shm) as tmp:
        path = os.path.join(tmp, "clips.npy")
        np.save(path, frames_host[:distinct_clips])
        ctx = mp.get_context("spawn")
        pool = ctx.Pool(cores, initializer=_pool_init, initargs=(path, wl_small))
        try:
            pool.map(_pool_clip, range(cores), chunksize=1)            # imports, page-in, one clip each: outside the timed part
            done, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s:                 # rounds of two clips per worker until the budget is spent
                done += sum(pool.map(_pool_clip, range(done, done + 2 * cores), chunksize=1))
            dt = time.perf_counter() - t0
        finally:
            pool.close()                                               # workers leave through their own exit, not through Pool.terminate()'s SIGTERM
            pool.join()
    grids = done * (1 if wl["bin"] == "bilinear" else (frames_host.shape[1] - 1) // (wl["tb"] * wl["fpb"]))
    return {"value": grids / dt, "unit": "voxel grids/s", "cores": cores, "kind": "port",
            "sample": f"{done} clip simulations ({min(distinct_clips, frames_host.shape[0])} distinct clips of the batch) over {cores} single-threaded worker "
                      f"processes ({detail}), oracle/v2v_oracle.py NumPy port, {dt:.1f} s"}


def cpu_baseline_c(frames_host, wl):
    """Secondary: the scalar C twin (table-driven) with OpenMP over clips on the cores libgomp takes (the affinity mask).  Times the
    library call alone: round 5's 112 grids/s "on 256 cores" was mostly the wrapper's single-threaded NumPy validation of the float32
    content (astype + array_equal + min + max over 0.5 G elements) inside the timed bracket."""
    from oracle import clib, v2v_oracle as O
    clib.lib().oracle_omp_set_threads(host_cores()[0])       # libgomp's default is the machine's core count, not what the cgroup lets this process use
    threads = int(clib.lib().oracle_omp_max_threads())
    bm = clib.BIN_BILINEAR if wl["bin"] == "bilinear" else clib.BIN_SUM
    timing = {}
    t0 = time.perf_counter()
    if wl["model"] in ("esim", "pipeline", "train_batch"):
        clib.esim_voxel(frames_host, wl["params"], O.load_luts(), rng_mode=clib.RNG_PHILOX, seed=1, bin_mode=bm,
                        num_bins=wl["tb"], frames_per_bin=wl["fpb"], check_integer=False, timing=timing)
    else:
        clib.v2e_voxel(frames_host, clib.v2e_params(*wl["params"]), O.load_luts(), seed=1, bin_mode=bm, num_bins=wl["tb"],
                       frames_per_bin=wl["fpb"])
    wall = time.perf_counter() - t0
    dt = timing.get("call_s", wall)
    grids = frames_host.shape[0] * (1 if wl["bin"] == "bilinear" else (frames_host.shape[1] - 1) // (wl["tb"] * wl["fpb"]))
    return {"value": grids / dt, "unit": "voxel grids/s", "cores": threads, "kind": "port",
            "sample": f"{frames_host.shape[0]} clips, oracle/v2v_oracle.c scalar C port (table-driven), OpenMP over clips on {threads} threads "
                      f"({host_cores()[1]}), library call {dt:.2f} s (with the Python wrapper's buffers {wall:.2f} s)"}


class Workload:
    """One benchmark workload resident on `dev`: inputs, the output buffer, the step closure and its byte accounting."""

    def __init__(self, name, dev, rank, world, batch=0):
        import torch
        from v2v_amd import esim, sharding
        wl = WORKLOADS[name]
        self.name, self.wl, self.dev = name, wl, dev
        b = self.b = batch or wl["b"]
        n, h, w, bin_mode, tb, fpb, params = wl["n"], wl["h"], wl["w"], wl["bin"], wl["tb"], wl["fpb"], wl["params"]
        tdtype = getattr(torch, wl["dtype"])
        self.clip_id0 = clip_id0 = sharding.weak_shard(b, rank, world).lo          # batch shard: global clip ids, no exchange
        self.raw = None
        if wl["model"] == "pipeline":
            import numpy as np
            from v2v_amd import frontend
            sh, sw = wl["src_hw"]
            # decoded BGR frames [B,T,720,1280,3] with three DIFFERENT channels (round 3 fed B = G = R, which cannot catch a swapped weight)
            self.raw = raw = torch.empty((b, n, sh, sw, 3), dtype=torch.uint8, device=dev)
            for ch in range(3):
                raw[..., ch] = esim.synth_clips(b, n, sh, sw, dtype=torch.uint8, seed=20240001 + 7919 * ch, clip_id0=clip_id0, device=dev)
            g = np.random.default_rng(20240001 + rank)
            keep_h = int(sh * 0.54)                                                          # keep_top_percentile (v2v_datasets.py:73)
            min_scale = max(0, h / keep_h, h / sw)
            scale = g.uniform(min_scale, max(1.3, min_scale), size=b)                        # :260-272
            cb = (h / scale).astype(np.int64)
            table = np.stack([[g.integers(0, keep_h - c + 1), g.integers(0, sw - c + 1), c, int(g.random() > 0.5)] for c in cb]).astype(np.int32)
            idx = np.tile(np.arange(n, dtype=np.int32), (b, 1))
            table_d, idx_d = torch.as_tensor(table, device=dev), torch.as_tensor(idx, device=dev)
            cb_max = int(cb.max())
            self.frames = frontend.prepare_clips_batch(raw, table_d, idx_d, h, "gray", validate=False, max_crop_before=cb_max)[1]
            src_bytes = int((cb.astype(np.int64) ** 2).sum()) * 3 * n
        else:
            self.frames = esim.synth_clips(b, n, h, w, dtype=tdtype, seed=20240001, clip_id0=clip_id0, device=dev)
        shape = (b, (n - 1) // (tb * fpb), tb, h, w) if bin_mode == "sum" else (b, tb, h, w)
        self.out = out = torch.empty(shape, dtype=torch.float32, device=dev)
        self.alg_bytes = esim.algorithmic_bytes(tdtype, b, n, h, w, bin_mode, tb, fpb)
        self.grids_per_step = b * (shape[1] if bin_mode == "sum" else 1)
        frames = self.frames

        if wl["model"] == "pipeline":
            ptensor = torch.tensor(params, dtype=torch.float64, device=dev)
            self.kernel_name = "frontend_tile_kernel + esim_voxel_kernel"
            self.alg_bytes += src_bytes + b * n * h * w              # source crop regions read once + uint8 clips written once
            consumer = None
            if wl.get("consumer"):
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from e2vid_consumer import E2VIDShapedConsumer, forward_sequence
                torch.manual_seed(0)
                fused, c_last = str(wl["consumer"]).startswith("fused"), str(wl["consumer"]).endswith("_cl")
                if fused and c_last:
                    # the PRODUCT's network: v2v_amd.unet.E2VIDRecurrent (the reference's module tree and state_dict keys, every
                    # layer on the device kernels, bfloat16 NHWC inside); random init, as the reference starts training
                    from v2v_amd.unet import E2VIDRecurrent
                    net = E2VIDRecurrent(dict(num_bins=tb, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3,
                                              base_num_channels=32, num_residual_blocks=2, use_upsample_conv=True,
                                              final_activation="", norm=None)).to(dev).eval()

                    class _Seq:                                              # forward_sequence's protocol
                        reads_any_layout = True

                        def reset_states(self):
                            net.reset_states()

                        def __call__(self, x):
                            return net(x)["image"]

                        def forward_sequence(self, events):                  # the whole time loop in one call, two streams (v2v_amd/unet.py)
                            return net.forward_sequence(events)
                    consumer = _Seq()
                else:
                    consumer = E2VIDShapedConsumer(num_bins=tb, fused_convlstm=fused).to(dev).eval()
                    if c_last:
                        consumer = consumer.to(memory_format=torch.channels_last)
                self.kernel_name += (" + E2VID-shaped UNet forward (bf16 autocast" + (", channels_last" if c_last else "")
                                     + (", ConvLSTM / residual / 5x5 convolutions / upsampling on the device kernels)" if fused else ")"))

            def step():
                gray = frontend.prepare_clips_batch(raw, table_d, idx_d, h, "gray", validate=False, max_crop_before=cb_max)[1]
                esim.esim_voxel_batch(gray, ptensor, bin_mode=bin_mode, num_bins=tb, frames_per_bin=fpb, rng_mode="philox",
                                      seed=20240001, clip_id0=clip_id0, out=out, validate=False, no_noise=False)
                if consumer is not None:
                    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                        forward_sequence(consumer, out, channels_last=c_last)
        elif wl["model"] == "train_batch":
            from v2v_amd import _lib, postops
            ptensor = torch.tensor([params] * b, dtype=torch.float64, device=dev)
            keys = torch.stack([torch.full((b,), 20240001, dtype=torch.int64), torch.arange(b, dtype=torch.int64) + clip_id0], 1).to(dev)
            stats = torch.empty((b, _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device=dev)
            elems = shape[1] * tb * h * w
            self.kernel_name = "esim_voxel_kernel (writer statistics) + count_pick_kernel" + (" + normalize_pad_rows_kernel" if wl["apply"] else "")
            if wl["apply"]:
                self.alg_bytes += 2 * out.numel() * 4                # the scaling pass reads and writes the grid once more
            self.scales = None

            def step():
                esim.esim_voxel_batch(frames, ptensor, bin_mode=bin_mode, num_bins=tb, frames_per_bin=fpb, rng_mode="philox", clip_keys=keys, out=out,
                                      validate=False, no_noise=False, stats=stats)
                self.scales = postops.scales_from_stats(stats, elems)
                if wl["apply"]:
                    postops.apply_scales(out, self.scales, 1, inplace=True)
        elif wl["model"] == "esim":
            ptensor = torch.tensor(params, dtype=torch.float64, device=dev)
            self.kernel_name = "esim_voxel_kernel"
            no_noise = params[2] == 0 and params[3] <= 0
            symmetric = params[0] == params[1]          # the workload's parameters are host constants: what EventEmulator(pos, neg) knows too

            def step():
                esim.esim_voxel_batch(frames, ptensor, bin_mode=bin_mode, num_bins=tb, frames_per_bin=fpb, rng_mode=wl.get("rng", "philox"),
                                      seed=20240001, clip_id0=clip_id0, out=out, validate=False, no_noise=no_noise, symmetric=symmetric)
        else:
            from v2v_amd import v2e
            vparams = v2e.make_params(*params)
            self.kernel_name = "v2e_voxel_kernel (+ v2e_shot_sum_kernel pre-pass)"

            def step():
                v2e.v2e_voxel_batch(frames, vparams, bin_mode=bin_mode, num_bins=tb, frames_per_bin=fpb, rng_mode="philox",
                                    seed=20240001, clip_id0=clip_id0, out=out)
        self.step = step

    def parity(self):
        """Light guard outside the timed region: the first, a middle and the last clip of this rank against the C oracle."""
        import numpy as np
        from oracle import clib, v2v_oracle as O
        wl = self.wl
        bm = clib.BIN_BILINEAR if wl["bin"] == "bilinear" else clib.BIN_SUM
        verdicts = []
        for c in sorted({0, self.b // 2, self.b - 1}):
            host = self.frames[c:c + 1].cpu().numpy()
            if wl["model"] == "train_batch":
                # the oracle's grid of this clip, its k-th values by sorting (model/train_utils.py:153-160), and -- normalised workload --
                # where(v > 0, v / pos_max, v / neg_max); device side: the scales off the writer's statistics and the grid as it stands
                want, _ = clib.esim_voxel(host, wl["params"], O.load_luts(), rng_mode=clib.RNG_PHILOX, seed=20240001, clip_id0=self.clip_id0 + c,
                                          bin_mode=bm, num_bins=wl["tb"], frames_per_bin=wl["fpb"])
                flat = np.sort(want.ravel())
                neg, pos = max(-flat[int(0.01 * flat.size) - 1], 1.0), max(flat[int(0.99 * flat.size) - 1], 1.0)
                ok = np.array_equal(self.scales[c].cpu().numpy(), np.array([neg, pos], dtype=np.float32))
                ref = np.where(want > 0, want.astype(np.float32) / np.float32(pos), want.astype(np.float32) / np.float32(neg)) if wl["apply"] else want
                ok = ok and np.array_equal(self.out[c:c + 1].cpu().numpy(), ref.astype(np.float32))
                verdicts.append("ok" if ok else "MISMATCH")
                continue
            if wl["model"] in ("esim", "pipeline", "train_batch"):
                want, _ = clib.esim_voxel(host, wl["params"], O.load_luts(), rng_mode=clib.RNG_PHILOX, seed=20240001,
                                          clip_id0=self.clip_id0 + c, bin_mode=bm, num_bins=wl["tb"], frames_per_bin=wl["fpb"])
            else:
                want, _ = clib.v2e_voxel(host, clib.v2e_params(*wl["params"]), O.load_luts(), seed=20240001, clip_id0=self.clip_id0 + c,
                                         bin_mode=bm, num_bins=wl["tb"], frames_per_bin=wl["fpb"])
            got = self.out[c:c + 1].cpu().numpy().astype(np.float64)
            verdicts.append("ok" if np.allclose(got, want, rtol=1e-5, atol=1e-5) else "MISMATCH")
        return verdicts[0] if len(set(verdicts)) == 1 else "; ".join(verdicts)

    def free(self):
        self.frames = self.out = self.raw = self.step = None


# ---------------------------------------------------------------------------------------------------------------- host-fed streams
# BASELINE configs 4 and 5 are STREAMS: the clips arrive in host memory and cross PCIe before the kernels see them.  The headline workload
# shards a device-resident batch (trivially N x); these two put the host side -- worker processes, page-locked rings, one PCIe link per
# rank -- inside the timed region, so that a multi-GPU run measures something that can fail to scale (DESIGN.md §7 names what to check).
STREAM_WORKLOADS = {
    # the reference's training shape through the one-line integration level: RingLoader per rank (9 fork()ed workers writing pre-generated
    # 201x128x128 uint8 clips into page-locked shared slots; one H2D copy, simulator + statistics + scales + scaling pass + frames per batch)
    "train_loader_b12": dict(kind="ring", b=12, n=201, h=128, w=128, tb=5, fpb=1, workers=9),
    # config 4 per rank: decoded 720p BGR frames resident in PAGE-LOCKED HOST memory -> HostStager (copy of batch k+1 under the kernels of
    # batch k) -> GPU front-end (cvtColor, crop, resize to 256x256, flip) -> fused simulator, 40 frames -> 5 temporal-bilinear bins
    "cfg4_stream_staged": dict(kind="cfg4", b=8, n=40, h=256, w=256, tb=5, fpb=1, src_hw=(720, 1280), params=DATASET_STYLE, zero_copy=False),
    # the same stream without the staging copy: the front-end kernel stages each clip's CROP RECTANGLE straight out of the page-locked host
    # frames over PCIe (v2v_amd/frontend.py: zero-copy host input) -- only the bytes the resize reads cross the link
    "cfg4_stream": dict(kind="cfg4", b=8, n=40, h=256, w=256, tb=5, fpb=1, src_hw=(720, 1280), params=DATASET_STYLE, zero_copy=True),
}


class StreamWorkload:
    """One rank's host-fed stream.  step() = one batch from host memory to finished voxel grids on the device (asynchronous on the GPU side;
    the caller synchronises at the end of the timed region).  h2d_bytes() = bytes that crossed PCIe per step so far."""

    def __init__(self, name, dev, rank, world, steps_needed):
        import torch
        from v2v_amd import esim
        self.name, self.cfg, self.dev = name, STREAM_WORKLOADS[name], dev
        cfg = self.cfg
        b, n, h, w, tb = cfg["b"], cfg["n"], cfg["h"], cfg["w"], cfg["tb"]
        self.b = b
        self._tmp = None
        if cfg["kind"] == "ring":
            import tempfile
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import loader_bench
            from torch.utils.data import DistributedSampler
            from v2v_amd.loader import RingLoader
            self._tmp = tempfile.TemporaryDirectory()
            n_samples = (steps_needed + 4) * b * world
            ds = loader_bench.make_dataset(self._tmp.name, (n_samples + 1) // 2, loader_bench.PooledFrameSource(), defer_sim=True)   # 2 samples per listed video
            sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=0) if world > 1 else None
            self.loader = RingLoader(ds, batch_size=b, sampler=sampler, shuffle=sampler is None, num_workers=cfg["workers"], drop_last=True, pad_to=16,
                                     normalize=True, device=dev)
            self._it = iter(self.loader)
            self.grids_per_step = b * ((n - 1) // (tb * cfg["fpb"]))
            self.alg_bytes = esim.algorithmic_bytes(torch.uint8, b, n, h, w, "sum", tb, cfg["fpb"])
            self.kernel_name = "esim_voxel_kernel (writer statistics) + count_pick_kernel + normalize_pad_rows_kernel + clip_frames kernel, fed by RingLoader"
            self.last = None

            def step():
                self.last = next(self._it)                                       # events [12,40,5,128,128] + frame on the device, normalised
            self.step = step
            self.gpu_only_ms = lambda: loader_bench.gpu_ms_of_batch(b, dev)
        else:
            import numpy as np
            from v2v_amd import frontend, staging
            sh, sw = cfg["src_hw"]
            clip_id0 = rank * b
            raw = torch.empty((b, n, sh, sw, 3), dtype=torch.uint8, device=dev)
            for ch in range(3):
                raw[..., ch] = esim.synth_clips(b, n, sh, sw, dtype=torch.uint8, seed=20240001 + 7919 * ch, clip_id0=clip_id0, device=dev)
            self.host = raw.cpu().pin_memory()                                   # the "decoded video" of this rank, page-locked
            g = np.random.default_rng(20240001 + rank)
            keep_h = int(sh * 0.54)                                              # keep_top_percentile (v2v_datasets.py:73)
            min_scale = max(0, h / keep_h, h / sw)
            scale = g.uniform(min_scale, max(1.3, min_scale), size=b)            # :260-272
            cb = (h / scale).astype(np.int64)
            table = np.stack([[g.integers(0, keep_h - c + 1), g.integers(0, sw - c + 1), c, int(g.random() > 0.5)] for c in cb]).astype(np.int32)
            table_d = torch.as_tensor(table, device=dev)
            idx_d = torch.as_tensor(np.tile(np.arange(n, dtype=np.int32), (b, 1)), device=dev)
            cb_max = int(cb.max())
            ptensor = torch.tensor(cfg["params"], dtype=torch.float64, device=dev)
            self.out = out = torch.empty((b, tb, h, w), dtype=torch.float32, device=dev)
            stager = staging.HostStager(dev)
            self.raw_bytes = raw.numel()
            self.grids_per_step = b
            self.alg_bytes = esim.algorithmic_bytes(torch.uint8, b, n, h, w, "bilinear", tb, cfg["fpb"]) + int((cb ** 2).sum()) * 3 * n + b * n * h * w
            self.kernel_name = "frontend_tile_kernel + esim_voxel_kernel, fed by HostStager from page-locked host frames"
            self.steps_done = 0

            def compute(raw_d):
                gray = frontend.prepare_clips_batch(raw_d, table_d, idx_d, h, "gray", validate=False, max_crop_before=cb_max)[1]
                esim.esim_voxel_batch(gray, ptensor, bin_mode="bilinear", num_bins=tb, frames_per_bin=cfg["fpb"], rng_mode="philox", seed=20240001,
                                      clip_id0=clip_id0, out=out, validate=False, no_noise=False)
            self.crop_bytes = int((cb ** 2).sum()) * 3 * n
            if cfg["zero_copy"]:
                self.kernel_name = "frontend_tile_kernel (reads the crop rectangles out of page-locked host frames) + esim_voxel_kernel"

                def step():
                    compute(self.host)                                           # no copy: the kernel's staging loads ARE the PCIe transfer
                    self.steps_done += 1
            else:
                self._handle = stager.stage(self.host)

                def step():
                    cur, self._handle = self._handle, stager.stage(self.host)   # batch k+1 crosses PCIe under batch k's kernels
                    compute(stager.ready(cur))
                    self.steps_done += 1
            self.step = step

            def gpu_only_ms():
                for _ in range(2):
                    compute(raw)
                ms = time_launches(lambda: compute(raw), 10, torch)
                return sum(ms) / len(ms)
            self.gpu_only_ms = gpu_only_ms

    def h2d_bytes_per_step(self):
        if self.cfg["kind"] == "ring":
            return self.loader.bytes_copied / max(1, self.loader.batches_copied)
        return float(self.crop_bytes if self.cfg["zero_copy"] else self.raw_bytes)   # zero-copy: algorithmic (the rectangles once; tile halos re-read a little)

    def close(self):
        if self.cfg["kind"] == "ring":
            self._it = self.last = None
            self.loader.close()
        if self._tmp is not None:
            self._tmp.cleanup()


def run_stream(args, torch, dev, rank, local_rank, world, dist, backend):
    """`--workload train_loader_b12 | cfg4_stream`: every rank runs its own host-fed stream; same barrier + max-over-ranks protocol, same line
    (`value` = voxel grids of all ranks per second), plus `stream`: aggregate samples/s, and per rank ms per step and PCIe GB/s."""
    from v2v_amd import sharding
    S = StreamWorkload(args.workload, dev, rank, world, args.warmup + args.steps)
    for _ in range(args.warmup):
        S.step()
    torch.cuda.synchronize()
    sharding.barrier(dist, local_rank)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        S.step()
    torch.cuda.synchronize()
    my_elapsed = time.perf_counter() - t0
    sharding.barrier(dist, local_rank)
    elapsed = sharding.max_over_ranks(dist, my_elapsed, dev)
    per_rank_ms = sharding.gather_floats(dist, my_elapsed / args.steps * 1e3, dev)
    h2d = S.h2d_bytes_per_step()
    per_rank_gbps = sharding.gather_floats(dist, h2d / (my_elapsed / args.steps) / 1e9, dev)
    gpu_ms = S.gpu_only_ms() if rank == 0 else None                        # the same launches on device-resident input: what the kernels alone take
    if rank == 0:
        cfg = S.cfg
        achieved = S.alg_bytes / (gpu_ms * 1e-3) / 1e9
        line = {
            "metric": "voxel grids/sec", "value": S.grids_per_step * world * args.steps / elapsed, "unit": "voxel grids/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "dist_backend": backend, "dist_world_size": world,
            "ms_per_step_per_rank": [round(v, 4) for v in per_rank_ms],
            "config": {"workload": args.workload, "host_fed": True, "clips_per_gpu_per_step": cfg["b"], "frames": cfg["n"], "height": cfg["h"], "width": cfg["w"],
                       "num_bins": cfg["tb"], "bin_mode": "sum" if cfg["kind"] == "ring" else "bilinear", "input_dtype": "uint8",
                       "source": ("pre-generated 201x128x128 uint8 clips, 9 fork()ed workers per rank -> page-locked shared ring (video decode excluded)" if cfg["kind"] == "ring"
                                  else "decoded 1280x720 BGR frames in page-locked host memory; " + ("the front-end kernel reads the crop rectangles over PCIe (zero-copy)" if cfg.get("zero_copy") else "whole frames cross PCIe (HostStager)")),
                       "sharding": f"one stream per rank, {world} rank(s), no collective"},
            "stream": {"samples_per_s": cfg["b"] * world * args.steps / elapsed, "h2d_bytes_per_step_per_rank": h2d,
                       "pcie_GBps_per_rank": [round(v, 2) for v in per_rank_gbps], "gpu_only_ms_per_step": gpu_ms,
                       "gpu_busy_fraction_rank0": gpu_ms / per_rank_ms[0]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": S.kernel_name, "algorithmic_bytes_per_launch": S.alg_bytes, "kernel_ms_avg": gpu_ms,
                         "note": "the step's kernels timed on device-resident input (HIP events); the stream itself is PCIe / host bound, see `stream`"},
            "cpu_baseline": None, "parity_check": None, "extra": None,
        }
        sys.stderr.flush()
        print(line_text(line), flush=True)
    S.close()


def time_launches(step, steps, torch):
    """Per-launch HIP events on the stream the kernels are launched on (torch's current stream)."""
    # ONE event per step boundary (the end of step i is the start of step i + 1): half the event packets between the launches of the
    # two-events-per-step form, whose ~10 us gaps were 2 % of a 0.56 ms step in the kernel trace
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]   # in launch order (callers sort their own copy for percentiles)


def maybe_graph(step, torch, dev, want):
    """Multi-launch steps (front-end + simulator [+ consumer]) are replayed from one captured hipGraph: the C-ABI entry
    points only enqueue kernels on the caller's stream (no allocation, no synchronisation), so they capture as they are."""
    if not want:
        return step, False
    try:
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        for _ in range(2):
            graph.replay()
        torch.cuda.synchronize()
        return graph.replay, True
    except Exception as exc:  # noqa: BLE001 - capture is an optimisation; fall back to eager launches
        print(f"[bench] hipGraph capture unavailable ({type(exc).__name__}: {exc}); eager launches", file=sys.stderr, flush=True)
        return step, False


BF16_DENSE_PEAK_TFLOPS = 2500.0   # MI355X dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def convlstm_roofline(torch, dev):
    """The consumer-side matrix-core kernel at the three encoder levels of the E2VID-shaped
    network for 8 x 256 x 256 input: per-launch HIP-event time, algorithmic FLOPs (2 * pixels * 18C * 4C), fraction of the dense
    bf16 peak, and a parity check of the hidden state against stock PyTorch fp32 on the same bf16-rounded operands."""
    from v2v_amd import convlstm as CL
    out = {"bound": "mfma", "peak_TFLOPs": BF16_DENSE_PEAK_TFLOPS, "kernel": "convlstm_step_kernel", "shapes": {}}
    for (b, c, h, w) in ((8, 64, 128, 128), (8, 128, 64, 64), (8, 256, 32, 32)):
        g = torch.Generator(device="cpu").manual_seed(c)
        x = torch.randn((b, c, h, w), generator=g).to(dev)
        hp = torch.tanh(torch.randn((b, c, h, w), generator=g)).to(dev)
        cp = torch.randn((b, c, h, w), generator=g).to(dev)
        wgt = ((torch.rand((4 * c, 2 * c, 3, 3), generator=g) * 2 - 1) * (3.0 / (18 * c) ** 0.5)).to(dev)
        bias = ((torch.rand((4 * c,), generator=g) * 2 - 1) * 0.5).to(dev)
        packed = CL.pack_gate_weights(wgt)
        xn, hn = CL.nchw_to_nhwc_bf16(x), CL.nchw_to_nhwc_bf16(hp)
        cn = cp.permute(0, 2, 3, 1).contiguous()
        step = lambda: CL.convlstm_step(xn, hn, cn, packed, bias, nchw_dtype=torch.float32)   # noqa: E731
        for _ in range(3):
            got = step()
        ms = sorted(time_launches(step, 20, torch))
        r16 = lambda v: v.to(torch.bfloat16).float()   # noqa: E731
        gates = torch.nn.functional.conv2d(torch.cat([r16(x), r16(hp)], 1), r16(wgt), bias, padding=1)
        gi, gr, go, gg = gates.chunk(4, 1)
        c_ref = torch.sigmoid(gr) * cp + torch.sigmoid(gi) * torch.tanh(gg)
        h_ref = torch.sigmoid(go) * torch.tanh(c_ref)
        err = float((got[2] - h_ref).abs().max())
        flops = 2.0 * b * h * w * (18 * c) * (4 * c)
        avg = sum(ms) / len(ms)
        out["shapes"][f"{b}x{c}x{h}x{w}"] = {"kernel_ms_avg": avg, "kernel_ms_p50": ms[len(ms) // 2], "algorithmic_GFLOP": flops / 1e9,
                                           "achieved_TFLOPs": flops / (avg * 1e-3) / 1e12, "frac_of_bf16_peak": flops / (avg * 1e-3) / 1e12 / BF16_DENSE_PEAK_TFLOPS,
                                           "max_abs_err_vs_fp32_torch": err, "parity_check": "ok" if err < 2e-3 else "MISMATCH"}
    return out


def load_valu(name, kernel_ms):
    """VALU-issue view of a launch, next to the HBM one (the noise-on kernels are issue-bound, not bandwidth-bound): the kernel's
    dynamic vector-instruction count per launch (SQ_INSTS_VALU of the rocprofv3 --pmc pass in profiles/r06|r05|r04|r03/<workload>/summary.json --
    static per binary, like `traffic`) over this run's kernel time, against what 4 SIMDs x 256 CUs can issue at the 2.4 GHz peak
    clock: one wave-instruction per 2 cycles for the cheap class (add/sub/mul/logic/shift/mov) and per 4 cycles for everything else
    (all float64, fma, convert, compare, select, packed: profiles/valu_rates_ubench.txt).  The true ceiling of a kernel lies between
    the two by its instruction mix and is lowered further by the clock the chip holds under load (~1.7-1.9 GHz here)."""
    insts = rel = None
    for tag in ("r06", "r05", "r04", "r03"):       # the latest profile of this workload's kernels
        try:
            rel = f"profiles/{tag}/{name}/summary.json"
            insts = json.load(open(os.path.join(ROOT, rel)))["sq_counters_per_step"]["SQ_INSTS_VALU"]
            break
        except Exception:  # noqa: BLE001
            continue
    if insts is None:
        return None
    simds, clock = 1024, 2.4e9
    rate = insts / (kernel_ms * 1e-3)
    return {"bound": "valu-issue", "wave_instructions_per_launch": insts, "achieved_Ginstr_per_s": rate / 1e9,
            "peak_Ginstr_per_s_2cycle_class": simds * clock / 2 / 1e9, "peak_Ginstr_per_s_4cycle_class": simds * clock / 4 / 1e9,
            "frac_of_2cycle_peak": rate / (simds * clock / 2), "frac_of_4cycle_peak": rate / (simds * clock / 4),
            "source": f"static: {rel} (SQ_INSTS_VALU, rocprofv3 --pmc pass) over this run's kernel time"}


def load_traffic(name):
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")     # rocprofv3 --pmc passes, see profiles/README.md
    try:
        return json.load(open(tpath)).get(name, {}).get("hbm_bytes_per_launch")
    except Exception:  # noqa: BLE001
        return None


MAX_LINE_BYTES = 4096      # the contract line stays far below what the driver's stdout tail keeps (round 4's 21.9 KB line was cut: parsed = null)

# secondary workloads of the default run (one of each kind); --full adds the rest
ALSO_DEFAULT = ["cfg2_dataset_style", "cfg2_u8", "cfg3_v2e_f32_256x32x256x256_bilinear5", "cfg4_pipeline_720p_to_256_40f_bilinear5",
                "train_u8_12x201x128x128_sum5", "train_batch_normalised_12x201x128x128", "cfg5_fused_convlstm_channels_last"]


def measure_secondary(torch, dev, full, lap, kernels_only=False):
    """Rank 0 at N = 1, after the headline's launches have warmed the clocks and outside the timed region: each secondary workload's own
    kernel time (3 warm-up + 12 timed launches), algorithmic bytes and fraction of the HBM peak; then the loader at the training
    shape (tools/loader_bench.py) and the consumer's ConvLSTM step against the bf16 matrix peak.  Never the headline value."""
    also = {}
    for name in (ALSO_MEASURED if full else ALSO_DEFAULT):
        try:
            S = Workload(name, dev, 0, 1)
            for _ in range(3):
                S.step()
            s_step, s_graph = maybe_graph(S.step, torch, dev, S.wl["model"] in ("pipeline", "train_batch"))
            ms = sorted(time_launches(s_step, 12, torch))
            avg = sum(ms) / len(ms)
            ach = S.alg_bytes / (avg * 1e-3) / 1e9
            also[name] = {"kernel": S.kernel_name, "kernel_ms_avg": avg, "kernel_ms_p50": ms[len(ms) // 2], "grids_per_s": S.grids_per_step / (avg * 1e-3),
                          "algorithmic_bytes_per_launch": S.alg_bytes, "achieved_GBps": ach, "frac_of_hbm_peak": ach / HBM_PEAK_GBPS,
                          "traffic_bytes_per_launch": load_traffic(name), "valu_issue": load_valu(name, avg), "sim_params": S.wl["params"], "input_dtype": S.wl["dtype"],
                          "bin_mode": S.wl["bin"], "launch": "hipGraph replay" if s_graph else "eager", "parity_check": S.parity()}
            S.free()
            del S, s_step
            torch.cuda.empty_cache()
        except Exception as exc:  # noqa: BLE001 - a secondary figure must not take the headline down
            also[name] = {"error": f"{type(exc).__name__}: {exc}"}
        lap(name)
    if kernels_only:
        return also
    try:
        also["convlstm_step_mfma"] = convlstm_roofline(torch, dev)
    except Exception as exc:  # noqa: BLE001
        also["convlstm_step_mfma"] = {"error": f"{type(exc).__name__}: {exc}"}
    lap("convlstm_step_mfma")
    try:
        # what train.py receives at the reference's training shape (B = 12, 201 x 128 x 128 -> [12,40,5,128,128]) at each integration level:
        # YAML only (train.py untouched), + one line of train.py (RingLoader), the reference's deployment (NumPy port in workers)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import loader_bench
        also["train_loader_b12_201x128x128"] = loader_bench.measure(batches=200 if full else 60, workers=9, batch=12, dev=dev,
                                                                    simulating_batches=40 if full else 0, cpu_port_budget_s=60.0 if full else 8.0,
                                                                    consumer_batches=20 if full else 6, yaml_only_host_return_batches=10 if full else 0)
    except Exception as exc:  # noqa: BLE001
        also["train_loader_b12_201x128x128"] = {"error": f"{type(exc).__name__}: {exc}"}
    lap("train_loader")
    return also


def measure_host_input(torch, dev, W, wl, kern_avg_ms):
    """Host-resident input: the boundary takes device pointers, so a host pipeline pays the PCIe copy first.  Pageable copy then launch
    (no overlap), and page-locked double buffers on a copy stream (v2v_amd/staging.py).  Reported for context, never `value`."""
    try:
        n_host = min(W.b, 32)
        host_batch = W.frames[:n_host].cpu()
        t_h = time.perf_counter()
        dev_copy = host_batch.to(dev)
        torch.cuda.synchronize()
        h2d_s = time.perf_counter() - t_h
        h2d_gbps = host_batch.numel() * host_batch.element_size() / h2d_s / 1e9
        per_batch_s = W.frames.numel() * W.frames.element_size() / (h2d_gbps * 1e9) + kern_avg_ms * 1e-3
        pinned = host_batch.pin_memory()
        t_h = time.perf_counter()
        dev_copy = pinned.to(dev, non_blocking=True)
        torch.cuda.synchronize()
        pin_gbps = host_batch.numel() * host_batch.element_size() / (time.perf_counter() - t_h) / 1e9
        from v2v_amd import esim, staging
        stager = staging.HostStager(dev)
        sub_out = torch.empty((n_host,) + tuple(W.out.shape[1:]), dtype=torch.float32, device=dev)
        ptensor = torch.tensor(wl["params"], dtype=torch.float64, device=dev)

        def sim(frames_d):
            esim.esim_voxel_batch(frames_d, ptensor, bin_mode=wl["bin"], num_bins=wl["tb"], frames_per_bin=wl["fpb"], seed=20240001,
                                  clip_id0=0, out=sub_out, validate=False, no_noise=False)
        h = stager.stage(pinned)
        sim(stager.ready(h))
        torch.cuda.synchronize()
        reps = 8
        t_h = time.perf_counter()
        h = stager.stage(pinned)
        for _ in range(reps):
            cur, h = h, stager.stage(pinned)
            sim(stager.ready(cur))
        torch.cuda.synchronize()
        staged_s = (time.perf_counter() - t_h) / reps
        del dev_copy
        return {"h2d_GBps_pageable": h2d_gbps, "h2d_GBps_pinned": pin_gbps,
                "pcie_inclusive_grids_per_s": W.grids_per_step / per_batch_s,
                "pcie_inclusive_overlapped_grids_per_s": n_host * (W.grids_per_step // W.b) / staged_s,
                "note": "host-resident float32 clips: `pcie_inclusive` = pageable copy then launch, no overlap; `overlapped` = page-locked "
                        "double buffers on a copy stream (v2v_amd/staging.py), copy of batch k+1 under the launch on batch k, "
                        f"{n_host}-clip batches; PCIe-bound either way; reported for context, never `value`"}
    except Exception as exc:  # noqa: BLE001
        return {"error": f"{type(exc).__name__}: {exc}"}


def write_sidecar(obj, path):
    """Everything that is not the contract line: `bench_extra.json` next to bench.py (and a copy under gpurun_out/ when that exists, so
    a gpurun call brings it home).  Returns the path named in the line, or None when nothing could be written."""
    path = path or os.path.join(ROOT, "bench_extra.json")
    written = None
    for target in (path, os.path.join(ROOT, "gpurun_out", "bench_extra.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None):
        if not target:
            continue
        try:
            with open(target, "w") as f:
                json.dump(obj, f, indent=1)
            written = written or os.path.relpath(target, ROOT)
        except OSError as exc:
            print(f"[bench] could not write {target}: {exc}", file=sys.stderr, flush=True)
    return written


def contract_line(*, workload, wl, clips_per_gpu, grids_per_step, alg_bytes, kernel_name, world, steps, warmup, elapsed_s, per_rank_ms, kern_ms_sorted,
                  backend, dist_world, use_graph, traffic, cpu, parity):
    """THE contract line as a dict (pure function: tests/test_host_logic.py builds it for 8 ranks without a GPU): small, the contract's
    keys + `config` (shape keys, no prose) + `roofline` + `cpu_baseline` + `parity_check`; everything else goes to the sidecar file."""
    kern_avg_ms = sum(kern_ms_sorted) / len(kern_ms_sorted)
    achieved = alg_bytes / (kern_avg_ms * 1e-3) / 1e9
    hints = []
    if wl["model"] == "esim" and wl["params"][0] == wl["params"][1]:
        hints.append("V2V_FLAG_SYMMETRIC")
    if wl["model"] == "esim" and wl["params"][2] == 0 and wl["params"][3] <= 0:
        hints.append("V2V_FLAG_NO_NOISE")
    n = len(kern_ms_sorted)
    return {
        "metric": "voxel grids/sec", "value": grids_per_step * world * steps / elapsed_s, "unit": "voxel grids/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": elapsed_s / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "dist_backend": backend, "dist_world_size": dist_world,
        "ms_per_step_per_rank": [round(v, 6) for v in per_rank_ms],
        "config": {"workload": workload, "simulator": wl["model"], "clips_per_gpu": clips_per_gpu, "frames": wl["n"], "height": wl["h"], "width": wl["w"],
                   "input_dtype": wl["dtype"], "output_dtype": "float32", "state_dtype": "float64",
                   "bin_mode": wl["bin"], "num_bins": wl["tb"], "frames_per_bin": wl["fpb"], "sim_params": wl["params"],
                   "rng": "philox4x32", "kernel_hints": hints, "sharding": f"batch over {world} GPU(s), no collective",
                   "grid": [wl["tb"], wl["h"], wl["w"]], "launch": "hipGraph replay" if use_graph else "eager"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "kernel": kernel_name, "algorithmic_bytes_per_launch": alg_bytes,
                     "kernel_ms_avg": kern_avg_ms, "kernel_ms_p10": kern_ms_sorted[n // 10],
                     "kernel_ms_p50": kern_ms_sorted[n // 2], "kernel_ms_p90": kern_ms_sorted[(n * 9) // 10]},
        "cpu_baseline": cpu,
        "parity_check": parity,
        "extra": None,
    }


def line_text(line):
    """JSON text of the contract line, never above MAX_LINE_BYTES: a line the driver cannot parse is worth nothing, so the optional keys
    are shed first (they stay in the sidecar)."""
    text = json.dumps(line)
    if len(text) > MAX_LINE_BYTES:
        line = dict(line)
        for key in ("train_loader_samples_per_s", "ms_per_step_per_rank", "parity_check", "extra", "stream"):
            line.pop(key, None)
        line["config"] = {"workload": line["config"]["workload"], "clips_per_gpu": line["config"]["clips_per_gpu"]}
        text = json.dumps(line)
    return text


def self_launch(n_gpus):
    """Re-run this script as `n_gpus` ranks under torch.distributed.run (child process; this one never initialises the GPU)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n_gpus} without WORLD_SIZE: launching {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS) + sorted(STREAM_WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="override clips per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads of the default run")
    ap.add_argument("--kernels-only", action="store_true", help="secondary workloads: kernel timings only (no loader bench, no ConvLSTM roofline, no host-input leg)")
    ap.add_argument("--full", action="store_true", help="every secondary workload, 200-batch loader bench, NumPy process-pool baseline (minutes)")
    ap.add_argument("--extra-out", default=None, help="sidecar file for everything that is not the contract line (default: bench_extra.json next to bench.py)")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of the single-core NumPy-port baseline")
    ap.add_argument("--cpu-pool-budget", type=float, default=6.0, help="seconds of the all-cores NumPy-port baseline (timed part)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: every rank uses cuda:0 (tests the N>1 code path on a 1-GPU box)")
    ap.add_argument("--hip-graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the step from a captured hipGraph (auto: multi-launch pipeline workloads only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU under
        # torch.distributed.run) BEFORE anything in this process touches the GPU, relay rank 0's JSON line and exit with the
        # launcher's code.  A --gpus N request never quietly measures one GPU.
        sys.exit(self_launch(args.gpus))

    import torch
    from v2v_amd import sharding
    rank, local_rank, world = sharding.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} (or plain `python bench.py --gpus {args.gpus}`)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the hot path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = sharding.init_process_group(args.backend, dev)  # RCCL; used for the barrier + max-over-ranks only
    backend = dist.get_backend() if dist is not None else None

    if args.workload in STREAM_WORKLOADS:
        run_stream(args, torch, dev, rank, local_rank, world, dist, backend)
        if dist is not None:
            dist.destroy_process_group()
        return

    W = Workload(args.workload, dev, rank, world, args.batch)
    wl = W.wl
    step = W.step
    for _ in range(args.warmup):
        step()
    step, use_graph = maybe_graph(step, torch, dev, args.hip_graph == "on" or (args.hip_graph == "auto" and wl["model"] in ("pipeline", "train_batch")))
    sharding.barrier(dist, local_rank)
    t0 = time.perf_counter()
    kern_trace = time_launches(step, args.steps, torch)           # ends with torch.cuda.synchronize(): this rank's K steps are done
    my_elapsed = time.perf_counter() - t0
    kern_ms = sorted(kern_trace)
    sharding.barrier(dist, local_rank)                            # the closing bracket; the job's time is the MAX over ranks of their own K steps
    elapsed = sharding.max_over_ranks(dist, my_elapsed, dev)
    per_rank_ms = sharding.gather_floats(dist, my_elapsed / args.steps * 1e3, dev)        # every rank's own ms per step
    kern_avg_ms = sum(kern_ms) / len(kern_ms)

    # Everything below is outside the timed region.  Order: secondary kernel workloads first (the headline's launches have just
    # warmed the clocks), then the parity guard and the CPU baselines (the GPU idles under those).
    extra = {"seconds": {}}
    t_sec = time.perf_counter()

    def lap(name):
        nonlocal t_sec
        now = time.perf_counter()
        extra["seconds"][name] = round(now - t_sec, 2)
        t_sec = now

    # the same step again, OUTSIDE the timed region, once the clocks have settled (the chip ramps for ~30 ms after idle: the K timed launches
    # of a short run sit in that ramp): 100 more launches, their median -- context for `frac`, never `value`
    settled_ms = None
    if rank == 0 and world == 1 and not args.no_also:
        tail = sorted(time_launches(step, 100, torch))
        settled_ms = tail[len(tail) // 2]
    secondary = rank == 0 and world == 1 and args.workload == DEFAULT_WORKLOAD and not args.no_also and not args.batch
    if secondary:
        extra["also_measured"] = measure_secondary(torch, dev, args.full, lap, args.kernels_only)
        if not args.kernels_only:
            extra["host_input"] = measure_host_input(torch, dev, W, wl, kern_avg_ms)
            lap("host_input")

    parity = cpu = cpu_all = None
    if rank == 0:
        try:
            from oracle import clib
            clib.build()
            parity = W.parity()
            lap("parity")
            if world == 1 and not args.no_cpu_baseline:
                sample = W.frames[: min(W.b, 256)].cpu().numpy()
                cpu = cpu_baseline(sample, wl, budget_s=args.cpu_budget)
                lap("cpu_baseline")
                try:
                    cpu_all = cpu_baseline_all_cores(sample, wl, budget_s=args.cpu_pool_budget)
                except Exception as exc:  # noqa: BLE001 - reported beside the headline, never instead of it
                    cpu_all = {"error": f"{type(exc).__name__}: {exc}"}
                lap("cpu_baseline_all_cores")
                extra["cpu_baseline_c_omp"] = cpu_baseline_c(sample, wl)
                lap("cpu_baseline_c_omp")
                del sample
        except Exception as exc:  # the oracle is a checker; never let it take the measurement down
            parity = f"unchecked ({type(exc).__name__}: {exc})"

    if rank == 0:
        line = contract_line(workload=args.workload, wl=wl, clips_per_gpu=W.b, grids_per_step=W.grids_per_step, alg_bytes=W.alg_bytes, kernel_name=W.kernel_name,
                             world=world, steps=args.steps, warmup=args.warmup, elapsed_s=elapsed, per_rank_ms=per_rank_ms, kern_ms_sorted=kern_ms,
                             backend=backend, dist_world=(dist.get_world_size() if dist is not None else 1), use_graph=use_graph,
                             traffic=load_traffic(args.workload), cpu=cpu, parity=parity)
        if cpu_all is not None:
            line["cpu_baseline_all_cores"] = cpu_all                        # north_star: the CPU path on the box's host cores, core count stated
        if settled_ms:
            line["roofline"]["kernel_ms_settled_p50"] = settled_ms          # median of 100 launches after the timed region (clocks settled)
            line["roofline"]["frac_settled"] = W.alg_bytes / (settled_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
        extra["headline"] = {"kernel_ms_trace": [round(v, 4) for v in kern_trace], "valu_issue": load_valu(args.workload, kern_avg_ms),
                             "measured_ceilings_GBps": MEASURED_CEILINGS_GBPS,
                             "traffic_source": "static: profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload, collected "
                                               "separately; not re-measured in this run)",
                             "sim_params_order": "pos_thres, neg_thres, base_noise_std, hot_pixel_fraction, hot_pixel_std"
                             + (" = EventEmulator() constructor defaults of the reference (noise on)" if wl["params"] == REF_DEFAULTS else ""),
                             "rng": "philox4x32 on device (10 rounds per-clip fields, 7 rounds per-step noise fields), Gaussians by direct table inversion (2 per word)",
                             "note": "noise-on launches are VALU-issue-bound, not HBM-bound (DESIGN.md); `frac` is still quoted against the HBM peak; "
                                     "`valu_issue` gives the same launch against the vector-issue ceilings"}
        # the training-shape loader figures by integration level (BASELINE.md quotes the reference's deployment there): four numbers in
        # the line itself, the rest of tools/loader_bench.py's output in the sidecar
        lv = ((extra.get("also_measured") or {}).get("train_loader_b12_201x128x128") or {}).get("integration_levels_samples_per_s")
        if lv:
            line["train_loader_samples_per_s"] = {k: (round(v, 1) if isinstance(v, float) else v) for k, v in lv.items() if k != "decode_caveat" and v is not None}
            line["train_loader_samples_per_s"]["note"] = "B=12, 201x128x128 -> [12,40,5,128,128]; video decode excluded at every level"
            e2 = (extra["also_measured"]["train_loader_b12_201x128x128"].get("ring_loader_feeding_e2vid") or {}).get("samples_per_s")
            if e2:
                line["train_loader_samples_per_s"]["ring_loader_feeding_e2vid"] = round(e2, 1)
        line["extra"] = write_sidecar(dict(line, **extra), args.extra_out)
        sys.stderr.flush()
        print(line_text(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
