"""-m gpu parity at the FULL sizes BASELINE.json names (configs 2, 3, 4), not at toy shapes: the whole batch goes through
one launch exactly as bench.py times it, then the first, a middle and the last clip (32-bit index arithmetic breaks at
the END of a 2.1 GB batch, not at clip 0) are compared with the C oracle, and per-clip ON/OFF totals of the WHOLE batch
with the oracle's totals (OpenMP over the host cores).  Bars: integer counts / SUM voxels bit-exact; float32 bilinear voxels
rtol = atol = 1e-5 (north_star)."""
import numpy as np
import pytest

from oracle import v2v_oracle as O

gpu = pytest.mark.gpu
SEED = 20240001


def _sample_ids(b):
    return sorted({0, b // 2 - 1, b // 2, b - 1})


@gpu
@pytest.mark.parametrize("dt_name", ["float32", "uint8"])
@pytest.mark.parametrize("params", [[0.2, 0.2, 0.0, 0.0, 0.0], [0.2, 0.3, 0.05, 5e-4, 1.0], [0.2, 0.2, 0.1, 1e-3, 0.1]],
                         ids=["fixedC_clean", "asym_noisy", "ref_defaults"])
def test_cfg2_full_batch_256x32x256x256(oracle_c, luts, dt_name, params):
    """BASELINE config 2: 256 clips of 32x256x256, 5 temporal-bilinear bins, one launch.  `ref_defaults` = the reference's own
    EventEmulator() constructor defaults (data/v2v_core_esim.py:8-16) = bench.py's headline parameter set: the host list has
    pos == neg, so the launch takes the symmetric-only noise instance (V2V_FLAG_SYMMETRIC), exactly the instance bench.py times."""
    import torch
    from v2v_amd import esim
    b, n, h, w, tb = 256, 32, 256, 256, 5
    dt = getattr(torch, dt_name)
    frames = esim.synth_clips(b, n, h, w, dtype=dt, seed=SEED, clip_id0=0)
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    out = esim.esim_voxel_batch(frames, params, bin_mode="bilinear", num_bins=tb, seed=SEED, clip_id0=0, counts=counts)
    torch.cuda.synchronize()
    assert out.shape == (b, tb, h, w) and out.dtype == torch.float32 and bool(torch.isfinite(out).all())
    # (1) sampled clips, every voxel
    for c in _sample_ids(b):
        host = frames[c:c + 1].cpu().numpy()
        want, tot = oracle_c.esim_voxel(host, params, luts, seed=SEED, clip_id0=c, bin_mode=oracle_c.BIN_BILINEAR, num_bins=tb)
        np.testing.assert_allclose(out[c].cpu().numpy().astype(np.float64), want[0], rtol=1e-5, atol=1e-5, err_msg=f"clip {c}")
        assert np.array_equal(counts[c].cpu().numpy(), tot[0]), f"clip {c} ON/OFF totals"
    # (2) ON/OFF totals of all 256 clips (bit-exact integers) -- the whole batch through the oracle, OpenMP over clips
    host_all = frames.cpu().numpy()
    _, totals = oracle_c.esim_voxel(host_all, params, luts, seed=SEED, clip_id0=0, bin_mode=oracle_c.BIN_BILINEAR, num_bins=tb)
    assert np.array_equal(counts.cpu().numpy(), totals)
    # (3) size-independent property: a sub-batch taken from the END of the batch reproduces its clips bit for bit
    tail = esim.esim_voxel_batch(frames[-3:], params, bin_mode="bilinear", num_bins=tb, seed=SEED, clip_id0=b - 3)
    assert torch.equal(tail, out[-3:])


@gpu
@pytest.mark.parametrize("dt_name", ["float32", "uint8"])
def test_batch_beyond_2_to_31_elements(oracle_c, luts, dt_name):
    """MAXIMUM sizes: a batch sized for 288 GB of HBM rather than for config 2 -- 1,056 clips of 32 x 256 x 256 = 2.21e9 input elements
    (8.9 GB as float32), so clip 1,024 STARTS at element 2^31 and every index past it needs 64-bit arithmetic (config 2's 2^29 elements
    only take the byte offsets to 2^31).  One launch; the clips either side of the boundary and the last one against the C oracle,
    and the tail of the batch against the same clips run as their own small batch."""
    import torch
    from v2v_amd import esim
    b, n, h, w, tb = 1056, 32, 256, 256, 5
    params = [0.2, 0.3, 0.05, 5e-4, 1.0]
    assert b * n * h * w > 2 ** 31
    frames = esim.synth_clips(b, n, h, w, dtype=getattr(torch, dt_name), seed=SEED, clip_id0=0)
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    out = esim.esim_voxel_batch(frames, params, bin_mode="bilinear", num_bins=tb, seed=SEED, clip_id0=0, counts=counts)
    torch.cuda.synchronize()
    assert out.shape == (b, tb, h, w) and bool(torch.isfinite(out[-40:]).all())
    for c in (0, 1023, 1024, b - 1):
        host = frames[c:c + 1].cpu().numpy()
        want, tot = oracle_c.esim_voxel(host, params, luts, seed=SEED, clip_id0=c, bin_mode=oracle_c.BIN_BILINEAR, num_bins=tb)
        np.testing.assert_allclose(out[c].cpu().numpy().astype(np.float64), want[0], rtol=1e-5, atol=1e-5, err_msg=f"clip {c}")
        assert np.array_equal(counts[c].cpu().numpy(), tot[0]), f"clip {c} ON/OFF totals"
    tail = esim.esim_voxel_batch(frames[-33:], params, bin_mode="bilinear", num_bins=tb, seed=SEED, clip_id0=b - 33)
    assert torch.equal(tail, out[-33:])
    # SUM bins through the 1-pixel mapping (its own index arithmetic) on the same buffer: 31 pairs do not divide by 5, so 30 of them
    sub = frames[:, :31]                                           # a strided view: the launcher takes the clip stride from the tensor or copies
    s4 = esim.esim_voxel_batch(sub, params, bin_mode="sum", num_bins=5, frames_per_bin=3, seed=SEED, clip_id0=0)
    s1 = esim.esim_voxel_batch(sub[-12:], params, bin_mode="sum", num_bins=5, frames_per_bin=3, seed=SEED, clip_id0=b - 12, mapping="1px")
    assert torch.equal(s4[-12:], s1)
    if dt_name == "uint8":
        # the `frame` assembly (v / 255 of picked frames) and the normalise + pad pass over the same oversized batch: every sample is
        # independent, so the tail must equal the tail run alone
        from v2v_amd import loader, postops
        pick = list(range(3, 32, 4))
        fr = loader.clip_frames_f32(frames, pick)
        assert fr.shape == (b, len(pick), 1, h, w) and torch.equal(fr[-5:], loader.clip_frames_f32(frames[-5:], pick))
        assert torch.equal(fr[1024, 2, 0].cpu(), frames[1024, pick[2]].cpu().float() / 255)    # torch's CPU division (the GPU op multiplies by 1/255)
        del fr
        grid = s4.reshape(b, 2, 5, h, w)                                                   # 1,056 samples x 2 x 5 x 256 x 256: 6.9e8 voxels
        big = torch.cat([grid, grid, grid, grid])                                            # 4,224 samples: 2.77e9 voxels
        nrm = postops.normalize_and_pad(big, normalize=True, method="count")
        assert torch.equal(nrm[-7:], postops.normalize_and_pad(big[-7:].contiguous(), normalize=True, method="count"))
        assert torch.equal(nrm[3 * b + 5], nrm[5])


@gpu
def test_cfg2_full_batch_sum_mode_exact(oracle_c, luts):
    """Same batch, N = 31 frames -> (N-1) = 30 = 2 x 5 x 3: SUM binning (the voxel grid V2V trains on), bit-exact integers."""
    import torch
    from v2v_amd import esim
    b, n, h, w = 256, 31, 256, 256
    frames = esim.synth_clips(b, n, h, w, dtype=torch.uint8, seed=SEED, clip_id0=0)
    p = [0.25, 0.2, 0.05, 5e-4, 1.0]
    out = esim.esim_voxel_batch(frames, p, bin_mode="sum", num_bins=5, frames_per_bin=3, seed=SEED, clip_id0=0)
    assert out.shape == (b, 2, 5, h, w)
    for c in _sample_ids(b):
        want, _ = oracle_c.esim_voxel(frames[c:c + 1].cpu().numpy(), p, luts, seed=SEED, clip_id0=c, bin_mode=oracle_c.BIN_SUM, num_bins=5,
                                      frames_per_bin=3)
        assert np.array_equal(out[c].cpu().numpy().astype(np.float64), want[0]), f"clip {c}"


@gpu
def test_cfg4_shape_sum_mode_per_clip_device_params(oracle_c, luts):
    """The config-4 shape (256 clips x 41 uint8 frames -> 8 grids of 5 SUM bins) with PER-CLIP parameters resident on the
    device as a [B,5] tensor, drawn as imgs_to_voxels draws them (data/v2v_datasets.py:368-386): what SimulatingCollator hands
    the kernel.  A device tensor gives the launcher no host knowledge (no symmetric / no-noise hint): the general instance."""
    import torch
    from v2v_amd import esim
    b, n, h, w = 256, 41, 256, 256
    frames = esim.synth_clips(b, n, h, w, dtype=torch.uint8, seed=SEED, clip_id0=0)
    g = np.random.default_rng(41)
    pos = g.uniform(0.05, 2.0, b)
    gap = g.uniform(1.0, 1.5, b)
    neg = np.clip(np.where(g.random(b) > 0.5, pos * gap, pos / gap), 0.05, 2.0)
    params = np.stack([pos, neg, g.uniform(0, 0.2, b), g.uniform(0, 1e-3, b), g.uniform(0, 0.2, b)], axis=1)
    params[7, 1] = params[7, 0]                                  # one clip with equal thresholds inside an asymmetric batch
    params[11, 2:] = 0.0                                         # and one noise-free clip
    pdev = torch.tensor(params, dtype=torch.float64, device="cuda")
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    out = esim.esim_voxel_batch(frames, pdev, bin_mode="sum", num_bins=5, frames_per_bin=1, seed=SEED, clip_id0=0, counts=counts)
    torch.cuda.synchronize()
    assert out.shape == (b, 8, 5, h, w)
    for c in sorted({0, 7, 11, b // 2, b - 1}):
        want, tot = oracle_c.esim_voxel(frames[c:c + 1].cpu().numpy(), params[c], luts, seed=SEED, clip_id0=c, bin_mode=oracle_c.BIN_SUM,
                                        num_bins=5, frames_per_bin=1)
        assert np.array_equal(out[c].cpu().numpy().astype(np.float64), want[0]), f"clip {c}"
        assert np.array_equal(counts[c].cpu().numpy(), tot[0]), f"clip {c} totals"
    assert float(out.sum()) == float((counts[:, 0] - counts[:, 1]).sum())          # SUM grids are the signed counts: totals agree exactly


V2E_NOISY = [24, "pn_related", 0.5, 0.1, 0.0, 0.1, 30, 0.1, 0, 5.0, 0.1, 0.1]


@gpu
@pytest.mark.parametrize("dt_name", ["float32", "uint8"])
def test_cfg3_v2e_full_batch_256x32x256x256(oracle_c, luts, dt_name):
    """BASELINE config 3: the v2e model (per-pixel thresholds, low-pass, leak, shot noise) on the same 256-clip batch; three
    sampled clips against the C oracle: float64 counts exact, float32 bilinear grid to 1e-5."""
    import torch
    from v2v_amd import esim, v2e
    b, n, h, w, tb = 256, 32, 256, 256, 5
    frames = esim.synth_clips(b, n, h, w, dtype=getattr(torch, dt_name), seed=SEED, clip_id0=0)
    vp = v2e.make_params(*V2E_NOISY)
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    out = v2e.v2e_voxel_batch(frames, vp, bin_mode="bilinear", num_bins=tb, seed=SEED, clip_id0=0, counts=counts)
    torch.cuda.synchronize()
    assert out.shape == (b, tb, h, w) and bool(torch.isfinite(out).all())
    for c in (0, b // 2, b - 1):
        host = frames[c:c + 1].cpu().numpy()
        want, tot = oracle_c.v2e_voxel(host, oracle_c.v2e_params(*V2E_NOISY), luts, seed=SEED, clip_id0=c, bin_mode=oracle_c.BIN_BILINEAR,
                                       num_bins=tb)
        assert np.array_equal(counts[c].cpu().numpy(), tot[0]), f"clip {c}: ON/OFF totals"
        np.testing.assert_allclose(out[c].cpu().numpy().astype(np.float64), want[0], rtol=1e-5, atol=1e-5, err_msg=f"clip {c}")
    # per-pair counts of the last clip, bit-exact (float64 SUM mode, 31 planes)
    c = b - 1
    sub = v2e.v2e_voxel_batch(frames[c:c + 1], vp, bin_mode="sum", num_bins=n - 1, seed=SEED, clip_id0=c, out_dtype=torch.float64)
    want, _ = oracle_c.v2e_voxel(frames[c:c + 1].cpu().numpy(), oracle_c.v2e_params(*V2E_NOISY), luts, seed=SEED, clip_id0=c,
                                 bin_mode=oracle_c.BIN_SUM, num_bins=n - 1)
    assert np.array_equal(sub.cpu().numpy(), want)


@gpu
def test_v2e_batch_beyond_2_to_31_elements(oracle_c, luts):
    """The v2e model on 1,056 uint8 clips of 32 x 256 x 256 (2.21e9 input elements; the frame-sum pre-pass and the simulator both index
    past 2^31): the clips either side of element 2^31 and the last one against the C oracle, the tail against its own small batch."""
    import torch
    from v2v_amd import esim, v2e
    b, n, h, w, tb = 1056, 32, 256, 256, 5
    frames = esim.synth_clips(b, n, h, w, dtype=torch.uint8, seed=SEED, clip_id0=0)
    vp = v2e.make_params(*V2E_NOISY)
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    out = v2e.v2e_voxel_batch(frames, vp, bin_mode="bilinear", num_bins=tb, seed=SEED, clip_id0=0, counts=counts)
    torch.cuda.synchronize()
    for c in (1023, 1024, b - 1):
        want, tot = oracle_c.v2e_voxel(frames[c:c + 1].cpu().numpy(), oracle_c.v2e_params(*V2E_NOISY), luts, seed=SEED, clip_id0=c,
                                       bin_mode=oracle_c.BIN_BILINEAR, num_bins=tb)
        assert np.array_equal(counts[c].cpu().numpy(), tot[0]), f"clip {c}: ON/OFF totals"
        np.testing.assert_allclose(out[c].cpu().numpy().astype(np.float64), want[0], rtol=1e-5, atol=1e-5, err_msg=f"clip {c}")
    tail = v2e.v2e_voxel_batch(frames[-20:], vp, bin_mode="bilinear", num_bins=tb, seed=SEED, clip_id0=b - 20)
    assert torch.equal(tail, out[-20:])


@gpu
@pytest.mark.parametrize("n,bin_mode,b", [(41, "sum", 6), (40, "bilinear", 6), (41, "sum", 24), (40, "bilinear", 24)])
def test_cfg4_720p_to_256_pipeline(oracle_c, luts, n, bin_mode, b):
    """BASELINE config 4 at its real geometry: decoded 1280x720x3 frames, keep_top_percentile 0.54, crop -> 256x256, flip,
    N = 41 frames (SUM, the reference's assert holds) and N = 40 (temporal-bilinear): GPU front-end against the OpenCV-algorithm
    restatement (oracle/frontend_oracle.py; parity with cv2 itself is unpinned -- OpenCV is not in the image), then the
    simulator on the front-end's output against the C oracle, on sampled clips.  b = 24 is the batch bench.py times per GPU (the
    tiled front-end's batch table with four frames per block)."""
    import torch
    from oracle import frontend_oracle as FO
    from v2v_amd import esim, frontend
    sh, sw, crop, tb = 720, 1280, 256, 5
    g = np.random.default_rng(404 + n)
    # decoded BGR frames with real colour content (three different channels), resident in HBM
    base = esim.synth_clips(b, n, sh, sw, dtype=torch.uint8, seed=SEED + 7, clip_id0=0)
    raw = torch.stack([base, base.flip(-1), 255 - base], dim=-1).contiguous()                       # [B,T,720,1280,3]
    keep_h = int(sh * 0.54)                                                                           # v2v_datasets.py:73
    min_scale = max(0, crop / keep_h, crop / sw)
    scale = g.uniform(min_scale, max(1.3, min_scale), size=b)                                        # :260-272
    cb = (crop / scale).astype(np.int64)
    table = np.stack([[g.integers(0, keep_h - c + 1), g.integers(0, sw - c + 1), c, int(g.random() > 0.5)] for c in cb]).astype(np.int32)
    idx = np.stack([np.sort(g.integers(0, n, size=n)) if i % 2 else np.arange(n) for i in range(b)]).astype(np.int32)   # pauses repeat frames
    _, gray = frontend.prepare_clips_batch(raw, table, idx, crop, "gray")
    params = np.stack([[0.2 + 0.01 * i, 0.3, 0.05, 5e-4, 1.0] for i in range(b)])
    kw = dict(bin_mode=bin_mode, num_bins=tb, seed=SEED, clip_id0=100)
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    out = esim.esim_voxel_batch(gray, params, counts=counts, **kw)
    torch.cuda.synchronize()
    for c in (0, b // 2, b - 1):
        _, want_gray = FO.frontend(raw[c].cpu().numpy(), int(table[c, 2]), int(table[c, 0]), int(table[c, 1]), bool(table[c, 3]), crop, idx[c])
        assert np.array_equal(gray[c].cpu().numpy(), want_gray), f"front-end clip {c}"
        bm = oracle_c.BIN_SUM if bin_mode == "sum" else oracle_c.BIN_BILINEAR
        want, tot = oracle_c.esim_voxel(want_gray[None], params[c], luts, seed=SEED, clip_id0=100 + c, bin_mode=bm, num_bins=tb)
        got = out[c].cpu().numpy().astype(np.float64)
        if bin_mode == "sum":
            assert np.array_equal(got, want[0]), f"clip {c}"
        else:
            np.testing.assert_allclose(got, want[0], rtol=1e-5, atol=1e-5, err_msg=f"clip {c}")
        assert np.array_equal(counts[c].cpu().numpy(), tot[0])


@pytest.mark.gpu
def test_cfg4_stream_zero_copy_from_page_locked_host_frames_at_full_geometry():
    """BASELINE config 4 as a host-fed stream at bench.py's per-rank geometry (`--workload cfg4_stream`): 8 clips x 40 decoded 1280x720 BGR
    frames (885 MB) in PAGE-LOCKED HOST memory; the front-end kernel reads the crop rectangles straight out of it over PCIe.  The gray
    clips and the 5-bin grids of the simulator behind them equal those of the device-resident frames bit for bit, for every clip."""
    import torch
    from v2v_amd import esim, frontend
    b, n, sh, sw, crop = 8, 40, 720, 1280, 256
    g = np.random.default_rng(909)
    base = esim.synth_clips(b, n, sh, sw, dtype=torch.uint8, seed=SEED + 11, clip_id0=0)
    raw = torch.stack([base, base.flip(-1), 255 - base], dim=-1).contiguous()
    del base
    keep_h = int(sh * 0.54)
    scale = g.uniform(max(crop / keep_h, crop / sw), 1.3, size=b)
    cb = (crop / scale).astype(np.int64)
    table = np.stack([[g.integers(0, keep_h - c + 1), g.integers(0, sw - c + 1), c, int(g.random() > 0.5)] for c in cb]).astype(np.int32)
    idx = np.tile(np.arange(n, dtype=np.int32), (b, 1))
    _, want_gray = frontend.prepare_clips_batch(raw, table, idx, crop, "gray")
    host = raw.cpu().pin_memory()
    del raw
    torch.cuda.empty_cache()
    _, got_gray = frontend.prepare_clips_batch(host, table, idx, crop, "gray")
    assert got_gray.is_cuda and torch.equal(got_gray, want_gray)
    p = [0.2, 0.3, 0.05, 5e-4, 1.0]
    kw = dict(bin_mode="bilinear", num_bins=5, seed=SEED, clip_id0=7)
    assert torch.equal(esim.esim_voxel_batch(got_gray, p, **kw), esim.esim_voxel_batch(want_gray, p, **kw))


@pytest.mark.gpu
def test_cfg5_pipeline_feeds_the_consumer():
    """BASELINE config 5 at its per-GPU geometry: 8 clips of 41 decoded 1280x720x3 frames -> GPU front-end -> simulator (SUM, 5 bins
    -> [8,8,5,256,256]) -> the E2VID-shaped recurrent network of tools/e2vid_consumer.py over the 8 time steps, with its three
    ConvLSTM blocks on the fused matrix-core kernel, against the all-stock network with the same weights.  Floating point:
    reference = the stock FP32 network; bar = what the stock network itself loses under bf16 autocast on the same input (the
    fused blocks keep the cell state in fp32, so they must not be worse than that by more than 25 %), and half the
    prediction's spread as an absolute sanity bound."""
    import os
    import sys
    import torch
    from v2v_amd import esim, frontend
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from e2vid_consumer import E2VIDShapedConsumer, forward_sequence
    b, n, sh, sw, crop, tb = 8, 41, 720, 1280, 256, 5
    g = np.random.default_rng(505)
    base = esim.synth_clips(b, n, sh, sw, dtype=torch.uint8, seed=SEED + 9, clip_id0=0)
    raw = torch.stack([base, base.flip(-1), 255 - base], dim=-1).contiguous()
    keep_h = int(sh * 0.54)
    cb = (crop / g.uniform(crop / keep_h, 1.3, size=b)).astype(np.int64)
    table = np.stack([[g.integers(0, keep_h - c + 1), g.integers(0, sw - c + 1), c, int(g.random() > 0.5)] for c in cb]).astype(np.int32)
    idx = np.tile(np.arange(n, dtype=np.int32), (b, 1))
    _, gray = frontend.prepare_clips_batch(raw, table, idx, crop, "gray")
    params = np.stack([[0.2 + 0.01 * i, 0.25, 0.05, 5e-4, 0.5] for i in range(b)])
    voxels = esim.esim_voxel_batch(gray, params, bin_mode="sum", num_bins=tb, seed=SEED, clip_id0=0)          # [8,8,5,256,256]
    assert voxels.shape == (b, (n - 1) // tb, tb, crop, crop) and float(voxels.abs().sum()) > 0
    from e2vid_consumer import reference_to_stock_keys
    from seeded_weights import seeded_state
    from v2v_amd.unet import E2VIDRecurrent
    kw = dict(num_bins=tb, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32, num_residual_blocks=2,
              use_upsample_conv=True, final_activation="", norm=None)
    product = E2VIDRecurrent(kw).cuda().eval()                                    # the package's network, reference state_dict keys
    ref_sd = {k: torch.from_numpy(v) for k, v in seeded_state({k: tuple(v.shape) for k, v in product.unetrecurrent.state_dict().items()},
                                                              1805, 1.7).items()}       # golden G18's weights: O(1) activations
    product.unetrecurrent.load_state_dict(ref_sd, strict=True)
    stock = E2VIDShapedConsumer(num_bins=tb).cuda().eval()                        # float32 yardstick, pinned to the reference by G18 on the CPU
    stock.load_state_dict(reference_to_stock_keys(ref_sd), strict=True)
    fused = E2VIDShapedConsumer(num_bins=tb, fused_convlstm=True).cuda().eval()
    fused.load_stock_state_dict(stock.state_dict())
    with torch.no_grad():
        want32 = forward_sequence(stock, voxels)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            got_stock16 = forward_sequence(stock, voxels)
            got_fused16 = forward_sequence(fused, voxels)
            fused_cl = fused.to(memory_format=torch.channels_last)                       # NHWC network: the step kernel works in place
            got_fused_cl = forward_sequence(fused_cl, voxels, channels_last=True)
        product.reset_states()
        got_product = [product(voxels[:, t])["image"] for t in range(voxels.shape[1])]
    # ABSOLUTE bar for the package network (v2v_amd.unet.E2VIDRecurrent, every layer on the device kernels) against the float32
    # stock network on G18's weights: prediction std ~0.8; bf16 operands through 8 recurrent steps of a 15-layer network:
    # |err| <= 6e-2 max, 8e-3 rms at every step (G18's 64x64 / 3-step case holds 4e-2 / 8e-3 against the reference itself)
    for t, (w32, p) in enumerate(zip(want32, got_product)):
        assert p.dtype == torch.float32 and p.shape == w32.shape
        d = (p - w32).abs()
        assert float(d.max()) <= 6e-2 and float((d ** 2).mean().sqrt()) <= 8e-3, f"step {t}: max {float(d.max()):.4g} rms {float((d ** 2).mean().sqrt()):.4g} std {float(w32.std()):.3g}"
    for t, (w32, s16, f16, fcl) in enumerate(zip(want32, got_stock16, got_fused16, got_fused_cl)):
        spread = float(w32.std())
        assert f16.shape == w32.shape == (b, 1, crop, crop)
        err_f, err_s = float((f16.float() - w32).abs().max()), float((s16.float() - w32).abs().max())
        assert err_f < 1.25 * err_s + 1e-3 and err_f < 0.5 * spread, f"step {t}: fused {err_f:.4g}, stock autocast {err_s:.4g}, spread {spread:.4g}"
        err_c = float((fcl.float() - w32).abs().max())
        assert err_c < 1.25 * err_s + 1e-3 and err_c < 0.5 * spread, f"step {t}: fused channels_last {err_c:.4g}, stock autocast {err_s:.4g}"


@gpu
@pytest.mark.parametrize("b,h,w", [(1, 192, 240), (1, 272, 352), (3, 48, 80), (1, 16, 16), (5, 16, 32), (2, 48, 208)])
def test_network_at_real_data_frame_sizes(b, h, w):
    """The package network at the sizes the reference EVALUATES on, batch 1: HQF / IJRR 180 x 240 and MVSEC 260 x 346 frames padded to
    multiples of 16 (model/train_utils.py:322-326) -- pixel counts that are no multiple of any workgroup tile at the deeper levels (e.g.
    24 x 30 = 720 at level 3), so the last tile of nearly every launch is partial -- and a few small odd batches down to a 2 x 2 level-3
    map.  Three recurrent steps as the step loop, as forward_sequence and as its hipGraph replay, against the all-stock float32 network
    on G18's weights; same absolute bar as config 5 (6e-2 max, 8e-3 rms)."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from e2vid_consumer import E2VIDShapedConsumer, forward_sequence, reference_to_stock_keys
    from seeded_weights import seeded_state
    from v2v_amd.unet import E2VIDRecurrent
    kw = dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32, num_residual_blocks=2,
              use_upsample_conv=True, final_activation="", norm=None)
    product = E2VIDRecurrent(kw).cuda().eval()
    ref_sd = {k: torch.from_numpy(v) for k, v in seeded_state({k: tuple(v.shape) for k, v in product.unetrecurrent.state_dict().items()}, 1805, 1.7).items()}
    product.unetrecurrent.load_state_dict(ref_sd, strict=True)
    stock = E2VIDShapedConsumer(num_bins=5).cuda().eval()
    stock.load_state_dict(reference_to_stock_keys(ref_sd), strict=True)
    g = torch.Generator().manual_seed(b * 1000 + h + w)
    voxels = torch.round(torch.randn((b, 3, 5, h, w), generator=g) * 1.5).cuda()
    with torch.no_grad():
        want = torch.stack(forward_sequence(stock, voxels), dim=1)
        product.reset_states()
        loop = torch.stack([product(voxels[:, t])["image"] for t in range(3)], dim=1)
        product.reset_states()
        seq = product.forward_sequence(voxels)
        graph = product.forward_sequence(voxels, graph=True).clone()
    assert torch.equal(loop, seq) and torch.equal(seq, graph) and bool(torch.isfinite(loop).all())
    d = (loop.float() - want).abs()
    assert float(d.max()) <= 6e-2 and float((d ** 2).mean().sqrt()) <= 8e-3, f"max {float(d.max()):.4g} rms {float((d ** 2).mean().sqrt()):.4g} std {float(want.std()):.3g}"


@gpu
def test_training_batch_packed_clips_with_statistics(oracle_c, luts):
    """The reference's training shape (config/train_v2v_e2vid_10k.yaml:50-76: B = 12, 201 frames of 128 x 128, 40 x 5 SUM bins) as the
    loader launches it: every decoded frame stored once + the pause-index row (proba_pause_when_running 0.0102 / _when_paused 0.9791),
    per-clip parameters and RNG keys on the device, writer statistics.  First / middle / last clip against the scalar C oracle on the
    GATHERED clip (exact), all twelve statistics rows against the histogram of the written grid, scales against its sorted k-th values."""
    import torch
    from v2v_amd import _lib, esim, postops
    b, n, h, w, tb = 12, 201, 128, 128, 5
    g = np.random.default_rng(77)
    fidx = np.zeros((b, n), np.int32)
    for c in range(b):                                             # the reference's pause chain (data/v2v_datasets.py:292-300)
        idx, paused = 0, False
        for f in range(n):
            fidx[c, f] = idx
            u = g.random()
            if paused:
                paused = not (u > 0.9791)
            elif u < 0.0102:
                paused = True
            if not paused:
                idx += 1
    stored = fidx[:, -1] + 1
    video = esim.synth_clips(b, n, h, w, dtype=torch.uint8, seed=SEED + 11, clip_id0=0)            # decoded frames: the first `stored` of each
    offs = np.concatenate([[0], np.cumsum((stored * h * w + 15) // 16 * 16)])[:-1].astype(np.int64)
    flat = torch.zeros(int(offs[-1] + stored[-1] * h * w), dtype=torch.uint8, device="cuda")
    for c in range(b):
        flat[offs[c]:offs[c] + stored[c] * h * w] = video[c, :stored[c]].reshape(-1)
    params = np.stack([[g.uniform(0.05, 2), 0, g.uniform(0, 0.1), g.uniform(0, 1e-3), g.uniform(0, 10)] for _ in range(b)])
    params[:, 1] = params[:, 0] * g.uniform(1, 1.5, size=b)
    keys = np.ascontiguousarray(np.stack([g.integers(0, 1 << 62, size=b), np.arange(b) + 1000]).T.astype(np.int64))
    stats = torch.zeros((b, _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device="cuda")
    got = esim.esim_voxel_packed(flat, torch.from_numpy(offs).cuda(), torch.from_numpy(fidx).cuda(), h, w, torch.from_numpy(params).cuda(),
                                 torch.from_numpy(keys).cuda(), num_bins=tb, pad_to=16, stats=stats)
    assert got.shape == (b, 40, tb, h, w) and stored.min() < n                                       # some clip did pause
    vh = video.cpu().numpy()
    for c in (0, b // 2, b - 1):
        gathered = vh[c][fidx[c]][None]
        want, _ = oracle_c.esim_voxel(gathered, params[c], luts, seed=int(keys[c, 0]), clip_id0=int(keys[c, 1]), bin_mode=oracle_c.BIN_SUM, num_bins=tb)
        assert np.array_equal(got[c].cpu().numpy(), want[0].astype(np.float32)), c
    iv = got.long().reshape(b, -1)
    assert torch.equal(iv.float().reshape(got.shape), got)
    m = iv.shape[1]
    st = stats.cpu().numpy()
    for c in range(b):
        hist = torch.bincount(iv[c].clamp(-256, 256) + 256, minlength=516)[:516].cpu().numpy()
        hist[256] = 0
        assert np.array_equal(st[c], hist), c
    srt = torch.sort(got.reshape(b, -1), dim=1).values
    lo, hi = srt[:, int(0.01 * m) - 1].cpu().numpy(), srt[:, int(0.99 * m) - 1].cpu().numpy()
    sc = postops.scales_from_stats(stats, m).cpu().numpy()
    assert np.array_equal(sc, np.stack([np.maximum(-lo, 1), np.maximum(hi, 1)], 1).astype(np.float32))
