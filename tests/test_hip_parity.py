"""HIP path (through the C ABI, include/v2v_hip.h) against the CPU oracle and the committed golden
vectors.  Needs a real MI355X: run with `pytest -m gpu`.  Integer event counts must be bit-exact;
float32 voxel values within 1e-5 (the tolerance BASELINE.json's north_star states)."""
import numpy as np
import pytest
import torch

from oracle import v2v_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["4px", "2px", "1px"])
def _both_mappings(request, monkeypatch):
    """Every parity case runs through both families of instances (4 pixels / 1 pixel per work-item): left alone, shapes this
    small would all take the 1-pixel mapping the launcher picks for small batches."""
    from v2v_amd import esim
    monkeypatch.setattr(esim, "DEFAULT_MAPPING", request.param)

RTOL = ATOL = 1e-5   # north_star: "within 1e-5 on fp32 voxel values"


@pytest.fixture(scope="module")
def E():
    from v2v_amd import esim
    assert torch.cuda.is_available()
    return esim


def _replay_tensors(fields):
    return [torch.from_numpy(np.ascontiguousarray(f))[None] for f in fields]


# ------------------------------------------------------------------ goldens from the reference
@pytest.mark.parametrize("tag,cp,cn", [("sym", 0.2, 0.2), ("asym", 0.31, 0.47)])
@pytest.mark.parametrize("seed", [5, 6])
@pytest.mark.parametrize("dt_tag,dt", [("u8", np.uint8), ("f32", np.float32)])
def test_g2_clean_replay_bit_exact(E, golden, tag, cp, cn, seed, dt_tag, dt):
    g = golden("g2_esim_clean.npz")
    video = g["video"].astype(dt)
    want = g[f"{tag}_s{seed}_{dt_tag}"].astype(np.float64)
    np.random.seed(seed)
    got = E.EventEmulator(cp, cn, 0.0, 0.0, 0.0, False, rng="numpy").video_to_voxel(video)
    assert got.dtype == np.float64 and got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("ext", [False, True])
@pytest.mark.parametrize("dt_tag,dt", [("u8", np.uint8), ("f32", np.float32)])
def test_g4_noisy_replay(E, golden, ext, dt_tag, dt):
    g = golden("g4_esim_noisy.npz")
    video = g["video"].astype(dt)
    want = g[f"ext{int(ext)}_{dt_tag}"]
    np.random.seed(int(g["seed"]))
    got = E.EventEmulator(*g["params"], put_noise_external=ext, rng="numpy").video_to_voxel(video)
    # float64 output requested -> the kernel's float64 state is written as is: bit-exact even with float noise
    assert np.array_equal(got, want)


@pytest.mark.parametrize("ext", [False, True])
@pytest.mark.parametrize("dt_tag,dt", [("u8", np.uint8), ("f32", np.float32)])
def test_g11_philox_native_equals_reference_on_same_fields(E, golden, ext, dt_tag, dt):
    g = golden("g11_philox_fed.npz")
    video = g["video"].astype(dt)
    want = g[f"ext{int(ext)}_{dt_tag}"]
    em = E.EventEmulator(*g["params"], put_noise_external=ext, seed=int(g["seed"]), rng="philox",
                         clip_id=int(g["clip_id"]))
    got = em.video_to_voxel(video)
    assert np.array_equal(got, want)
    # torch in -> torch float32 out
    got32 = em.video_to_voxel(torch.from_numpy(video).cuda())
    assert got32.dtype == torch.float32 and got32.is_cuda
    np.testing.assert_allclose(got32.cpu().numpy(), want, rtol=RTOL, atol=ATOL)


def test_g6_imgs_to_voxels_sum_binning(E, golden):
    g = golden("g6_imgs_to_voxels.npz")
    video = g["video"]
    for seed_key, par_key, vox_key, fpb, scale in (("seed", "params", "voxels", 1, False), ("seed2", "params2", "voxels2", 2, True)):
        np.random.seed(int(g[seed_key]))
        for _ in range(3):
            np.random.uniform()          # thres_1, gap, rand()>0.5 draws of v2v_datasets.py:369-372
        for _ in range(3):
            np.random.uniform()          # noise parameter draws :379-381
        fields = E.draw_numpy_replay_fields(*video.shape)
        out = E.esim_voxel_batch(torch.from_numpy(video)[None].cuda(), g[par_key], bin_mode="sum", num_bins=5,
                                 frames_per_bin=fpb, rng_mode="replay", replay=_replay_tensors(fields),
                                 out_dtype=torch.float64)
        assert np.array_equal(out[0].cpu().numpy(), g[vox_key])


@pytest.mark.parametrize("k", [2, 7, 31, 39])
def test_g7_bilinear_weights_through_kernel(E, golden, k):
    """Drive the kernel so that its per-pair counts are known, then compare the binned result with the
    reference's events_to_voxel output on the same counts (golden G7)."""
    g = golden("g7_bilinear.npz")
    counts = g[f"counts_K{k}"].astype(np.float64)       # [K,16,16] in -6..6
    want = g[f"voxel_K{k}"]
    # external-noise mode with zero thresholds crossing: make the simulator emit exactly `counts` by feeding
    # them as replayed base noise (std = 1) on a constant video with huge thresholds
    video = np.full((k + 1, 16, 16), 128, dtype=np.uint8)
    zeros = np.zeros((16, 16))
    fields = (zeros + 0.5, zeros + 1.0, zeros, counts)
    out = E.esim_voxel_batch(torch.from_numpy(video)[None].cuda(), [1e6, 1e6, 1.0, 0.0, 0.0], bin_mode="bilinear",
                             num_bins=5, rng_mode="replay", replay=_replay_tensors(fields), put_noise_external=True,
                             out_dtype=torch.float64)
    assert np.array_equal(out[0].cpu().numpy(), want)
    out32 = E.esim_voxel_batch(torch.from_numpy(video)[None].cuda(), [1e6, 1e6, 1.0, 0.0, 0.0], bin_mode="bilinear",
                               num_bins=5, rng_mode="replay", replay=_replay_tensors(fields), put_noise_external=True)
    np.testing.assert_allclose(out32[0].cpu().numpy(), want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("sign,asym", [(1, False), (-1, False), (1, True), (-1, True)])
def test_g5_floor_divide_near_ties_through_kernel(E, golden, sign, asym):
    """Each (a,b) near-tie of the reference's np.floor_divide becomes one pixel whose potential is +-a exactly and whose threshold on that
    side is b: the kernel must report +-q.  Both polarities and both threshold selections (C+ == C-: registers; C+ != C-: the LDS pair by
    polarity) -- round 5 runs the whole quotient chain in signed form, so the OFF side and its +-1 fix-up are exercised like the ON side."""
    g = golden("g5_floor_divide.npz")
    a, b, q = g["a"], g["b"], g["q"]
    sel = a >= b
    a, b, q = a[sel], b[sel], q[sel]
    n = a.size
    frames = torch.from_numpy(np.full((n, 2, 1, 4), 77, dtype=np.uint8)).cuda()      # constant video: diff = 0, the potential stays what the noise makes it
    other = b * 1.75 if asym else b                                                    # the threshold of the side that does not fire
    pos, neg = (b, other) if sign > 0 else (other, b)
    params = np.stack([pos, neg, np.ones(n), np.zeros(n), np.zeros(n)], axis=1)
    # potential init = u * (pos + neg) - neg = 0 with u = neg / (pos + neg) is not exact in general; inject +-a through the base noise on top of
    # an init that IS exact: u = 0.5 with pos == neg gives 0; for the asymmetric case start from u = 0 (potential = -neg) and add neg back
    if asym:
        u_init = np.zeros((n, 1, 4))
        g_base = np.broadcast_to((sign * a + neg)[:, None, None, None], (n, 1, 1, 4)).copy()   # -neg + (neg + sign*a): exact only if it rounds back
        exact = (-neg + (sign * a + neg)) == sign * a
    else:
        u_init = np.full((n, 1, 4), 0.5)
        g_base = np.broadcast_to((sign * a)[:, None, None, None], (n, 1, 1, 4)).copy()
        exact = np.ones(n, dtype=bool)
    assert exact.sum() > 0.4 * n                                                        # enough ties survive the asymmetric construction
    out = E.esim_voxel_batch(frames, params, bin_mode="sum", num_bins=1, rng_mode="replay",
                             replay=[torch.from_numpy(x) for x in (u_init, np.ones((n, 1, 4)), np.zeros((n, 1, 4)), g_base)],
                             out_dtype=torch.float64)
    got = out[:, 0, 0, 0, 0].cpu().numpy()
    assert np.array_equal(got[exact], sign * q[exact])


# ------------------------------------------------------------------ HIP vs C oracle, seeded inputs
@pytest.mark.parametrize("dt", [np.uint8, np.float32])
@pytest.mark.parametrize("bin_mode,nb,fpb,n", [("sum", 5, 1, 11), ("sum", 5, 2, 21), ("bilinear", 5, 1, 32), ("bilinear", 3, 1, 9)])
@pytest.mark.parametrize("ext", [False, True])
def test_hip_vs_c_oracle_philox(E, oracle_c, luts, dt, bin_mode, nb, fpb, n, ext):
    b, h, w = 3, 40, 64
    video = np.stack([O.synth_clip_s1(n, h, w, seed=100 + i, dtype=dt) for i in range(b)])
    params = np.array([[0.2, 0.2, 0.05, 5e-3, 1.0], [0.31, 0.47, 0.0, 0.0, 0.0], [0.07, 0.11, 0.02, 0.01, 0.3]])
    bm = oracle_c.BIN_SUM if bin_mode == "sum" else oracle_c.BIN_BILINEAR
    want, totals = oracle_c.esim_voxel(video, params, luts, noise_external=ext, rng_mode=oracle_c.RNG_PHILOX,
                                       seed=0xC0FFEE1234, clip_id0=17, bin_mode=bm, num_bins=nb, frames_per_bin=fpb)
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    got64 = E.esim_voxel_batch(torch.from_numpy(video).cuda(), params, bin_mode=bin_mode, num_bins=nb,
                               frames_per_bin=fpb, rng_mode="philox", seed=0xC0FFEE1234, clip_id0=17,
                               put_noise_external=ext, out_dtype=torch.float64, counts=counts)
    assert np.array_equal(got64.cpu().numpy(), want)
    assert np.array_equal(counts.cpu().numpy(), totals)
    got32 = E.esim_voxel_batch(torch.from_numpy(video).cuda(), params, bin_mode=bin_mode, num_bins=nb,
                               frames_per_bin=fpb, rng_mode="philox", seed=0xC0FFEE1234, clip_id0=17,
                               put_noise_external=ext)
    assert got32.dtype == torch.float32
    if not ext and bin_mode == "sum":
        assert np.array_equal(got32.cpu().numpy(), want)                  # integer counts: exact in fp32 too
    else:
        np.testing.assert_allclose(got32.cpu().numpy(), want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("h,w", [(1, 1), (3, 5), (7, 9), (16, 18), (33, 31), (1080, 1920), (719, 1279)])
def test_ragged_sizes_scalar_path(E, oracle_c, luts, h, w):
    video = O.synth_clip_s1(6, h, w, seed=9, dtype=np.uint8)[None]
    p = [0.15, 0.25, 0.04, 0.02, 0.5]
    for dt in (np.uint8, np.float32):
        v = video.astype(dt)
        want, _ = oracle_c.esim_voxel(v, p, luts, seed=3, clip_id0=1, bin_mode=oracle_c.BIN_SUM, num_bins=5, frames_per_bin=1)
        got = E.esim_voxel_batch(torch.from_numpy(v).cuda(), p, seed=3, clip_id0=1, out_dtype=torch.float64)
        assert np.array_equal(got.cpu().numpy(), want)


def test_strided_input_view(E, oracle_c, luts):
    big = torch.from_numpy(np.stack([O.synth_clip_s1(16, 32, 32, seed=i, dtype=np.uint8) for i in range(4)])).cuda()
    view = big[1:3, 2:13]                      # clip_stride 16*1024, frame_stride 1024, offset not at 0
    want, _ = oracle_c.esim_voxel(view.cpu().numpy(), [0.2, 0.2, 0, 0, 0], luts, seed=1, bin_mode=oracle_c.BIN_SUM,
                                  num_bins=5, frames_per_bin=2)
    got = E.esim_voxel_batch(view, [0.2, 0.2, 0, 0, 0], seed=1, num_bins=5, frames_per_bin=2, out_dtype=torch.float64)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("bin_mode,nb,fpb,n", [("bilinear", 1, 1, 9), ("bilinear", 2, 1, 9), ("bilinear", 2, 1, 3), ("bilinear", 12, 1, 6), ("bilinear", 64, 1, 4),
                                               ("sum", 1, 1, 2), ("sum", 1, 8, 9), ("sum", 8, 1, 9), ("sum", 1, 1, 9)])
def test_bin_count_extremes(E, oracle_c, luts, bin_mode, nb, fpb, n):
    """The ends of the binning parameters: one or two temporal-bilinear bins (no interior segment), more bins than frame pairs (most bins
    empty), two frame pairs only; SUM with a single pair, a single plane of 8 pairs, one bin per pair -- HIP == C oracle, float64 exact."""
    video = np.stack([O.synth_clip_s1(n, 24, 36, seed=60 + i, dtype=np.uint8) for i in range(2)])
    p = [0.15, 0.22, 0.05, 5e-3, 1.0]
    bm = oracle_c.BIN_SUM if bin_mode == "sum" else oracle_c.BIN_BILINEAR
    want, tot = oracle_c.esim_voxel(video, p, luts, seed=21, clip_id0=7, bin_mode=bm, num_bins=nb, frames_per_bin=fpb)
    counts = torch.zeros((2, 2), dtype=torch.int64, device="cuda")
    got = E.esim_voxel_batch(torch.from_numpy(video).cuda(), p, bin_mode=bin_mode, num_bins=nb, frames_per_bin=fpb, seed=21, clip_id0=7,
                             out_dtype=torch.float64, counts=counts)
    assert got.shape == want.shape and np.array_equal(got.cpu().numpy(), want) and np.array_equal(counts.cpu().numpy(), tot)
    got32 = E.esim_voxel_batch(torch.from_numpy(video.astype(np.float32)).cuda(), p, bin_mode=bin_mode, num_bins=nb, frames_per_bin=fpb, seed=21, clip_id0=7)
    np.testing.assert_allclose(got32.cpu().numpy(), want, rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("dt", [np.uint8, np.float32])
@pytest.mark.parametrize("ext", [False, True])
def test_parameter_extremes(E, oracle_c, luts, dt, ext):
    """Per-clip parameters at the ends of what the arithmetic takes: thresholds of 1e-3 (thousands of events per step, quotients far from
    0 / 1), of 5 and 7 (a few hot-pixel events only), every pixel hot with a standard deviation of 50, base noise larger than any signal, and a clean
    asymmetric clip beside them in the same launch -- HIP == C oracle, float64 counts exact, ON/OFF totals exact."""
    params = np.array([[1e-3, 1.3e-3, 0.0, 0.0, 0.0], [5.0, 7.0, 0.05, 1e-3, 1.0], [0.2, 0.2, 2.0, 1.0, 50.0], [1e-3, 1e-3, 0.5, 0.5, 5.0],
                       [0.31, 0.47, 0.0, 0.0, 0.0], [0.05, 0.9, 0.3, 0.02, 20.0]])
    b = len(params)
    video = np.stack([O.synth_clip_s1(11, 24, 40, seed=80 + i, dtype=dt) for i in range(b)])
    for bin_mode, bm, nb, fpb in (("sum", oracle_c.BIN_SUM, 5, 2), ("bilinear", oracle_c.BIN_BILINEAR, 5, 1)):
        want, tot = oracle_c.esim_voxel(video, params, luts, noise_external=ext, seed=0x5EED, clip_id0=3, bin_mode=bm, num_bins=nb, frames_per_bin=fpb)
        counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
        got = E.esim_voxel_batch(torch.from_numpy(video).cuda(), params, bin_mode=bin_mode, num_bins=nb, frames_per_bin=fpb, seed=0x5EED, clip_id0=3,
                                 put_noise_external=ext, out_dtype=torch.float64, counts=counts)
        assert np.array_equal(got.cpu().numpy(), want), bin_mode
        assert np.array_equal(counts.cpu().numpy(), tot) and tot[1].sum() < 1000 and tot[0].sum() > 1e5


def test_long_clips(E, oracle_c, luts):
    """MAXIMUM clip lengths: 3,001 (float32 grid) / 1,501 (float64) frames into 5 bilinear bins (the per-pair weight table and segment starts
    fill LDS next to the Gaussian table), 10,001 frames into SUM bins (no per-pair table), both against the C oracle; one pair more than the tables hold is refused with
    the status the header names (V2V_ERR_SHAPE -> ValueError), not mis-run."""
    p = [0.2, 0.3, 0.05, 5e-3, 1.0]
    for n, dt, tol in ((3001, torch.float32, dict(rtol=RTOL, atol=ATOL)), (1501, torch.float64, dict(rtol=1e-12, atol=1e-9))):   # 8 / 16 bytes of weights per pair
        v = O.synth_clip_s1(n, 8, 12, seed=4, dtype=np.uint8)[None]
        want, tot = oracle_c.esim_voxel(v, p, luts, seed=8, clip_id0=2, bin_mode=oracle_c.BIN_BILINEAR, num_bins=5)
        counts = torch.zeros((1, 2), dtype=torch.int64, device="cuda")
        got = E.esim_voxel_batch(torch.from_numpy(v).cuda(), p, bin_mode="bilinear", num_bins=5, seed=8, clip_id0=2, out_dtype=dt, counts=counts)
        assert np.array_equal(counts.cpu().numpy(), tot)
        np.testing.assert_allclose(got.cpu().numpy(), want, **tol)
    v = O.synth_clip_s1(10001, 4, 8, seed=5, dtype=np.uint8)[None]
    want, _ = oracle_c.esim_voxel(v, p, luts, seed=8, clip_id0=3, bin_mode=oracle_c.BIN_SUM, num_bins=5, frames_per_bin=4)
    got = E.esim_voxel_batch(torch.from_numpy(v).cuda(), p, bin_mode="sum", num_bins=5, frames_per_bin=4, seed=8, clip_id0=3, out_dtype=torch.float64)
    assert got.shape == (1, 500, 5, 4, 8) and np.array_equal(got.cpu().numpy(), want)
    with pytest.raises(ValueError, match="split the clip"):
        E.esim_voxel_batch(torch.zeros((1, 20001, 4, 4), dtype=torch.uint8, device="cuda"), p, bin_mode="bilinear", num_bins=5)


def test_shard_invariance_and_batch_independence(E):
    """Clip b of a batch == the same clip run alone with clip_id0 = b: what makes 8-GPU sharding exact."""
    frames = E.synth_clips(6, 11, 32, 64, dtype=torch.uint8, seed=77)
    p = [0.2, 0.2, 0.05, 1e-2, 1.0]
    full = E.esim_voxel_batch(frames, p, seed=99, clip_id0=40)
    for lo, hi in ((0, 2), (2, 6)):
        part = E.esim_voxel_batch(frames[lo:hi], p, seed=99, clip_id0=40 + lo)
        assert torch.equal(part, full[lo:hi])
    again = E.synth_clips(2, 11, 32, 64, dtype=torch.uint8, seed=77, clip_id0=3)
    assert torch.equal(again, frames[3:5])


def test_non_finite_and_out_of_range_float_frames(E):
    """Degenerate float32 frames -- NaN, +-inf, negative, beyond 255, huge -- behave as in the reference's NumPy arithmetic
    (data/v2v_core_esim.py:33-58): a NaN log value silences the pixel (every compare is false), an infinite potential passes the compare and
    np.floor_divide(inf, C) is NaN, so that step's count is NaN and the pixel is silent afterwards.  NaN-aware equality on every voxel; the
    neighbouring pixels of the same work-item are untouched."""
    import warnings
    v = O.synth_clip_s1(9, 4, 8, seed=4, dtype=np.float32)
    v[3, 0, 0] = np.nan; v[2, 0, 1] = np.inf; v[4, 0, 2] = -np.inf; v[5, 0, 3] = -3.0; v[1, 0, 4] = 300.0; v[6, 0, 5] = 1e30; v[2, 0, 6] = 0.0
    v[2:4, 2, 3] = np.inf; v[7, 3, 7] = np.nan
    np.random.seed(3)
    fields = O.draw_replay_fields(*v.shape)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = O.esim_video_to_voxel(v, 0.2, 0.3, 0.0, 0.0, 0.0, rng=O.ReplayRNG(fields[:2], [fields[2]] + list(fields[3])))
    assert np.isnan(want).sum() >= 3
    got = E.esim_voxel_batch(torch.from_numpy(v)[None].cuda(), [0.2, 0.3, 0, 0, 0], bin_mode="sum", num_bins=8, rng_mode="replay",
                             replay=_replay_tensors(fields), out_dtype=torch.float64)[0, 0].cpu().numpy()
    assert np.array_equal(got, want, equal_nan=True), np.argwhere(~((got == want) | (np.isnan(got) & np.isnan(want))))
    got32 = E.esim_voxel_batch(torch.from_numpy(v)[None].cuda(), [0.2, 0.3, 0, 0, 0], bin_mode="sum", num_bins=8, rng_mode="replay",
                               replay=_replay_tensors(fields))[0, 0].cpu().numpy()
    assert np.array_equal(got32.astype(np.float64), want, equal_nan=True)
    # the same frames with the reference's noise on (replayed fields), internal and external
    for ext in (False, True):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wantn = O.esim_video_to_voxel(v, 0.2, 0.3, 0.05, 5e-3, 1.0, put_noise_external=ext, rng=O.ReplayRNG(fields[:2], [fields[2]] + list(fields[3])))
        gotn = E.esim_voxel_batch(torch.from_numpy(v)[None].cuda(), [0.2, 0.3, 0.05, 5e-3, 1.0], bin_mode="sum", num_bins=8, rng_mode="replay",
                                  replay=_replay_tensors(fields), put_noise_external=ext, out_dtype=torch.float64)[0, 0].cpu().numpy()
        assert np.isnan(wantn).any() and np.array_equal(gotn, wantn, equal_nan=True), ext


def test_generic_fp32_content_tolerance(E):
    """Non-integer float32 content: the oracle's float32 pow/log are NumPy SIMD kernels that device
    powf/logf match only to 1-2 ulp, so counts may flip on <= 1e-5 of pixel-steps (SURVEY §7 hard part 4)."""
    g = np.random.default_rng(0)
    video = (O.synth_clip_s1(9, 64, 64, seed=4, dtype=np.float32) + g.uniform(0, 0.9, size=(9, 64, 64))).astype(np.float32)
    video = np.clip(video, 0, 255)
    np.random.seed(3)
    fields = O.draw_replay_fields(*video.shape)
    want = O.esim_video_to_voxel(video, 0.2, 0.2, 0.0, 0.0, 0.0, rng=O.ReplayRNG(fields[:2], [fields[2]] + list(fields[3])))
    got = E.esim_voxel_batch(torch.from_numpy(video)[None].cuda(), [0.2, 0.2, 0, 0, 0], bin_mode="sum", num_bins=8,
                             rng_mode="replay", replay=_replay_tensors(fields), out_dtype=torch.float64)[0, 0].cpu().numpy()
    bad = np.count_nonzero(got != want)
    assert bad <= max(1, int(1e-5 * want.size) + 1), bad
    assert np.abs(got - want).max() <= 1


def test_full_size_properties_config2(E):
    """BASELINE config 2 shape (a 16-clip slice of it): properties that need no oracle.
    (i) zero-motion video -> no events once the initial potential has discharged;
    (ii) event totals == sum of |voxel| in SUM mode; (iii) bilinear bins sum to the total signed count."""
    b, n, h, w = 16, 32, 256, 256
    frames = E.synth_clips(b, n, h, w, dtype=torch.float32)
    assert frames.min() >= 0 and frames.max() <= 255 and torch.equal(frames, frames.round())
    p = [0.2, 0.2, 0.0, 0.0, 0.0]
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    raw = E.esim_voxel_batch(frames, p, bin_mode="sum", num_bins=n - 1, seed=5, counts=counts)      # [B,1,31,H,W]
    assert torch.equal(raw, raw.round())
    pos = raw.clamp(min=0).sum(dim=(1, 2, 3, 4)).to(torch.int64)
    neg = (-raw).clamp(min=0).sum(dim=(1, 2, 3, 4)).to(torch.int64)
    assert torch.equal(counts[:, 0], pos) and torch.equal(counts[:, 1], neg)
    bil = E.esim_voxel_batch(frames, p, bin_mode="bilinear", num_bins=5, seed=5)
    torch.testing.assert_close(bil.sum(dim=1), raw[:, 0].sum(dim=1), rtol=1e-5, atol=1e-4)
    still = frames[:, :1].expand(-1, n, -1, -1).contiguous()
    z = E.esim_voxel_batch(still, p, bin_mode="sum", num_bins=n - 1, seed=5)
    assert not z.any()


def test_error_behaviour(E):
    f = torch.zeros((1, 21, 8, 8), dtype=torch.uint8, device="cuda")
    with pytest.raises(AssertionError):
        E.esim_voxel_batch(f[:, :20], [0.2, 0.2, 0, 0, 0], num_bins=5)           # reference assert, v2v_datasets.py:365
    with pytest.raises(ValueError):
        E.esim_voxel_batch(f.to(torch.int32), [0.2, 0.2, 0, 0, 0])
    with pytest.raises(ValueError):
        E.esim_voxel_batch(f, [0.0, 0.2, 0, 0, 0])
    with pytest.raises(ValueError):
        E.esim_voxel_batch(f, [0.2, 0.2, 0, 0, 0], rng_mode="replay")
    with pytest.raises(ValueError):
        E.esim_voxel_batch(f.cpu(), [0.2, 0.2, 0, 0, 0])
    from v2v_amd import _lib
    import ctypes as C
    rc = _lib.lib().v2v_esim_voxel_hip(None, 0, 1, 2, 1, 1, 2, 1, None, 0, 0, 0, 0, 0, None, 0, 1, 1, None, 1, None, None)
    assert rc == _lib.ERR_NULL and b"NULL" in _lib.lib().v2v_last_error()
    out = E.esim_voxel_batch(f[:0], [0.2, 0.2, 0, 0, 0])
    assert out.shape == (0, 4, 5, 8, 8)


def test_float64_non_integer_video_is_a_stated_deviation(luts):
    """The reference runs non-integer float64 video through float64 pow/log; the fused kernel's generic path is float32.
    The wrapper says so (RuntimeWarning) and the deviation stays inside the stated bound: <= 1e-4 of the pixel-steps differ
    from the float64 NumPy restatement on the same np.random stream, each by one count."""
    import warnings
    from v2v_amd import esim as E
    g = np.random.default_rng(8)
    base = g.uniform(5, 250, size=(48, 64))
    video = np.stack([np.clip(base + 9.0 * np.sin(0.35 * k + base / 40.0), 0, 255) for k in range(12)])       # float64, non-integer
    p = [0.21, 0.27, 0.03, 1e-3, 0.4]
    np.random.seed(4242)
    want = O.esim_video_to_voxel(video, *p, put_noise_external=False, rng=O.GlobalNumpyRNG, use_lut=False)
    np.random.seed(4242)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        got = E.EventEmulator(*p, rng="numpy").video_to_voxel(video)
    assert any(issubclass(r.category, RuntimeWarning) and "float32 log path" in str(r.message) for r in rec)
    diff = got != want
    assert diff.mean() <= 1e-4 and (np.abs(got - want)[diff] <= 1).all()
    # integer-valued float64 content takes the exact table path: no warning, no difference
    vi = np.round(video)
    np.random.seed(7)
    want_i = O.esim_video_to_voxel(vi, *p, put_noise_external=False, rng=O.GlobalNumpyRNG, use_lut=False)
    np.random.seed(7)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        got_i = E.EventEmulator(*p, rng="numpy").video_to_voxel(vi)
    assert np.array_equal(got_i, want_i)
