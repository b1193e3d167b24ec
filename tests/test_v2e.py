"""v2e model (BASELINE config 3): oracle vs the reference's golden outputs (CPU), HIP vs goldens and oracle (GPU)."""
import numpy as np
import pytest

from oracle import v2v_oracle as O

ARG_NAMES = ("FPS", "threshold_model", "thres_mean_mean", "thres_mean_std", "thres_diff_mean", "thres_diff_std", "cutoff_hz",
             "leak_rate_hz", "refractory_period_s", "shot_noise_rate_hz", "leak_jitter_fraction", "noise_rate_cov_decades")
CASES = ["pn_clean_u8", "pn_noisy_u8", "pn_noisy_f32", "si_leak_u8", "si_cut_f32", "sti_noisy_u8", "sti_shot_f32"]


def _case(g, name):
    a = g[f"{name}__args"]
    args = [a[0], O.V2E_MODELS[int(a[1])]] + [float(x) for x in a[2:]]
    video = g["video"].astype(np.dtype(str(g[f"{name}__dtype"])))
    fields = {k: g[f"{name}__{k}"] for k in ("pos_thres", "neg_thres", "noise_rate", "leak_randn", "shot_pos", "shot_neg")
              if f"{name}__{k}" in g}
    if args[1] != "spatial_temporal_independent":          # static thresholds are stored as a stack of one
        fields["pos_thres"], fields["neg_thres"] = fields["pos_thres"][0], fields["neg_thres"][0]
    return video, args, fields, g[f"{name}__voxels"].astype(np.float64)


class _Replay:
    """Feeds the recorded draws back in order (normal -> already-clipped thresholds cannot be un-clipped, so the
    oracle is driven through its own RNG hooks with the raw stream instead: see test below)."""


@pytest.mark.parametrize("name", CASES)
def test_g9_numpy_oracle_reproduces_reference_from_seed(golden, name):
    """Same seed, same global stream -> the NumPy restatement must equal the reference's output."""
    g = golden("g9_v2e.npz")
    video, args, fields, want = _case(g, name)
    rec = {}
    got = O.v2e_video_to_voxel(video, *args, seed=11, use_lut=True, record=rec)
    assert np.array_equal(got, want)
    # and the fields it drew are the ones the reference drew (pins the draw ORDER)
    pt = np.stack(rec["pos_thres"])
    assert np.array_equal(pt[1:] if pt.shape[0] > 1 else pt[0], fields["pos_thres"])
    if "leak_randn" in fields:
        assert np.array_equal(np.stack(rec["leak_randn"]), fields["leak_randn"])
    if "shot_pos" in fields:
        assert np.array_equal(np.stack(rec["shot_pos"]), fields["shot_pos"]) and np.array_equal(np.stack(rec["shot_neg"]), fields["shot_neg"])
    # np.exp(float32) may differ by an ulp across hosts; the golden holds the reference's bits
    assert np.all(np.abs(rec["noise_rate"] - fields["noise_rate"]) <= np.spacing(fields["noise_rate"]))


def _replay_arrays(fields, k, hw):
    return {"pos_thres": np.ascontiguousarray(fields["pos_thres"].reshape(-1, hw), dtype=np.float64),
            "neg_thres": np.ascontiguousarray(fields["neg_thres"].reshape(-1, hw), dtype=np.float64),
            "stride": hw if fields["pos_thres"].ndim == 3 else 0,
            "noise_rate": np.ascontiguousarray(fields["noise_rate"].reshape(hw), dtype=np.float32),
            "leak_randn": np.ascontiguousarray(fields["leak_randn"].reshape(k, hw)) if "leak_randn" in fields else None,
            "shot_pos": np.ascontiguousarray(fields.get("shot_pos", np.zeros((k, hw), np.int64)).reshape(k, hw), dtype=np.int64),
            "shot_neg": np.ascontiguousarray(fields.get("shot_neg", np.zeros((k, hw), np.int64)).reshape(k, hw), dtype=np.int64)}


@pytest.mark.parametrize("name", CASES)
def test_g9_c_oracle_replay(golden, oracle_c, luts, name):
    g = golden("g9_v2e.npz")
    video, args, fields, want = _case(g, name)
    k, hw = video.shape[0] - 1, video.shape[1] * video.shape[2]
    got, totals = oracle_c.v2e_voxel(video[None], oracle_c.v2e_params(*args), luts, rng_mode=oracle_c.RNG_REPLAY,
                                     bin_mode=oracle_c.BIN_SUM, num_bins=k, replay=_replay_arrays(fields, k, hw))
    assert np.array_equal(got[0, 0], want)


def test_refractory_known_answers(oracle_c, luts):
    """The reference's refractory line raises; intended semantics = min(count, int(dt/refractory)).
    Hand-checkable: a 0 -> 255 step with tiny thresholds saturates every pixel at the cap."""
    video = np.zeros((3, 4, 4), dtype=np.uint8)
    video[1:] = 255
    args = [24, "spatial_independent", 0.05, 0.0, 0.0, 0.0, 0, 0, 1 / 240, 0, 0.0, 0.0]
    got = O.v2e_video_to_voxel(video, *args, seed=1, use_lut=True)
    assert np.all(got[0] == int((1 / 24) / (1 / 240))) and int((1 / 24) / (1 / 240)) in (9, 10)
    c, _ = oracle_c.v2e_voxel(video[None], oracle_c.v2e_params(*args), luts, seed=1, bin_mode=oracle_c.BIN_SUM, num_bins=2)
    assert np.array_equal(c[0, 0, 0], got[0])       # std = 0 -> thresholds are exactly 0.05 in native mode too


def test_native_det_functions_accuracy(oracle_c):
    import ctypes as C
    L = oracle_c.lib()
    L.oracle_exp_neg.restype = C.c_double
    L.oracle_exp_neg.argtypes = [C.c_double]
    L.oracle_expf_det.restype = C.c_float
    L.oracle_expf_det.argtypes = [C.c_float]
    L.oracle_poisson_inv.restype = C.c_double
    L.oracle_poisson_inv.argtypes = [C.c_double, C.c_double]
    assert max(abs(L.oracle_exp_neg(x) - np.exp(-x)) / np.exp(-x) for x in np.linspace(0, 60, 601)) < 1e-15
    assert max(abs(L.oracle_expf_det(float(x)) - np.exp(x)) / np.exp(x) for x in np.linspace(-6, 6, 601)) < 3e-7
    u = np.random.default_rng(0).random(20000)
    for lam in (0.05, 0.104, 1.7):
        x = np.array([L.oracle_poisson_inv(lam, float(v)) for v in u])
        assert abs(x.mean() - lam) < 4 * np.sqrt(lam / u.size) and abs(x.var() - lam) < 0.1 * lam + 0.01


# ------------------------------------------------------------------ G14: the reference run on the device-native fields
G14_CASES = ["pn_noisy_u8", "pn_noisy_f32", "pn_shot_only_f32", "si_leak_u8", "sti_noisy_u8", "sti_shot_f32"]


def _case14(g, name):
    a = g[f"{name}__args"]
    args = [a[0], O.V2E_MODELS[int(a[1])]] + [float(x) for x in a[2:]]
    video = g["video"].astype(np.dtype(str(g[f"{name}__dtype"])))
    return video, args, int(g["seed"]), int(g["clip_id"]), g[f"{name}__voxels"].astype(np.float64)


@pytest.mark.parametrize("name", G14_CASES)
def test_g14_c_oracle_native_equals_reference_on_native_fields(golden, oracle_c, luts, name):
    """Native mode of the C oracle (Philox fields, fixed-point frame mean, float32 shot-noise means, inversion sampler)
    against the REFERENCE fed with the same fields (tests/golden/make_goldens.py::g14_v2e_native).  The reference
    normalises the shot-noise mean with a float64 np.mean and keeps it in float64; the native path uses 2^32 fixed-point
    sums and float32 means (relative difference ~3e-7), so a count can only flip when a 16-bit uniform falls within ~3e-8
    of a sampler threshold: expected flips over this fixture's 1.8e4 samples: < 1e-3.  Required: exact."""
    g = golden("g14_v2e_native.npz")
    video, args, seed, clip, want = _case14(g, name)
    k = video.shape[0] - 1
    got, totals = oracle_c.v2e_voxel(video[None], oracle_c.v2e_params(*args), luts, seed=seed, clip_id0=clip,
                                     bin_mode=oracle_c.BIN_SUM, num_bins=k)
    assert np.array_equal(got[0, 0], want)
    assert totals[0, 0] == int(np.clip(want, 0, None).sum()) or args[9] > 0     # with shot noise ON and OFF can cancel in a pixel


def test_native_poisson_sampler_statistics(oracle_c):
    """The native float32 inversion sampler on its own 16-bit uniforms against NumPy's Poisson (v2v_core_v2e.py:102-103):
    mean, variance and the probabilities of 0 / 1 / 2+ events for the means config 3 produces (and a large one)."""
    n = 1 << 20
    u = oracle_c.philox_uniform16_field(77, 1, 16 + 8 * 3 + 3, n, low=0)
    u2 = oracle_c.philox_uniform16_field(77, 1, 16 + 8 * 3 + 3, n, low=1)
    assert abs(u.mean() - 0.5) < 1e-3 and abs(np.corrcoef(u, u2)[0, 1]) < 4e-3 and u.min() > 0 and u.max() < 1
    ref = np.random.default_rng(5)
    for lam in (0.02, 0.104, 0.35, 1.7):
        x = oracle_c.poisson_inv_f32(np.full(n, lam, np.float32), u)
        y = ref.poisson(lam, size=n)
        se = np.sqrt(lam / n)
        assert abs(x.mean() - lam) < 5 * se + 2e-5 and abs(x.var() - lam) < 12 * se * max(1, lam) + 1e-4
        for c in (0, 1):
            p_true = np.exp(-lam) * lam ** c
            assert abs((x == c).mean() - p_true) < 5 * np.sqrt(p_true / n) + 2e-5
            assert abs((x == c).mean() - (y == c).mean()) < 8 * np.sqrt(p_true / n) + 4e-5
        assert abs((x >= 2).mean() - (1 - np.exp(-lam) * (1 + lam))) < 5 * np.sqrt(lam * lam / n) + 2e-5


# ------------------------------------------------------------------ GPU
gpu = pytest.mark.gpu


@gpu
@pytest.mark.parametrize("name", G14_CASES)
def test_hip_native_equals_reference_on_native_fields(golden, name):
    """HIP native mode against the reference itself run on the device-native fields (golden G14): exact counts."""
    import torch
    from v2v_amd import v2e
    g = golden("g14_v2e_native.npz")
    video, args, seed, clip, want = _case14(g, name)
    k = video.shape[0] - 1
    counts = torch.zeros((1, 2), dtype=torch.int64, device="cuda")
    out = v2e.v2e_voxel_batch(torch.from_numpy(video)[None].cuda(), v2e.make_params(*args), bin_mode="sum", num_bins=k, seed=seed,
                              clip_id0=clip, out_dtype=torch.float64, counts=counts)
    assert np.array_equal(out[0, 0].cpu().numpy(), want)
    out32 = v2e.v2e_voxel_batch(torch.from_numpy(video)[None].cuda(), v2e.make_params(*args), bin_mode="sum", num_bins=k, seed=seed,
                                clip_id0=clip)
    assert np.array_equal(out32[0, 0].cpu().numpy().astype(np.float64), want)


@gpu
def test_hip_native_shot_noise_statistics():
    """Shot-noise-only model on a constant clip: every event is a shot event.  Device totals against the Poisson law of
    v2v_core_v2e.py:65-105 (rate/2 x dt per polarity, pixel and frame; the per-frame mean normalisation makes the frame mean
    exactly that) and against np.random.poisson draws of the same means."""
    import torch
    from v2v_amd import v2e
    b, n, h, w = 64, 2, 128, 128                               # first frame pair only: later pairs also fire real events that
    video = torch.full((b, n, h, w), 90, dtype=torch.uint8, device="cuda")   # compensate the shot events fed back into the base (:547-548)
    rate, fps = 6.0, 24
    args = [fps, "pn_related", 0.5, 0.1, 0.0, 0.1, 0, 0, 0, rate, 0.1, 0.1]
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    out = v2e.v2e_voxel_batch(video, v2e.make_params(*args), bin_mode="sum", num_bins=n - 1, seed=2024, clip_id0=0, out_dtype=torch.float64,
                              counts=counts)
    lam = rate / 2 / fps                                       # mean shot events per pixel, frame and polarity
    samples = b * (n - 1) * h * w
    tot = counts.cpu().numpy().sum(axis=0).astype(np.float64)
    for t in tot:                                              # ON and OFF totals: Poisson(samples * lam)
        assert abs(t / samples - lam) < 5 * np.sqrt(lam / samples) + 1e-5, (t / samples, lam)
    vox = out.cpu().numpy()                                    # ON - OFF per pixel: Skellam(lam_p, lam_n), means spread by the thresholds
    assert abs(vox.mean()) < 5 * np.sqrt(2 * lam / samples) and abs(vox.var() - 2 * lam) < 0.03 * 2 * lam
    ref = np.random.default_rng(1)
    sk = ref.poisson(lam, size=samples) - ref.poisson(lam, size=samples)
    for v in (-1, 0, 1):
        assert abs((vox == v).mean() - (sk == v).mean()) < 2e-3
    # per-clip totals are independent draws: their spread is Poisson too
    per_clip = counts.cpu().numpy().astype(np.float64)
    assert abs(per_clip.var(axis=0).mean() / (h * w * lam) - 1) < 0.5


@gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_replay_equals_reference_golden(golden, name):
    import torch
    from v2v_amd import v2e
    g = golden("g9_v2e.npz")
    video, args, fields, want = _case(g, name)
    params = v2e.make_params(*args)
    k = video.shape[0] - 1
    rep = {kk: torch.from_numpy(np.ascontiguousarray(a))[None] for kk, a in fields.items()}
    out = v2e.v2e_voxel_batch(torch.from_numpy(video)[None].cuda(), params, bin_mode="sum", num_bins=k, rng_mode="replay",
                              replay=rep, out_dtype=torch.float64)
    assert np.array_equal(out[0, 0].cpu().numpy(), want)


@gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_video_to_voxel_dropin_numpy_stream(golden, name):
    """Full drop-in call: seed -> host draws in the reference's order -> GPU.  np.exp(float32) of the leak-rate
    factor is host-SIMD dependent (<= 1 ulp), so allow a vanishing number of flipped counts."""
    from v2v_amd import v2e
    g = golden("g9_v2e.npz")
    video, args, fields, want = _case(g, name)
    got = v2e.video_to_voxel(video, *args, seed=11, rng="numpy")
    assert got.dtype == np.float64 and got.shape == want.shape
    assert np.count_nonzero(got != want) <= 2


@gpu
@pytest.mark.parametrize("dt", [np.uint8, np.float32])
@pytest.mark.parametrize("model,cutoff,leak,refr,shot", [("pn_related", 30, 0.1, 0, 5.0), ("spatial_independent", 0, 0, 0, 0),
                                                          ("spatial_temporal_independent", 30, 0.1, 1 / 240, 5.0),
                                                          ("pn_related", 0, 0.1, 0, 0)])
@pytest.mark.parametrize("bin_mode", ["sum", "bilinear"])
def test_hip_philox_equals_c_oracle(oracle_c, luts, dt, model, cutoff, leak, refr, shot, bin_mode):
    import torch
    from v2v_amd import v2e
    b, n, h, w = 3, 11, 24, 40
    video = np.stack([O.synth_clip_s1(n, h, w, seed=300 + i, dtype=dt) for i in range(b)])
    args = [24, model, 0.5, 0.1, 0.0, 0.1, cutoff, leak, refr, shot, 0.1, 0.1]
    bm = oracle_c.BIN_SUM if bin_mode == "sum" else oracle_c.BIN_BILINEAR
    want, totals = oracle_c.v2e_voxel(video, oracle_c.v2e_params(*args), luts, seed=0xABCDEF123, clip_id0=5, bin_mode=bm,
                                      num_bins=5, frames_per_bin=2 if bin_mode == "sum" else 1)
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    got = v2e.v2e_voxel_batch(torch.from_numpy(video).cuda(), v2e.make_params(*args), bin_mode=bin_mode, num_bins=5,
                              frames_per_bin=2 if bin_mode == "sum" else 1, seed=0xABCDEF123, clip_id0=5,
                              out_dtype=torch.float64, counts=counts)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(counts.cpu().numpy(), totals)
    got32 = v2e.v2e_voxel_batch(torch.from_numpy(video).cuda(), v2e.make_params(*args), bin_mode=bin_mode, num_bins=5,
                                frames_per_bin=2 if bin_mode == "sum" else 1, seed=0xABCDEF123, clip_id0=5)
    np.testing.assert_allclose(got32.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    # batch / shard invariance
    part = v2e.v2e_voxel_batch(torch.from_numpy(video[1:]).cuda(), v2e.make_params(*args), bin_mode=bin_mode, num_bins=5,
                               frames_per_bin=2 if bin_mode == "sum" else 1, seed=0xABCDEF123, clip_id0=6, out_dtype=torch.float64)
    assert torch.equal(part, got[1:])


@gpu
@pytest.mark.parametrize("h,w", [(1, 1), (3, 5), (33, 31), (16, 18), (719, 1279)])
def test_hip_v2e_ragged_and_large_frames(oracle_c, luts, h, w):
    """Frame sizes that are not multiples of the 4-pixel work-item or of the workgroup (the scalar tail paths, a partly empty last
    workgroup, the per-frame sums of the shot-noise pre-pass over a ragged frame), up to an odd near-720p frame: C oracle, exact counts."""
    import torch
    from v2v_amd import v2e
    args = [24, "pn_related", 0.5, 0.1, 0.0, 0.1, 30, 0.1, 0, 5.0, 0.1, 0.1]
    for dt in (np.uint8, np.float32):
        video = O.synth_clip_s1(7, h, w, seed=41, dtype=dt)[None]
        want, totals = oracle_c.v2e_voxel(video, oracle_c.v2e_params(*args), luts, seed=77, clip_id0=9, bin_mode=oracle_c.BIN_SUM, num_bins=3, frames_per_bin=2)
        counts = torch.zeros((1, 2), dtype=torch.int64, device="cuda")
        got = v2e.v2e_voxel_batch(torch.from_numpy(video).cuda(), v2e.make_params(*args), bin_mode="sum", num_bins=3, frames_per_bin=2, seed=77, clip_id0=9,
                                  out_dtype=torch.float64, counts=counts)
        assert np.array_equal(got.cpu().numpy(), want) and np.array_equal(counts.cpu().numpy(), totals)


@gpu
@pytest.mark.parametrize("bin_mode,nb,fpb,n", [("bilinear", 1, 1, 9), ("bilinear", 2, 1, 3), ("bilinear", 12, 1, 6), ("sum", 1, 1, 2), ("sum", 1, 8, 9), ("sum", 8, 1, 9)])
def test_hip_v2e_bin_count_extremes(oracle_c, luts, bin_mode, nb, fpb, n):
    """The v2e kernel's own binning at the ends of its parameters (one / two bilinear bins, more bins than pairs, a single pair, one plane
    of 8 pairs, a bin per pair): C oracle, float64 exact."""
    import torch
    from v2v_amd import v2e
    args = [24, "pn_related", 0.5, 0.1, 0.0, 0.1, 30, 0.1, 0, 5.0, 0.1, 0.1]
    video = np.stack([O.synth_clip_s1(n, 24, 36, seed=70 + i, dtype=np.uint8) for i in range(2)])
    bm = oracle_c.BIN_SUM if bin_mode == "sum" else oracle_c.BIN_BILINEAR
    want, tot = oracle_c.v2e_voxel(video, oracle_c.v2e_params(*args), luts, seed=31, clip_id0=2, bin_mode=bm, num_bins=nb, frames_per_bin=fpb)
    counts = torch.zeros((2, 2), dtype=torch.int64, device="cuda")
    got = v2e.v2e_voxel_batch(torch.from_numpy(video).cuda(), v2e.make_params(*args), bin_mode=bin_mode, num_bins=nb, frames_per_bin=fpb, seed=31, clip_id0=2,
                              out_dtype=torch.float64, counts=counts)
    assert got.shape == want.shape and np.array_equal(got.cpu().numpy(), want) and np.array_equal(counts.cpu().numpy(), tot)


@gpu
@pytest.mark.parametrize("dt", [np.uint8, np.float32])
@pytest.mark.parametrize("name,args", [
    ("tiny_thresholds_wide_spread", [24, "pn_related", 0.02, 0.05, 0.0, 0.02, 0, 0, 0, 0, 0.1, 0.1]),
    ("shot_rate_100hz_low_fps", [5, "pn_related", 0.5, 0.1, 0.0, 0.1, 30, 0.1, 0, 100.0, 0.1, 1.0]),
    ("leak_50hz_full_jitter", [24, "spatial_independent", 0.3, 0.05, 0.1, 0.05, 0, 50.0, 0, 0.0, 1.0, 0.1]),
    ("refractory_longer_than_the_clip", [24, "spatial_temporal_independent", 0.2, 0.05, 0.0, 0.05, 200, 0.1, 10.0, 5.0, 0.1, 0.1]),
    ("lowpass_1hz_at_240fps", [240, "pn_related", 0.1, 0.02, 0.05, 0.02, 1, 0.1, 1 / 480, 1.0, 0.1, 0.1]),
])
def test_hip_v2e_parameter_extremes(oracle_c, luts, dt, name, args):
    """The v2e model's parameters at the ends of what its arithmetic takes (thresholds of a few hundredths with a spread larger than the mean,
    a shot-noise rate of 100 Hz at 5 fps -- 10 expected noise events per pixel, frame and polarity -- a 50 Hz leak with full jitter, a
    refractory period longer than the clip, a 1 Hz low-pass at 240 fps): HIP == C oracle, float64 counts and ON/OFF totals exact."""
    import torch
    from v2v_amd import v2e
    video = np.stack([O.synth_clip_s1(13, 20, 28, seed=90 + i, dtype=dt) for i in range(2)])
    want, tot = oracle_c.v2e_voxel(video, oracle_c.v2e_params(*args), luts, seed=0xFACE, clip_id0=4, bin_mode=oracle_c.BIN_SUM, num_bins=4, frames_per_bin=3)
    counts = torch.zeros((2, 2), dtype=torch.int64, device="cuda")
    got = v2e.v2e_voxel_batch(torch.from_numpy(video).cuda(), v2e.make_params(*args), bin_mode="sum", num_bins=4, frames_per_bin=3, seed=0xFACE, clip_id0=4,
                              out_dtype=torch.float64, counts=counts)
    assert np.array_equal(got.cpu().numpy(), want), name
    assert np.array_equal(counts.cpu().numpy(), tot), name


@gpu
def test_hip_native_shot_noise_at_large_rates_and_its_refusal():
    """The native Poisson sampler inverts up to 64 events per pixel, frame and polarity.  Inside its stated domain (shot_noise_rate_hz <= 32 fps,
    i.e. <= 16 expected events) the per-pixel mean is the law's; beyond it the entry point refuses (V2V_ERR_PARAM -> ValueError) instead of
    truncating silently -- the replay mode takes np.random.poisson's counts at any rate."""
    import torch
    from v2v_amd import v2e
    b, n, h, w = 16, 2, 128, 128
    video = torch.full((b, n, h, w), 90, dtype=torch.uint8, device="cuda")
    for rate, fps in ((40.0, 5), (160.0, 5)):                                     # 4 and 16 expected events per pixel, frame and polarity
        args = [fps, "pn_related", 0.5, 0.1, 0.0, 0.1, 0, 0, 0, rate, 0.1, 0.1]
        counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
        v2e.v2e_voxel_batch(video, v2e.make_params(*args), bin_mode="sum", num_bins=1, seed=7, counts=counts)
        lam, samples = rate / 2 / fps, b * h * w
        for t in counts.cpu().numpy().sum(axis=0) / samples:
            assert abs(t - lam) < 5 * np.sqrt(lam / samples) + 2e-3 * lam, (rate, t, lam)
    with pytest.raises(ValueError, match="REPLAY"):
        v2e.v2e_voxel_batch(video, v2e.make_params(5, "pn_related", 0.5, 0.1, 0.0, 0.1, 0, 0, 0, 161.0, 0.1, 0.1), bin_mode="sum", num_bins=1, seed=7)


@gpu
def test_hip_v2e_errors():
    import torch
    from v2v_amd import v2e
    with pytest.raises(ValueError):
        v2e.make_params(24, "spatial_independent_temporal_changing", 0.5, 0.1, 0, 0.1, 0, 0, 0, 0, 0.1, 0.1)
    f = torch.zeros((1, 8, 8, 8), dtype=torch.uint8, device="cuda")
    p = v2e.make_params(24, "pn_related", 0.5, 0.1, 0, 0.1, 0, 0, 0, 0, 0.1, 0.1)
    with pytest.raises(AssertionError):
        v2e.v2e_voxel_batch(f, p, num_bins=5)
    with pytest.raises(ValueError):
        v2e.v2e_voxel_batch(f, p, num_bins=7, rng_mode="replay")


@gpu
def test_hip_lowpass_table_falls_back_when_dt_over_tau_is_not_one_float32(oracle_c, luts):
    """The specialised float32 instances tabulate the low-pass factors for ONE float32(dt / tau) (true for every ordinary frame
    rate / cut-off: the float64 wobble of i/fps - (i-1)/fps vanishes in the cast).  This cut-off is CONSTRUCTED so that dt / tau
    straddles a float32 rounding boundary (two distinct values over 31 frames at 24 fps): the launcher must notice and take the
    run-time-feature kernel -- same results as the oracle either way, and as the ordinary cut-off right beside it."""
    import torch
    from v2v_amd import esim, v2e
    fps, cutoff = 24.0, 26.738033171514086
    i = np.arange(1, 32, dtype=np.float64)
    tau = 1 / (np.pi * 2 * cutoff)
    assert len(set(((i / fps - (i - 1) / fps) / tau).astype(np.float32).tolist())) == 2          # the premise of this test
    frames = esim.synth_clips(3, 32, 64, 64, dtype=torch.float32, seed=31, clip_id0=0)
    for co in (cutoff, 30.0):
        args = [fps, "pn_related", 0.5, 0.1, 0.0, 0.1, co, 0.1, 0, 5.0, 0.1, 0.1]
        got = v2e.v2e_voxel_batch(frames, v2e.make_params(*args), bin_mode="sum", num_bins=31, seed=9, clip_id0=0)
        want, _ = oracle_c.v2e_voxel(frames.cpu().numpy(), oracle_c.v2e_params(*args), luts, seed=9, clip_id0=0,
                                     bin_mode=oracle_c.BIN_SUM, num_bins=31)
        assert np.array_equal(got.cpu().numpy().astype(np.float64), want), co
