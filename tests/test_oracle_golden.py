"""The CPU oracle (oracle/v2v_oracle.py + oracle/v2v_oracle.c) against the golden vectors captured
from the imported reference (tests/golden/make_goldens.py).  Runs without a GPU."""
import numpy as np
import pytest

from oracle import v2v_oracle as O


# ---------------------------------------------------------------- G1 log tables
def test_g1_luts_match_direct_formula_here(luts):
    # On this host NumPy may dispatch a different SIMD kernel than where the goldens were made; the
    # direct formula must agree to <= 1 ulp, and the LUT (not the formula) is what parity uses.
    d64 = O.esim_log_direct(np.arange(256, dtype=np.uint8))
    d32 = O.esim_log_direct(np.arange(256, dtype=np.float32))
    assert d64.dtype == np.float64 and d32.dtype == np.float32
    assert np.all(np.abs(d64 - luts["lut64"]) <= np.spacing(np.abs(luts["lut64"])))
    assert np.all(np.abs(d32 - luts["lut32"]) <= np.spacing(np.abs(luts["lut32"])))
    assert np.array_equal(O.v2e_linlog_direct(np.arange(256, dtype=np.uint8)), luts["v2e32"]) or \
        np.all(np.abs(O.v2e_linlog_direct(np.arange(256)) - luts["v2e32"]) <= np.spacing(np.abs(luts["v2e32"])))


def test_g1_lut_monotone_and_endpoints(luts):
    assert np.all(np.diff(luts["lut64"]) > 0) and np.all(np.diff(luts["lut32"]) > 0)
    assert luts["lut64"][0] == np.log(0.001)
    assert abs(luts["lut64"][255] - np.log(1.001)) < 1e-15


# ---------------------------------------------------------------- G2/G3 noise-free ESIM
@pytest.mark.parametrize("tag,cp,cn", [("sym", 0.2, 0.2), ("asym", 0.31, 0.47)])
@pytest.mark.parametrize("seed", [5, 6])
@pytest.mark.parametrize("dt_tag,dt", [("u8", np.uint8), ("f32", np.float32)])
def test_g2_esim_clean(golden, oracle_c, luts, tag, cp, cn, seed, dt_tag, dt):
    g = golden("g2_esim_clean.npz")
    video = g["video"].astype(dt)
    want = g[f"{tag}_s{seed}_{dt_tag}"].astype(np.float64)
    np.random.seed(seed)
    got = O.esim_video_to_voxel(video, cp, cn, 0.0, 0.0, 0.0, False, use_lut=True)
    assert np.array_equal(got, want)
    # scalar C twin, replaying the same MT19937 fields
    np.random.seed(seed)
    fields = O.draw_replay_fields(*video.shape)
    k = video.shape[0] - 1
    vox, totals = oracle_c.esim_voxel(video[None], [cp, cn, 0.0, 0.0, 0.0], luts, rng_mode=oracle_c.RNG_REPLAY,
                                      bin_mode=oracle_c.BIN_SUM, num_bins=k, frames_per_bin=1, replay=fields)
    assert np.array_equal(vox[0, 0], want)
    assert totals[0, 0] == np.sum(want[want > 0]) and totals[0, 1] == -np.sum(want[want < 0])


# ---------------------------------------------------------------- G4 noisy ESIM (internal + external)
@pytest.mark.parametrize("ext", [False, True])
@pytest.mark.parametrize("dt_tag,dt", [("u8", np.uint8), ("f32", np.float32)])
def test_g4_esim_noisy(golden, oracle_c, luts, ext, dt_tag, dt):
    g = golden("g4_esim_noisy.npz")
    video = g["video"].astype(dt)
    p = g["params"]
    want = g[f"ext{int(ext)}_{dt_tag}"]
    np.random.seed(int(g["seed"]))
    got = O.esim_video_to_voxel(video, *p, put_noise_external=ext, use_lut=True)
    assert np.array_equal(got, want)
    np.random.seed(int(g["seed"]))
    fields = O.draw_replay_fields(*video.shape)
    k = video.shape[0] - 1
    vox, _ = oracle_c.esim_voxel(video[None], p, luts, noise_external=ext, rng_mode=oracle_c.RNG_REPLAY,
                                 bin_mode=oracle_c.BIN_SUM, num_bins=k, frames_per_bin=1, replay=fields)
    assert np.array_equal(vox[0, 0], want)


# ---------------------------------------------------------------- G5 np.floor_divide
def test_g5_floor_divide(golden, oracle_c):
    g = golden("g5_floor_divide.npz")
    assert np.array_equal(oracle_c.floor_divide(g["a"], g["b"]), g["q"])
    py = np.array([O.floor_divide_scalar(float(a), float(b)) for a, b in zip(g["a"], g["b"])])
    assert np.array_equal(py, g["q"])
    assert np.array_equal(np.floor_divide(g["a"], g["b"]), g["q"])       # numpy on this host agrees
    # the survey's listed near-tie: floor(a/b) would say 20
    assert O.floor_divide_scalar(24.550921417593624, 1.2275460708796813) == 19.0
    assert np.floor(24.550921417593624 / 1.2275460708796813) == 20.0


# ---------------------------------------------------------------- G6 imgs_to_voxels
def test_g6_imgs_to_voxels(golden):
    g = golden("g6_imgs_to_voxels.npz")
    np.random.seed(int(g["seed"]))
    params, vox = O.imgs_to_voxels(g["video"], 5, 1, use_lut=True)
    keys = ["pos_thres", "neg_thres", "base_noise_std", "hot_pixel_fraction", "hot_pixel_std"]
    assert np.array_equal(np.array([params[k] for k in keys]), g["params"])
    assert vox.shape == (4, 5, 32, 32) and np.array_equal(vox, g["voxels"])
    np.random.seed(int(g["seed2"]))
    params2, vox2 = O.imgs_to_voxels(g["video"], 5, 2, scale_noise_strength=True, use_lut=True)
    assert np.array_equal(np.array([params2[k] for k in keys]), g["params2"])
    assert np.array_equal(vox2, g["voxels2"])
    with pytest.raises(AssertionError):
        O.imgs_to_voxels(g["video"][:20], 5, 1)


# ---------------------------------------------------------------- G7 temporal-bilinear composition
@pytest.mark.parametrize("k", [2, 7, 31, 39])
def test_g7_bilinear(golden, oracle_c, luts, k):
    g = golden("g7_bilinear.npz")
    assert np.array_equal(O.bilinear_weights(k, 5), g[f"w_K{k}"])
    counts = g[f"counts_K{k}"].astype(np.float64)
    assert np.array_equal(O.bin_bilinear(counts, 5), g[f"voxel_K{k}"])
    w = O.bilinear_weights(k, 5)
    assert np.allclose(w.sum(axis=0), 1.0) and w.min() >= 0.0


# ---------------------------------------------------------------- G8 event-list voxelisers
def test_g8_make_voxel(golden, oracle_c):
    g = golden("g8_make_voxel.npz")
    evs = [g["ts"], g["xs"], g["ys"], g["ps"]]
    assert np.array_equal(O.make_voxel(evs, 5, 16, 24, False), g["discrete"])
    assert np.array_equal(O.make_voxel(evs, 5, 16, 24, True), g["interpolated"])
    assert np.array_equal(O.make_voxel([a[:0] for a in evs], 5, 16, 24, True), g["empty"])
    assert not g["empty"].any()
    ts_us = ((g["ts"] - g["ts"][0]) * 1e6).astype(np.int64)
    assert np.array_equal(oracle_c.make_voxel(ts_us, g["xs"], g["ys"], g["ps"], 5, 16, 24, False), g["discrete"])
    assert np.array_equal(oracle_c.make_voxel(ts_us, g["xs"], g["ys"], g["ps"], 5, 16, 24, True), g["interpolated"])
    pf = (g["ps"] * 2 - 1).astype(np.float64)
    assert np.array_equal(O.events_to_voxel(g["xs"], g["ys"], g["ts"], pf, 5, (16, 24)), g["events_to_voxel"])


# ---------------------------------------------------------------- Philox / native fields
def test_philox_known_answers(oracle_c):
    # Random123 kat_vectors, philox4x32 10 rounds
    kats = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
            ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
            ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
             [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kats:
        assert oracle_c.philox4x32(ctr, key) == want
        got = O.philox4x32(*[np.uint32(c) for c in ctr], key[0], key[1])
        assert [int(x) for x in got] == want


def _icdf_table(path):
    """Parse a generated gauss_icdf.inc: 8192 C99 hex float literals."""
    import re
    vals = [float.fromhex(v) for v in re.findall(r"-?0x[0-9a-f.]+p[-+]?\d+", open(path).read())]
    assert len(vals) == 8192
    return np.array(vals, dtype=np.float64).astype(np.float32)


def test_icdf_table_is_shared_and_rederivable():
    """The Gaussian generator's table is DATA: the library's and the oracle's copies are the same text, and the committed
    values are what tools/gen_gauss_icdf.py derives from scipy.special.ndtri today."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    a, b = os.path.join(root, "v2v_amd", "csrc", "v2v_gauss_icdf.inc"), os.path.join(root, "oracle", "gauss_icdf.inc")
    assert open(a).read() == open(b).read()
    tab = _icdf_table(b)
    spec = importlib.util.spec_from_file_location("gen_gauss_icdf", os.path.join(root, "tools", "gen_gauss_icdf.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    z = gen.table()
    # ndtri may move by an ulp between scipy builds: allow one float32 ulp, require almost all entries identical
    assert (tab == z).mean() > 0.999 and np.allclose(tab, z, rtol=2e-7, atol=0)


def test_icdf_exhaustive(oracle_c):
    """All 2^16 half-words, in both halves of the word: value = table[(n >> 2) & 0x1FFF] with the sign of bit 15 (the two low
    bits unused), accuracy against float64 Phi^-1 on the 2^14 midpoint grid, and the exact moments of the distribution."""
    import os
    from scipy.special import ndtri
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tab = _icdf_table(os.path.join(root, "oracle", "gauss_icdf.inc"))
    n = np.arange(1 << 16, dtype=np.uint32)
    hi, _ = oracle_c.gauss16_many(n << np.uint32(16))
    _, lo = oracle_c.gauss16_many(n)
    assert np.array_equal(hi.view(np.uint32), lo.view(np.uint32))              # both halves go through the same map
    idx = ((n >> 2) & 0x1FFF).astype(np.int64)
    want = np.where(n & 0x8000, -tab[idx], tab[idx]).astype(np.float32)
    assert np.array_equal(hi.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(hi[:32768], -hi[32768:]) and hi[:32768].min() > 0   # sign bit = top bit, never zero
    z = ndtri(0.5 + (idx[:32768] + 0.5) / 16384.0)
    assert np.abs(hi[:32768].astype(np.float64) / z - 1.0).max() < 2e-7       # float32 rounding of the exact quantile
    g = hi.astype(np.float64)
    assert abs(g.mean()) == 0.0 and abs((g ** 2).mean() - 1.0) < 1e-4 and abs((g ** 4).mean() - 3.0) < 4e-3 and 4.0 < abs(g).max() < 4.01
    # two deviates of one word: independent halves
    g0, g1 = oracle_c.gauss16(0x12345678)
    assert g0 == hi[0x1234] and g1 == hi[0x5678]


def test_native_fields_statistics(oracle_c):
    r7 = oracle_c.noise_rounds()                                                    # the per-step fields' round count
    g = oracle_c.philox_gauss_field(2024, 3, 5, 1 << 18, rounds=r7)
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1) < 0.01
    assert abs(((g - g.mean()) ** 3).mean()) < 0.03 and abs((g ** 4).mean() - 3) < 0.08
    gb = oracle_c.philox_gauss_field(2024, 3, 5, 1 << 18, comp=1, rounds=r7)        # second deviate of every word
    assert abs(gb.mean()) < 0.01 and abs(gb.std() - 1) < 0.01 and abs(np.corrcoef(g, gb)[0, 1]) < 0.01
    assert abs(np.corrcoef(g * g, gb * gb)[0, 1]) < 0.01                            # the two halves of a word are independent
    u = oracle_c.philox_uniform_field(2024, 3, 0, 1 << 16)
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    assert np.array_equal(u, O.philox_uniform53(2024, 3, 0, 1 << 16))
    # independence across clips / fields / pixels: no duplicated blocks
    g2 = oracle_c.philox_gauss_field(2024, 4, 5, 1 << 12)
    assert abs(np.corrcoef(g[:1 << 12], g2)[0, 1]) < 0.06
    from scipy import stats
    assert stats.kstest(g.astype(np.float64), "norm").pvalue > 1e-3
    assert stats.kstest(gb.astype(np.float64), "norm").pvalue > 1e-3


# ---------------------------------------------------------------- G11: reference run on the native fields
@pytest.mark.parametrize("ext", [False, True])
@pytest.mark.parametrize("dt_tag,dt", [("u8", np.uint8), ("f32", np.float32)])
def test_g11_philox_fed(golden, oracle_c, luts, ext, dt_tag, dt):
    g = golden("g11_philox_fed.npz")
    seed, clip = int(g["seed"]), int(g["clip_id"])
    video = g["video"].astype(dt)
    want = g[f"ext{int(ext)}_{dt_tag}"]
    assert np.array_equal(oracle_c.philox_gauss_field(seed, clip, 3, 1024, rounds=oracle_c.noise_rounds()), g["gauss_field3"])
    assert np.array_equal(oracle_c.philox_gauss_field(seed, clip, 3, 1024, comp=1, rounds=oracle_c.noise_rounds()), g["gauss_field3b"])
    assert np.array_equal(oracle_c.philox_uniform_field(seed, clip, 0, 1024), g["uniform_field0"])
    got = O.esim_video_to_voxel(video, *g["params"], put_noise_external=ext,
                                rng=O.PhiloxFieldRNG(seed, clip), use_lut=True)
    assert np.array_equal(got, want)
    k = video.shape[0] - 1
    vox, _ = oracle_c.esim_voxel(video[None], g["params"], luts, noise_external=ext, rng_mode=oracle_c.RNG_PHILOX,
                                 seed=seed, clip_id0=clip, bin_mode=oracle_c.BIN_SUM, num_bins=k, frames_per_bin=1)
    assert np.array_equal(vox[0, 0], want)


# ---------------------------------------------------------------- C twin == numpy restatement, binning modes
def test_c_twin_binning_modes(oracle_c, luts):
    video = O.synth_clip_s1(21, 24, 20, seed=5, dtype=np.uint8)
    p = [0.15, 0.2, 0.03, 0.01, 0.4]
    counts = O.esim_video_to_voxel(video, *p, rng=O.PhiloxFieldRNG(77, 2), use_lut=True)
    vs, _ = oracle_c.esim_voxel(video[None], p, luts, seed=77, clip_id0=2, bin_mode=oracle_c.BIN_SUM,
                                num_bins=5, frames_per_bin=2)
    assert np.array_equal(vs[0], O.bin_sum(counts, 5, 2))
    vb, _ = oracle_c.esim_voxel(video[None], p, luts, seed=77, clip_id0=2, bin_mode=oracle_c.BIN_BILINEAR, num_bins=5)
    assert np.array_equal(vb[0], O.bin_bilinear(counts, 5))
