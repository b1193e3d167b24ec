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


def test_box_muller_exact_fma_semantics(oracle_c):
    """The 16+16-bit fp32 Box-Muller (gauss16) is defined with single-rounded fmaf; re-derive values in exact
    rational arithmetic (a third, independent statement of the sequence) to prove the C build really single-rounds
    (i.e. host == device definition)."""
    from fractions import Fraction as Fr

    def rnd(x):          # round a Fraction to nearest-even float32
        return np.float32(float(x)) if abs(float(x)) < 1e30 else np.float32(x)

    def f32_round(fr):
        # exact: go through float64 only when harmless -> use integer scaling instead
        if fr == 0:
            return np.float32(0.0)
        import math
        s = -1 if fr < 0 else 1
        fr = abs(fr)
        e = math.floor(math.log2(float(fr)))
        while Fr(2) ** e > fr:
            e -= 1
        while Fr(2) ** (e + 1) <= fr:
            e += 1
        q = fr / Fr(2) ** (e - 23)
        n = q.numerator // q.denominator
        rem = q - n
        if rem > Fr(1, 2) or (rem == Fr(1, 2) and n % 2 == 1):
            n += 1
        return np.float32(s * float(Fr(n) * Fr(2) ** (e - 23)))

    def fma(a, b, c):
        return f32_round(Fr(float(a)) * Fr(float(b)) + Fr(float(c)))

    def mul(a, b):
        return f32_round(Fr(float(a)) * Fr(float(b)))

    def add(a, b):
        return f32_round(Fr(float(a)) + Fr(float(b)))

    F = np.float32

    def as_u32(x):
        return int(np.array(x, dtype=np.float32).view(np.uint32))

    def as_f32(u):
        return np.array(u & 0xFFFFFFFF, dtype=np.uint32).view(np.float32)[()]

    def horner(coefs, v):          # coefs highest degree first
        p = F(float.fromhex(coefs[0]))
        for c in coefs[1:]:
            p = fma(p, v, F(float.fromhex(c)))
        return p

    def gauss16_ref(w):
        xh = add(F(w >> 16), F(0.5))
        xb = as_u32(xh)
        ef = F((xb >> 23) - 143)
        f = add(as_f32((xb & 0x7FFFFF) | 0x3F800000), F(-1.0))
        L = horner(["-0x1.57869cp-6", "0x1.bb3e08p-4", "-0x1.10adbap-2", "0x1.cc4bd8p-2", "-0x1.4fa778p-1", "0x1.ff5d72p-1",
                    "-0x1.fffc7ap+0", "-0x1.9cde6p-22"], f)
        t = fma(ef, F(float.fromhex("-0x1.62e43p+0")), L)
        th = mul(t, F(float.fromhex("0x1.007aa6p-1")))
        y = as_f32(0x5f374000 - (as_u32(t) >> 1))
        q = fma(-th, mul(y, y), F(float.fromhex("0x1.804d8ep+0")))
        y = mul(y, q)
        q = fma(-th, mul(y, y), F(float.fromhex("0x1.803d52p+0")))
        r = mul(mul(y, q), t)
        x = fma(F(w & 0xFFFF), F(float.fromhex("0x1.921fb6p-15")), F(float.fromhex("-0x1.921e24p+0")))
        z = mul(x, x)
        sn = mul(x, horner(["-0x1.12b318p-12", "0x1.813e8ap-7", "-0x1.e2b092p-3", "0x1.6a09d4p+0"], z))
        c = horner(["0x1.12ae8p-15", "-0x1.00cc2ap-9", "0x1.e2aebap-5", "-0x1.6a09bap-1", "0x1.6a09e6p+0"], z)
        t1 = mul(r, sn)
        return fma(-t1, sn, r), mul(t1, c)

    g = np.random.default_rng(3)
    for w in [0, 0xFFFFFFFF, 0xFFFF0000, 0x0000FFFF, 0x80008000, 0x7FFF7FFF] + [int(v) for v in g.integers(0, 2**32, size=40)]:
        want = gauss16_ref(w)
        got = oracle_c.gauss16(w)
        # value equality (the rational emulation does not track the sign of a zero result)
        assert got[0] == np.float32(want[0]) and got[1] == np.float32(want[1]), (hex(w), got, want)


def test_native_fields_statistics(oracle_c):
    r7 = oracle_c.noise_rounds()                                                    # the per-step fields' round count
    g = oracle_c.philox_gauss_field(2024, 3, 5, 1 << 18, rounds=r7)
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1) < 0.01
    assert abs(((g - g.mean()) ** 3).mean()) < 0.03 and abs((g ** 4).mean() - 3) < 0.08
    gb = oracle_c.philox_gauss_field(2024, 3, 5, 1 << 18, comp=1, rounds=r7)        # second normal of every pair
    assert abs(gb.mean()) < 0.01 and abs(gb.std() - 1) < 0.01 and abs(np.corrcoef(g, gb)[0, 1]) < 0.01
    assert abs(np.corrcoef(g * g, gb * gb)[0, 1]) < 0.01                            # radius shared, yet independent (Box-Muller)
    u = oracle_c.philox_uniform_field(2024, 3, 0, 1 << 16)
    assert 0 <= u.min() and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    assert np.array_equal(u, O.philox_uniform53(2024, 3, 0, 1 << 16))
    # independence across clips / fields / pixels: no duplicated blocks
    g2 = oracle_c.philox_gauss_field(2024, 4, 5, 1 << 12)
    assert abs(np.corrcoef(g[:1 << 12], g2)[0, 1]) < 0.06
    from scipy import stats
    assert stats.kstest(g.astype(np.float64), "norm").pvalue > 1e-3
    assert stats.kstest(gb.astype(np.float64), "norm").pvalue > 1e-3


def test_gauss16_exhaustive_accuracy(oracle_c):
    """Every one of the 2^16 radii and 2^16 angles of the 16+16-bit Box-Muller against float64 math, and the exact moments of
    the discrete generator (it is separable: g0 = r[n] * cos2x[a], g1 = r[n] * sin2x[a])."""
    n = np.arange(1 << 16, dtype=np.uint32)
    # radius: angle word a such that cos 2x ~ 1: a = 32768 -> x = pi/131072 -> g0 = r cos(2x) ~ r (1 - 1.2e-9)
    g0, _ = oracle_c.gauss16_many((n << np.uint32(16)) | np.uint32(32768))
    r_true = np.sqrt(-2.0 * np.log((n.astype(np.float64) + 0.5) / 65536.0))
    rel = np.abs(g0 / r_true - 1.0)
    assert rel[:65000].max() < 1e-5, rel[:65000].max()             # r > 0.128 (99.2 % of the radii): 1e-5 relative
    # the rest: -2 ln u = 2ln2 + L(f) cancels towards u -> 1, the absolute error of L (4e-7) shows: |dr| <= 5e-5 on
    # radii below 0.13, never negative
    assert np.abs(g0 - r_true).max() < 5e-5 and np.abs(g0 - r_true)[:65500].max() < 5e-6 and g0.min() > 0
    # angle: fixed radius word n = 0x8000 (r ~ 1.1774)
    a = np.arange(1 << 16, dtype=np.uint32)
    c0, s0 = oracle_c.gauss16_many(np.uint32(0x8000 << 16) | a)
    r0 = np.sqrt(-2.0 * np.log((0x8000 + 0.5) / 65536.0))
    th = 2.0 * (np.pi * (a.astype(np.float64) + 0.5) / 65536.0 - np.pi / 2)
    assert np.abs(c0 / r0 - np.cos(th)).max() < 8e-6 and np.abs(s0 / r0 - np.sin(th)).max() < 8e-6   # angular grid step: 9.6e-5
    # exact moments of the discrete distribution (float64 sums over the separable grids)
    r = g0.astype(np.float64) / np.cos(2 * np.pi / 131072)
    c, s = c0.astype(np.float64) / r0, s0.astype(np.float64) / r0
    for trig in (c, s):
        m1, m2, m4 = trig.mean() * r.mean(), (trig ** 2).mean() * (r ** 2).mean(), (trig ** 4).mean() * (r ** 4).mean()
        assert abs(m1) < 1e-6 and abs(m2 - 1.0) < 2e-4 and abs(m4 - 3.0) < 3e-3, (m1, m2, m4)
    assert abs((c * s).mean()) < 1e-7                                # the two normals of a pair are uncorrelated


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
