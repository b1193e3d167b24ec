"""Randomised parity sweep: HIP (through the C ABI) vs the scalar C oracle over random shapes, dtypes, parameters,
binning modes and batch splits.  Float64 output must be bit-exact, float32 output within 1e-5 (relative, and absolute in
units of the grid's largest magnitude), totals exact."""
import os

import numpy as np
import pytest
import torch

from oracle import v2v_oracle as O

pytestmark = pytest.mark.gpu

# V2V_FUZZ_SCALE=k multiplies the number of random cases (soak runs on the GPU box; the default suite stays short)
_SCALE = int(os.environ.get("V2V_FUZZ_SCALE", "1"))


@pytest.mark.parametrize("case", range(24 * _SCALE))
def test_random_esim_case(oracle_c, luts, case):
    from v2v_amd import esim as E
    g = np.random.default_rng(1000 + case)
    b = int(g.integers(1, 5))
    h, w = int(g.integers(1, 40)), int(g.integers(1, 70))
    if g.random() < 0.5:
        w = (w + 3) // 4 * 4                                   # exercise the 4-pixel vector path too
    dt = np.uint8 if g.random() < 0.5 else np.float32
    bilinear = g.random() < 0.5
    if bilinear:
        nb, fpb, k = int(g.integers(1, 8)), 1, int(g.integers(2, 20))
    else:
        nb, fpb = int(g.integers(1, 6)), int(g.integers(1, 4))
        k = nb * fpb * int(g.integers(1, 4))
    n = k + 1
    kind = g.integers(0, 3)
    if kind == 0:
        video = g.integers(0, 256, size=(b, n, h, w)).astype(dt)                       # white noise: worst case for ties
    elif kind == 1:
        video = np.stack([O.synth_clip_s1(n, h, w, seed=int(g.integers(1 << 30)), dtype=dt) for _ in range(b)])
    else:
        video = np.repeat(g.integers(0, 256, size=(b, 1, h, w)), n, axis=1).astype(dt)  # static scene
    params = np.stack([[g.uniform(0.05, 1.0), g.uniform(0.05, 1.0), g.choice([0.0, g.uniform(0, 0.2)]),
                        g.choice([0.0, g.uniform(0, 0.05)]), g.uniform(0, 1.0)] for _ in range(b)])
    if g.random() < 0.3:
        params[:, 1] = params[:, 0]                           # symmetric-threshold specialisation
    ext = bool(g.random() < 0.3)
    seed, cid0 = int(g.integers(1 << 62)), int(g.integers(1 << 20))
    bm = oracle_c.BIN_BILINEAR if bilinear else oracle_c.BIN_SUM
    want, totals = oracle_c.esim_voxel(video, params, luts, noise_external=ext, seed=seed, clip_id0=cid0, bin_mode=bm,
                                       num_bins=nb, frames_per_bin=fpb)
    # small shapes would all take the 1-pixel mapping: pin one at random so that both families of instances are swept
    kw = dict(bin_mode="bilinear" if bilinear else "sum", num_bins=nb, frames_per_bin=fpb, seed=seed, clip_id0=cid0,
              put_noise_external=ext, mapping=["4px", "2px", "1px", "auto"][int(g.integers(0, 4))])
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    got = E.esim_voxel_batch(torch.from_numpy(video).cuda(), params, out_dtype=torch.float64, counts=counts, **kw)
    assert np.array_equal(got.cpu().numpy(), want), (b, n, h, w, dt, kw)
    assert np.array_equal(counts.cpu().numpy(), totals)
    got32 = E.esim_voxel_batch(torch.from_numpy(video).cuda(), params, **kw)
    # float32 grid = the same terms accumulated in float32: a bin whose +/- contributions cancel keeps the rounding of its
    # largest partial sum, so the absolute bound scales with the grid's magnitude (soak case 5798: 1.4e-5 on 0.26 beside 12.0)
    np.testing.assert_allclose(got32.cpu().numpy(), want, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(want).max())))
    if b > 1:                                                  # any split of the batch gives the same clips
        cut = int(g.integers(1, b))
        tail = E.esim_voxel_batch(torch.from_numpy(video[cut:]).cuda(), params[cut:], out_dtype=torch.float64,
                                  **dict(kw, clip_id0=cid0 + cut))
        assert torch.equal(tail, got[cut:])


@pytest.mark.parametrize("case", range(12 * _SCALE))
def test_random_v2e_case(oracle_c, luts, case):
    """v2e model, native RNG: HIP vs the scalar C oracle over random shapes / models / parameter combinations."""
    from v2v_amd import v2e
    g = np.random.default_rng(5000 + case)
    b, h, w = int(g.integers(1, 4)), int(g.integers(2, 30)), int(g.integers(2, 50))
    if g.random() < 0.5:
        w = (w + 3) // 4 * 4
    dt = np.uint8 if g.random() < 0.5 else np.float32
    bilinear = g.random() < 0.4
    nb = int(g.integers(1, 6))
    fpb = 1 if bilinear else int(g.integers(1, 3))
    k = int(g.integers(2, 12)) if bilinear else nb * fpb * int(g.integers(1, 3))
    video = np.stack([O.synth_clip_s1(k + 1, h, w, seed=int(g.integers(1 << 30)), dtype=dt) for _ in range(b)])
    model = list(O.V2E_MODELS)[int(g.integers(0, 3))]
    args = [float(g.choice([24, 30, 29.97])), model, float(g.uniform(0.2, 0.8)), float(g.uniform(0, 0.15)), float(g.uniform(-0.1, 0.1)),
            float(g.uniform(0, 0.15)), float(g.choice([0, 30, 100])), float(g.choice([0, 0.1, 1.0])), float(g.choice([0, 1 / 240, 1 / 60])),
            float(g.choice([0, 5.0, 20.0])), float(g.uniform(0, 0.3)), float(g.uniform(0, 0.3))]
    wrap = bool(g.random() < 0.5)
    seed, cid0 = int(g.integers(1 << 62)), int(g.integers(1 << 20))
    bm = oracle_c.BIN_BILINEAR if bilinear else oracle_c.BIN_SUM
    want, totals = oracle_c.v2e_voxel(video, oracle_c.v2e_params(*args, uint8_wrap=wrap), luts, seed=seed, clip_id0=cid0,
                                      bin_mode=bm, num_bins=nb, frames_per_bin=fpb)
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    got = v2e.v2e_voxel_batch(torch.from_numpy(video).cuda(), v2e.make_params(*args, uint8_wrap=wrap),
                              bin_mode="bilinear" if bilinear else "sum", num_bins=nb, frames_per_bin=fpb, seed=seed, clip_id0=cid0,
                              out_dtype=torch.float64, counts=counts)
    assert np.array_equal(got.cpu().numpy(), want), (args, dt, h, w)
    assert np.array_equal(counts.cpu().numpy(), totals)
    # float32 grid: the production dtype (feature-specialised kernel instances when the launch qualifies)
    got32 = v2e.v2e_voxel_batch(torch.from_numpy(video).cuda(), v2e.make_params(*args, uint8_wrap=wrap),
                                bin_mode="bilinear" if bilinear else "sum", num_bins=nb, frames_per_bin=fpb, seed=seed, clip_id0=cid0,
                                out_dtype=torch.float32, counts=counts.zero_()).cpu().numpy()
    assert np.array_equal(counts.cpu().numpy(), totals)
    if bilinear:
        np.testing.assert_allclose(got32, want, rtol=1e-5, atol=1e-5 * max(1.0, float(np.abs(want).max())))
    else:
        assert np.array_equal(got32, want.astype(np.float32)), (args, dt, h, w)      # integer counts: exact


@pytest.mark.parametrize("case", range(6 * _SCALE))
def test_random_event_lists(case):
    from v2v_amd import voxel
    g = np.random.default_rng(9000 + case)
    n, h, w, nb = int(g.integers(1, 30000)), int(g.integers(1, 100)), int(g.integers(1, 100)), int(g.integers(1, 9))
    ts = np.sort(g.uniform(0.0, g.uniform(1e-3, 2.0), size=n)) + g.uniform(0, 100)
    xs, ys, ps = g.integers(0, w, n), g.integers(0, h, n), g.integers(0, 2, n)
    assert np.array_equal(voxel.make_voxel([ts, xs, ys, ps], h, w, nb, False), O.make_voxel([ts, xs, ys, ps], nb, h, w, False))
    np.testing.assert_allclose(voxel.make_voxel([ts, xs, ys, ps], h, w, nb, True), O.make_voxel([ts, xs, ys, ps], nb, h, w, True),
                               rtol=1e-11, atol=1e-11)
    if n > 1 and ts[-1] > ts[0] and nb > 1:
        pf = (ps * 2 - 1).astype(np.float64)
        np.testing.assert_allclose(voxel.events_to_voxel(xs, ys, ts, pf, nb, (h, w)), O.events_to_voxel(xs, ys, ts, pf, nb, (h, w)),
                                   rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("case", range(20 * _SCALE))
def test_random_frontend_tiles(case):
    """LDS-tiled front-end vs the OpenCV-algorithm restatement (and the gather kernel) on random frames, rectangles,
    crop sizes (both tile widths), flips; includes up-scaling, >2x down-scaling (skipped source rows) and 1-pixel crops."""
    from oracle import frontend_oracle as F
    from v2v_amd import frontend
    g = np.random.default_rng(7000 + case)
    hs, ws = int(g.integers(8, 260)), int(g.integers(8, 300))
    crop = int(g.integers(1, 200)) if case % 4 else int(g.integers(129, 300))          # every 4th case: the 256-wide tile
    cb = int(g.integers(1, min(hs, ws) + 1))
    if case % 5 == 0 and 2 * crop <= min(hs, ws):
        cb = 2 * crop                                                                    # OpenCV's area shortcut
    mi, mj = int(g.integers(0, hs - cb + 1)), int(g.integers(0, ws - cb + 1))
    flip = bool(g.integers(0, 2))
    t = 2
    raw = g.integers(0, 256, size=(t, hs, ws, 3), dtype=np.uint8)
    idxes = [1, 0, 1]
    _, want = F.frontend(raw, cb, mi, mj, flip, crop, idxes, None, None, "gray")
    raw_d = torch.from_numpy(raw).cuda()
    _, gray = frontend.prepare_clip(raw_d, cb, mi, mj, flip, crop, idxes, want_imgs=False)
    assert np.array_equal(gray.cpu().numpy(), want), (hs, ws, crop, cb, mi, mj, flip)
    _, gather = frontend.prepare_clip(raw_d, cb, mi, mj, flip, crop, idxes, want_imgs=True)
    assert torch.equal(gray, gather)


@pytest.mark.parametrize("case", range(12 * _SCALE))
def test_random_frontend_modes(case):
    """The tiled front-end's other instances against the restatement on random frames: colour mode gray_in_bgr_out (three channels
    through both passes, resized frames + the reference's float64 bgr_to_gray), per-frame shake offsets (resize to crop + the
    largest offset, cut per frame), both cvtColor weight sets."""
    from oracle import frontend_oracle as F
    from v2v_amd import frontend
    g = np.random.default_rng(11000 + case)
    hs, ws = int(g.integers(8, 220)), int(g.integers(8, 260))
    crop = int(g.integers(1, 150)) if case % 3 else int(g.integers(129, 200))
    t = 3
    shake = case % 2 == 1
    di = g.integers(-3, 4, size=t) if shake else None
    dj = g.integers(-3, 4, size=t) if shake else None
    cb = int(g.integers(1, min(hs, ws) + 1))
    mi, mj = int(g.integers(0, hs - cb + 1)), int(g.integers(0, ws - cb + 1))
    flip = bool(g.integers(0, 2))
    mode = "gray_in_bgr_out" if case % 4 < 2 else "gray"
    ver = "cv3" if case % 5 == 0 else "cv4"
    raw = g.integers(0, 256, size=(t, hs, ws, 3), dtype=np.uint8)
    idxes = [2, 0, 1, 1]
    want_imgs, want_gray = F.frontend(raw, cb, mi, mj, flip, crop, idxes, di, dj, mode, ver)
    raw_d = torch.from_numpy(raw).cuda()
    imgs, gray = frontend.prepare_clip(raw_d, cb, mi, mj, flip, crop, idxes, all_di=di, all_dj=dj, color_mode=mode, want_imgs=True, cv_version=ver)
    assert np.array_equal(gray.cpu().numpy(), want_gray), (hs, ws, crop, cb, mi, mj, flip, mode, ver, shake)
    assert np.array_equal(imgs.cpu().numpy(), want_imgs)
    _, gray2 = frontend.prepare_clip(raw_d, cb, mi, mj, flip, crop, idxes, all_di=di, all_dj=dj, color_mode=mode, want_imgs=False, cv_version=ver)
    assert torch.equal(gray2, gray)                                   # the tiled kernel (gray clip only) and the one that also returns frames


@pytest.mark.parametrize("case", range(16 * _SCALE))
def test_random_packed_clips_with_statistics(oracle_c, luts, case):
    """The loader's launch (v2v_esim_voxel_ex_hip: per-clip frame index, packed clip offsets, writer statistics) over random shapes, pause
    patterns, parameters and mappings: grid == the scalar C oracle run on the GATHERED clips (exact: SUM bins hold integers), statistics
    == the histogram of that grid, scales == its sorted k-th values."""
    from v2v_amd import _lib, esim as E, postops
    g = np.random.default_rng(9000 + case)
    b = int(g.integers(1, 6))
    h = int(g.integers(2, 40))
    w = int(g.integers(1, 18)) * 4                             # rows of whole 4-pixel groups (the dataset's crops are multiples of 16)
    nb, fpb = int(g.integers(1, 6)), int(g.integers(1, 3))
    k = nb * fpb * int(g.integers(1, 5))
    n = k + 1
    stored = [int(g.integers(1, n + 1)) for _ in range(b)]
    clips = [g.integers(0, 256, size=(u, h, w), dtype=np.uint8) if g.random() < 0.5 else
             O.synth_clip_s1(u, h, w, seed=int(g.integers(1 << 30)), dtype=np.uint8) for u in stored]
    fidx = np.stack([np.sort(np.concatenate([np.arange(u), g.integers(0, u, n - u)])) for u in stored]).astype(np.int32)
    offs, parts, pos = [], [], 0
    for c in clips:
        offs.append(pos)
        pad = (-c.size) % 16
        parts += [c.ravel(), np.zeros(pad, np.uint8)]
        pos += c.size + pad
    gathered = np.stack([c[i] for c, i in zip(clips, fidx)])
    params = np.stack([[g.uniform(0.05, 1.0), g.uniform(0.05, 1.0), g.uniform(0, 0.2), g.choice([0.0, g.uniform(0, 0.004)]), g.uniform(0, 30.0)]
                       for _ in range(b)])
    keys = np.stack([[int(g.integers(1 << 62)), int(g.integers(1 << 30))] for _ in range(b)]).astype(np.int64)
    want = np.stack([oracle_c.esim_voxel(gathered[i:i + 1], params[i], luts, seed=int(keys[i, 0]), clip_id0=int(keys[i, 1]), bin_mode=oracle_c.BIN_SUM,
                                         num_bins=nb, frames_per_bin=fpb)[0][0] for i in range(b)])
    stats = torch.zeros((b, _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device="cuda")
    pad_to = int(g.choice([1, 16]))
    got = E.esim_voxel_packed(torch.from_numpy(np.concatenate(parts)).cuda(), torch.tensor(offs, dtype=torch.int64).cuda(), torch.from_numpy(fidx).cuda(),
                              h, w, torch.from_numpy(params).cuda(), torch.from_numpy(keys).cuda(), num_bins=nb, frames_per_bin=fpb, pad_to=pad_to,
                              stats=stats, mapping=["4px", "2px", "1px", "auto"][int(g.integers(0, 4))])
    assert np.array_equal(got[..., :h, :w].cpu().numpy(), want.astype(np.float32)), (b, n, h, w, stored)
    assert not got[..., h:, :].any() and not got[..., :, w:].any()
    st = stats.cpu().numpy()
    for i in range(b):
        iv = want[i].astype(np.int64).ravel()
        hist = np.bincount(np.clip(iv, -256, 256) + 256, minlength=516)[:516]
        hist[256] = 0
        assert np.array_equal(st[i], hist), i
    m = want[0].size
    if m >= 100:
        flat = np.sort(want.reshape(b, -1), axis=1)
        lo, hi = flat[:, int(0.01 * m) - 1], flat[:, int(0.99 * m) - 1]
        sc = postops.scales_from_stats(stats, m).cpu().numpy()
        for i in range(b):                                     # exact inside the counting range, NaN (never a wrong value) beyond it
            assert (np.isnan(sc[i, 0]) and lo[i] < -255) or sc[i, 0] == max(-lo[i], 1.0), (i, sc[i], lo[i])
            assert (np.isnan(sc[i, 1]) and hi[i] > 255) or sc[i, 1] == max(hi[i], 1.0), (i, sc[i], hi[i])
