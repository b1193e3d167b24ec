"""Randomised parity sweep: HIP (through the C ABI) vs the scalar C oracle over random shapes, dtypes, parameters,
binning modes and batch splits.  Float64 output must be bit-exact, float32 output within 1e-5, totals exact."""
import numpy as np
import pytest
import torch

from oracle import v2v_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", range(24))
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
    kw = dict(bin_mode="bilinear" if bilinear else "sum", num_bins=nb, frames_per_bin=fpb, seed=seed, clip_id0=cid0,
              put_noise_external=ext)
    counts = torch.zeros((b, 2), dtype=torch.int64, device="cuda")
    got = E.esim_voxel_batch(torch.from_numpy(video).cuda(), params, out_dtype=torch.float64, counts=counts, **kw)
    assert np.array_equal(got.cpu().numpy(), want), (b, n, h, w, dt, kw)
    assert np.array_equal(counts.cpu().numpy(), totals)
    got32 = E.esim_voxel_batch(torch.from_numpy(video).cuda(), params, **kw)
    np.testing.assert_allclose(got32.cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    if b > 1:                                                  # any split of the batch gives the same clips
        cut = int(g.integers(1, b))
        tail = E.esim_voxel_batch(torch.from_numpy(video[cut:]).cuda(), params[cut:], out_dtype=torch.float64,
                                  **dict(kw, clip_id0=cid0 + cut))
        assert torch.equal(tail, got[cut:])
