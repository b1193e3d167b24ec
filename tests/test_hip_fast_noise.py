"""The device-native base-noise field as the simulator injects it (external-noise mode exposes it exactly): standard normal,
uncorrelated across pixels and time steps, equal to the C oracle's field bit for bit; and V2V_RNG_PHILOX_FAST, kept as an
alias of V2V_RNG_PHILOX, gives the same launch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _noise_field(E, rng_mode, std=0.7, n=9, h=128, w=256, seed=11):
    """External-noise mode on a constant video with unreachable thresholds: voxel[k] == base_noise_std * g[k] exactly."""
    video = torch.full((1, n, h, w), 128, dtype=torch.uint8, device="cuda")
    out = E.esim_voxel_batch(video, [1e6, 1e6, std, 0.0, 0.0], bin_mode="sum", num_bins=n - 1, rng_mode=rng_mode, seed=seed,
                             put_noise_external=True, out_dtype=torch.float64)
    return (out[0, 0] / std).cpu().numpy()


def test_fast_noise_is_standard_normal():
    from scipy import stats
    from v2v_amd import esim as E
    g = _noise_field(E, "philox_fast").reshape(-1)
    assert g.size == 8 * 128 * 256
    assert abs(g.mean()) < 4 / np.sqrt(g.size) and abs(g.std() - 1) < 4 / np.sqrt(2 * g.size)
    assert abs(stats.skew(g)) < 0.02 and abs(stats.kurtosis(g)) < 0.04
    assert stats.kstest(g[:100000], "norm").pvalue > 1e-3
    # tails: the 2^-14 midpoint grid of the table inversion ends at Phi^-1(1 - 2^-15) = 4.009
    assert 3.8 < np.abs(g).max() < 4.01
    # no correlation between neighbouring pixels / consecutive pairs; and it IS the exact field up to the hardware functions' ulps
    f = g.reshape(8, 128, 256)
    assert abs(np.corrcoef(f[:, :, :-1].ravel(), f[:, :, 1:].ravel())[0, 1]) < 0.01
    assert abs(np.corrcoef(f[:-1].ravel(), f[1:].ravel())[0, 1]) < 0.01
    exact = _noise_field(E, "philox").reshape(-1)
    assert np.array_equal(g, exact)                                   # the alias
    # and it is the oracle's field: pair k = member k&1 of block 3 + (k>>1), Philox4x32-7
    from oracle import clib
    want = np.stack([clib.philox_gauss_field(11, 0, 3 + (k >> 1), 128 * 256, 0, k & 1, clib.noise_rounds()) for k in range(8)])
    assert np.array_equal(exact.reshape(8, -1).astype(np.float32), want)


def test_fast_noise_event_statistics_match_exact_mode():
    from v2v_amd import esim as E
    frames = E.synth_clips(8, 32, 128, 128, dtype=torch.uint8, seed=5)
    p = [0.2, 0.25, 0.08, 1e-3, 1.0]
    tot = {}
    for mode in ("philox", "philox_fast"):
        c = torch.zeros((8, 2), dtype=torch.int64, device="cuda")
        v = E.esim_voxel_batch(frames, p, bin_mode="bilinear", num_bins=5, rng_mode=mode, seed=9, counts=c)
        tot[mode] = (c.sum(0).cpu().numpy().astype(np.float64), float(v.abs().sum()))
    on_off_exact, on_off_fast = tot["philox"][0], tot["philox_fast"][0]
    assert np.array_equal(on_off_fast, on_off_exact)
    assert abs(tot["philox_fast"][1] - tot["philox"][1]) / tot["philox"][1] < 0.01
    # noise-free launches ignore the mode entirely
    a = E.esim_voxel_batch(frames, [0.2, 0.25, 0, 0, 0], bin_mode="bilinear", rng_mode="philox", seed=9)
    b = E.esim_voxel_batch(frames, [0.2, 0.25, 0, 0, 0], bin_mode="bilinear", rng_mode="philox_fast", seed=9)
    assert torch.equal(a, b)
