"""Voxel post-ops: normalize_batch_voxel + pad, against goldens produced by the reference's own
function (model/train_utils.py:147-166, compiled from the reference file by tests/golden/make_goldens.py)."""
import numpy as np
import pytest


def _oracle_normalize(v):
    """NumPy restatement: k-th smallest (1-based) at int(0.99*M) / int(0.01*M), clamp >= 1, where(v>0, v/pos, v/neg)."""
    b = v.shape[0]
    flat = v.reshape(b, -1)
    m = flat.shape[1]
    max_k, min_k = int(0.99 * m), int(0.01 * m)
    srt = np.sort(flat, axis=1)
    pos = np.maximum(srt[:, max_k - 1], 1).reshape(b, 1, 1, 1, 1).astype(np.float32)
    neg = np.maximum(-srt[:, min_k - 1], 1).reshape(b, 1, 1, 1, 1).astype(np.float32)
    return np.where(v > 0, v / pos, v / neg).astype(np.float32)


def test_oracle_matches_reference_golden(golden):
    g = golden("g13_normalize_batch_voxel.npz")
    assert np.array_equal(_oracle_normalize(g["counts"]), g["counts_norm"])
    assert np.array_equal(_oracle_normalize(g["soft"]), g["soft_norm"])


@pytest.mark.gpu
def test_hip_normalize_equals_reference_golden(golden):
    import torch
    from v2v_amd import postops
    g = golden("g13_normalize_batch_voxel.npz")
    for key in ("counts", "soft"):
        got = postops.normalize_batch_voxel(torch.from_numpy(g[key]).cuda())
        assert got.shape == g[key].shape and got.dtype == torch.float32
        assert np.array_equal(got.cpu().numpy(), g[f"{key}_norm"])              # exact k-th value, IEEE division


@pytest.mark.gpu
def test_hip_pad_and_fused(golden):
    import torch
    from v2v_amd import postops
    g = golden("g13_normalize_batch_voxel.npz")
    v = torch.from_numpy(g["soft"]).cuda()                                       # [2,2,5,17,19] -> pad to 32x32
    padded = postops.pad_events(v, 16)
    assert padded.shape == (2, 2, 5, 32, 32)
    assert torch.equal(padded[..., :17, :19], v) and not padded[..., 17:, :].any() and not padded[..., :, 19:].any()
    fused = postops.normalize_and_pad(v, True, 16)
    assert np.array_equal(fused[..., :17, :19].cpu().numpy(), g["soft_norm"]) and not fused[..., 17:, :].any()
    big = torch.round(torch.randn((4, 8, 5, 128, 128), device="cuda") * 3)
    want = _oracle_normalize(big.cpu().numpy())
    assert np.array_equal(postops.normalize_batch_voxel(big).cpu().numpy(), want)
    with pytest.raises(ValueError):
        postops.normalize_batch_voxel(torch.zeros((1, 1, 1, 3, 3), device="cuda"))      # < 100 elements: kthvalue(0) raises in torch


@pytest.mark.gpu
def test_hip_counting_select_equals_reference_golden(golden):
    """Integer-valued voxels through the counting path (1 + 1 reads instead of 3 + 1): same bits as the reference's function."""
    import torch
    from v2v_amd import postops
    g = golden("g13_normalize_batch_voxel.npz")
    got = postops.normalize_batch_voxel(torch.from_numpy(g["counts"]).cuda(), method="count").cpu().numpy()
    assert np.array_equal(got[[0, 2]], g["counts_norm"][[0, 2]])
    assert np.isnan(got[1]).all()                                                # G13's sample 1 is scaled by 0.2: not integer-valued
    # non-integer content is refused loudly: NaN for that sample, never a silently wrong scale
    bad = postops.normalize_batch_voxel(torch.from_numpy(g["soft"]).cuda(), method="count")
    assert bool(torch.isnan(bad).any(dim=(1, 2, 3, 4)).all())
    big = torch.round(torch.randn((4, 8, 5, 128, 128), device="cuda") * 3)
    big[1] *= 40                                                                 # values beyond +-255 in one sample only
    out = postops.normalize_batch_voxel(big, method="count").cpu().numpy()
    want = _oracle_normalize(big.cpu().numpy())
    assert np.isnan(out[1]).all() and np.array_equal(out[[0, 2, 3]], want[[0, 2, 3]])


@pytest.mark.gpu
def test_simulator_writes_the_padded_layout_and_normalises_in_place(oracle_c, luts):
    """f-2 end to end: the simulator writes its SUM-mode grids straight into the x16-padded buffer (odd frame size 36 x 52 ->
    48 x 64), the counting normaliser runs in place on it (pad zeros excluded from the k-th values), and the result equals
    simulate -> normalize_batch_voxel (NumPy restatement pinned by G13) -> zero-pad."""
    import torch
    from oracle import v2v_oracle as O
    from v2v_amd import esim, postops
    b, n, h, w = 3, 21, 36, 52
    video = np.stack([O.synth_clip_s1(n, h, w, seed=40 + i, dtype=np.uint8) for i in range(b)])
    p = [0.12, 0.15, 0.05, 1e-3, 0.5]
    frames = torch.from_numpy(video).cuda()
    plain = esim.esim_voxel_batch(frames, p, bin_mode="sum", num_bins=5, frames_per_bin=2, seed=5)                  # [3,2,5,36,52]
    want_counts, _ = oracle_c.esim_voxel(video, p, luts, seed=5, bin_mode=oracle_c.BIN_SUM, num_bins=5, frames_per_bin=2)
    assert np.array_equal(plain.cpu().numpy(), want_counts.astype(np.float32))
    padded = esim.esim_voxel_batch(frames, p, bin_mode="sum", num_bins=5, frames_per_bin=2, seed=5, pad_to=16)      # [3,2,5,48,64]
    assert padded.shape == (b, 2, 5, 48, 64) and torch.equal(padded[..., :h, :w], plain)
    assert not padded[..., h:, :].any() and not padded[..., :, w:].any()
    bil = esim.esim_voxel_batch(frames.float(), p, bin_mode="bilinear", num_bins=5, seed=5, pad_to=16)
    assert torch.equal(bil[..., :h, :w], esim.esim_voxel_batch(frames.float(), p, bin_mode="bilinear", num_bins=5, seed=5)) and not bil[..., h:, :].any()
    out = postops.normalize_and_pad(padded.clone(), True, 16, method="count", valid_hw=(h, w), inplace=True)
    want = np.zeros((b, 2, 5, 48, 64), dtype=np.float32)
    want[..., :h, :w] = _oracle_normalize(want_counts.astype(np.float32))
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(postops.normalize_and_pad(plain, True, 16, method="radix").cpu().numpy(), want)           # the general path agrees
