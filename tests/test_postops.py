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
