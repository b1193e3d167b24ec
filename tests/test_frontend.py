"""Decode-side front-end.  OpenCV is absent, so parity is UNPINNED against cv2: the oracle restates
OpenCV's published 8-bit algorithm and the HIP kernel must match that restatement bit for bit; plus properties."""
import numpy as np
import pytest

from oracle import frontend_oracle as F


def test_oracle_resize_properties():
    g = np.random.default_rng(0)
    img = g.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    assert np.array_equal(F.cv_resize_linear_u8(img, 53, 37), img)                       # scale 1 is the identity
    assert (F.cv_resize_linear_u8(np.full((40, 60), 77, np.uint8), 25, 31) == 77).all()   # constants survive
    ramp = np.tile(np.arange(0, 200, 2, dtype=np.uint8), (8, 1))
    up = F.cv_resize_linear_u8(ramp, 250, 8)
    assert np.all(np.diff(up.astype(int), axis=1) >= 0) and up.min() == 0 and up.max() == 198      # monotone, in range
    box = F.cv_resize_linear_u8(img[:36, :52, 0], 26, 18)                                # exact 2x: box average
    s = img[:36, :52, 0].astype(int)
    assert np.array_equal(box, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))


def test_oracle_gray_matches_reference_numpy_on_samples():
    g = np.random.default_rng(1)
    im = g.integers(0, 256, size=(8, 64, 64, 3), dtype=np.uint8)
    ref = np.dot(im[..., :3], [0.5870, 0.1140, 0.2989]).astype(np.uint8)      # the reference's own line (v2v_datasets.py:21)
    assert np.count_nonzero(ref != F.bgr_to_gray_scalar(im)) <= max(1, int(2e-5 * ref.size))


@pytest.mark.gpu
@pytest.mark.parametrize("color_mode", ["gray", "gray_in_bgr_out"])
@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("crop_before,shake", [(100, False), (64, False), (32, False), (45, True), (83, True)])
def test_hip_frontend_equals_oracle(color_mode, flip, crop_before, shake):
    import torch
    from v2v_amd import frontend
    g = np.random.default_rng(crop_before + flip)
    t, hs, ws, crop = 7, 120, 160, 32
    raw = g.integers(0, 256, size=(t, hs, ws, 3), dtype=np.uint8)
    idxes = [0, 1, 1, 2, 3, 3, 3, 4, 5, 6]
    di = dj = None
    if shake:
        di, dj = g.integers(-3, 4, size=t), g.integers(-2, 5, size=t)
    want_imgs, want_gray = F.frontend(raw, crop_before, 11, 23, flip, crop, idxes, di, dj, color_mode)
    imgs, gray = frontend.prepare_clip(torch.from_numpy(raw).cuda(), crop_before, 11, 23, flip, crop, idxes, di, dj, color_mode)
    assert np.array_equal(gray.cpu().numpy(), want_gray)
    assert np.array_equal(imgs.cpu().numpy(), want_imgs)


@pytest.mark.gpu
def test_hip_frontend_feeds_simulator():
    import torch
    from v2v_amd import esim, frontend
    raw = torch.randint(0, 256, (12, 96, 128, 3), dtype=torch.uint8, device="cuda")
    _, gray = frontend.prepare_clip(raw, 80, 4, 9, True, 64, list(range(11)), want_imgs=False)
    assert gray.shape == (11, 64, 64) and gray.dtype == torch.uint8
    vox = esim.esim_voxel_batch(gray[None], [0.2, 0.2, 0, 0, 0], num_bins=5, frames_per_bin=2, seed=1)
    assert vox.shape == (1, 1, 5, 64, 64) and torch.equal(vox, vox.round())
    with pytest.raises(ValueError):
        frontend.prepare_clip(raw, 200, 4, 9, False, 64, [0])           # crop rectangle outside the frame


@pytest.mark.gpu
def test_hip_frontend_batch_equals_single_clip_calls():
    import torch
    from v2v_amd import frontend
    g = np.random.default_rng(3)
    b, t, hs, ws, crop = 3, 6, 90, 120, 32
    raw = torch.from_numpy(g.integers(0, 256, size=(b, t, hs, ws, 3), dtype=np.uint8)).cuda()
    table = np.array([[5, 7, 64, 0], [0, 0, 80, 1], [20, 40, 41, 1]], dtype=np.int32)
    idx = np.array([[0, 1, 2, 3, 4, 5, 5], [0, 0, 1, 2, 3, 4, 5], [0, 1, 1, 1, 2, 3, 4]], dtype=np.int32)
    for mode in ("gray", "gray_in_bgr_out"):
        imgs, gray = frontend.prepare_clips_batch(raw, table, idx, crop, mode, want_imgs=True)
        for c in range(b):
            i1, g1 = frontend.prepare_clip(raw[c], int(table[c, 2]), int(table[c, 0]), int(table[c, 1]), bool(table[c, 3]), crop,
                                           idx[c], color_mode=mode)
            assert torch.equal(gray[c], g1) and torch.equal(imgs[c], i1)
    with pytest.raises(ValueError):
        frontend.prepare_clips_batch(raw, np.array([[0, 0, 200, 0]] * 3), idx, crop)
