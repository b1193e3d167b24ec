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


def _all_colours():
    b, g, r = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    return np.stack([b, g, r], axis=-1)                                          # [256,256,256,3], index (b, g, r)


def test_oracle_bgr_to_gray_equals_reference_on_all_colours(golden):
    """bgr_to_gray (v2v_datasets.py:19-22) is byte work: exact.  The restated order fma(r,w2, fma(g,w1, b*w0)) against golden G15
    = the reference's function on all 2^24 colours; and against this host's own np.dot on the same 4-D stack."""
    import hashlib
    g15 = golden("g15_bgr_to_gray.npz")
    got = F.bgr_to_gray_scalar(_all_colours())
    assert hashlib.sha256(got.tobytes()).hexdigest() == str(g15["sha256"])
    assert np.array_equal(got[g15["planes_b"]], g15["planes"])
    im = np.random.default_rng(1).integers(0, 256, size=(8, 64, 64, 3), dtype=np.uint8)
    ref = np.dot(im[..., :3], [0.5870, 0.1140, 0.2989]).astype(np.uint8)      # the reference's own line (v2v_datasets.py:21)
    assert np.array_equal(ref, F.bgr_to_gray_scalar(im))


@pytest.mark.gpu
def test_hip_bgr_to_gray_equals_reference_on_all_colours(golden):
    """The GPU front-end's gray_in_bgr_out conversion on an image holding every one of the 2^24 colours (identity resize)."""
    import hashlib
    import torch
    from v2v_amd import frontend
    g15 = golden("g15_bgr_to_gray.npz")
    img = _all_colours().reshape(1, 4096, 4096, 3)
    _, gray = frontend.prepare_clip(torch.from_numpy(img).cuda(), 4096, 0, 0, False, 4096, [0], color_mode="gray_in_bgr_out")
    got = gray.cpu().numpy().reshape(256, 256, 256)
    assert np.array_equal(got[g15["planes_b"]], g15["planes"])
    assert hashlib.sha256(got.tobytes()).hexdigest() == str(g15["sha256"])


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
    if color_mode == "gray":                # gray output only: the LDS-tiled kernel (with the per-frame shake offsets since round 3)
        _, gray_t = frontend.prepare_clip(torch.from_numpy(raw).cuda(), crop_before, 11, 23, flip, crop, idxes, di, dj, color_mode, want_imgs=False)
        assert np.array_equal(gray_t.cpu().numpy(), want_gray)


@pytest.mark.gpu
@pytest.mark.parametrize("flip", [False, True])
def test_hip_frontend_tiled_with_shake_at_training_size(flip):
    """color_mode 'gray' + shake_frames > 0 (data/v2v_datasets.py:145-161, 217-224) on the LDS-tiled kernel at the training geometry:
    resize to 128 + the largest offset in each direction (need_h != need_w), flip, per-frame cut-out; partial tiles, up- and
    down-scaling, the exact-2x area shortcut (crop_before = 2 * need in both directions only without shake)."""
    import torch
    from v2v_amd import frontend
    g = np.random.default_rng(17 + flip)
    t, hs, ws, crop = 9, 300, 420, 128
    raw = g.integers(0, 256, size=(t, hs, ws, 3), dtype=np.uint8)
    raw_d = torch.from_numpy(raw).cuda()
    idxes = [0, 1, 2, 2, 3, 4, 5, 6, 7, 8, 8]
    for cb in (97, 128, 200, 256, 277):
        di, dj = g.integers(-4, 5, size=t), g.integers(-6, 3, size=t)
        _, want = F.frontend(raw, cb, 13, 31, flip, crop, idxes, di, dj, "gray")
        _, got = frontend.prepare_clip(raw_d, cb, 13, 31, flip, crop, idxes, di, dj, "gray", want_imgs=False)
        assert np.array_equal(got.cpu().numpy(), want), cb


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


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["gray", "gray_in_bgr_out"])
def test_hip_frontend_device_resident_tables_are_clamped_into_bounds(mode):
    """Tables that live on the device (validate=False: no host round trip, the hipGraph-captured pipeline) were never seen by the host.  Whatever
    they hold -- negative corners, a rectangle past the frame, a crop larger than the frame (and than the bound the LDS tile was sized for), frame numbers outside
    the decoded clip -- every read stays inside the clip's frames: the result is that of the table clamped into bounds, never an
    out-of-bounds access.  Valid rows of the same call are untouched."""
    import torch
    from v2v_amd import frontend
    g = np.random.default_rng(5)
    b, t, hs, ws, crop, cb_max = 5, 6, 90, 120, 32, 80
    raw = torch.from_numpy(g.integers(0, 256, size=(b, t, hs, ws, 3), dtype=np.uint8)).cuda()
    bad = np.array([[-5, 10, 64, 0], [70, 100, 64, 1], [3, 4, 5000, 0], [20, 30, 0, 1], [5, 7, 64, 0]], dtype=np.int32)
    idx_bad = np.array([[0, 1, -3, 3, 4, 5, 99]] * b, dtype=np.int32)
    cb = np.clip(bad[:, 2], 1, min(hs, ws))                     # the bound only sizes the LDS tile: larger crops take the unstaged path
    fixed = np.stack([np.clip(bad[:, 0], 0, hs - cb), np.clip(bad[:, 1], 0, ws - cb), cb, bad[:, 3]], axis=1).astype(np.int32)
    idx_fixed = np.clip(idx_bad, 0, t - 1)
    want_imgs, want = frontend.prepare_clips_batch(raw, fixed, idx_fixed, crop, mode, want_imgs=True, max_crop_before=cb_max)
    dev = lambda a: torch.from_numpy(a).cuda().contiguous()   # noqa: E731
    got_imgs, got = frontend.prepare_clips_batch(raw, dev(bad), dev(idx_bad), crop, mode, want_imgs=True, validate=False, max_crop_before=cb_max)
    torch.cuda.synchronize()
    assert torch.equal(got, want) and torch.equal(got_imgs, want_imgs)
    assert torch.equal(got[4, :2], frontend.prepare_clip(raw[4], 64, 5, 7, False, crop, [0, 1], color_mode=mode)[1])


# ---- LDS-tiled kernel (gray output only, no shake): same bits as the oracle and as the per-pixel gather kernel
@pytest.mark.gpu
@pytest.mark.parametrize("flip", [False, True])
@pytest.mark.parametrize("hs,ws,crop,crop_before,mi,mj", [
    (120, 160, 32, 100, 11, 23),       # one partial tile, downscale 3.1x
    (120, 161, 32, 64, 3, 5),          # OpenCV's 2x2 area shortcut (crop_before == 2*crop), odd row pitch
    (300, 400, 200, 290, 5, 17),       # several tiles, partial last column tile and row tile, scale 1.45
    (300, 401, 256, 197, 40, 101),     # upscale 0.77 (config-4 lower bound), two column tiles, odd row pitch
    (300, 400, 130, 131, 0, 0),        # scale ~1: border taps clamp (s1 == s0) at the right/bottom edge
    (64, 64, 7, 64, 0, 0),             # crop smaller than one 4-pixel group row; whole frame
    (700, 700, 64, 600, 50, 60),       # rectangle larger than the LDS row budget -> per-pixel fallback inside the tile kernel
    (700, 700, 64, 70, 500, 600),      # same launch geometry, small rectangle -> staged
])
def test_hip_frontend_tiled_equals_oracle(flip, hs, ws, crop, crop_before, mi, mj):
    import torch
    from v2v_amd import frontend
    g = np.random.default_rng(hs + ws + crop + crop_before + flip)
    t = 3
    raw = g.integers(0, 256, size=(t, hs, ws, 3), dtype=np.uint8)
    idxes = [2, 0, 1, 1]
    _, want_gray = F.frontend(raw, crop_before, mi, mj, flip, crop, idxes, None, None, "gray")
    raw_d = torch.from_numpy(raw).cuda()
    none_imgs, gray = frontend.prepare_clip(raw_d, crop_before, mi, mj, flip, crop, idxes, want_imgs=False)     # tiled kernel
    assert none_imgs is None
    assert np.array_equal(gray.cpu().numpy(), want_gray)
    _, gray_gather = frontend.prepare_clip(raw_d, crop_before, mi, mj, flip, crop, idxes, want_imgs=True)       # gather kernel
    assert torch.equal(gray, gray_gather)


@pytest.mark.gpu
def test_hip_frontend_tiled_batch_mixed_rectangles():
    """One launch with staged, area-shortcut and over-budget clips; rectangle touching the end of the buffer."""
    import torch
    from v2v_amd import frontend
    g = np.random.default_rng(11)
    b, t, hs, ws, crop = 4, 2, 700, 700, 64
    raw = g.integers(0, 256, size=(b, t, hs, ws, 3), dtype=np.uint8)
    table = np.array([[0, 0, 70, 0], [100, 200, 128, 1], [0, 0, 700, 1], [630, 630, 70, 0]], dtype=np.int32)
    idx = np.array([[0, 1, 1]] * b, dtype=np.int32)
    _, gray = frontend.prepare_clips_batch(torch.from_numpy(raw).cuda(), table, idx, crop, "gray", want_imgs=False)
    for c in range(b):
        _, want = F.frontend(raw[c], int(table[c, 2]), int(table[c, 0]), int(table[c, 1]), bool(table[c, 3]), crop, idx[c], None, None, "gray")
        assert np.array_equal(gray[c].cpu().numpy(), want), c


@pytest.mark.gpu
@pytest.mark.parametrize("color_mode", ["gray", "gray_in_bgr_out"])
def test_hip_frontend_reads_page_locked_host_frames_zero_copy(color_mode):
    """BASELINE config 4's stream: the decoded frames sit in PAGE-LOCKED HOST memory and the front-end kernel stages each clip's crop
    rectangle straight out of it over PCIe (no copy of the whole frames, no host-side slicing).  Same results as with device-resident
    frames, bit for bit; pageable host memory is refused."""
    import torch
    from v2v_amd import frontend
    g = np.random.default_rng(12)
    b, t, hs, ws, crop = 3, 4, 360, 640, 96
    raw = torch.from_numpy(g.integers(0, 256, size=(b, t, hs, ws, 3), dtype=np.uint8))
    table = np.array([[5, 7, 120, 0], [100, 300, 190, 1], [0, 0, 96, 1]], dtype=np.int32)
    idx = np.array([[0, 1, 1, 2, 3]] * b, dtype=np.int32)
    want_imgs, want_gray = frontend.prepare_clips_batch(raw.cuda(), table, idx, crop, color_mode, want_imgs=color_mode != "gray")
    pinned = raw.pin_memory()
    for rep in range(3):
        got_imgs, got_gray = frontend.prepare_clips_batch(pinned, table, idx, crop, color_mode, want_imgs=color_mode != "gray")
        assert got_gray.is_cuda and torch.equal(got_gray, want_gray), rep
        if want_imgs is not None:
            assert torch.equal(got_imgs, want_imgs), rep
    with pytest.raises(ValueError):
        frontend.prepare_clips_batch(raw, table, idx, crop, color_mode)              # pageable host memory: not device-accessible


def test_bgr2gray_known_answers():
    """Hand-worked known answers of the two fixed-point BGR2GRAY forms (oracle/frontend_oracle.py header): OpenCV >= 4.0
    (B*3735 + G*19235 + R*9798 + 16384) >> 15 and OpenCV 2.x/3.x (B*1868 + G*9617 + R*4899 + 8192) >> 14.
        (B,G,R) = (255,0,0):  cv4 (952425 + 16384) >> 15 = 968809 >> 15 = 29          cv3 (476340 + 8192) >> 14 = 484532 >> 14 = 29
        (0,255,0):            cv4 (4904925 + 16384) >> 15 = 4921309 >> 15 = 150        cv3 (2452335 + 8192) >> 14 = 2460527 >> 14 = 150
        (0,0,255):            cv4 (2498490 + 16384) >> 15 = 2514874 >> 15 = 76         cv3 (1249245 + 8192) >> 14 = 1257437 >> 14 = 76
        (255,255,255):        weights sum to 32768 / 16384 -> 255 in both
        (0,5,0):              cv4 (96175 + 16384) >> 15 = 112559 >> 15 = 3             cv3 (48085 + 8192) >> 14 = 56277 >> 14 = 3
        (1,1,0):              cv4 (22970 + 16384) >> 15 = 39354 >> 15 = 1              cv3 (11485 + 8192) >> 14 = 19677 >> 14 = 1
        (0,23,0):             cv4 (442405 + 16384) >> 15 = 458789 >> 15 = 14 (14.0011) cv3 (221191 + 8192) >> 14 = 229383 >> 14 = 14 (14.0004)
        (203,0,0):            cv4 (758205 + 16384) >> 15 = 774589 >> 15 = 23 (23.64)   cv3 (379204 + 8192) >> 14 = 387396 >> 14 = 23 (23.64)
        (20,4,2):             cv4 (74700 + 76940 + 19596 + 16384) >> 15 = 187620 >> 15 = 5 (5.73)
                              cv3 (37360 + 38468 + 9798 + 8192) >> 14 = 93818 >> 14 = 5 (5.73)
    and the two forms DIFFER on some colours (that is why the version matters): the test finds them and checks one by hand:
        (B,G,R) = (0,0,3):    cv4 (29394 + 16384) >> 15 = 45778 >> 15 = 1 (1.397)      cv3 (14697 + 8192) >> 14 = 22889 >> 14 = 1 -> equal;
        (125,0,0):            cv4 (466875 + 16384) >> 15 = 483259 >> 15 = 14 (14.748)  cv3 (233500 + 8192) >> 14 = 241692 >> 14 = 14 (14.752)"""
    px = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 5, 0], [1, 1, 0], [0, 23, 0], [203, 0, 0], [20, 4, 2],
                   [0, 0, 3], [125, 0, 0]], dtype=np.uint8)
    want = np.array([29, 150, 76, 255, 3, 1, 14, 23, 5, 1, 14], dtype=np.uint8)
    assert np.array_equal(F.cv_bgr2gray_u8(px, "cv4"), want) and np.array_equal(F.cv_bgr2gray_u8(px, "cv3"), want)
    # all 2^24 colours: the forms agree except where the weighted sum sits within 2^-15 of a rounding boundary
    v = np.arange(256, dtype=np.uint8)
    b, g, r = np.meshgrid(v, v, v, indexing="ij")
    allc = np.stack([b.ravel(), g.ravel(), r.ravel()], axis=1)
    g4, g3 = F.cv_bgr2gray_u8(allc, "cv4"), F.cv_bgr2gray_u8(allc, "cv3")
    diff = np.flatnonzero(g4 != g3)
    assert 0 < diff.size < allc.shape[0] // 100 and int(np.abs(g4.astype(int) - g3.astype(int)).max()) == 1
    bb, gg, rr = (int(x) for x in allc[diff[0]])
    assert ((bb * 3735 + gg * 19235 + rr * 9798 + 16384) >> 15) == int(g4[diff[0]])
    assert ((bb * 1868 + gg * 9617 + rr * 4899 + 8192) >> 14) == int(g3[diff[0]])
    with pytest.raises(ValueError):
        F.cv_bgr2gray_u8(px, "cv2")


@pytest.mark.gpu
def test_hip_frontend_gray_versions():
    """The kernel's two BGR2GRAY forms (gather kernel and LDS-tiled kernel) against the restatement, default = OpenCV 4.x."""
    import torch
    from v2v_amd import frontend as FE
    g = np.random.default_rng(44)
    raw = g.integers(0, 256, size=(3, 96, 160, 3), dtype=np.uint8)
    raw_d = torch.from_numpy(raw).cuda()
    idx = [0, 1, 1, 2]
    for ver in ("cv4", "cv3"):
        _, want = F.frontend(raw, 80, 7, 21, True, 64, idx, None, None, "gray", cv_version=ver)
        _, got = FE.prepare_clip(raw_d, 80, 7, 21, True, 64, idx, color_mode="gray", want_imgs=False, cv_version=ver)      # tiled kernel
        assert np.array_equal(got.cpu().numpy(), want), ver
        imgs, got2 = FE.prepare_clip(raw_d, 80, 7, 21, True, 64, idx, color_mode="gray", want_imgs=True, cv_version=ver)    # gather kernel
        assert np.array_equal(got2.cpu().numpy(), want) and np.array_equal(imgs.cpu().numpy()[..., 0], want), ver
    _, default = FE.prepare_clip(raw_d, 80, 7, 21, True, 64, idx, color_mode="gray", want_imgs=False)
    _, want4 = F.frontend(raw, 80, 7, 21, True, 64, idx, None, None, "gray")
    assert np.array_equal(default.cpu().numpy(), want4)
