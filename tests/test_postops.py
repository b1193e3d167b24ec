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
    with pytest.raises(ValueError, match="65535"):                                       # one grid row per sample: a clear refusal, not a launch error
        postops.normalize_and_pad(torch.zeros((65536, 1, 1, 10, 10), device="cuda"), True, 16)
    many = torch.round(torch.randn((65535, 1, 1, 10, 12), device="cuda") * 3)
    assert torch.equal(postops.normalize_and_pad(many, True, 16)[-3:], postops.normalize_and_pad(many[-3:].contiguous(), True, 16))


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


def _np_stats(vox):
    """NumPy restatement of the writer's statistics words for ONE sample's valid voxels (include/v2v_hip.h: v2v_esim_voxel_stats_hip)."""
    iv = vox.astype(np.int64).ravel()
    assert np.array_equal(iv, vox.ravel())
    h = np.bincount(np.clip(iv, -256, 256) + 256, minlength=516)[:516].astype(np.int64)
    h[256] = 0                                                                       # zeros are not counted: the reader derives them
    return h


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,shape,fpb,mapping", [("uint8", (5, 41, 36, 52), 1, "4px"), ("uint8", (3, 21, 32, 32), 2, "1px"),
                                                      ("uint8", (3, 21, 32, 34), 1, "2px"), ("float32", (4, 21, 64, 64), 1, "4px"),
                                                      ("float32", (2, 41, 32, 32), 4, "auto")])
def test_simulator_writer_statistics_give_exact_scales(oracle_c, luts, dtype, shape, fpb, mapping):
    """f-2 "in the writer": the statistics the simulator accumulates while it stores its SUM-mode planes (+-1 by ballots, |v| >= 2 by LDS
    atomics, hot pixels beyond +-255 in the overflow words) equal the histogram of the grid it wrote; the scales read off them equal
    the reference's k-th values (NumPy restatement pinned by G13), and scaling in place equals normalize_batch_voxel + zero padding.
    Hot pixels with hundreds of events per frame (hot_pixel_std 10, config/train_v2v_e2vid_10k.yaml:75) must not disturb any of it."""
    import torch
    from oracle import v2v_oracle as O
    from v2v_amd import _lib, esim, postops
    b, n, h, w = shape
    video = np.stack([O.synth_clip_s1(n, h, w, seed=70 + i, dtype=np.uint8) for i in range(b)]).astype(dtype)
    params = np.array([[0.11 + 0.03 * i, 0.14 + 0.02 * i, 0.03, 4e-3, 40.0] for i in range(b)])        # 0.4 % hot pixels, most of them beyond +-255
    params[-1] = [0.2, 0.2, 0.0, 0.0, 0.0]                                           # a noise-free symmetric clip in the same batch
    keys = torch.tensor([[900 + i, i] for i in range(b)], dtype=torch.int64)
    frames = torch.from_numpy(video).cuda()
    stats = torch.full((b, _lib.VOXEL_STATS_WORDS), 7, dtype=torch.int32, device="cuda")    # the launch zeroes it itself
    vox = esim.esim_voxel_batch(frames, torch.from_numpy(params).cuda(), bin_mode="sum", num_bins=5, frames_per_bin=fpb, clip_keys=keys,
                                pad_to=16, stats=stats, mapping=mapping)
    plain = esim.esim_voxel_batch(frames, torch.from_numpy(params).cuda(), bin_mode="sum", num_bins=5, frames_per_bin=fpb, clip_keys=keys, pad_to=16)
    assert torch.equal(vox, plain)                                                   # the statistics do not touch the grid
    valid = vox[..., :h, :w].cpu().numpy()
    st = stats.cpu().numpy()
    for i in range(b):
        assert np.array_equal(st[i], _np_stats(valid[i])), i
    assert st[:-1, [0, 512]].sum() > 0                                               # hot pixels did overflow +-255 somewhere
    scales = postops.scales_from_stats(stats, valid[0].size)
    flat = np.sort(valid.reshape(b, -1), axis=1)
    m = flat.shape[1]
    want_scales = np.stack([np.maximum(-flat[:, int(0.01 * m) - 1], 1), np.maximum(flat[:, int(0.99 * m) - 1], 1)], 1).astype(np.float32)
    assert np.array_equal(scales.cpu().numpy(), want_scales)
    want = np.zeros(tuple(vox.shape), dtype=np.float32)
    want[..., :h, :w] = _oracle_normalize(valid)
    out = postops.apply_scales(vox.clone(), scales, 16, valid_hw=(h, w), inplace=True)
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(postops.apply_scales(vox[..., :h, :w], scales, 16).cpu().numpy(), want)                       # pad + scale in one pass
    assert np.array_equal(postops.normalize_and_pad(vox.clone(), True, 16, method="count", valid_hw=(h, w), inplace=True).cpu().numpy(), want)
    assert np.array_equal(postops.normalize_and_pad(vox[..., :h, :w], True, 16, method="radix").cpu().numpy(), want)


@pytest.mark.gpu
def test_statistics_refuse_what_they_cannot_count_and_flag_overflowing_ranks():
    import torch
    from v2v_amd import _lib, esim, postops
    frames = esim.synth_clips(2, 11, 32, 32, dtype=torch.uint8, seed=3)
    stats = torch.zeros((2, _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        esim.esim_voxel_batch(frames.float(), [0.2, 0.2, 0.0, 0.0, 0.0], bin_mode="bilinear", num_bins=5, stats=stats)
    with pytest.raises(ValueError):
        esim.esim_voxel_batch(frames, [0.2, 0.2, 0.1, 0.0, 0.0], bin_mode="sum", num_bins=5, put_noise_external=True, stats=stats)
    with pytest.raises(ValueError):
        esim.esim_voxel_batch(frames, [0.2, 0.2, 0.1, 0.0, 0.0], bin_mode="sum", num_bins=5, out_dtype=torch.float64, stats=stats)
    # 5 % of the voxels beyond +255: the 99 % rank lies among the overflow counts -> NaN scale for that side of that sample only
    st = torch.zeros((2, _lib.VOXEL_STATS_WORDS), dtype=torch.int32)
    st[:, 257] = 100
    st[:, 255] = 300
    st[1, 512] = 500
    sc = postops.scales_from_stats(st.cuda(), 10000).cpu().numpy()
    assert np.array_equal(sc[0], [1.0, 1.0]) and sc[1, 0] == 1.0 and np.isnan(sc[1, 1])
    st[0, 513] = 1                                                                   # a flagged clip: both NaN
    assert np.isnan(postops.scales_from_stats(st.cuda(), 10000).cpu().numpy()[0]).all()
    # V2V_FLAG_SYMMETRIC broken by a clip: NaN planes AND a flagged statistic
    p = torch.tensor([[0.2, 0.2, 0.05, 0.0, 0.0], [0.2, 0.3, 0.05, 0.0, 0.0]], dtype=torch.float64, device="cuda")
    big = esim.synth_clips(2, 11, 64, 64, dtype=torch.uint8, seed=3)
    stats = torch.zeros((2, _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device="cuda")
    vox = esim.esim_voxel_batch(big, p, bin_mode="sum", num_bins=5, symmetric=True, stats=stats, mapping="4px")
    assert torch.isnan(vox[1]).all() and not torch.isnan(vox[0]).any()
    assert int(stats[1, 513]) != 0 and int(stats[0, 513]) == 0


@pytest.mark.gpu
def test_consumer_reader_applies_the_scales_while_it_converts():
    """normalize_batch_voxel folded into the E2VID head's input conversion: raw voxels + scales through v2v_to_nhwc8_bf16_scaled_hip ==
    the normalised tensor through the plain conversion, bit for bit (same float32 division, same bf16 rounding) -- and through the whole
    network (v2v_amd.unet.E2VIDRecurrent(event_tensor, event_scales))."""
    import torch
    from v2v_amd import convlstm as CL, postops
    g = torch.Generator().manual_seed(5)
    vox = torch.round(torch.randn((3, 4, 5, 64, 64), generator=g) * 4).cuda()                    # [B,T,C,H,W] integer counts
    scales = torch.tensor([[3.0, 7.0], [1.0, 2.0], [5.0, 1.0]], device="cuda")
    normed = postops.apply_scales(vox, scales, 16)
    v = vox.cpu().numpy()
    s = scales.cpu().numpy()
    want = np.where(v > 0, v / s[:, 1].reshape(3, 1, 1, 1, 1), v / s[:, 0].reshape(3, 1, 1, 1, 1)).astype(np.float32)
    assert np.array_equal(normed.cpu().numpy(), want)
    for t in range(4):
        a = CL.to_nhwc8_bf16(vox[:, t], scales)
        b = CL.to_nhwc8_bf16(normed[:, t])
        assert torch.equal(a, b) and float(a.float().abs().max()) > 0
    from v2v_amd.unet import E2VIDRecurrent
    torch.manual_seed(0)
    net = E2VIDRecurrent(dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
                              num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)).cuda().eval()
    with torch.no_grad():
        net.reset_states()
        raw = [net(vox[:, t], scales)["image"] for t in range(2)]
        net.reset_states()
        ref = [net(normed[:, t])["image"] for t in range(2)]
    assert all(torch.equal(x, y) for x, y in zip(raw, ref))
