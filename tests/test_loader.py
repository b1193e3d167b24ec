"""RingLoader (v2v_amd/loader.py): the batch loader of the drop-in path.  CPU tests cover the host half (slot layout, slot
assignment, ConcatDataset resolution, the sample written into a slot == the deferred sample of __getitem__ for the same np.random
draws); the GPU tests compare whole batches with SimulatingCollator / the per-sample path bit for bit."""
import numpy as np
import pytest
import torch


def _frames(ds, sample_idx, start, end, crop_before, min_i, min_j, flip, need_h, need_w):
    g = np.random.default_rng(1000 + start + 7 * int(sample_idx))     # own generator: must not touch the global stream
    c = 3 if ds.color_mode == "gray_in_bgr_out" else 1
    base = g.uniform(0, 255, size=(need_h, need_w, c))
    out = []
    for _ in range(end - start):
        base = np.clip(base + g.normal(0, 6, size=base.shape), 0, 255)
        f = base.astype(np.uint8)
        out.append((f[:, ::-1] if flip else f).copy())
    return out


def _make_ds(tmp_path, n_videos=6, **cfg):
    from v2v_amd.datasets import WebvidDatasetV2
    lst = tmp_path / "videos.txt"
    lst.write_text("".join(f"clip_{i}.mp4 {300 + 10 * i} 0.2 0.3\n" for i in range(n_videos)))
    base = {"video_list_file": str(lst), "sequence_length": 4, "crop_size": 32, "data_source_name": "webvid",
            "frame_source": _frames, "video_size": (1280, 720), "video_reader": "opencv"}
    base.update(cfg)
    return WebvidDatasetV2(str(tmp_path), base)


@pytest.mark.parametrize("extra", [{}, {"output_additional_frame": True}, {"output_additional_evs": True}, {"color_mode": "gray_in_bgr_out"},
                                   {"shake_frames": 5, "shake_std": 2.0}, {"video_degrade": "hdr", "degrade_ratio": 1.0}, {"fixed_seed": 5},
                                   {"proba_pause_when_running": 0.2, "proba_pause_when_paused": 0.7}])
def test_host_sample_into_equals_deferred_getitem(tmp_path, extra):
    """The slot writer consumes np.random exactly like __getitem__(defer_sim) and leaves the same clip, parameters and key -- in the plain
    form (the gathered clip) and in the PACKED form (every decoded frame once + the index row the simulator gathers through)."""
    ds = _make_ds(tmp_path, defer_sim=True, **extra)
    n, hw = ds.frames_per_seq + 1, ds.crop_size
    pick = ds.frame_pick()
    colour = ds.color_mode != "gray"
    for idx in (4, 0, 2):
        np.random.seed(100 + idx)
        want = ds[idx]
        state_after = np.random.get_state()[1].copy()
        for packed in (False, True):
            clip, params, key = np.full((n, hw, hw), 255, np.uint8), np.zeros(5), np.zeros(2, np.int64)
            cframes = np.zeros((len(pick), hw, hw, 3), np.uint8) if colour else None
            fidx = np.zeros(n, np.int32) if packed else None
            np.random.seed(100 + idx)
            res = ds.host_sample_into(idx, clip, params, key, cframes, fidx)
            assert np.array_equal(np.random.get_state()[1], state_after)             # same number of draws
            v2e, n_stored = res if packed else (res, n)
            gathered = clip[fidx] if packed else clip
            assert np.array_equal(gathered, want["sim_frames"].numpy())
            if packed:
                assert 1 <= n_stored <= n and fidx[0] == 0 and fidx[-1] == n_stored - 1 and set(np.diff(fidx)) <= {0, 1}
                assert (clip[n_stored:] == 255).all()                                # nothing written past the stored frames
            assert np.array_equal(params, want["sim_params"].numpy()) and np.array_equal(key, want["sim_key"].numpy())
            assert v2e == want["v2e_params"]
            # the frames the loader builds on the device: stored frame fidx[pick] / 255 (gray) or the colour frames / 255
            if cframes is None:
                got = torch.from_numpy(gathered[pick]).float().unsqueeze(1) / 255
            else:
                got = torch.from_numpy(cframes).float().permute(0, 3, 1, 2) / 255
            assert torch.equal(got, want["frame"])


def _frames_array(*args):
    return np.stack(_frames(*args))


@pytest.mark.parametrize("extra", [{}, {"color_mode": "gray_in_bgr_out"}, {"video_degrade": "hdr", "degrade_ratio": 1.0}, {"shake_frames": 5, "shake_std": 2.0}])
def test_array_frame_source_equals_list_frame_source(tmp_path, extra):
    """A frame source may hand out ONE [T,h,w,C] array instead of a list of frames (pre-decoded stores): same samples, through
    __getitem__ and through the slot writer, with and without the per-frame paths (shake, degradations) in the way."""
    a = _make_ds(tmp_path, defer_sim=True, **extra)
    b = _make_ds(tmp_path, defer_sim=True, frame_source=_frames_array, **extra)
    n, hw = a.frames_per_seq + 1, a.crop_size
    colour = a.color_mode != "gray"
    for idx in (1, 3):
        np.random.seed(7 + idx)
        sa = a[idx]
        np.random.seed(7 + idx)
        sb = b[idx]
        assert torch.equal(sa["sim_frames"], sb["sim_frames"]) and torch.equal(sa["frame"], sb["frame"]) and sa["v2e_params"] == sb["v2e_params"]
        outs = []
        for ds in (a, b):
            clip, params, key = np.zeros((n, hw, hw), np.uint8), np.zeros(5), np.zeros(2, np.int64)
            cframes = np.zeros((len(ds.frame_pick()), hw, hw, 3), np.uint8) if colour else None
            fidx = np.zeros(n, np.int32)
            np.random.seed(7 + idx)
            _, stored = ds.host_sample_into(idx, clip, params, key, cframes, fidx)
            outs.append((clip, params, key, cframes, fidx, stored))
        for x, y in zip(*outs):
            assert np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y
        assert np.array_equal(outs[1][0][outs[1][4]], sa["sim_frames"].numpy())


def test_slot_layout_is_aligned_and_disjoint():
    from v2v_amd.loader import _SlotLayout
    lay = _SlotLayout(12, 201, 128, 128, 40, True)
    offs = [lay.off_offsets, lay.off_fidx, lay.off_pick, lay.off_params, lay.off_keys, lay.off_used, lay.off_stored, lay.off_cframes, lay.off_clips, lay.nbytes]
    assert all(o % 256 == 0 for o in offs) and offs == sorted(offs) and len(set(offs)) == len(offs)
    buf = np.zeros(lay.nbytes, dtype=np.uint8)
    views = lay.views(buf)
    for i, v in enumerate(views):
        v[...] = i + 1
    for i, v in enumerate(views):
        assert (v == i + 1).all()                                                    # no view overlaps another
    offsets, fidx, pick, params, keys, used, cframes, clips, stored = views
    assert stored.shape == (12,) and stored.dtype == np.int32
    assert offsets.shape == (12,) and fidx.shape == (12, 201) and pick.shape == (12, 40) and params.shape == (12, 5) and keys.shape == (12, 2)
    assert used.shape == (1,) and cframes.shape == (12, 40, 128, 128, 3) and clips.shape == (12 * 201 * 128 * 128,)
    assert _SlotLayout(12, 201, 128, 128, 40, False).views(np.zeros(lay.nbytes, np.uint8))[6] is None
    # every packed clip is rounded up to 16 bytes: the clip region holds `batch` ROUNDED clips (a crop size that is not a multiple of 4
    # with no paused clip in the batch overflowed the region before round 5)
    odd = _SlotLayout(7, 21, 30, 30, 4, False)
    per_clip = (21 * 30 * 30 + 15) // 16 * 16
    assert 21 * 30 * 30 % 16 != 0 and odd.views(np.zeros(odd.nbytes, np.uint8))[7].size >= 7 * per_clip
    assert 6 * per_clip + 21 * 30 * 30 <= odd.views(np.zeros(odd.nbytes, np.uint8))[7].size            # the last clip's room, all clips unpaused


def test_conv_layer_refuses_scales_it_would_drop():
    """ConvLayer.forward(x, skip, scales) applies normalize_batch_voxel's scales in the head kernel only; a layer that is not the head
    (more than 8 input channels, or another kernel size) must refuse them instead of running on raw events."""
    from v2v_amd.convlstm import ConvLayer
    layer = ConvLayer(64, 64, kernel_size=5, padding=2)
    assert not layer.head
    with torch.no_grad(), pytest.raises(ValueError, match="head kernel only"):
        layer(torch.zeros(1, 64, 8, 8), None, torch.ones(1, 2))


def test_slot_batch_sampler_and_leaf_resolution(tmp_path):
    from torch.utils.data import BatchSampler, ConcatDataset, SequentialSampler
    from v2v_amd.loader import _SlotBatchSampler, _leaf, _leaves
    a, b = _make_ds(tmp_path, n_videos=3), _make_ds(tmp_path, n_videos=5)
    cat = ConcatDataset([ConcatDataset([a]), ConcatDataset([b])])                     # data/data_interface.py:19-27 nesting
    assert [d for d in _leaves(cat)] == [a, b]
    assert _leaf(cat, 0) == (a, 0) and _leaf(cat, 2) == (a, 2) and _leaf(cat, 3) == (b, 0) and _leaf(cat, 7) == (b, 4) and _leaf(cat, -1) == (b, 4)
    counter = [0]
    s = _SlotBatchSampler(BatchSampler(SequentialSampler(range(8)), 3, True), 4, counter)
    epoch1, epoch2 = list(s), list(s)
    assert len(s) == 2 and epoch1 == [[(0, 0, 0), (1, 0, 1), (2, 0, 2)], [(3, 1, 0), (4, 1, 1), (5, 1, 2)]]
    assert [b[0][1] for b in epoch2] == [2, 3] and counter[0] == 4                   # slots keep rotating across epochs


def test_choose_normalize_method():
    from v2v_amd.loader import choose_normalize_method as ch
    p = np.array([[0.2, 0.3, 0.05, 1e-3, 10.0], [0.5, 0.5, 0.1, 5e-4, 3.0]])
    assert ch(p, 1, False) == "count"                            # hot pixels may exceed 255: overflow bins, far from the 1 % ranks
    assert ch(p, 1, True) == "radix"                             # external noise: non-integer voxels
    assert ch(np.array([[0.02, 0.3, 0.0, 0.0, 0.0]]), 1, False) == "radix"     # 6.91 / 0.02 > 255
    assert ch(np.array([[0.2, 0.3, 0.0, 0.02, 1.0]]), 1, False) == "radix"     # 2 % hot pixels could reach the 1 % ranks
    assert ch(p, 8, False) == "radix"                            # 8 frames per bin


def test_create_dataloader_falls_back_to_torch_for_other_datasets():
    """train.py:52-65's signature; datasets that are not the simulator's (validation sets) get the reference's own DataLoader."""
    from torch.utils.data import DataLoader, RandomSampler, TensorDataset
    from v2v_amd.loader import create_dataloader
    ds = TensorDataset(torch.arange(10).float())
    dl = create_dataloader(ds, {"num_workers": 0, "pin_memory": False}, 2, None)
    assert isinstance(dl, DataLoader) and isinstance(dl.sampler, RandomSampler) and dl.drop_last and len(dl) == 5


# ----------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_ring_loader_with_a_crop_size_that_is_not_a_multiple_of_4_and_no_pauses(tmp_path):
    """n*h*w % 16 != 0 and no clip of the batch pauses: every packed clip is rounded up to 16 bytes and the last sample's room must still
    lie inside the clip region (it did not before round 5: a reshape error inside the worker)."""
    from torch.utils.data import default_collate
    from v2v_amd.datasets import SimulatingCollator
    from v2v_amd.loader import RingLoader
    ds = _make_ds(tmp_path, defer_sim=True, fixed_seed=3, crop_size=30, proba_pause_when_running=0.0)
    assert ((ds.frames_per_seq + 1) * 30 * 30) % 16 != 0
    col = SimulatingCollator.from_configs(dict(ds.__dict__, num_bins=5), output_device="cuda", pad_to=16, normalize=False)
    for workers in (0, 2):
        loader = RingLoader(ds, batch_size=6, num_workers=workers, drop_last=True, pad_to=16)
        batch = next(iter(loader))
        want = col.simulate(default_collate([ds[j] for j in range(6)]))
        assert torch.equal(batch["events"], want["events"]) and torch.equal(batch["frame"], want["frame"])
        loader.close()


@pytest.mark.gpu
def test_ring_loader_refuses_numpy_replay_datasets_and_create_dataloader_routes_them_to_torch(tmp_path):
    """`sim_rng: numpy` (bit-exact replay of the reference's np.random stream) lives on the per-sample path; the ring always draws
    device-native noise, so it must not accept such a dataset silently."""
    from torch.utils.data import DataLoader
    from v2v_amd.loader import RingLoader, create_dataloader
    ds = _make_ds(tmp_path, sim_rng="numpy")
    with pytest.raises(TypeError, match="sim_rng"):
        RingLoader(ds, batch_size=2)
    dl = create_dataloader(ds, {"num_workers": 0, "pin_memory": False}, 2, None)
    assert isinstance(dl, DataLoader)
    batch = next(iter(dl))
    assert batch["events"].shape == (2, 4, 5, 32, 32)


@pytest.mark.gpu
def test_a_new_iter_retires_the_abandoned_epoch_at_once(tmp_path):
    """iter(loader) is a plain method: the abandoned epoch's DataLoader iterator is shut down when the new one is requested, not at the
    new iterator's first next(); the old generator then refuses to continue."""
    from v2v_amd.loader import RingLoader
    ds = _make_ds(tmp_path, n_videos=12, defer_sim=True, fixed_seed=1)
    loader = RingLoader(ds, batch_size=2, num_workers=2, drop_last=True)
    it1 = iter(loader)
    first = next(it1)
    old_dl_iter = loader._it
    it2 = iter(loader)                                                   # no next() yet
    assert loader._it is not old_dl_iter and getattr(old_dl_iter, "_shutdown", True)
    with pytest.raises(RuntimeError, match="retired"):
        next(it1)
    again = next(it2)
    assert torch.equal(first["events"], again["events"])                 # fixed_seed: the epoch restarts from sample 0
    loader.close()


@pytest.mark.gpu
def test_packed_launch_checks_the_frame_index_rows():
    """v2v_esim_voxel_ex_hip with stored_frames: a row that names a frame outside its clip (or a clip that does not fit the buffer) gives
    NaN planes and the statistics' flag word for THAT clip -- no out-of-bounds read, the other clips bit-identical to the clean launch."""
    from v2v_amd import _lib, esim, postops
    g = np.random.default_rng(5)
    b, n, h, w = 4, 11, 32, 32
    stored = [n, 5, 7, n]
    clips_h = [g.integers(0, 256, (u, h, w), dtype=np.uint8) for u in stored]
    fidx = np.stack([np.sort(np.concatenate([np.arange(u), g.integers(0, u, n - u)])) for u in stored]).astype(np.int32)
    offs = np.cumsum([0] + [c.size for c in clips_h[:-1]]).astype(np.int64)
    flat_d = torch.from_numpy(np.concatenate([c.ravel() for c in clips_h])).cuda()
    offs_d, stored_d = torch.from_numpy(offs).cuda(), torch.tensor(stored, dtype=torch.int32).cuda()
    params = torch.tensor([[0.2, 0.25, 0.05, 1e-3, 2.0]] * b, dtype=torch.float64).cuda()
    keys = torch.tensor([[7 + i, i] for i in range(b)], dtype=torch.int64).cuda()

    def run(fi, st=stored_d, off=offs_d, mapping="auto"):
        stats = torch.zeros((b, _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device="cuda")
        vox = esim.esim_voxel_packed(flat_d, off, torch.from_numpy(fi).cuda(), h, w, params, keys, num_bins=5, stats=stats, stored_frames=st, mapping=mapping)
        return vox, stats
    clean, st_clean = run(fidx)
    assert not torch.isnan(clean).any() and int(st_clean[:, _lib.VOXEL_STATS_WORDS - 3].sum()) == 0
    unchecked, _ = run(fidx, st=None)
    assert torch.equal(clean, unchecked)                                 # the check changes nothing for valid rows
    for mapping in ("auto", "4px", "2px", "1px"):
        for bad_clip, bad_val in ((1, 5), (2, -1), (1, 2**31 - 1)):      # == stored, negative, huge
            bad = fidx.copy()
            bad[bad_clip, 6] = bad_val
            vox, st = run(bad, mapping=mapping)
            assert torch.isnan(vox[bad_clip]).all() and int(st[bad_clip, 513]) == 1
            ok = [i for i in range(b) if i != bad_clip]
            assert torch.equal(vox[ok], clean[ok]) and int(st[ok][:, 513].sum()) == 0
            scales = postops.scales_from_stats(st, 10 * h * w)
            assert torch.isnan(scales[bad_clip]).all() and not torch.isnan(scales[ok]).any()
    # a clip that does not fit `frames` (offset + stored frames beyond the buffer; a negative offset)
    for bad_off in (int(flat_d.numel()) - 3 * h * w, -16):
        off2 = offs.copy()
        off2[3] = bad_off
        vox, st = run(fidx, off=torch.from_numpy(off2).cuda())
        assert torch.isnan(vox[3]).all() and int(st[3, 513]) == 1 and torch.equal(vox[:3], clean[:3])
    with pytest.raises(ValueError):
        esim.esim_voxel_packed(flat_d, offs_d, torch.from_numpy(fidx).cuda(), h, w, params, keys, stored_frames=stored_d.long())
    # the `frame` assembly gathers through picks too (v2v_clip_frames_f32_bounded_hip): a pick outside its clip -> that output frame is NaN
    from v2v_amd.loader import clip_frames_packed
    pick = np.stack([fidx[:, 2], fidx[:, 7], fidx[:, 10]], 1).astype(np.int32)
    good = clip_frames_packed(flat_d, offs_d, torch.from_numpy(pick).cuda(), h, w, stored_frames=stored_d)
    assert torch.equal(good, clip_frames_packed(flat_d, offs_d, torch.from_numpy(pick).cuda(), h, w)) and not torch.isnan(good).any()
    bad_pick = pick.copy()
    bad_pick[1, 1], bad_pick[2, 0] = 5, -3                              # clip 1 holds 5 frames, clip 2 seven
    got = clip_frames_packed(flat_d, offs_d, torch.from_numpy(bad_pick).cuda(), h, w, stored_frames=stored_d)
    assert torch.isnan(got[1, 1]).all() and torch.isnan(got[2, 0]).all()
    mask = torch.ones(got.shape[:2], dtype=torch.bool)
    mask[1, 1] = mask[2, 0] = False
    assert torch.equal(got[mask], good[mask])


@pytest.mark.gpu
def test_clip_frames_equals_cpu_division_on_all_values():
    from v2v_amd.loader import clip_frames_f32
    vals = torch.arange(256, dtype=torch.uint8).repeat(4 * 5 * 8 * 32 // 256 + 1)[: 4 * 5 * 8 * 32]
    src = vals.reshape(4, 5, 8, 32)
    pick = [4, 0, 2]
    got = clip_frames_f32(src.cuda(), pick).cpu()
    assert got.shape == (4, 3, 1, 8, 32) and torch.equal(got, (src[:, pick].float() / 255).unsqueeze(2))
    # colour frames (HWC in, CHW out), odd sizes, strided source, identity pick
    g = torch.Generator().manual_seed(0)
    col = torch.randint(0, 256, (3, 6, 7, 9, 3), dtype=torch.uint8, generator=g)
    got = clip_frames_f32(col.cuda(), frames=4).cpu()
    assert torch.equal(got, col[:, :4].float().permute(0, 1, 4, 2, 3) / 255)
    sub = col.cuda()[:, 1:5]                                     # clip stride != T * frame stride
    assert torch.equal(clip_frames_f32(sub, [3, 1]).cpu(), col[:, [4, 2]].float().permute(0, 1, 4, 2, 3) / 255)
    with pytest.raises(IndexError):
        clip_frames_f32(src.cuda(), [5])
    # more output frames than one grid holds rows (65,535): the entry point sends whole clips out in several launches
    many = torch.randint(0, 256, (9000, 9, 4, 8), dtype=torch.uint8, generator=g)
    pick8 = [8, 0, 3, 3, 5, 1, 7, 2]
    got = clip_frames_f32(many.cuda(), pick8).cpu()
    assert got.shape == (9000, 8, 1, 4, 8) and torch.equal(got, (many[:, pick8].float() / 255).unsqueeze(2))
    odd = many.reshape(9000, 9, 32)[:, :, :31].reshape(9000, 9, 1, 31).contiguous()          # 31 bytes per frame: the byte-wise kernel
    assert torch.equal(clip_frames_f32(odd.cuda(), pick8).cpu(), (odd[:, pick8].float() / 255).unsqueeze(2))


@pytest.mark.gpu
@pytest.mark.parametrize("extra,workers", [({}, 2), ({"color_mode": "gray_in_bgr_out"}, 0), ({"output_additional_evs": True}, 2),
                                           ({"output_additional_frame": True, "crop_size": 36}, 0), ({"put_noise_external": True}, 2),
                                           ({"proba_pause_when_running": 0.3, "proba_pause_when_paused": 0.6, "shake_frames": 4, "shake_std": 1.5}, 2)])
def test_ring_loader_equals_simulating_collator(tmp_path, extra, workers):
    """Whole batches: RingLoader == default_collate + SimulatingCollator over the deferred samples (hence == the per-sample path,
    tests/test_hip_dataset_events.py::test_simulating_collator_equals_per_sample_path), bit for bit, with forked workers too."""
    from torch.utils.data import ConcatDataset, default_collate
    from v2v_amd.datasets import SimulatingCollator
    from v2v_amd.loader import RingLoader
    ds = _make_ds(tmp_path, defer_sim=True, fixed_seed=77, **extra)     # fixed_seed: a sample is a pure function of its index
    wrapped = ConcatDataset([ConcatDataset([ds])])
    col = SimulatingCollator.from_configs(dict(ds.__dict__, num_bins=5), output_device="cuda", pad_to=16, normalize=True)
    loader = RingLoader(wrapped, batch_size=3, num_workers=workers, drop_last=True, pad_to=16, normalize=True)
    assert len(loader) == 2
    for epoch in range(2):
        n = 0
        for bi, batch in enumerate(loader):
            want = col.simulate(default_collate([ds[3 * bi + j] for j in range(3)]))
            assert set(batch) == {"frame", "events", "data_source_idx", "v2e_params"}
            assert batch["events"].is_cuda and batch["frame"].is_cuda
            assert torch.equal(batch["events"], want["events"]) and torch.equal(batch["frame"], want["frame"])
            assert batch["events"].shape[-1] % 16 == 0 and batch["frame"].shape[1] == len(ds.frame_pick())
            assert torch.equal(batch["data_source_idx"], want["data_source_idx"]) and batch["data_source_idx"].dtype == torch.int64
            for k in want["v2e_params"]:
                assert torch.equal(batch["v2e_params"][k], want["v2e_params"][k])
            n += 1
        assert n == 2
    loader.close()


@pytest.mark.gpu
def test_ring_loader_scales_mode_and_hot_pixels(tmp_path):
    """normalize='scales': raw events + (neg_max, pos_max) per sample; scaling them afterwards equals normalize=True, which equals the
    reference's normalize_batch_voxel on the raw events (NumPy restatement pinned by G13) -- with the training configuration's hot pixels
    (hundreds of events per frame on a few pixels, config/train_v2v_e2vid_10k.yaml:75 draws hot_pixel_std up to 10; more here) in the batch."""
    from v2v_amd import postops
    from v2v_amd.loader import RingLoader
    cfg = dict(defer_sim=True, fixed_seed=11, hot_pixel_std_range=[60, 80], hot_pixel_fraction_range=[0.003, 0.004], threshold_range=[0.1, 0.3])
    ds = _make_ds(tmp_path, **cfg)
    raw = next(iter(RingLoader(ds, batch_size=4, num_workers=0, pad_to=16, normalize="scales")))
    done = next(iter(RingLoader(ds, batch_size=4, num_workers=0, pad_to=16, normalize=True)))
    plain = next(iter(RingLoader(ds, batch_size=4, num_workers=0, pad_to=16, normalize=False)))
    assert torch.equal(raw["events"], plain["events"]) and "event_scales" not in done and raw["event_scales"].shape == (4, 2)
    assert float(raw["events"].abs().max()) > 255                                                  # hot pixels beyond the counting range
    assert torch.equal(postops.apply_scales(raw["events"], raw["event_scales"], 16), done["events"])
    v = plain["events"].cpu().numpy()
    flat = np.sort(v.reshape(4, -1), axis=1)
    m = flat.shape[1]
    pos = np.maximum(flat[:, int(0.99 * m) - 1], 1).reshape(4, 1, 1, 1, 1)
    neg = np.maximum(-flat[:, int(0.01 * m) - 1], 1).reshape(4, 1, 1, 1, 1)
    assert np.array_equal(done["events"].cpu().numpy(), np.where(v > 0, v / pos, v / neg).astype(np.float32))
    assert np.array_equal(raw["event_scales"].cpu().numpy(), np.concatenate([neg.reshape(4, 1), pos.reshape(4, 1)], 1).astype(np.float32))


@pytest.mark.gpu
def test_create_dataloader_is_train_pys_with_a_ring_loader(tmp_path):
    """The reference's create_dataloader(dataset, configs, batch_size, local_rank) (train.py:52-65) over the plugin loader's ConcatDataset
    nesting: a RingLoader with the same sampler type, drop_last, len() and batch dict, batches on the GPU, epochs restartable."""
    from torch.utils.data import ConcatDataset, RandomSampler
    from v2v_amd.loader import RingLoader, create_dataloader
    ds = ConcatDataset([ConcatDataset([_make_ds(tmp_path, n_videos=7)])])
    dl = create_dataloader(ds, {"num_workers": 2, "persistent_workers": True, "pin_memory": True, "normalize_in_loader": True}, 3, None)
    assert isinstance(dl, RingLoader) and isinstance(dl.sampler, RandomSampler) and len(dl) == 2 and dl.drop_last
    for epoch in range(2):
        seen = 0
        for batch in dl:
            assert batch["events"].shape == (3, 4, 5, 32, 32) and batch["events"].is_cuda and batch["frame"].shape == (3, 4, 1, 32, 32)
            ev = batch["events"]                                       # normalised by the 1 % / 99 % k-th values of every sample (train_utils.py:147-166)
            per_sample = ev.reshape(3, -1)
            assert float((per_sample > 1).float().mean(1).max()) <= 0.0101 and float((per_sample < -1).float().mean(1).max()) <= 0.0101
            assert float(per_sample.abs().max()) >= 1.0 - 1e-6          # ... and something sits at the k-th value
            assert batch["data_source_idx"].tolist() == [11, 11, 11] and set(batch["v2e_params"]) == {"pos_thres", "neg_thres", "base_noise_std", "hot_pixel_fraction", "hot_pixel_std"}
            seen += 1
        assert seen == 2
    dl.close()


@pytest.mark.gpu
def test_packed_clips_launch_equals_the_gathered_launch():
    """v2v_esim_voxel_ex_hip with a frame index and clip offsets (every decoded frame stored once, clips of different stored lengths
    packed back to back) == v2v_esim_voxel_padded_hip on the gathered clips, bit for bit, for every work-item mapping, with the
    writer's statistics; the packed frame assembly equals the plain one."""
    from v2v_amd import _lib, esim
    from v2v_amd.loader import clip_frames_f32, clip_frames_packed
    g = np.random.default_rng(3)
    b, n, h, w = 5, 21, 32, 48
    stored = [int(g.integers(3, n + 1)) for _ in range(b)]
    stored[0], stored[1] = n, 1                                                      # no pause at all / a video that never moves
    clips_h = [g.integers(0, 256, (u, h, w), dtype=np.uint8) for u in stored]
    fidx = np.stack([np.sort(np.concatenate([np.arange(u), g.integers(0, u, n - u)])) if u < n else np.arange(n) for u in stored]).astype(np.int32)
    offs, flat, pos = [], [], 0
    for c in clips_h:
        offs.append(pos)
        flat.append(c.ravel())
        pad = (-c.size) % 16
        flat.append(np.zeros(pad, np.uint8))
        pos += c.size + pad
    flat_d = torch.from_numpy(np.concatenate(flat)).cuda()
    offs_d, fidx_d = torch.tensor(offs, dtype=torch.int64).cuda(), torch.from_numpy(fidx).cuda()
    gathered = torch.from_numpy(np.stack([c[i] for c, i in zip(clips_h, fidx)])).cuda()
    params = torch.tensor([[0.2 + 0.02 * i, 0.25, 0.05, 1e-3, 2.0] for i in range(b)], dtype=torch.float64).cuda()
    keys = torch.tensor([[77 + i, 10 * i] for i in range(b)], dtype=torch.int64).cuda()
    for mapping in ("4px", "2px", "1px", "auto"):
        st_a = torch.zeros((b, _lib.VOXEL_STATS_WORDS), dtype=torch.int32, device="cuda")
        st_b = torch.zeros_like(st_a)
        want = esim.esim_voxel_batch(gathered, params, bin_mode="sum", num_bins=5, clip_keys=keys, no_noise=False, pad_to=16, stats=st_a, mapping=mapping)
        got = esim.esim_voxel_packed(flat_d, offs_d, fidx_d, h, w, params, keys, num_bins=5, pad_to=16, stats=st_b, mapping=mapping)
        assert torch.equal(got, want) and torch.equal(st_a, st_b) and float(want.abs().sum()) > 0
    pick = torch.from_numpy(fidx[:, [5, 10, 20]].copy()).cuda()
    assert torch.equal(clip_frames_packed(flat_d, offs_d, pick, h, w), clip_frames_f32(gathered, [5, 10, 20]))
    with pytest.raises(ValueError):                                                   # indexed launches have no float64 / external-noise instances
        esim.esim_voxel_packed(flat_d, offs_d, fidx_d.long(), h, w, params, keys)


@pytest.mark.gpu
def test_create_dataloader_under_two_ddp_ranks(tmp_path):
    """train.py under torchrun: two processes, each `create_dataloader(dataset, configs, batch_size, local_rank)` -> DistributedSampler +
    RingLoader.  The ranks receive disjoint halves of the samples, their union is the dataset, and set_epoch reshuffles."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket() as sock:                                   # an ephemeral port, like bench.self_launch: no clash with a concurrent session
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "ddp_loader_rank.py"), str(tmp_path)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in (0, 1))
    for epoch in range(2):
        a, b = set(r0[epoch]), set(r1[epoch])
        assert len(r0[epoch]) == 6 and len(r1[epoch]) == 6 and not (a & b) and len(a | b) == 12      # 12 samples, 2 ranks, drop_last, batch 3
    assert r0[0] != r0[1]                                                                               # set_epoch changed the order


@pytest.mark.gpu
def test_ring_loader_partial_last_batch(tmp_path):
    """drop_last=False: the epoch's last batch is smaller; every sample still equals the per-sample path."""
    from torch.utils.data import default_collate
    from v2v_amd.datasets import SimulatingCollator
    from v2v_amd.loader import RingLoader
    ds = _make_ds(tmp_path, n_videos=7, defer_sim=True, fixed_seed=21)
    col = SimulatingCollator.from_configs({"num_bins": 5}, output_device="cuda", pad_to=16, normalize=True)
    loader = RingLoader(ds, batch_size=3, num_workers=2, drop_last=False, pad_to=16, normalize=True)
    sizes = []
    for bi, batch in enumerate(loader):
        idx = list(range(3 * bi, min(3 * bi + 3, 7)))
        want = col.simulate(default_collate([ds[i] for i in idx]))
        sizes.append(batch["events"].shape[0])
        assert torch.equal(batch["events"], want["events"]) and torch.equal(batch["frame"], want["frame"])
        assert batch["data_source_idx"].shape == (len(idx),) and all(v.shape == (len(idx),) for v in batch["v2e_params"].values())
    assert sizes == [3, 3, 1]
    loader.close()


@pytest.mark.gpu
def test_ring_loader_survives_an_abandoned_epoch(tmp_path):
    """The training loop breaks out of an epoch (validation, early stop): the next epoch starts clean -- no worker of the abandoned one
    writes into a slot that is handed out again, every batch still equals the per-sample path."""
    from torch.utils.data import default_collate
    from v2v_amd.datasets import SimulatingCollator
    from v2v_amd.loader import RingLoader
    ds = _make_ds(tmp_path, n_videos=12, defer_sim=True, fixed_seed=9)
    col = SimulatingCollator.from_configs({"num_bins": 5}, output_device="cuda")
    for persistent in (False, True):
        loader = RingLoader(ds, batch_size=2, num_workers=3, persistent_workers=persistent)
        for bi, batch in enumerate(loader):
            if bi == 1:
                break                                                             # 4 more batches are in flight in the workers
        for epoch in range(2):
            for bi, batch in enumerate(loader):
                want = col.simulate(default_collate([ds[2 * bi], ds[2 * bi + 1]]))
                assert torch.equal(batch["events"], want["events"]) and torch.equal(batch["frame"], want["frame"]), (persistent, epoch, bi)
            assert bi == 5
        loader.close()


@pytest.mark.gpu
def test_ring_loader_retires_an_older_iterator(tmp_path):
    """The ring's slots serve one epoch at a time: a second iter(loader) retires the first one loudly instead of sharing slots with it."""
    from v2v_amd.loader import RingLoader
    ds = _make_ds(tmp_path, n_videos=8, defer_sim=True, fixed_seed=4)
    loader = RingLoader(ds, batch_size=2, num_workers=0)
    first = iter(loader)
    next(first)
    second = iter(loader)
    assert next(second)["events"].shape[0] == 2
    with pytest.raises(RuntimeError, match="retired"):
        next(first)
    assert sum(1 for _ in second) == 3                                            # the newer iterator finishes its epoch
    loader.close()


@pytest.mark.gpu
def test_train_pys_loop_trains_a_stock_model_on_ring_loader_batches(tmp_path):
    """train.py:71-88 verbatim in miniature: `batch[k] = v.to(device)` for every tensor (a no-op for the GPU tensors, a copy for
    data_source_idx), a stock torch model run over the sequence, loss.backward(), optimizer.step() -- the batches a RingLoader hands out are
    ordinary leaf tensors a training loop can consume."""
    from v2v_amd.loader import create_dataloader
    ds = _make_ds(tmp_path, n_videos=6)
    dl = create_dataloader(ds, {"num_workers": 2, "normalize_in_loader": True, "pad_events_to": 16}, 2, None)
    device = torch.device("cuda")
    model = torch.nn.Sequential(torch.nn.Conv2d(5, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 1, 3, padding=1)).to(device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for batch in dl:
        opt.zero_grad()
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                batch[k] = v.to(device)
        assert batch["events"].data_ptr() != 0 and batch["data_source_idx"].is_cuda and not batch["events"].requires_grad
        loss = 0.0
        for t in range(batch["events"].shape[1]):
            loss = loss + torch.nn.functional.mse_loss(model(batch["events"][:, t]), batch["frame"][:, t])
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert len(losses) == 3 and all(np.isfinite(losses))
    dl.close()


@pytest.mark.gpu
def test_ring_loader_without_page_locking_still_delivers_the_same_batches(tmp_path, monkeypatch):
    """RLIMIT_MEMLOCK too small for the ring: hipHostRegister fails, the loader warns and runs out of pageable shared memory (the runtime
    stages the copies) -- slower, same batches."""
    from torch.utils.data import default_collate
    from v2v_amd.datasets import SimulatingCollator
    from v2v_amd import loader as L
    ds = _make_ds(tmp_path, n_videos=6, defer_sim=True, fixed_seed=3)
    col = SimulatingCollator.from_configs({"num_bins": 5}, output_device="cuda")

    class _NoPin:
        def __init__(self, rt):
            self.rt = rt

        def cudaHostRegister(self, *a):
            return 2                                                               # hipErrorOutOfMemory

        def __getattr__(self, k):
            return getattr(self.rt, k)

    real = torch.cuda.cudart()
    monkeypatch.setattr(torch.cuda, "cudart", lambda: _NoPin(real))
    with pytest.warns(RuntimeWarning, match="stays pageable"):
        loader = L.RingLoader(ds, batch_size=2, num_workers=2)
    monkeypatch.undo()
    assert not loader._registered
    for bi, batch in enumerate(loader):
        want = col.simulate(default_collate([ds[2 * bi], ds[2 * bi + 1]]))
        assert torch.equal(batch["events"], want["events"]) and torch.equal(batch["frame"], want["frame"])
    assert bi == 2
    loader.close()


@pytest.mark.gpu
def test_ring_loader_batches_through_the_network_in_one_graphed_call(tmp_path):
    """BASELINE config 5 at miniature size, as tools/loader_bench.py runs it: RingLoader(normalize='scales') -> E2VIDRecurrent.forward_sequence
    (events, event_scales, graph=True) -- the reference's reset + time loop (model/train_utils.py:309-345) captured once and replayed -- gives,
    for every batch, the images of the step-by-step loop on the normalised events."""
    from v2v_amd import postops
    from v2v_amd.loader import RingLoader
    from v2v_amd.unet import E2VIDRecurrent
    ds = _make_ds(tmp_path, n_videos=8, defer_sim=True, fixed_seed=3)
    torch.manual_seed(0)
    net = E2VIDRecurrent(dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
                              num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)).cuda().eval()
    loader = RingLoader(ds, batch_size=4, num_workers=2, drop_last=True, pad_to=16, normalize="scales")
    n = 0
    with torch.no_grad():
        for batch in loader:
            ev, sc = batch["events"], batch["event_scales"]
            got = net.forward_sequence(ev, sc, graph=True).clone()
            normed = postops.apply_scales(ev, sc, 16)                          # what the head divides by while it reads the raw events
            net.reset_states()
            want = torch.stack([net(normed[:, t])["image"] for t in range(ev.shape[1])], 1)
            assert got.shape == (4, ev.shape[1], 1, 32, 32) and torch.equal(got, want)
            n += 1
    assert n == 2 and len(net._sequence_graphs) == 1
    loader.close()
