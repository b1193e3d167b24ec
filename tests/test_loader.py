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
                                   {"shake_frames": 5, "shake_std": 2.0}, {"video_degrade": "hdr", "degrade_ratio": 1.0}, {"fixed_seed": 5}])
def test_host_sample_into_equals_deferred_getitem(tmp_path, extra):
    """The slot writer consumes np.random exactly like __getitem__(defer_sim) and leaves the same clip, parameters and key."""
    from v2v_amd.loader import _SlotLayout
    ds = _make_ds(tmp_path, defer_sim=True, **extra)
    n, hw = ds.frames_per_seq + 1, ds.crop_size
    pick = ds.frame_pick()
    lay = _SlotLayout(3, n, hw, hw, len(pick), ds.color_mode != "gray")
    buf = np.zeros(lay.nbytes, dtype=np.uint8)
    clips, cframes, params, keys = lay.views(buf)
    for pos, idx in enumerate((4, 0, 2)):
        np.random.seed(100 + idx)
        want = ds[idx]
        state_after = np.random.get_state()[1].copy()
        np.random.seed(100 + idx)
        v2e = ds.host_sample_into(idx, clips[pos], params[pos], keys[pos], cframes[pos] if cframes is not None else None)
        assert np.array_equal(np.random.get_state()[1], state_after)                 # same number of draws
        assert np.array_equal(clips[pos], want["sim_frames"].numpy())
        assert np.array_equal(params[pos], want["sim_params"].numpy()) and np.array_equal(keys[pos], want["sim_key"].numpy())
        assert v2e == want["v2e_params"]
        # the frames the loader will build on the device: clip[pick] / 255 (gray) or the colour frames / 255
        if cframes is None:
            got = torch.from_numpy(clips[pos][pick]).float().unsqueeze(1) / 255
        else:
            got = torch.from_numpy(cframes[pos]).float().permute(0, 3, 1, 2) / 255
        assert torch.equal(got, want["frame"])


def test_slot_layout_is_aligned_and_disjoint():
    from v2v_amd.loader import _SlotLayout
    lay = _SlotLayout(12, 201, 128, 128, 40, True)
    offs = [lay.off_clips, lay.off_frames, lay.off_params, lay.off_keys, lay.nbytes]
    assert all(o % 256 == 0 for o in offs) and offs == sorted(offs)
    buf = np.zeros(lay.nbytes, dtype=np.uint8)
    clips, cframes, params, keys = lay.views(buf)
    clips[:] = 1
    cframes[:] = 2
    params[:] = 3.0
    keys[:] = 4
    assert (clips == 1).all() and (cframes == 2).all() and (params == 3.0).all() and (keys == 4).all()
    assert clips.shape == (12, 201, 128, 128) and cframes.shape == (12, 40, 128, 128, 3) and params.shape == (12, 5) and keys.shape == (12, 2)


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


# ----------------------------------------------------------------------------------------------------------------- GPU
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


@pytest.mark.gpu
@pytest.mark.parametrize("extra,workers", [({}, 2), ({"color_mode": "gray_in_bgr_out"}, 0), ({"output_additional_evs": True}, 2),
                                           ({"output_additional_frame": True, "crop_size": 36}, 0)])
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
