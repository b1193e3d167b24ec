"""Real-data side of `make_voxel`: the Monash-layout reader, the TestH5Dataset drop-in (data/testh5.py:14-173)
and the cached-voxel writer (scripts/esim_to_voxel.py:17-56 over DynamicH5Dataset, data/dataset.py:176-231,375-427) against
golden G16 = the REFERENCE's own loaders run on the committed fixture sequence (tests/golden/make_goldens.py::g16)."""
import os

import numpy as np
import pytest
import torch

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g16_monash_sequence.npz")
CFGS = {"a": {"sequence_length": 4, "num_bins": 5, "dataset_name": "hqf"},
        "b": {"sequence_length": 5, "warm_up_length": 1, "num_bins": 3, "interpolate_bins": True, "output_additional_frame": True,
              "output_additional_evs": True, "image_range": 1, "dataset_name": "hqf"}}


def test_monash_store_reads_the_layout(golden):
    from v2v_amd import monash
    g = golden("g16_monash_sequence.npz")
    with monash.open_sequence(FIX) as f:
        assert f.image_keys == ["image%09d" % i for i in range(9)] and f.image(f.image_keys[3]).shape == (36, 48)
        assert int(f.image_attr("image000000004", "event_idx")) == int(g["images/event_idx"][4])
        assert np.array_equal(f.events("ts", 10, 20), g["events/ts"][10:20]) and f.attr("source") == "hqf" and not f.has_flow()
        assert list(f.attr("sensor_resolution")) == [36, 48] and int(f.attr("num_imgs")) == 9


@pytest.mark.parametrize("tag", ["a", "b"])
def test_testh5_sample_table_equals_reference(golden, tag):
    """Host logic only (no GPU): the (begin, real_begin, end) table of data/testh5.py:43-52 incl. warm-up overlap."""
    from v2v_amd.testh5 import TestH5Dataset
    g = golden("g16_monash_sequence.npz")
    ds = TestH5Dataset(FIX, CFGS[tag])
    assert len(ds) == int(g[f"th5_{tag}__len"]) and np.array_equal(np.array(ds.samples), g[f"th5_{tag}__samples"])
    assert (ds.H, ds.W) == (36, 48) and ds.sequence_name == "g16_monash_sequence"


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_testh5_getitem_equals_reference(golden, tag):
    """Every sample of the fixture sequence: same dict as the reference's __getitem__.  Discrete bins: exact; interpolated bins:
    the same float64 terms summed by atomics in another order, then cast to float32 like the reference does (1e-6)."""
    from v2v_amd.testh5 import TestH5Dataset
    g = golden("g16_monash_sequence.npz")
    ds = TestH5Dataset(FIX, CFGS[tag])
    for i in range(len(ds)):
        s = ds[i]
        assert set(s) == {"frame", "events", "data_source_idx", "sequence_name", "real_begin_idx", "frame_idx"}
        assert s["frame"].dtype == torch.float32 and np.array_equal(s["frame"].numpy(), g[f"th5_{tag}__{i}__frame"])
        want = g[f"th5_{tag}__{i}__events"]
        assert s["events"].dtype == torch.float32 and s["events"].shape == want.shape
        if CFGS[tag].get("interpolate_bins"):
            np.testing.assert_allclose(s["events"].numpy(), want, rtol=1e-6, atol=1e-6)
        else:
            assert np.array_equal(s["events"].numpy(), want)
        assert np.array_equal(np.stack([s["real_begin_idx"].numpy(), s["frame_idx"].numpy()]), g[f"th5_{tag}__{i}__meta"])
        assert int(s["data_source_idx"]) == int(g[f"th5_{tag}__{i}__source"]) and s["data_source_idx"].dtype == torch.int64
        assert s["sequence_name"] == ["g16_monash_sequence"] * s["frame_idx"].numel()
    # make_voxel(self, evs) itself, on one interval (data/testh5.py:60-90)
    lo, hi = int(g["images/event_idx"][0]), int(g["images/event_idx"][1])
    one = ds.make_voxel([g["events/ts"][lo:hi], g["events/xs"][lo:hi], g["events/ys"][lo:hi], g["events/ps"][lo:hi]])
    assert one.shape == (ds.num_bins, 36, 48) and one.dtype == np.float64


@pytest.mark.gpu
@pytest.mark.parametrize("bilinear", [False, True])
def test_voxel_cache_equals_reference_loader(golden, tmp_path, bilinear):
    """scripts/esim_to_voxel.py:17-56: all `between_frames` grids of the sequence in one segmented launch == the stacked,
    float32-cast items of the reference's DynamicH5Dataset (incl. the < 3 events -> empty grid rule and dt / timestamps)."""
    from v2v_amd import voxel_cache
    g = golden("g16_monash_sequence.npz")
    tag = "bil" if bilinear else "nobi"
    out = tmp_path / f"cache_{tag}.npz"
    data = voxel_cache.convert(FIX, str(out), temporal_bilinear=bilinear)
    z = np.load(out)
    for k in ("frames", "flow", "events", "timestamps", "dt"):
        assert z[k].dtype == np.float32 and np.array_equal(z[k], data[k])
    assert np.array_equal(z["frames"], g[f"cache_{tag}__frames"]) and bool(g[f"cache_{tag}__flow_is_zero"]) and not z["flow"].any()
    assert np.array_equal(z["timestamps"], g[f"cache_{tag}__timestamps"]) and np.array_equal(z["dt"], g[f"cache_{tag}__dt"])
    want = g[f"cache_{tag}__events"]
    assert z["events"].shape == want.shape == (8, 5, 36, 48)
    if bilinear:
        np.testing.assert_allclose(z["events"], want, rtol=1e-5, atol=1e-5)      # float32 atomics: torch's index_put_ order differs
    else:
        assert np.array_equal(z["events"], want)
    assert not z["events"][2].any() and not z["events"][3].any() and not z["events"][4].any()    # 0, 1 and 2 events: empty grids


@pytest.mark.gpu
def test_event_kernels_edge_semantics():
    """Index and time-span corner cases of the scatter kernels follow the reference's primitives: np.add.at / index_put_ wrap a
    negative index once, np.ravel_multi_index raises; an event of the segmented form outside every interval is ignored even if
    it points outside the sensor; a zero time span gives NaN bilinear weights in every bin of the touched pixel."""
    from v2v_amd import voxel
    h, w, nb = 6, 8, 4
    ts = np.array([0.0, 0.1, 0.2, 0.3]); xs = np.array([1, -1, 3, 2]); ys = np.array([0, 2, -2, 5]); ps = np.array([1, 0, 1, 1])
    want = np.zeros((nb, h, w))
    pol = ps.astype(np.int8) * 2 - 1
    tus = ((ts - ts[0]) * 1e6).astype(np.int64)
    np.add.at(want, (np.floor(tus / ((tus[-1] + 0.001) / nb)).astype(np.uint8), ys, xs), pol)       # data/testh5.py:70-75, negative indices wrap
    assert np.array_equal(voxel.make_voxel([ts, xs, ys, ps], h, w, nb, interpolate_bins=False), want)
    with pytest.raises(IndexError):
        voxel.make_voxel([ts, xs - 20, ys, ps], h, w, nb, interpolate_bins=False)
    with pytest.raises((IndexError, ValueError)):
        voxel.events_to_voxel(xs, ys, ts, pol.astype(float), nb, (h, w))                            # ravel_multi_index: negative -> error
    # segmented: events 0 and 3 lie outside [1, 3) and point anywhere
    seg = voxel.make_voxels_segmented([ts, np.array([99, 1, 3, -77]), np.array([99, 2, 1, 0]), ps], [1, 3], h, w, nb)
    assert seg.shape == (1, nb, h, w) and seg.sum() == 0 and np.abs(seg).sum() == 2
    # zero time span: NaN in every bin of the pixel, zero elsewhere (event_utils.py:713-719 with dt == 0)
    v = voxel.events_to_voxel(np.array([2, 2]), np.array([1, 1]), np.array([0.5, 0.5]), np.array([1.0, 1.0]), nb, (h, w))
    assert np.isnan(v[:, 1, 2]).all() and np.nansum(np.abs(v)) == 0 and np.isnan(v).sum() == nb
    t32 = voxel.events_to_voxel_torch(torch.tensor([2]), torch.tensor([1]), torch.tensor([0.5]), torch.tensor([1.0]), nb, sensor_size=(h, w))
    assert bool(torch.isnan(t32[:, 1, 2]).all()) and int(torch.isnan(t32).sum()) == nb
    t32 = voxel.events_to_voxel_torch(torch.tensor([-1, 2]), torch.tensor([-1, 0]), torch.tensor([0.0, 1.0]), torch.tensor([1.0, 1.0]), nb, sensor_size=(h, w))
    assert float(t32[0, h - 1, w - 1]) == 1.0 and float(t32[nb - 1, 0, 2]) == 1.0                    # index_put_ wraps


def _h5_copy(tmp_path, monkeypatch):
    """The fixture sequence under an .h5 name + tests/fake_h5py.py injected as `h5py` (none in this image)."""
    import shutil
    import sys
    import fake_h5py
    monkeypatch.setitem(sys.modules, "h5py", fake_h5py)
    path = tmp_path / "g16_monash_sequence.h5"
    shutil.copy(FIX, path)
    return str(path)


def test_h5_branch_of_the_monash_store_equals_the_npz_form(tmp_path, monkeypatch, golden):
    """v2v_amd/monash.py:H5Sequence (the branch real Monash .h5 files take) through a stand-in h5py: the same access pattern gives the same
    arrays, attributes and image keys as the .npz form the other tests pin, and TestH5Dataset builds the same sample table from it."""
    from v2v_amd import monash
    from v2v_amd.testh5 import TestH5Dataset
    path = _h5_copy(tmp_path, monkeypatch)
    with monash.open_sequence(path) as h, monash.open_sequence(FIX) as z:
        assert isinstance(h, monash.H5Sequence) and isinstance(z, monash.NpzSequence)
        assert h.image_keys == z.image_keys and not h.has_flow()
        for key in h.image_keys:
            assert np.array_equal(h.image(key), z.image(key)) and int(h.image_attr(key, "event_idx")) == int(z.image_attr(key, "event_idx"))
            assert float(h.image_attr(key, "timestamp")) == float(z.image_attr(key, "timestamp"))
        for name in ("ts", "xs", "ys", "ps"):
            assert np.array_equal(h.events(name), z.events(name)) and np.array_equal(h.events(name, 7, 31), z.events(name, 7, 31))
        assert h.attr("source") == z.attr("source") and list(h.attr("sensor_resolution")) == list(z.attr("sensor_resolution"))
        assert h.attr("no_such_attribute", 5) == 5
    for tag in ("a", "b"):
        a, b = TestH5Dataset(path, CFGS[tag]), TestH5Dataset(FIX, CFGS[tag])
        assert len(a) == len(b) and a.samples == b.samples and (a.H, a.W) == (b.H, b.W) and a.sequence_name == b.sequence_name


@pytest.mark.gpu
def test_h5_branches_end_to_end(tmp_path, monkeypatch, golden):
    """TestH5Dataset.__getitem__ and the cached-voxel writer over the .h5 container (stand-in h5py): same samples / same file contents as the
    .npz form that golden G16 pins."""
    from v2v_amd import voxel_cache
    from v2v_amd.testh5 import TestH5Dataset
    path = _h5_copy(tmp_path, monkeypatch)
    a, b = TestH5Dataset(path, CFGS["a"]), TestH5Dataset(FIX, CFGS["a"])
    for i in range(len(a)):
        sa, sb = a[i], b[i]
        assert set(sa) == set(sb)
        for k in sa:
            assert torch.equal(sa[k], sb[k]) if isinstance(sa[k], torch.Tensor) else sa[k] == sb[k], k
    out_h5, out_npz = tmp_path / "cache.h5", tmp_path / "cache.npz"
    d1 = voxel_cache.convert(path, str(out_h5), temporal_bilinear=False)
    d2 = voxel_cache.convert(FIX, str(out_npz), temporal_bilinear=False)
    import fake_h5py
    with fake_h5py.File(str(out_h5), "r") as f:
        z = np.load(out_npz)
        for k in ("frames", "flow", "events", "timestamps", "dt"):
            assert f[k][()].dtype == np.float32 and np.array_equal(f[k][()], z[k]) and np.array_equal(d1[k], d2[k])
        assert f.attrs["source"] == "esim" and list(f.attrs["sensor_resolution"]) == [36, 48]


# ---- the other two loaders of data/testh5.py on the same sequence: golden G19 = the reference's TestH5EventDataset / FPS_H5Dataset
EV_CFGS = {"a": {"sequence_length": 4, "num_bins": 5, "dataset_name": "hqf"},
           "b": {"sequence_length": 5, "warm_up_length": 1, "num_bins": 3, "output_additional_frame": True, "image_range": 1, "dataset_name": "evaid"}}
FPS_CFGS = {"a": {"sequence_length": 6, "num_bins": 5, "FPS": 100, "H": 36, "W": 48, "dataset_name": "evbird"},
            "b": {"sequence_length": 80, "num_bins": 3, "interpolate_bins": True, "FPS": 40, "H": 40, "W": 50, "dataset_name": "evbird"}}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_event_dataset_equals_reference(golden, tag):
    """TestH5EventDataset (data/testh5.py:305-381) returns raw event rows, no voxel grids: host logic only, exact (no GPU needed) --
    [x, y, t, p in {-1,+1}, 0] in float64 per image interval, one row of zeros for the empty interval, frames as TestH5Dataset."""
    from v2v_amd.testh5 import TestH5EventDataset
    g = golden("g19_event_and_fps_loaders.npz")
    ds = TestH5EventDataset(FIX, EV_CFGS[tag])
    assert len(ds) == int(g[f"ev_{tag}__len"])
    for i in range(len(ds)):
        s = ds[i]
        assert set(s) == {"frame", "events", "data_source_idx", "sequence_name", "real_begin_idx", "frame_idx"}
        assert s["frame"].dtype == torch.float32 and np.array_equal(s["frame"].numpy(), g[f"ev_{tag}__{i}__frame"])
        assert isinstance(s["events"], list) and len(s["events"]) == int(g[f"ev_{tag}__{i}__n"])
        for j, e in enumerate(s["events"]):
            assert e.dtype == torch.float64 and np.array_equal(e.numpy(), g[f"ev_{tag}__{i}__events{j}"]), (i, j)
        assert np.array_equal(np.stack([s["real_begin_idx"].numpy(), s["frame_idx"].numpy()]), g[f"ev_{tag}__{i}__meta"])
        assert int(s["data_source_idx"]) == int(g[f"ev_{tag}__{i}__source"]) and s["data_source_idx"].dtype == torch.int64
    assert any(e.shape == (1, 5) and not e.any() for i in range(len(ds)) for e in ds[i]["events"])      # the fixture's empty interval


@pytest.mark.parametrize("tag", ["a", "b"])
def test_fps_dataset_cut_table_equals_reference(golden, tag):
    """FPS_H5Dataset's constructor (data/testh5.py:452-481): number of cuts, np.searchsorted borders, sample table -- host logic only."""
    from v2v_amd.testh5 import FPS_H5Dataset
    g = golden("g19_event_and_fps_loaders.npz")
    ds = FPS_H5Dataset(FIX, FPS_CFGS[tag])
    assert len(ds) == int(g[f"fps_{tag}__len"]) and np.array_equal(np.array(ds.samples), g[f"fps_{tag}__samples"])
    assert np.array_equal(np.asarray(ds.event_idx), g[f"fps_{tag}__event_idx"]) and (ds.H, ds.W) == (FPS_CFGS[tag]["H"], FPS_CFGS[tag]["W"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_fps_dataset_getitem_equals_reference(golden, tag):
    """Every sample: the reference's dict (events [L,Tb,H,W] float32, data_source_idx, sequence_name), one segmented launch per sample.
    Discrete bins exact; interpolated bins to 1e-6 (float64 atomics in another order, then the reference's float32 cast)."""
    from v2v_amd.testh5 import FPS_H5Dataset
    g = golden("g19_event_and_fps_loaders.npz")
    ds = FPS_H5Dataset(FIX, FPS_CFGS[tag])
    for i in range(len(ds)):
        s = ds[i]
        want = g[f"fps_{tag}__{i}__events"]
        assert set(s) == {"events", "data_source_idx", "sequence_name"} and s["events"].dtype == torch.float32 and s["events"].shape == want.shape
        if FPS_CFGS[tag].get("interpolate_bins"):
            np.testing.assert_allclose(s["events"].numpy(), want, rtol=1e-6, atol=1e-6)
        else:
            assert np.array_equal(s["events"].numpy(), want)
        assert int(s["data_source_idx"]) == int(g[f"fps_{tag}__{i}__source"]) and s["sequence_name"] == ["g16_monash_sequence"] * want.shape[0]


# ---- MVSEC-style flow sequences and voxel caches: golden G20 = the reference's TestH5FlowDataset / TestH5CacheDataset
FLOW_CFGS = {"a": {"sequence_length": 4, "num_bins": 5, "dataset_name": "mvsec"},
             "b": {"sequence_length": 3, "num_bins": 3, "interpolate_bins": True, "output_additional_frame": True, "output_additional_evs": True, "image_range": 1,
                   "dataset_name": "mvsec", "max_samples": 2}}


def _flow_fixture(tmp_path, golden):
    """G16's events and images + G20's flow maps as one .npz sequence (the layout v2v_amd.monash documents)."""
    g16, g20 = golden("g16_monash_sequence.npz"), golden("g20_flow_and_cache_loaders.npz")
    path = tmp_path / "indoor_flying1.npz"
    np.savez(path, **{k: g16[k] for k in g16 if k.split("/")[0] in ("events", "images", "attrs")}, **{k: g20[k] for k in g20 if k.startswith("flow/")})
    return str(path), g20


@pytest.mark.parametrize("tag", ["a", "b"])
def test_flow_dataset_sample_table_equals_reference(tmp_path, golden, tag):
    from v2v_amd.testh5 import TestH5FlowDataset
    path, g = _flow_fixture(tmp_path, golden)
    ds = TestH5FlowDataset(path, FLOW_CFGS[tag])
    assert len(ds) == int(g[f"flow_{tag}__len"]) and np.array_equal(np.array(ds.samples), g[f"flow_{tag}__samples"]) and (ds.H, ds.W) == (36, 48)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a", "b"])
def test_flow_dataset_getitem_equals_reference(tmp_path, golden, tag):
    """TestH5FlowDataset.__getitem__ (data/testh5.py:218-303): frames by the maps' image_idx (clamped), voxel grids of the events between
    consecutive maps (an empty and a 3-event interval in the fixture), the maps, the extra leading frame / grid of configuration b."""
    from v2v_amd.testh5 import TestH5FlowDataset
    path, g = _flow_fixture(tmp_path, golden)
    ds = TestH5FlowDataset(path, FLOW_CFGS[tag])
    for i in range(len(ds)):
        s = ds[i]
        assert set(s) == {"frame", "events", "flow", "data_source_idx", "sequence_name", "frame_idx"}
        assert s["frame"].dtype == torch.float32 and np.array_equal(s["frame"].numpy(), g[f"flow_{tag}__{i}__frame"])
        assert np.array_equal(s["flow"].numpy(), g[f"flow_{tag}__{i}__flow"]) and np.array_equal(s["frame_idx"].numpy(), g[f"flow_{tag}__{i}__frame_idx"])
        want = g[f"flow_{tag}__{i}__events"]
        assert s["events"].dtype == torch.float32 and s["events"].shape == want.shape
        if FLOW_CFGS[tag].get("interpolate_bins"):
            np.testing.assert_allclose(s["events"].numpy(), want, rtol=1e-6, atol=1e-6)
        else:
            assert np.array_equal(s["events"].numpy(), want)
        assert int(s["data_source_idx"]) == int(g[f"flow_{tag}__{i}__source"]) and s["sequence_name"] == ["indoor_flying1"] * s["frame_idx"].numel()


def test_cache_dataset_equals_reference(tmp_path, golden):
    """TestH5CacheDataset (data/testh5.py:383-446) over a cache in the .npz form: sample table, slices, per-item source indices; a
    configuration that disagrees with the cache's attributes is refused by the same asserts."""
    from v2v_amd.testh5 import TestH5CacheDataset
    g = golden("g20_flow_and_cache_loaders.npz")
    path = tmp_path / "bike_bay_hdr.npz"
    np.savez(path, frames=g["cache__frames"], events=g["cache__events"], **{"attrs/num_bins": np.array(5), "attrs/interpolate_bins": np.array(False)})
    ds = TestH5CacheDataset(str(path), {"sequence_length": 3, "num_bins": 5, "dataset_name": "hqf"})
    assert len(ds) == int(g["cache__len"]) and np.array_equal(np.array(ds.samples), g["cache__samples"]) and (ds.H, ds.W) == (36, 48)
    for i in range(len(ds)):
        s = ds[i]
        assert set(s) == {"frame", "events", "data_source_idx", "sequence_name"}
        assert np.array_equal(s["frame"].numpy(), g[f"cache__{i}__frame"]) and np.array_equal(s["events"].numpy(), g[f"cache__{i}__events"])
        assert np.array_equal(s["data_source_idx"].numpy(), g[f"cache__{i}__source"]) and s["sequence_name"] == ["bike_bay_hdr"] * len(s["data_source_idx"])
    with pytest.raises(AssertionError):
        TestH5CacheDataset(str(path), {"num_bins": 3})
    with pytest.raises(AssertionError):
        TestH5CacheDataset(str(path), {"num_bins": 5, "interpolate_bins": True})


@pytest.mark.gpu
def test_cache_writer_produces_what_the_reference_reader_was_given(tmp_path, golden):
    """v2v_amd.voxel_cache.testh5_to_cache on G16's sequence == the cache golden G20 fed to the reference's reader (= the reference's
    TestH5Dataset items, stacked), and reading it back through TestH5CacheDataset gives the golden samples."""
    from v2v_amd.testh5 import TestH5CacheDataset
    from v2v_amd.voxel_cache import testh5_to_cache
    g = golden("g20_flow_and_cache_loaders.npz")
    out = tmp_path / "bike_bay_hdr.npz"
    d = testh5_to_cache(FIX, str(out), {"num_bins": 5, "dataset_name": "hqf"})
    assert np.array_equal(d["frames"], g["cache__frames"]) and np.array_equal(d["events"], g["cache__events"])
    ds = TestH5CacheDataset(str(out), {"sequence_length": 3, "num_bins": 5})
    assert all(np.array_equal(ds[i]["events"].numpy(), g[f"cache__{i}__events"]) for i in range(len(ds)))


@pytest.mark.gpu
def test_flow_and_cache_loaders_over_the_h5_container(tmp_path, monkeypatch, golden):
    """The .h5 branches of the flow and cache loaders (stand-in h5py): same samples as over the .npz form that golden G20 pins; the cache
    written as .h5 reads back through TestH5CacheDataset."""
    import shutil
    import sys
    import fake_h5py
    from v2v_amd.testh5 import TestH5CacheDataset, TestH5FlowDataset
    from v2v_amd.voxel_cache import testh5_to_cache
    npz, g = _flow_fixture(tmp_path, golden)
    monkeypatch.setitem(sys.modules, "h5py", fake_h5py)
    h5 = tmp_path / "indoor_flying1.h5"
    shutil.copy(npz, h5)
    a, b = TestH5FlowDataset(str(h5), FLOW_CFGS["b"]), TestH5FlowDataset(npz, FLOW_CFGS["b"])
    assert len(a) == len(b) and a.samples == b.samples and a.flow_keys == b.flow_keys
    for i in range(len(a)):
        sa, sb = a[i], b[i]
        for k in sa:
            assert torch.equal(sa[k], sb[k]) if isinstance(sa[k], torch.Tensor) else sa[k] == sb[k], k
    seq = tmp_path / "bike_bay_hdr.h5"
    shutil.copy(FIX, seq)
    out = tmp_path / "cache" / "bike_bay_hdr.h5"
    out.parent.mkdir()
    testh5_to_cache(str(seq), str(out), {"num_bins": 5, "dataset_name": "hqf"})
    ds = TestH5CacheDataset(str(out), {"sequence_length": 3, "num_bins": 5})
    assert len(ds) == int(g["cache__len"]) and all(np.array_equal(ds[i]["events"].numpy(), g[f"cache__{i}__events"]) for i in range(len(ds)))


@pytest.mark.gpu
def test_evaluation_loop_on_a_real_data_sequence(golden):
    """The reference's evaluation path end to end on the fixture sequence (test_e2vid.py's loop over model/train_utils.py:318-345): a
    TestH5Dataset sample -> events [1,L,5,36,48] -> zero padding to multiples of 16 (postops.pad_events, :322-326) -> the package
    E2VIDRecurrent step by step at batch 1 (a 48 x 48 input: 6 x 6 = 36 pixels at the third level, far below any workgroup tile) -> crop.
    The step loop, forward_sequence and its hipGraph replay agree bit for bit and the padding does not leak into the kept region's
    finiteness; a second sample continues from the first one's states as the harness does (no reset in between)."""
    from v2v_amd import postops
    from v2v_amd.testh5 import TestH5Dataset
    from v2v_amd.unet import E2VIDRecurrent
    torch.manual_seed(3)
    net = E2VIDRecurrent(dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
                              num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)).cuda().eval()
    ds = TestH5Dataset(FIX, {"sequence_length": 4, "num_bins": 5, "dataset_name": "hqf"})
    net.reset_states()
    for i in range(2):
        s = ds[i]
        events = postops.pad_events(s["events"][None].cuda(), 16)                       # [1, L, 5, 48, 48]
        assert events.shape[-2:] == (48, 48) and torch.equal(events[..., :36, :48], s["events"][None].cuda())
        before = net.states
        with torch.no_grad():
            loop = torch.stack([net(events[:, t])["image"] for t in range(events.shape[1])], dim=1)
            after = net.states
            net.states = before
            seq = net.forward_sequence(events)
        assert torch.equal(loop, seq) and bool(torch.isfinite(loop).all()) and loop.shape == (1, events.shape[1], 1, 48, 48)
        assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(after, net.states))
        pred = loop[..., :36, :48]
        assert pred.shape[-2:] == tuple(s["frame"].shape[-2:])
    with torch.no_grad():
        net.reset_states()
        fresh = net.forward_sequence(postops.pad_events(ds[0]["events"][None].cuda(), 16))
        graph = net.forward_sequence(postops.pad_events(ds[0]["events"][None].cuda(), 16), graph=True)
    assert torch.equal(fresh, graph)
