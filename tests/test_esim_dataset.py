"""v2v_amd.esim_dataset.ESIMH5Dataset against golden G21 = the REFERENCE's data/esim_dataset.py:ESIMH5Dataset run on the stored cache after
random.seed / np.random.seed (host logic only: the voxels are cached, nothing is simulated)."""
import random

import numpy as np
import pytest
import torch

CFGS = {"a": {"sequence_length": 6, "random_crop_size": 16, "noise_std": 0.1, "hot_pixel_std": 0.1, "max_hot_pixel_fraction": 0.05},
        "b": {"sequence_length": 5, "step_size": 3, "random_crop_size": None, "random_flip": False, "noise_std": 0.7, "noise_fraction": 0.3,
              "proba_pause_when_running": 0.4, "proba_pause_when_paused": 0.6, "hot_pixel_std": 2.0, "max_hot_pixel_fraction": 0.1, "integer_noise": True}}


@pytest.fixture()
def cache(tmp_path, golden):
    g = golden("g21_esim_h5_dataset.npz")
    path = tmp_path / "seq.npz"
    np.savez(path, frames=g["frames"], flow=g["flow"], events=g["events"], **{"attrs/sensor_resolution": g["attrs/sensor_resolution"]})
    return str(path), g


@pytest.mark.parametrize("tag", ["a", "b"])
def test_esim_h5_dataset_equals_reference(cache, tag):
    """Sample table, crop / flip / pause schedule and both noise models: the same two seeds give the reference's tensors bit for bit
    (Gaussian noise on every cell; signed-Poisson noise on 30 % of the cells with 40 % / 60 % pause probabilities and 10 % hot pixels)."""
    from v2v_amd.esim_dataset import ESIMH5Dataset
    path, g = cache
    ds = ESIMH5Dataset(path, CFGS[tag])
    assert len(ds) == int(g[f"ds_{tag}__len"]) and np.array_equal(np.array(ds.samples), g[f"ds_{tag}__samples"])
    assert list(ds.sensor_resolution) == [20, 24] and ds.data_source_name == "esim"
    for i in range(len(ds)):
        random.seed(100 + i)
        np.random.seed(200 + i)
        s = ds[i]
        assert set(s) == {"frame", "flow", "events", "data_source_idx"} and int(s["data_source_idx"]) == int(g[f"ds_{tag}__{i}__source"])
        for k in ("frame", "flow", "events"):
            assert s[k].dtype == torch.float32 and np.array_equal(s[k].numpy(), g[f"ds_{tag}__{i}__{k}"]), (i, k)


def test_esim_h5_dataset_reads_what_the_cache_writer_stores(cache, tmp_path, monkeypatch):
    """The .h5 branch (stand-in h5py) gives the same samples as the .npz form, and the dataset survives pickling (DataLoader workers)."""
    import pickle
    import shutil
    import sys
    import fake_h5py
    from v2v_amd.esim_dataset import ESIMH5Dataset
    path, _ = cache
    monkeypatch.setitem(sys.modules, "h5py", fake_h5py)
    h5 = tmp_path / "seq.h5"
    shutil.copy(path, h5)
    a, b = ESIMH5Dataset(str(h5), CFGS["a"]), pickle.loads(pickle.dumps(ESIMH5Dataset(path, CFGS["a"])))
    for ds in (a, b):
        random.seed(1)
        np.random.seed(2)
        ds.out = ds[1]
    assert all(torch.equal(a.out[k], b.out[k]) for k in a.out)
