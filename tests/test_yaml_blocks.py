"""Every dataset block of the reference's experiment YAMLs (golden G24: the 99 `class_name` mappings of config/*.yaml, as data) against the
package's drop-in classes: the class exists under the same name with `data.` -> `v2v_amd.` (data/data_interface.py:7-27 resolves the name
with get_obj_from_str and calls `cls(path, configs)`), its constructor takes the block as it stands, and a sample comes out where no
GPU is needed for one (deferred simulation; raw event rows; cached voxels).  No GPU in this file."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
G16 = os.path.join(HERE, "golden", "g16_monash_sequence.npz")


def _blocks():
    z = np.load(os.path.join(HERE, "golden", "g24_yaml_dataset_blocks.npz"))
    return json.loads(str(z["blocks"]))


def _package_class(name):
    mod, cls = name.rsplit(".", 1)
    mod = {"data.v2v_datasets": "v2v_amd.datasets", "data.testh5": "v2v_amd.testh5", "data.esim_dataset": "v2v_amd.esim_dataset"}[mod]
    return getattr(importlib.import_module(mod), cls)


def test_every_class_name_of_the_reference_yamls_has_a_drop_in():
    blocks = _blocks()
    assert len(blocks) == 99
    names = sorted({b["block"]["class_name"] for b in blocks})
    assert names == ["data.esim_dataset.ESIMH5Dataset", "data.testh5.FPS_H5Dataset", "data.testh5.TestH5Dataset", "data.testh5.TestH5EventDataset",
                     "data.testh5.TestH5FlowDataset", "data.v2v_datasets.WebvidDatasetV2"]
    for n in names:
        assert callable(_package_class(n))


def test_every_webvid_block_builds_and_samples(tmp_path):
    """The 8 WebvidDatasetV2 blocks (the V2V training sets of the e2vid / eraft / etnet / evflow / hyper experiments and the ablations) as
    they stand, over a synthetic frame source: constructed, one sample drawn (simulation deferred: no GPU), shapes as the block implies."""
    from test_loader import _frames
    lst = tmp_path / "videos.txt"
    lst.write_text("".join(f"clip{i}.mp4 400 0.2 0.3\n" for i in range(4)))
    blocks = [b for b in _blocks() if b["block"]["class_name"] == "data.v2v_datasets.WebvidDatasetV2"]
    assert len(blocks) == 8
    for b in blocks:
        cfg = {k: v for k, v in b["block"].items() if k not in ("class_name", "data_file")}
        cfg.update(video_list_file=str(lst), frame_source=_frames, video_size=(640, 360), defer_sim=True)
        ds = _package_class(b["block"]["class_name"])(str(tmp_path), cfg)
        np.random.seed(1)
        s = ds[1]
        n_img = cfg["sequence_length"] + (1 if cfg.get("output_additional_frame") else 0)
        assert s["frame"].shape == (n_img, 1, cfg["crop_size"], cfg["crop_size"]), b["yaml"]
        per_bin = cfg["num_bins"] * cfg.get("frames_per_bin", 1)
        n_grids = cfg["sequence_length"] + (1 if cfg.get("output_additional_evs") else 0)
        assert s["sim_frames"].shape == (n_grids * per_bin + 1, cfg["crop_size"], cfg["crop_size"]) and s["sim_frames"].dtype == torch.uint8, b["yaml"]


def test_every_h5_block_builds_on_the_fixture_sequence(tmp_path, golden):
    """The 90 real-data blocks (TestH5Dataset x 79, TestH5FlowDataset x 6, TestH5EventDataset x 4, FPS_H5Dataset x 1) and the ESIM cache block:
    every constructor takes its block on the fixture sequence (G16 + G20's flow maps; G21's cache), the sample tables are non-empty, and
    the two classes without voxelisation return a sample here."""
    g16, g20, g21 = golden("g16_monash_sequence.npz"), golden("g20_flow_and_cache_loaders.npz"), golden("g21_esim_h5_dataset.npz")
    flow_path = tmp_path / "indoor_flying1.npz"
    np.savez(flow_path, **{k: g16[k] for k in g16 if k.split("/")[0] in ("events", "images", "attrs")}, **{k: g20[k] for k in g20 if k.startswith("flow/")})
    cache_path = tmp_path / "seq.npz"
    big = {k: np.concatenate([g21[k]] * 4) for k in ("frames", "flow", "events")}              # 56 cached items: room for the block's 40-step samples
    np.savez(cache_path, **big, **{"attrs/sensor_resolution": g21["attrs/sensor_resolution"]})
    seen = 0
    for b in _blocks():
        name = b["block"]["class_name"]
        if name == "data.v2v_datasets.WebvidDatasetV2":
            continue
        cfg = {k: v for k, v in b["block"].items() if k not in ("class_name", "data_file")}
        cls = _package_class(name)
        if name.endswith("ESIMH5Dataset"):
            cfg["random_crop_size"] = 16                                                         # the fixture's frames are 20 x 24 (the block crops 128 from 256 x 256 caches)
            ds = cls(str(cache_path), cfg)
            s = ds[0]
            assert len(ds) >= 1 and s["events"].shape == (cfg["sequence_length"], 5, 16, 16)
        else:
            ds = cls(str(flow_path) if name.endswith("FlowDataset") else G16, cfg)
            assert len(ds) >= 1 and ds.num_bins == cfg["num_bins"], (b["yaml"], b["path"])
            if name.endswith("EventDataset"):
                s = ds[0]
                assert len(s["events"]) == s["frame_idx"].numel() and s["events"][0].shape[1] == 5
        seen += 1
    assert seen == 91
