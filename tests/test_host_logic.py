"""Host-side logic of the drop-in surface that needs no GPU: config defaults, sample indexing, parameter draws."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mk_list(tmp_path, rows):
    p = tmp_path / "videos.txt"
    p.write_text("\n".join(" ".join(map(str, r)) for r in rows) + "\n")
    return str(p)


def test_config_defaults_mirror_reference(tmp_path):
    from v2v_amd.datasets import WebvidDatasetV2, data_sources
    ds = WebvidDatasetV2(str(tmp_path), {"video_list_file": _mk_list(tmp_path, [["a.mp4", 450, 0.2, 0.3]]),
                                         "data_source_name": "webvid"})
    # defaults table of data/v2v_datasets.py:26-92
    assert (ds.FPS, ds.L, ds.num_bins, ds.frames_per_bin) == (30, 40, 5, 1)
    assert ds.frames_per_img == 5 and ds.frames_per_seq == 200 and ds.step_size == 200
    assert ds.threshold_range == [0.05, 2] and ds.max_thres_pos_neg_gap == 1.5
    assert ds.base_noise_std_range == [0, 0.2] and ds.hot_pixel_fraction_range == [0, 0.001] and ds.hot_pixel_std_range == [0, 0.2]
    assert ds.keep_top_percentile == 0.54 and ds.max_resize_scale == 1.3 and ds.random_flip is True
    assert ds.data_source_idx == 11 == data_sources.index("webvid")            # utils/data.py:7
    # sample index: range(0, 450-200-1, 200) = [0, 200] capped by max_samples_per_shot = 1
    assert len(ds) == 1 and ds.sample_begin_idx[0] == 0


def test_sample_indexing_and_asserts(tmp_path):
    from v2v_amd.datasets import WebvidDatasetV2
    lst = _mk_list(tmp_path, [["a.mp4", 450, 0.2, 0.3], ["b.mp4", 120, 0.1, 0.1], ["c.mp4", 1000, 0.4, 0.5]])
    ds = WebvidDatasetV2("/data", {"video_list_file": lst, "sequence_length": 4, "max_samples_per_shot": 3, "step_size": 30})
    # frames_per_seq = 20: a -> range(0,429,30)[:3], b -> range(0,99,30)[:3], c -> [:3]
    assert list(ds.sample_begin_idx) == [0, 30, 60] * 3
    assert list(ds.sample_video_name[:3]) == ["a.mp4"] * 3 and ds.sample_pos_thres[3] == 0.1
    half = WebvidDatasetV2("/data", {"video_list_file": lst, "sequence_length": 4, "max_samples_per_shot": 3,
                                     "step_size": 30, "subsample_ratio": 0.5})
    assert len(half) == 4
    ev = WebvidDatasetV2("/data", {"video_list_file": lst, "sequence_length": 4, "output_additional_evs": True})
    assert ev.frames_per_seq == 25
    for bad in ({"video_reader": "pyav"}, {"color_mode": "rgb"}, {"sequence_length": 0}, {"video_degrade": "blur"}):
        with pytest.raises(AssertionError):
            WebvidDatasetV2("/data", dict({"video_list_file": lst}, **bad))


def test_param_sampling_order_matches_golden(golden):
    """The six draws of imgs_to_voxels (v2v_datasets.py:369-381) reproduce the reference's parameters for its seed."""
    from v2v_amd.datasets import sample_sim_params
    g = golden("g6_imgs_to_voxels.npz")
    keys = ["pos_thres", "neg_thres", "base_noise_std", "hot_pixel_fraction", "hot_pixel_std"]
    np.random.seed(int(g["seed"]))
    p = sample_sim_params([0.05, 2], 1.5, [0, 0.2], [0, 0.001], [0, 0.2])
    assert np.array_equal(np.array([p[k] for k in keys]), g["params"])
    np.random.seed(int(g["seed2"]))
    p2 = sample_sim_params([0.05, 2], 1.5, [0, 0.2], [0, 0.001], [0, 0.2], scale_noise_strength=True)
    assert np.array_equal(np.array([p2[k] for k in keys]), g["params2"])
    fixed = sample_sim_params([0.05, 2], 1.5, [0, 0.2], [0, 0.001], [0, 0.2], use_fixed_thresholds=True, pos_thres=0.3, neg_thres=0.4)
    assert fixed["pos_thres"] == 0.3 and fixed["neg_thres"] == 0.4


def test_bgr_to_gray_matches_oracle():
    from oracle import v2v_oracle as O
    from v2v_amd.datasets import bgr_to_gray
    img = np.random.default_rng(0).integers(0, 256, size=(3, 8, 9, 3), dtype=np.uint8)
    assert np.array_equal(bgr_to_gray(img), O.bgr_to_gray(img)) and bgr_to_gray(img).dtype == np.uint8


def test_events_to_voxel_rejects_broken_reference_branch():
    import torch
    from v2v_amd import voxel
    with pytest.raises(NotImplementedError):
        voxel.events_to_voxel([0], [0], [0.0], [1.0], 5, temporal_bilinear=False)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            voxel.make_voxel([np.zeros(0), np.zeros(0, int), np.zeros(0, int), np.zeros(0, int)], 4, 4)


def test_degrade_video_equals_reference_golden(golden, tmp_path):
    """The three NumPy modes of degrade_video (data/v2v_datasets.py:454-483) against golden G17, produced by the reference's own
    method after np.random.seed: same draws, same frames.  'subtitles' needs OpenCV's font rasteriser and says so."""
    from v2v_amd.datasets import WebvidDatasetV2
    g = golden("g17_degrade_video.npz")
    lst = _mk_list(tmp_path, [["a.mp4", 450, 0.2, 0.3]])
    for chan in (1, 3):
        clip = g[f"clip_c{chan}"]
        for mode in ("dirtyshotcut", "hdr", "ldr"):
            ds = WebvidDatasetV2(str(tmp_path), {"video_list_file": lst, "video_degrade": mode, "degrade_ratio": 1.0})
            for seed in (0, 1, 2):
                np.random.seed(seed)
                got = np.stack(ds.degrade_video([f.copy() for f in clip]))
                want = g[f"{mode}_c{chan}_s{seed}"]
                assert got.shape == want.shape and got.dtype == np.uint8 and np.array_equal(got, want), (mode, chan, seed)
    ds = WebvidDatasetV2(str(tmp_path), {"video_list_file": lst, "video_degrade": "subtitles", "degrade_ratio": 1.0})
    try:
        import cv2  # noqa: F401
    except ImportError:
        with pytest.raises(NotImplementedError):
            ds.degrade_video([f.copy() for f in g["clip_c3"]])


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """WORLD_SIZE=1 from a launcher but --gpus 2 asked: an error, never a 1-GPU number labelled n_gpus 2 (runs without a GPU)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert res.returncode != 0 and "WORLD_SIZE=1" in (res.stderr + res.stdout)


def test_pause_chain_draws_equal_the_scalar_loop(tmp_path):
    """_draw_geometry draws the pause chain's uniforms in ONE np.random.rand(n) call.  The reference's loop
    (data/v2v_datasets.py:292-300) calls np.random.rand() once per step (the other test of the if / elif short-circuits before its
    draw): restated here literally with scalar draws -- same frame indices, same stream state afterwards, for running-heavy,
    pause-heavy and extreme probabilities."""
    from v2v_amd.datasets import WebvidDatasetV2
    lst = _mk_list(tmp_path, [["a.mp4", 450, 0.2, 0.3]])
    for p_run, p_paused, extra in ((0.0102, 0.9791, {}), (0.3, 0.5, {"output_additional_evs": True}), (0.0, 0.98, {}), (1.0, 0.0, {}), (0.5, 1.0, {})):
        ds = WebvidDatasetV2(str(tmp_path), dict({"video_list_file": lst, "crop_size": 32, "video_size": (640, 360), "frame_source": lambda *a: [],
                                                  "proba_pause_when_running": p_run, "proba_pause_when_paused": p_paused, "fixed_crop": True,
                                                  "random_flip": False, "min_resize_scale": 1, "max_resize_scale": 1}, **extra))
        for seed in range(5):
            np.random.seed(seed)
            np.random.uniform(1, 1)                                                   # the resize-scale draw in front of the chain (:272)
            img_idxes, idx, is_pause = [], 0, False
            additional = ds.frames_per_img if ds.output_additional_evs else 0
            for _ in range(ds.L * ds.frames_per_img + 1 + additional):
                img_idxes.append(idx)
                if is_pause and np.random.rand() > ds.proba_pause_when_paused:
                    is_pause = False
                elif not is_pause and np.random.rand() < ds.proba_pause_when_running:
                    is_pause = True
                if not is_pause:
                    idx += 1
            want_state = np.random.get_state()
            np.random.seed(seed)
            _, start, end, _, _, _, _, got, _ = ds._draw_geometry(0)
            st = np.random.get_state()
            assert got == img_idxes and end - start == idx + 1
            assert np.array_equal(st[1], want_state[1]) and st[2:] == want_state[2:]


def test_bench_contract_line_for_eight_ranks_is_small_and_complete():
    """The line bench.py prints is built by a pure function: for 8 ranks (no multi-GPU node has run it yet) it carries the contract's keys,
    one time per rank, a whole-job aggregate over all ranks, and stays far below the 4 KB the driver's stdout tail keeps; an over-long
    line sheds its optional keys instead of becoming unparseable (round 4's 21.9 KB line was `parsed: null`)."""
    import json
    import bench
    wl = bench.WORKLOADS[bench.DEFAULT_WORKLOAD]
    kern = sorted(0.57 + 0.001 * i for i in range(20))
    cpu = {"value": 11.2, "unit": "voxel grids/s", "cores": 1, "kind": "port", "sample": "3 of the batch's clips (32x256x256 float32), NumPy port, single thread, 12.3 s"}
    line = bench.contract_line(workload=bench.DEFAULT_WORKLOAD, wl=wl, clips_per_gpu=256, grids_per_step=256, alg_bytes=2483027968, kernel_name="esim_voxel_kernel",
                               world=8, steps=20, warmup=5, elapsed_s=0.0118, per_rank_ms=[0.58 + 0.002 * r for r in range(8)], kern_ms_sorted=kern,
                               backend="nccl", dist_world=8, use_graph=False, traffic=2491741150.0, cpu=None, parity="ok")
    text = bench.line_text(line)
    d = json.loads(text)
    assert len(text.encode()) <= 2048 and "\n" not in text
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline", "parity_check", "dist_backend", "dist_world_size", "ms_per_step_per_rank"):
        assert key in d, key
    assert d["n_gpus"] == 8 and len(d["ms_per_step_per_rank"]) == 8 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert abs(d["value"] - 8 * 256 * 20 / 0.0118) < 1e-6 and abs(d["ms_per_step"] - 0.59) < 1e-9          # whole-job aggregate over all ranks
    assert d["roofline"]["bound"] == "hbm" and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / 8000.0) < 1e-12
    assert d["config"]["workload"] == bench.DEFAULT_WORKLOAD and "8 GPU(s)" in d["config"]["sharding"]
    # N = 1 with the CPU baseline, and a pathological line that would not fit
    one = bench.contract_line(workload=bench.DEFAULT_WORKLOAD, wl=wl, clips_per_gpu=256, grids_per_step=256, alg_bytes=2483027968, kernel_name="esim_voxel_kernel",
                              world=1, steps=20, warmup=5, elapsed_s=0.0117, per_rank_ms=[0.585], kern_ms_sorted=kern, backend=None, dist_world=1, use_graph=False,
                              traffic=2491741150.0, cpu=cpu, parity="ok")
    assert len(bench.line_text(one).encode()) <= 2048
    fat = dict(one, parity_check="x" * 10000)
    slim = json.loads(bench.line_text(fat))
    assert len(bench.line_text(fat).encode()) <= bench.MAX_LINE_BYTES and "parity_check" not in slim and slim["value"] == one["value"] and "roofline" in slim


_START_METHOD_CHILD = r"""
import json, multiprocessing, sys, warnings
sys.path.insert(0, sys.argv[1])
from torch.utils.data import DataLoader
from v2v_amd.datasets import WebvidDatasetV2, synthetic_frame_source
tmp = sys.argv[2]
open(tmp + "/videos.txt", "w").write("clip_a.mp4 450 0.2 0.3\n")
cfg = {"video_list_file": tmp + "/videos.txt", "sequence_length": 4, "crop_size": 32, "data_source_name": "webvid", "frame_source": synthetic_frame_source,
       "video_size": (1280, 720), "video_reader": "opencv"}
out = {"unset_before": multiprocessing.get_start_method(allow_none=True)}
plain = WebvidDatasetV2(tmp, cfg)
out["plain_leaves_it_unset"] = multiprocessing.get_start_method(allow_none=True) is None and plain.worker_start_method is None and plain.multiprocessing_context is None
ds = WebvidDatasetV2(tmp, dict(cfg, worker_start_method="spawn"))
out["after"] = multiprocessing.get_start_method(allow_none=True)
out["context"] = type(ds.multiprocessing_context).__name__
it = iter(DataLoader(ds, batch_size=1, num_workers=1))                   # what train.py builds (train.py:52-65): the default context is now spawn
out["popen"] = type(it._workers[0]._popen).__module__
it._shutdown_workers()
with warnings.catch_warnings(record=True) as w:                           # a second dataset asking for something else: warned, nothing forced
    warnings.simplefilter("always")
    WebvidDatasetV2(tmp, dict(cfg, worker_start_method="forkserver"))
out["warned"] = [str(x.message)[:60] for x in w if issubclass(x.category, RuntimeWarning)]
out["still"] = multiprocessing.get_start_method(allow_none=True)
try:
    WebvidDatasetV2(tmp, dict(cfg, worker_start_method="threads"))
    out["bad_refused"] = False
except AssertionError:
    out["bad_refused"] = True
print(json.dumps(out))
"""


def test_worker_start_method_yaml_key_sets_how_train_pys_dataloader_starts_workers(tmp_path):
    """`worker_start_method: spawn` in the dataset's YAML block: train.py's own DataLoader (train.py:52-65 passes no multiprocessing_context)
    then spawns its workers -- each owns a HIP context and simulates its samples itself -- with train.py untouched.  The key fixes the
    process-wide default ONLY while the program has not chosen one (no force=True: a later, different request is warned about and changes
    nothing); `dataset.multiprocessing_context` is the explicit handle.  Run in a fresh interpreter, as train.py is one."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", _START_METHOD_CHILD, root, str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["unset_before"] is None and out["plain_leaves_it_unset"]
    assert out["after"] == "spawn" and out["context"] == "SpawnContext" and "popen_spawn" in out["popen"]
    assert len(out["warned"]) == 1 and out["still"] == "spawn" and out["bad_refused"]


def test_worker_start_method_never_overrides_a_choice_the_program_already_made(tmp_path):
    """In THIS process (pytest has long fixed its start method by the time this runs, or fixes it here): the key warns and changes nothing."""
    import multiprocessing
    from v2v_amd.datasets import WebvidDatasetV2, synthetic_frame_source
    (tmp_path / "videos.txt").write_text("clip_a.mp4 450 0.2 0.3\n")
    cfg = {"video_list_file": str(tmp_path / "videos.txt"), "sequence_length": 4, "crop_size": 32, "data_source_name": "webvid",
           "frame_source": synthetic_frame_source, "video_size": (1280, 720), "video_reader": "opencv"}
    before = multiprocessing.get_start_method()                          # fixes the default (fork on Linux) if nothing has yet
    other = "spawn" if before != "spawn" else "forkserver"
    with pytest.warns(RuntimeWarning, match="already fixed"):
        ds = WebvidDatasetV2(str(tmp_path), dict(cfg, worker_start_method=other))
    assert multiprocessing.get_start_method() == before
    assert ds.multiprocessing_context.get_start_method() == other        # the explicit handle still says what the YAML asked for


def _npy_videos(tmp_path, n=3, t=60, h=96, w=160):
    g = np.random.default_rng(5)
    names = []
    for i in range(n):
        base = g.integers(0, 256, size=(1, h, w, 3)).astype(np.int16)
        walk = np.clip(base + np.cumsum(g.integers(-5, 6, size=(t, h, w, 3)), axis=0), 0, 255).astype(np.uint8)
        np.save(tmp_path / f"clip_{i}.npy", walk)
        names.append(f"clip_{i}.npy")
    (tmp_path / "videos.txt").write_text("".join(f"{nm} {t} 0.2 0.3\n" for nm in names))
    return names


def _oracle_frame_source(ds, sample_idx, start, end, crop_before, min_i, min_j, flip, need_h, need_w):
    """The reference's decode loop (data/v2v_datasets.py:188-213) written out on the OpenCV restatement: what the cv2 branch must equal."""
    from oracle import frontend_oracle as FO
    video = np.load(os.path.join(ds.dataset_path, ds.sample_video_name[sample_idx]))
    out = []
    for f in video[start:end]:
        if ds.color_mode == "gray":
            f = FO.cv_bgr2gray_u8(f, "cv4")
        f = f[min_i:min_i + crop_before, min_j:min_j + crop_before, ...]
        f = FO.cv_resize_linear_u8(f, need_w, need_h)
        if flip:
            f = np.ascontiguousarray(f[:, ::-1])
        out.append(f[..., None] if ds.color_mode == "gray" else f)
    return out


@pytest.mark.parametrize("extra", [{}, {"color_mode": "gray_in_bgr_out"}, {"shake_frames": 4, "shake_std": 1.5}, {"max_resize_scale": 1.3, "min_resize_scale": 0.4}])
def test_opencv_decode_branch_runs_against_a_stand_in_cv2(tmp_path, monkeypatch, extra):
    """`video_reader: opencv` without a frame_source -- the branch a real V2V installation takes (cv2.VideoCapture: seek, read loop, cvtColor
    BEFORE the crop, resize to need_w x need_h, flip after the resize, the trailing axis, release; _probe_size for the geometry draws).  No
    OpenCV in this image: tests/fake_cv2.py stands in (OpenCV's algorithms as restated in oracle/), and the samples must equal those of a
    frame_source that writes the reference's loop out -- same np.random consumption, same clip, same frames."""
    import sys
    import fake_cv2
    from v2v_amd.datasets import WebvidDatasetV2
    monkeypatch.setitem(sys.modules, "cv2", fake_cv2)
    fake_cv2.calls.clear()
    _npy_videos(tmp_path)
    cfg = {"video_list_file": str(tmp_path / "videos.txt"), "sequence_length": 3, "crop_size": 32, "data_source_name": "webvid", "video_reader": "opencv",
           "defer_sim": True, "proba_pause_when_running": 0.2, "proba_pause_when_paused": 0.6}
    cfg.update(extra)
    via_cv2 = WebvidDatasetV2(str(tmp_path), cfg)
    via_src = WebvidDatasetV2(str(tmp_path), dict(cfg, frame_source=_oracle_frame_source, video_size=(160, 96)))
    assert via_cv2._probe_size(str(tmp_path / "clip_0.npy")) == (160, 96)
    for idx in (0, 2, 1):
        np.random.seed(40 + idx)
        a = via_cv2[idx]
        state = np.random.get_state()[1].copy()
        np.random.seed(40 + idx)
        b = via_src[idx]
        assert np.array_equal(np.random.get_state()[1], state)
        assert torch.equal(a["sim_frames"], b["sim_frames"]) and torch.equal(a["frame"], b["frame"]) and a["v2e_params"] == b["v2e_params"]
        assert a["sim_frames"].shape == (16, 32, 32) and a["sim_frames"].dtype == torch.uint8 and float(a["sim_frames"].float().std()) > 1
    names = [c[0] for c in fake_cv2.calls]
    assert names.count("VideoCapture") == names.count("release") and "set_pos" in names      # every capture is released; the clip is sought, not read from frame 0


def test_bench_core_count_is_what_the_process_may_use():
    """bench.py's `cores`: the affinity mask cut by the cgroup CPU quota -- the pool's 1-GPU boxes report os.cpu_count() = 256 and grant 16."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.apply_cpu_quota(256, "m", "1600000 100000\n") == (16, "m, cgroup quota 16.0 CPUs")
    assert bench.apply_cpu_quota(8, "m", "max 100000\n") == (8, "m")
    assert bench.apply_cpu_quota(8, "m", "1600000 100000") == (8, "m, cgroup quota 16.0 CPUs")            # a quota above the mask changes nothing
    assert bench.apply_cpu_quota(8, "m", "50000 100000")[0] == 1 and bench.apply_cpu_quota(8, "m", "")[0] == 8
    n, detail = bench.host_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0)) and "affinity mask" in detail
