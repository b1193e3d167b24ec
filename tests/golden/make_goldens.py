#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ by IMPORTING THE REFERENCE (build container only).

    python tests/golden/make_goldens.py            # needs /root/reference, writes tests/golden/*.npz

The reference's Python never travels to the GPU box; these fixtures (inputs + the reference's own
outputs) do.  Everything here is data produced by calling reference functions -- no reference source
text is stored.  SURVEY.md §8(c) lists the vectors (G1..G11).

Reference entry points exercised:
  data/v2v_core_esim.py:26-69       EventEmulator.video_to_voxel          (G1-G4, G11)
  numpy.floor_divide                 as used at v2v_core_esim.py:51,54     (G5)
  data/v2v_datasets.py:363-410       WebvidDatasetV2.imgs_to_voxels        (G6)
  utils/event_utils.py:692-728       events_to_voxel                       (G7)
  data/testh5.py:60-90               TestH5Dataset.make_voxel              (G8)
  data/v2v_core_v2e.py:556-581       video_to_voxel (v2e)                  (G9)
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

for m in ("cv2", "h5py", "ffmpeg", "event_voxel_builder", "torchvision", "torchvision.transforms"):   # absent here; only IO / augmentation code touches them
    sys.modules.setdefault(m, types.ModuleType(m))
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["event_voxel_builder"].EventVoxelBuilder = object
sys.path.insert(0, REF)

from data.v2v_core_esim import EventEmulator  # noqa: E402  (pure numpy)
from data import v2v_core_v2e  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_eu = _load("ref_event_utils", os.path.join(REF, "utils/event_utils.py"))
ref_ds = _load("ref_v2v_datasets", os.path.join(REF, "data/v2v_datasets.py"))
ref_th5 = _load("ref_testh5", os.path.join(REF, "data/testh5.py"))

from oracle import v2v_oracle as O  # noqa: E402  (only for the synthetic input generator + Philox fields)


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


def g1_luts():
    u8 = np.arange(256, dtype=np.uint8).reshape(1, 16, 16)
    f32 = np.arange(256, dtype=np.float32).reshape(1, 16, 16)
    # the reference's own two lines (v2v_core_esim.py:33-34) applied to every intensity, both dtypes
    from data.v2v_core_esim import reverse_gamma_correction
    lut64 = np.log(0.001 + reverse_gamma_correction(u8) / 255.0).reshape(256)
    lut32 = np.log(0.001 + reverse_gamma_correction(f32) / 255.0).reshape(256)
    assert lut64.dtype == np.float64 and lut32.dtype == np.float32
    v2e32 = v2v_core_v2e.lin_log(np.arange(256, dtype=np.uint8))
    assert v2e32.dtype == np.float32
    save("g1_luts.npz", lut64=lut64, lut32=lut32, v2e32=v2e32)


def g2_g3_esim_clean():
    video = O.synth_clip_s1(8, 128, 128, seed=1234, dtype=np.uint8)        # S1 / BASELINE config 1
    out = {"video": video, "seeds": np.array([5, 6])}
    for tag, (cp, cn) in {"sym": (0.2, 0.2), "asym": (0.31, 0.47)}.items():
        for s in (5, 6):
            for dt_tag, dt in (("u8", np.uint8), ("f32", np.float32)):
                np.random.seed(s)
                v = EventEmulator(cp, cn, 0.0, 0.0, 0.0, False).video_to_voxel(video.astype(dt))
                assert np.array_equal(v, np.round(v)) and np.abs(v).max() < 127
                out[f"{tag}_s{s}_{dt_tag}"] = v.astype(np.int8)
    save("g2_esim_clean.npz", **out)


def g4_esim_noisy():
    video = O.synth_clip_s1(8, 32, 32, seed=77, dtype=np.uint8)
    out = {"video": video, "params": np.array([0.25, 0.4, 0.05, 0.02, 0.8]), "seed": np.array(42)}
    for ext in (False, True):
        for dt_tag, dt in (("u8", np.uint8), ("f32", np.float32)):
            np.random.seed(42)
            v = EventEmulator(0.25, 0.4, 0.05, 0.02, 0.8, ext).video_to_voxel(video.astype(dt))
            out[f"ext{int(ext)}_{dt_tag}"] = v
    save("g4_esim_noisy.npz", **out)


def g5_floor_divide():
    g = np.random.default_rng(2024)
    b = g.uniform(0.05, 3.0, size=1500)
    q = g.integers(1, 60, size=1500).astype(np.float64)
    a = q * b                                               # near-ties: a is within an ulp or two of q*b
    a = np.nextafter(a, np.where(g.random(1500) < 0.5, -np.inf, np.inf))
    a[::3] = (q * b)[::3]
    a2 = g.uniform(0.0, 40.0, size=500)
    b2 = g.uniform(0.05, 3.0, size=500)
    a = np.concatenate([a, a2, [24.550921417593624]])
    b = np.concatenate([b, b2, [1.2275460708796813]])
    save("g5_floor_divide.npz", a=a, b=b, q=np.floor_divide(a, b))


def g6_imgs_to_voxels():
    video = O.synth_clip_s1(21, 32, 32, seed=99, dtype=np.uint8)
    inst = object.__new__(ref_ds.WebvidDatasetV2)
    inst.load_configs({"sequence_length": 4, "num_bins": 5, "frames_per_bin": 1})
    np.random.seed(2025)
    params, vox = inst.imgs_to_voxels(video, 5, 1, 24)
    assert vox.shape == (4, 5, 32, 32)
    inst2 = object.__new__(ref_ds.WebvidDatasetV2)
    inst2.load_configs({"sequence_length": 2, "num_bins": 5, "frames_per_bin": 2, "scale_noise_strength": True})
    np.random.seed(2026)
    params2, vox2 = inst2.imgs_to_voxels(video, 5, 2, 24)
    assert vox2.shape == (2, 5, 32, 32)
    keys = ["pos_thres", "neg_thres", "base_noise_std", "hot_pixel_fraction", "hot_pixel_std"]
    save("g6_imgs_to_voxels.npz", video=video, seed=np.array(2025), voxels=vox,
         params=np.array([params[k] for k in keys]), seed2=np.array(2026), voxels2=vox2,
         params2=np.array([params2[k] for k in keys]))


def g7_bilinear():
    out = {}
    g = np.random.default_rng(7)
    for k in (2, 7, 31, 39):
        # weights: one unit event per pair at a single pixel
        xs = np.zeros(k, dtype=np.int64)
        ys = np.zeros(k, dtype=np.int64)
        ts = np.arange(k, dtype=np.float64)[:, None]
        wmat = np.zeros((5, k))
        for kk in range(k):
            ps = np.zeros((k, 1))
            ps[kk] = 1.0
            wmat[:, kk] = ref_eu.events_to_voxel(xs, ys, ts, ps, 5, sensor_size=(1, 1))[:, 0, 0]
        out[f"w_K{k}"] = wmat
        counts = g.integers(-6, 7, size=(k, 16, 16)).astype(np.float64)
        yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
        xs = np.tile(xx.reshape(-1), k)
        ys = np.tile(yy.reshape(-1), k)
        ts = np.repeat(np.arange(k, dtype=np.float64), 256)[:, None]
        ps = counts.reshape(-1)[:, None]
        out[f"counts_K{k}"] = counts.astype(np.int8)
        out[f"voxel_K{k}"] = ref_eu.events_to_voxel(xs, ys, ts, ps, 5, sensor_size=(16, 16))
    save("g7_bilinear.npz", **out)


def g8_make_voxel():
    g = np.random.default_rng(8)
    n = 500
    ts = np.sort(g.uniform(10.0, 10.05, size=n))
    xs = g.integers(0, 24, size=n)
    ys = g.integers(0, 16, size=n)
    ps = g.integers(0, 2, size=n)
    stub = types.SimpleNamespace(num_bins=5, H=16, W=24, interpolate_bins=False)
    disc = ref_th5.TestH5Dataset.make_voxel(stub, [ts, xs, ys, ps])
    stub.interpolate_bins = True
    interp = ref_th5.TestH5Dataset.make_voxel(stub, [ts, xs, ys, ps])
    empty = ref_th5.TestH5Dataset.make_voxel(stub, [ts[:0], xs[:0], ys[:0], ps[:0]])
    # generic events_to_voxel on a real event list (float ps in {-1,+1})
    pf = (ps * 2 - 1).astype(np.float64)
    e2v = ref_eu.events_to_voxel(xs, ys, ts[:, None], pf[:, None], 5, sensor_size=(16, 24))
    save("g8_make_voxel.npz", ts=ts, xs=xs, ys=ys, ps=ps, discrete=disc, interpolated=interp, empty=empty,
         events_to_voxel=e2v)


def g12_events_to_voxel_torch():
    """events_to_voxel_torch (utils/event_utils.py:466-507), both branches, float32 CPU tensors."""
    import torch
    g = np.random.default_rng(12)
    n = 600
    ts = np.sort(g.uniform(0.0, 0.03, size=n)).astype(np.float32)
    xs = g.integers(0, 24, size=n)
    ys = g.integers(0, 16, size=n)
    ps = (g.integers(0, 2, size=n) * 2 - 1).astype(np.float32)
    tt, tp = torch.from_numpy(ts), torch.from_numpy(ps)
    tx, ty = torch.from_numpy(xs), torch.from_numpy(ys)
    bil = ref_eu.events_to_voxel_torch(tx, ty, tt, tp, 5, sensor_size=(16, 24), temporal_bilinear=True)
    disc = ref_eu.events_to_voxel_torch(tx, ty, tt, tp, 5, sensor_size=(16, 24), temporal_bilinear=False)
    assert bil.dtype == torch.float32 and disc.dtype == torch.float32
    save("g12_events_to_voxel_torch.npz", ts=ts, xs=xs, ys=ys, ps=ps, bilinear=bil.numpy(), discrete=disc.numpy())


def g22_voxel_grid_lists():
    """The list-of-grids helpers around events_to_voxel_torch (utils/event_utils.py:378-464): voxel_grids_fixed_n_torch,
    voxel_grids_fixed_t_torch, events_to_voxel_timesync_torch, run by the REFERENCE on golden G12's 600 events (inputs not stored again)."""
    import torch
    z = np.load(os.path.join(HERE, "g12_events_to_voxel_torch.npz"))
    tt, tp, tx, ty = (torch.from_numpy(z[k]) for k in ("ts", "ps", "xs", "ys"))
    out = {}
    for tag, tb in (("bil", True), ("disc", False)):
        out[f"fixed_n_{tag}"] = torch.stack(ref_eu.voxel_grids_fixed_n_torch(tx, ty, tt, tp, 5, 100, sensor_size=(16, 24), temporal_bilinear=tb)).numpy()
        out[f"fixed_t_{tag}"] = torch.stack(ref_eu.voxel_grids_fixed_t_torch(tx, ty, tt, tp, 4, 0.007, sensor_size=(16, 24), temporal_bilinear=tb)).numpy()
        out[f"timesync_{tag}"] = ref_eu.events_to_voxel_timesync_torch(tx, ty, tt, tp, 3, 0.004, 0.0215, sensor_size=(16, 24), temporal_bilinear=tb).numpy()
    save("g22_voxel_grid_lists.npz", **out)


def g23_events_to_image():
    """events_to_image (utils/event_utils.py:155-174, NumPy, float64) and events_to_image_torch (:330-376, float32: plain accumulation,
    and bilinear splatting of fractional coordinates through interpolate_to_image :176-184, with and without the one-pixel padding) run by
    the REFERENCE.  Integer events = golden G12's; the fractional coordinates and real-valued weights are stored here."""
    import torch
    z = np.load(os.path.join(HERE, "g12_events_to_voxel_torch.npz"))
    g = np.random.default_rng(23)
    n = len(z["xs"])
    wts = g.normal(0, 1.5, size=n)
    fx = g.uniform(0, 24.6, size=n).astype(np.float32)                    # some beyond the 24-pixel width: clipped by the mask
    fy = g.uniform(0, 16.4, size=n).astype(np.float32)
    out = {"weights": wts, "fx": fx, "fy": fy}
    out["np_pol"] = ref_eu.events_to_image(z["xs"], z["ys"], z["ps"].astype(np.float64), sensor_size=(16, 24))
    out["np_weights"] = ref_eu.events_to_image(z["xs"], z["ys"], wts, sensor_size=(16, 24))
    tx, ty, tp = torch.from_numpy(z["xs"]), torch.from_numpy(z["ys"]), torch.from_numpy(z["ps"])
    out["torch_plain"] = ref_eu.events_to_image_torch(tx, ty, tp, sensor_size=(16, 24)).numpy()
    tw = torch.from_numpy(wts.astype(np.float32))
    out["torch_weights"] = ref_eu.events_to_image_torch(tx, ty, tw, sensor_size=(16, 24), padding=False).numpy()
    for pad in (True, False):
        out[f"torch_bilinear_pad{int(pad)}"] = ref_eu.events_to_image_torch(torch.from_numpy(fx), torch.from_numpy(fy), tw, sensor_size=(16, 24),
                                                                             interpolation="bilinear", padding=pad).numpy()
    save("g23_events_to_image.npz", **out)


def g13_normalize_batch_voxel():
    """normalize_batch_voxel (model/train_utils.py:147-166).  The module needs torchvision/torchmetrics/skimage, which
    are absent, so ONLY that function is compiled from the reference file (AST extraction at run time; no source kept)."""
    import ast
    import torch
    path = os.path.join(REF, "model/train_utils.py")
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "normalize_batch_voxel"]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fn, type_ignores=[]), path, "exec"), ns)
    g = np.random.default_rng(13)
    counts = np.round(g.normal(0, 2.5, size=(3, 4, 5, 20, 24))).astype(np.float32)          # integer-valued like SUM voxels
    counts[1] *= 0.2                                                                          # sample whose 1 %/99 % values are < 1
    soft = g.normal(0, 3.0, size=(2, 2, 5, 17, 19)).astype(np.float32)                       # bilinear / noisy voxels
    out = {"counts": counts, "counts_norm": ns["normalize_batch_voxel"](torch.from_numpy(counts)).numpy(),
           "soft": soft, "soft_norm": ns["normalize_batch_voxel"](torch.from_numpy(soft)).numpy()}
    save("g13_normalize_batch_voxel.npz", **out)


def g11_philox_fed():
    """The reference itself, run on the device-native Philox fields (monkey-patched np.random)."""
    from oracle import clib
    clib.build()
    video = O.synth_clip_s1(8, 32, 32, seed=314, dtype=np.uint8)
    seed, clip_id = 0x1234ABCD5678, 9
    out = {"video": video, "seed": np.array(seed, dtype=np.uint64), "clip_id": np.array(clip_id),
           "params": np.array([0.2, 0.3, 0.04, 0.05, 0.6])}
    real_rand, real_randn = np.random.rand, np.random.randn
    for ext in (False, True):
        for dt_tag, dt in (("u8", np.uint8), ("f32", np.float32)):
            rng = O.PhiloxFieldRNG(seed, clip_id)
            np.random.rand, np.random.randn = rng.rand, rng.randn
            try:
                v = EventEmulator(0.2, 0.3, 0.04, 0.05, 0.6, ext).video_to_voxel(video.astype(dt))
            finally:
                np.random.rand, np.random.randn = real_rand, real_randn
            out[f"ext{int(ext)}_{dt_tag}"] = v
    out["gauss_field3"] = clib.philox_gauss_field(seed, clip_id, 3, 1024, rounds=clib.noise_rounds())
    out["gauss_field3b"] = clib.philox_gauss_field(seed, clip_id, 3, 1024, comp=1, rounds=clib.noise_rounds())
    out["uniform_field0"] = clib.philox_uniform_field(seed, clip_id, 0, 1024)
    save("g11_philox_fed.npz", **out)


def g9_v2e():
    """The reference's v2e video_to_voxel (data/v2v_core_v2e.py:556-581) with np.random.{normal,randn,poisson}
    wrapped so the drawn fields are recorded in draw order (the values still come from the real seeded stream)."""
    video = O.synth_clip_s1(8, 24, 24, seed=909, dtype=np.uint8)
    cases = {
        "pn_clean_u8": (np.uint8, "pn_related", 0, 0, 0, 0),
        "pn_noisy_u8": (np.uint8, "pn_related", 30, 0.1, 0, 5.0),
        "pn_noisy_f32": (np.float32, "pn_related", 30, 0.1, 0, 5.0),
        "si_leak_u8": (np.uint8, "spatial_independent", 0, 0.1, 0, 0),
        "si_cut_f32": (np.float32, "spatial_independent", 30, 0, 0, 0),
        "sti_noisy_u8": (np.uint8, "spatial_temporal_independent", 30, 0.1, 0, 5.0),
        "sti_shot_f32": (np.float32, "spatial_temporal_independent", 0, 0, 0, 5.0),
    }
    out = {"video": video, "case_names": np.array(list(cases))}
    real = (np.random.normal, np.random.randn, np.random.poisson)
    for name, (dt, model, cutoff, leak, refr, shot) in cases.items():
        rec = {"normal": [], "randn": [], "poisson": []}

        def normal(loc=0.0, scale=1.0, size=None):
            v = real[0](loc=loc, scale=scale, size=size); rec["normal"].append(v); return v

        def randn(*shape):
            v = real[1](*shape); rec["randn"].append(v); return v

        def poisson(lam):
            v = real[2](lam); rec["poisson"].append(v); return v
        np.random.normal, np.random.randn, np.random.poisson = normal, randn, poisson
        try:
            vox = v2v_core_v2e.video_to_voxel(video.astype(dt), 24, model, 0.5, 0.1, 0.0, 0.1, cutoff, leak, refr, shot,
                                              0.1, 0.1, seed=11)
        finally:
            np.random.normal, np.random.randn, np.random.poisson = real
        out[f"{name}__args"] = np.array([24, list(O.V2E_MODELS).index(model), 0.5, 0.1, 0.0, 0.1, cutoff, leak, refr, shot, 0.1, 0.1])
        out[f"{name}__dtype"] = np.array(np.dtype(dt).name)
        out[f"{name}__voxels"] = vox.astype(np.int16)
        assert np.array_equal(vox, vox.astype(np.int16))
        # Replay-ready fields, derived from the recorded draws with the reference's own expressions
        # (_init :328-349, change_pos_neg_thres :392-399) -- derived HERE because np.exp(float32) is a SIMD
        # kernel whose last bit may differ on another host.
        import math
        temporal = model == "spatial_temporal_independent"
        normals = rec["normal"][2:] if temporal else rec["normal"]          # temporal: frame 0 pre-draw is discarded
        pts, nts = [], []
        for a, b in zip(normals[0::2], normals[1::2]):
            if model == "pn_related":
                pt, nt = a + (b / 2), a - (b / 2)
            else:
                pt, nt = a, b
            pts.append(np.clip(pt, a_min=0.01, a_max=None))
            nts.append(np.clip(nt, a_min=0.01, a_max=None))
        k = video.shape[0] - 1
        out[f"{name}__pos_thres"] = np.stack(pts[1:] if temporal else pts)   # temporal: per frame 1..N-1
        out[f"{name}__neg_thres"] = np.stack(nts[1:] if temporal else nts)
        out[f"{name}__noise_rate"] = np.exp(math.log(10) * 0.1 * rec["randn"][0].astype(np.float32))
        assert out[f"{name}__noise_rate"].dtype == np.float32
        if leak > 0:
            out[f"{name}__leak_randn"] = np.stack(rec["randn"][1:])
            assert len(rec["randn"]) == 1 + k
        if shot > 0:
            out[f"{name}__shot_pos"] = np.stack(rec["poisson"][0::2])
            out[f"{name}__shot_neg"] = np.stack(rec["poisson"][1::2])
            assert len(rec["poisson"]) == 2 * k
    save("g9_v2e.npz", **out)


def g14_v2e_native():
    """The reference's v2e video_to_voxel (data/v2v_core_v2e.py:556-581) run on the DEVICE-NATIVE random fields: np.random.normal /
    randn / poisson are replaced by the Philox fields of v2v_amd/csrc/v2v_v2e.hpp in the reference's own draw order (:334-349
    _init, :417-421 per-frame thresholds, :201 leak jitter, :102-103 shot noise), np.random.poisson by the native float32
    inversion sampler fed with the device's 16-bit uniforms, and np.exp on the float32 leak-rate array (:349) by the native
    expf (NumPy's float32 exp is a SIMD kernel no other implementation reproduces bit for bit).  Pins native mode -- the
    fixed-point frame mean, the float32 shot-noise means and the sampler -- to the reference itself."""
    from oracle import clib
    clib.build()
    F_THRES, F_NRATE, F0, FS, STREAM = 0, 2, 16, 8, 1
    video = O.synth_clip_s1(10, 32, 32, seed=1414, dtype=np.uint8)
    n, h, w = video.shape
    seed, clip_id = 0x0DDC0FFEE123, 5
    r7 = clib.noise_rounds()
    cases = {
        "pn_noisy_u8": (np.uint8, "pn_related", 30, 0.1, 5.0),
        "pn_noisy_f32": (np.float32, "pn_related", 30, 0.1, 5.0),
        "pn_shot_only_f32": (np.float32, "pn_related", 0, 0, 20.0),
        "si_leak_u8": (np.uint8, "spatial_independent", 0, 0.1, 0),
        "sti_noisy_u8": (np.uint8, "spatial_temporal_independent", 30, 0.1, 5.0),
        "sti_shot_f32": (np.float32, "spatial_temporal_independent", 0, 0, 5.0),
    }
    out = {"video": video, "case_names": np.array(list(cases)), "seed": np.array(seed, dtype=np.uint64), "clip_id": np.array(clip_id)}
    real = (np.random.normal, np.random.randn, np.random.poisson, np.exp)
    for name, (dt, model, cutoff, leak, shot) in cases.items():
        temporal = model == "spatial_temporal_independent"
        st = {"normal": 0, "randn": 0, "poisson": 0}

        def gauss(field, comp, rounds=10):
            return clib.philox_gauss_field(seed, clip_id, field, h * w, STREAM, comp, rounds).astype(np.float64).reshape(h, w)

        def normal(loc=0.0, scale=1.0, size=None):
            c = st["normal"]; st["normal"] += 1
            if temporal:       # draws 0,1: frame-0 pre-draw (:417-421, overwritten by _init); 2,3: _init; then 2 per frame i >= 1
                frame = 0 if c < 4 else (c - 4) // 2 + 1
                field = F_THRES if c in (2, 3) else F0 + FS * frame
            else:
                field = F_THRES
            return loc + scale * gauss(field, c & 1)

        def randn(*shape):
            c = st["randn"]; st["randn"] += 1
            if c == 0:
                return gauss(F_NRATE, 0)
            k = c - 1                                             # leak jitter of frame pair k: member k&1 of couple k>>1
            return gauss(F0 + FS * (k >> 1) + 2, k & 1, r7)

        def poisson(lam):
            c = st["poisson"]; st["poisson"] += 1
            i = c // 2 + 1                                        # frame index; even draw = ON, odd = OFF
            u = clib.philox_uniform16_field(seed, clip_id, F0 + FS * i + 3, h * w, STREAM, low=c & 1).reshape(h, w)
            return clib.poisson_inv_f32(np.asarray(lam, dtype=np.float64).astype(np.float32), u)

        def exp(x, *a, **k):
            x = np.asarray(x)
            return clib.expf_det(x).reshape(x.shape) if x.dtype == np.float32 else real[3](x, *a, **k)
        np.random.normal, np.random.randn, np.random.poisson, np.exp = normal, randn, poisson, exp
        try:
            vox = v2v_core_v2e.video_to_voxel(video.astype(dt), 24, model, 0.5, 0.1, 0.0, 0.1, cutoff, leak, 0, shot, 0.1, 0.1, seed=11)
        finally:
            np.random.normal, np.random.randn, np.random.poisson, np.exp = real
        assert st["normal"] == (4 + 2 * (n - 1) if temporal else 2) and st["randn"] == 1 + (n - 1 if leak > 0 else 0)
        assert st["poisson"] == (2 * (n - 1) if shot > 0 else 0)
        out[f"{name}__args"] = np.array([24, list(O.V2E_MODELS).index(model), 0.5, 0.1, 0.0, 0.1, cutoff, leak, 0, shot, 0.1, 0.1])
        out[f"{name}__dtype"] = np.array(np.dtype(dt).name)
        assert np.array_equal(vox, vox.astype(np.int16))
        out[f"{name}__voxels"] = vox.astype(np.int16)
        print(f"  g14 {name}: |events| = {int(np.abs(vox).sum())}")
    save("g14_v2e_native.npz", **out)


def g18_unet_modules():
    """The reference's own consumer modules run in float32 on seeded weights (tests/seeded_weights.py: the recipe, not the
    10.7 M values, is what the fixture shares with the GPU tests): model/submodules.py ConvLSTM (:179-235, two steps incl.
    prev_state=None), ResidualBlock (:143-177), ConvLayer 5x5 stride 2 (:6-33), UpsampleConvLayer (:68-96), and
    model/unet.py UNetRecurrent (:252-310) with the kwargs of config/train_v2v_e2vid_10k.yaml:21-30 over 3 time steps at 64x64.
    Stored: inputs, outputs, the seeds / gains, and the reference modules' state_dict keys + shapes (so the package module's
    key compatibility is checked without a GPU)."""
    import contextlib
    import io
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from seeded_weights import load_seeded
    import model.submodules as sm          # the reference's modules (REF is on sys.path)
    import model.unet as un
    torch.manual_seed(0)
    torch.set_num_threads(4)
    from seeded_weights import seeded_input as rnd          # inputs are a recipe too (NEP 19 streams): only OUTPUTS are stored
    g = np.random.Generator(np.random.PCG64(1818))
    out = {}
    with torch.no_grad():
        # ConvLSTM, C = 64, 16x16, two steps: prev_state None, then the returned state
        m = sm.ConvLSTM(64, 64, 3).eval()
        load_seeded(m, 1801, gain=1.0)
        x0, x1 = rnd(18010, 1, 64, 16, 16), rnd(18011, 1, 64, 16, 16)
        h1, c1 = m(torch.from_numpy(x0), None)
        h2, c2 = m(torch.from_numpy(x1), (h1, c1))
        out.update(convlstm__x_seeds=np.array([18010, 18011]), convlstm__x_shape=np.array(x0.shape), convlstm__h1=h1.numpy(), convlstm__c1=c1.numpy(), convlstm__h2=h2.numpy(),
                   convlstm__c2=c2.numpy(), convlstm__seed=np.array(1801))
        # ResidualBlock 256 -> 256 at 2 x 8x8 (the bottleneck of the 64x64 network; the kernels want >= 128 pixels per launch)
        m = sm.ResidualBlock(256, 256).eval()
        load_seeded(m, 1802, gain=1.0)
        x = rnd(18020, 2, 256, 8, 8)
        out.update(resblock__x_seed=np.array(18020), resblock__x_shape=np.array(x.shape), resblock__y=m(torch.from_numpy(x.copy())).numpy(),
                   resblock__seed=np.array(1802))
        # ConvLayer 5x5 stride 2, 64 -> 128 (the second encoder's convolution)
        m = sm.ConvLayer(64, 128, 5, stride=2, padding=2).eval()
        load_seeded(m, 1803, gain=1.0)
        x = rnd(18030, 2, 64, 16, 16)
        out.update(convlayer__x_seed=np.array(18030), convlayer__x_shape=np.array(x.shape), convlayer__y=m(torch.from_numpy(x)).numpy(),
                   convlayer__seed=np.array(1803))
        # UpsampleConvLayer 128 -> 64 (the second decoder)
        m = sm.UpsampleConvLayer(128, 64, 5, padding=2).eval()
        load_seeded(m, 1804, gain=1.0)
        x = rnd(18040, 2, 128, 8, 8)
        out.update(upsample__x_seed=np.array(18040), upsample__x_shape=np.array(x.shape), upsample__y=m(torch.from_numpy(x)).numpy(),
                   upsample__seed=np.array(1804))
        # UNetRecurrent, the training configuration, three time steps of integer voxel grids
        kwargs = dict(num_bins=5, skip_type="sum", recurrent_block_type="convlstm", num_encoders=3, base_num_channels=32,
                      num_residual_blocks=2, use_upsample_conv=True, final_activation="", norm=None)
        with contextlib.redirect_stdout(io.StringIO()):
            net = un.UNetRecurrent(dict(kwargs)).eval()
        gain = 1.7                                              # keeps the random-init activations O(1) through the depth of the net
        probe = load_seeded(net, 1805, gain=gain)
        out["unet__weight_probe"] = np.concatenate([probe[k].ravel()[:3] for k in list(probe)[::5]])   # the recipe must reproduce these bits
        vox = g.integers(-3, 4, size=(3, 2, 5, 64, 64)).astype(np.float32)      # [T, B = 2, bins, H, W]
        vox[g.random(vox.shape) < 0.6] = 0.0                    # sparse, like event counts
        imgs = [net(torch.from_numpy(vox[t]))["image"].numpy() for t in range(3)]
        sd = net.state_dict()
        out.update(unet__vox=vox.astype(np.int8), unet__images=np.stack(imgs), unet__seed=np.array(1805), unet__gain=np.array(gain),
                   unet__keys=np.array(list(sd.keys())), unet__shapes=np.array([",".join(map(str, v.shape)) for v in sd.values()]),
                   unet__hidden0_absmax=np.array([float(s[0].abs().max()) for s in net.states]),
                   unet__n_params=np.array(sum(v.numel() for v in sd.values())))
        print(f"  g18 unet: {int(out['unet__n_params'])} parameters, image range {np.stack(imgs).min():.3f} .. {np.stack(imgs).max():.3f}, "
              f"std {np.stack(imgs).std():.3f}")
    save("g18_unet_modules.npz", **out)


def g15_bgr_to_gray():
    """The reference's bgr_to_gray (data/v2v_datasets.py:19-22) on ALL 2^24 colours, passed as the 4-D [N,H,W,3] stack the
    reference passes (np.dot's evaluation order depends on the array's dimensionality): sha256 of the full [256,256,256]
    table (index b, g, r) + four b-planes for debugging."""
    import hashlib
    b, g, r = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    stack = np.stack([b, g, r], axis=-1)                      # [256,256,256,3]
    gray = ref_ds.bgr_to_gray(stack)
    assert gray.dtype == np.uint8 and gray.shape == (256, 256, 256)
    save("g15_bgr_to_gray.npz", sha256=np.array(hashlib.sha256(gray.tobytes()).hexdigest()), planes_b=np.array([0, 37, 128, 255]),
         planes=gray[[0, 37, 128, 255]])


class _FakeH5:
    """Just enough of h5py.File for the reference's real-data loaders (data/testh5.py:33-50,101-143; data/dataset.py:375-427),
    backed by an in-memory Monash-layout sequence.  Test infrastructure of the golden generator only."""

    class _Dset:
        def __init__(self, arr, attrs=None):
            self._a, self.attrs, self.shape = arr, dict(attrs or {}), arr.shape

        def __getitem__(self, k):
            return self._a[k]

        def __len__(self):
            return len(self._a)

    class _Group:
        def __init__(self, items):
            self._items = items

        def keys(self):
            return self._items.keys()

        def __iter__(self):
            return iter(self._items)

        def __len__(self):
            return len(self._items)

        def __getitem__(self, k):
            return self._items[k]

    store = None          # set by the golden function: {"events": {...}, "images": {name: (array, attrs)}, "attrs": {...}}

    def __init__(self, path, mode="r"):
        s = _FakeH5.store
        self.attrs = dict(s["attrs"])
        self._groups = {}
        if "images" in s:
            self._groups = {"events": _FakeH5._Group({k: _FakeH5._Dset(v) for k, v in s["events"].items()}),
                            "images": _FakeH5._Group({k: _FakeH5._Dset(a, at) for k, (a, at) in sorted(s["images"].items())})}
        if "flow" in s:
            self._groups["flow"] = _FakeH5._Group({k: _FakeH5._Dset(a, at) for k, (a, at) in sorted(s["flow"].items())})
        for k in ("frames", "events_cache", "flow_top"):      # a voxel cache: top-level datasets `frames`, `events` (, `flow`)
            if k in s:
                self._groups[{"events_cache": "events", "flow_top": "flow"}.get(k, k)] = _FakeH5._Dset(s[k])

    def keys(self):
        return self._groups.keys()

    def __getitem__(self, path):
        node = self
        for part in path.split("/"):
            node = node._groups[part] if node is self else node[part]
        return node

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def g19_event_and_fps_loaders():
    """The other two real-data loaders of data/testh5.py that sit on the event-list voxeliser's data, run by the REFERENCE on golden G16's
    sequence (served through _FakeH5 like G16): TestH5EventDataset.__getitem__ (:305-381: raw [n,5] event rows per image interval, no
    voxelisation) and FPS_H5Dataset (:448-520: a frame-less stream cut at a fixed rate, one make_voxel per cut).  Inputs are G16's
    arrays (not stored again); only the reference's outputs are."""
    z = np.load(os.path.join(HERE, "g16_monash_sequence.npz"))
    keys = [str(k) for k in z["images/keys"]]
    _FakeH5.store = {"events": {k: z[f"events/{k}"] for k in ("ts", "xs", "ys", "ps")},
                     "images": {k: (z["images/stack"][i], {"event_idx": z["images/event_idx"][i], "timestamp": z["images/timestamp"][i]}) for i, k in enumerate(keys)},
                     "attrs": {"sensor_resolution": z["attrs/sensor_resolution"], "num_events": int(z["attrs/num_events"]), "num_imgs": int(z["attrs/num_imgs"]), "source": "hqf"}}
    sys.modules["h5py"].File = _FakeH5
    out = {}
    cfgs = {"a": {"sequence_length": 4, "num_bins": 5, "dataset_name": "hqf"},
            "b": {"sequence_length": 5, "warm_up_length": 1, "num_bins": 3, "output_additional_frame": True, "image_range": 1, "dataset_name": "evaid"}}
    for tag, cfg in cfgs.items():
        ds = ref_th5.TestH5EventDataset("/fake/hqf_h5/bike_bay_hdr.h5", cfg)
        out[f"ev_{tag}__len"] = np.array(len(ds))
        for i in range(len(ds)):
            s = ds[i]
            assert s["sequence_name"] == ["bike_bay_hdr"] * len(s["frame_idx"])
            out[f"ev_{tag}__{i}__frame"] = s["frame"].numpy()
            out[f"ev_{tag}__{i}__n"] = np.array(len(s["events"]))
            for j, e in enumerate(s["events"]):
                assert e.dtype == torch_mod().float64
                out[f"ev_{tag}__{i}__events{j}"] = e.numpy()
            out[f"ev_{tag}__{i}__meta"] = np.stack([s["real_begin_idx"].numpy(), s["frame_idx"].numpy()])
            out[f"ev_{tag}__{i}__source"] = np.array(int(s["data_source_idx"]))
    fcfgs = {"a": {"sequence_length": 6, "num_bins": 5, "FPS": 100, "H": 36, "W": 48, "dataset_name": "evbird"},
             "b": {"sequence_length": 80, "num_bins": 3, "interpolate_bins": True, "FPS": 40, "H": 40, "W": 50, "dataset_name": "evbird"}}
    for tag, cfg in fcfgs.items():
        ds = ref_th5.FPS_H5Dataset("/fake/evbird/lhy_0.h5", cfg)
        out[f"fps_{tag}__len"] = np.array(len(ds))
        out[f"fps_{tag}__samples"] = np.array(ds.samples)
        out[f"fps_{tag}__event_idx"] = np.asarray(ds.event_idx)
        for i in range(len(ds)):
            s = ds[i]
            assert set(s) == {"events", "data_source_idx", "sequence_name"} and s["sequence_name"] == ["lhy_0"] * s["events"].shape[0]
            out[f"fps_{tag}__{i}__events"] = s["events"].numpy()
            out[f"fps_{tag}__{i}__source"] = np.array(int(s["data_source_idx"]))
    save("g19_event_and_fps_loaders.npz", **out)


def g20_flow_and_cache_loaders():
    """TestH5FlowDataset (data/testh5.py:175-303: MVSEC-style sequences, one sample item per optic-flow map: the events between two flow
    maps' `event_idx`, the image named by the map's `image_idx`, the map itself) and TestH5CacheDataset (:383-446: pre-built voxel
    caches) run by the REFERENCE.  The flow sequence is G16's events and images + seven random flow maps (stored here: the one new input);
    the cache is what TestH5Dataset produced for G16's configuration "a" (frames [n,H,W] and events [n,Tb,H,W] + the two attributes)."""
    import torch
    z = np.load(os.path.join(HERE, "g16_monash_sequence.npz"))
    keys = [str(k) for k in z["images/keys"]]
    g = np.random.default_rng(2020)
    H, W = int(z["attrs/sensor_resolution"][0]), int(z["attrs/sensor_resolution"][1])
    n_flow = 7
    flow = g.normal(0, 2.0, size=(n_flow, 2, H, W)).astype(np.float32)
    n_ev = int(z["attrs/num_events"])
    flow_event_idx = np.array([0, 300, 1350, 1350, 1353, 2900, n_ev], dtype=np.int64)              # an empty and a 3-event interval
    flow_image_idx = np.array([0, 1, 2, 4, 5, 7, 11], dtype=np.int64)                             # the last one beyond the images: clamped (:236)
    fkeys = ["flow{:09d}".format(i) for i in range(n_flow)]
    store = {"events": {k: z[f"events/{k}"] for k in ("ts", "xs", "ys", "ps")},
             "images": {k: (z["images/stack"][i], {"event_idx": z["images/event_idx"][i], "timestamp": z["images/timestamp"][i]}) for i, k in enumerate(keys)},
             "flow": {k: (flow[i], {"event_idx": flow_event_idx[i], "image_idx": flow_image_idx[i]}) for i, k in enumerate(fkeys)},
             "attrs": {"sensor_resolution": z["attrs/sensor_resolution"], "num_events": n_ev, "num_imgs": int(z["attrs/num_imgs"]), "source": "mvsec"}}
    _FakeH5.store = store
    sys.modules["h5py"].File = _FakeH5
    out = {"flow/stack": flow, "flow/keys": np.array(fkeys), "flow/event_idx": flow_event_idx, "flow/image_idx": flow_image_idx}
    cfgs = {"a": {"sequence_length": 4, "num_bins": 5, "dataset_name": "mvsec"},
            "b": {"sequence_length": 3, "num_bins": 3, "interpolate_bins": True, "output_additional_frame": True, "output_additional_evs": True, "image_range": 1,
                  "dataset_name": "mvsec", "max_samples": 2}}
    for tag, cfg in cfgs.items():
        ds = ref_th5.TestH5FlowDataset("/fake/mvsec/indoor_flying1.h5", cfg)
        out[f"flow_{tag}__len"] = np.array(len(ds))
        out[f"flow_{tag}__samples"] = np.array(ds.samples)
        for i in range(len(ds)):
            s = ds[i]
            assert set(s) == {"frame", "events", "flow", "data_source_idx", "sequence_name", "frame_idx"} and s["sequence_name"] == ["indoor_flying1"] * len(s["frame_idx"])
            for k in ("frame", "events", "flow", "frame_idx"):
                out[f"flow_{tag}__{i}__{k}"] = s[k].numpy()
            out[f"flow_{tag}__{i}__source"] = np.array(int(s["data_source_idx"]))
    # the cache: every item of TestH5Dataset(config a with one long sample) stacked, as a converter would store it
    _FakeH5.store = {k: store[k] for k in ("events", "images", "attrs")}
    full = ref_th5.TestH5Dataset("/fake/hqf_h5/bike_bay_hdr.h5", {"sequence_length": 100, "num_bins": 5, "dataset_name": "hqf"})[0]
    cache = {"frames": full["frame"].numpy()[:, 0], "events_cache": full["events"].numpy(), "attrs": {"num_bins": 5, "interpolate_bins": False}}
    _FakeH5.store = cache
    ds = ref_th5.TestH5CacheDataset("/fake/cache/bike_bay_hdr.h5", {"sequence_length": 3, "num_bins": 5, "dataset_name": "hqf"})
    out["cache__frames"], out["cache__events"] = cache["frames"], cache["events_cache"]
    out["cache__len"] = np.array(len(ds))
    out["cache__samples"] = np.array(ds.samples)
    for i in range(len(ds)):
        s = ds[i]
        assert set(s) == {"frame", "events", "data_source_idx", "sequence_name"} and s["sequence_name"] == ["bike_bay_hdr"] * len(s["data_source_idx"])
        out[f"cache__{i}__frame"], out[f"cache__{i}__events"] = s["frame"].numpy(), s["events"].numpy()
        out[f"cache__{i}__source"] = s["data_source_idx"].numpy()
    save("g20_flow_and_cache_loaders.npz", **out)


def g21_esim_h5_dataset():
    """ESIMH5Dataset (data/esim_dataset.py:49-152), the training loader over the cached voxels scripts/esim_to_voxel.py writes, run by the
    REFERENCE on a small cache after random.seed / np.random.seed: random crop, flip, pause schedule, Gaussian or integer noise on every
    step, hot pixels.  The cache (frames / flow / events) is stored; the samples are the reference's outputs."""
    import random
    ref_esim = _load("ref_esim_dataset", os.path.join(REF, "data/esim_dataset.py"))
    g = np.random.default_rng(2121)
    n, H, W = 14, 20, 24
    cache = {"frames": g.random((n, 1, H, W)).astype(np.float32), "flow": g.normal(0, 1, (n, 2, H, W)).astype(np.float32),
             "events_cache": np.round(g.normal(0, 1.5, (n, 5, H, W))).astype(np.float32), "attrs": {"sensor_resolution": np.array([H, W])}}
    cache["flow_top"] = cache.pop("flow")
    _FakeH5.store = cache
    sys.modules["h5py"].File = _FakeH5
    out = {"frames": cache["frames"], "flow": cache["flow_top"], "events": cache["events_cache"], "attrs/sensor_resolution": np.array([H, W])}
    cfgs = {"a": {"sequence_length": 6, "random_crop_size": 16, "noise_std": 0.1, "hot_pixel_std": 0.1, "max_hot_pixel_fraction": 0.05},
            "b": {"sequence_length": 5, "step_size": 3, "random_crop_size": None, "random_flip": False, "noise_std": 0.7, "noise_fraction": 0.3,
                  "proba_pause_when_running": 0.4, "proba_pause_when_paused": 0.6, "hot_pixel_std": 2.0, "max_hot_pixel_fraction": 0.1, "integer_noise": True}}
    for tag, cfg in cfgs.items():
        ds = ref_esim.ESIMH5Dataset("/fake/esim_h5/seq.h5", cfg)
        out[f"ds_{tag}__len"] = np.array(len(ds))
        out[f"ds_{tag}__samples"] = np.array(ds.samples)
        for i in range(len(ds)):
            random.seed(100 + i)
            np.random.seed(200 + i)
            s = ds[i]
            assert set(s) == {"frame", "flow", "events", "data_source_idx"}
            for k in ("frame", "flow", "events"):
                out[f"ds_{tag}__{i}__{k}"] = s[k].numpy()
            out[f"ds_{tag}__{i}__source"] = np.array(int(s["data_source_idx"]))
    save("g21_esim_h5_dataset.npz", **out)


def g24_yaml_dataset_blocks():
    """Every dataset block (a mapping with `class_name`) of the reference's 15 experiment YAMLs (config/*.yaml), as data: the configuration
    keys and values a drop-in dataset class must accept.  Stored as JSON inside an .npz (one string); no reference code."""
    import glob
    import json
    import yaml
    blocks = []
    for f in sorted(glob.glob(os.path.join(REF, "config", "*.yaml"))):
        def walk(o, path):
            if isinstance(o, dict):
                if "class_name" in o:
                    blocks.append({"yaml": os.path.basename(f), "path": path, "block": o})
                for k, v in o.items():
                    walk(v, path + "/" + str(k))
            elif isinstance(o, list):
                for i, v in enumerate(o):
                    walk(v, f"{path}[{i}]")
        walk(yaml.safe_load(open(f)), "")
    save("g24_yaml_dataset_blocks.npz", blocks=np.array(json.dumps(blocks, sort_keys=True)))


def torch_mod():
    import torch
    return torch


def g16_monash_sequence():
    """A small event sequence in the Monash HDF5 layout + what the REFERENCE's own loaders make of it:
    TestH5Dataset.__getitem__ (data/testh5.py:96-173, two configurations) and DynamicH5Dataset.__getitem__ as
    scripts/esim_to_voxel.py:29-50 stacks and casts it (temporal_bilinear False / True).  h5py is not installed, so the
    reference's `h5py.File` is served by _FakeH5 above from the arrays that are also stored in the fixture."""
    import torch
    g = np.random.default_rng(1616)
    H, W, n_img = 36, 48, 9
    counts = [700, 650, 0, 1, 2, 900, 800, 3, 640]                                 # events before image i (an empty, a 1-, a 2- and a 3-event interval)
    n_ev = int(sum(counts))
    img_ts = np.cumsum(g.uniform(0.02, 0.05, size=n_img)) + 10.0
    ts = np.concatenate([np.sort(g.uniform(img_ts[i - 1] if i else 10.0, img_ts[i], size=c)) for i, c in enumerate(counts)])
    ts[counts[0] + counts[1] + 1: counts[0] + counts[1] + 3] = ts[counts[0] + counts[1] + 1]        # the 2-event interval: equal timestamps (dt == 0)
    xs = g.integers(0, W, size=n_ev).astype(np.uint16)
    ys = g.integers(0, H, size=n_ev).astype(np.uint16)
    ps = (g.random(n_ev) < 0.5).astype(np.uint8)
    event_idx = np.cumsum(counts).astype(np.int64)
    images = g.integers(0, 256, size=(n_img, H, W)).astype(np.uint8)
    keys = ["image{:09d}".format(i) for i in range(n_img)]
    _FakeH5.store = {"events": {"ts": ts, "xs": xs, "ys": ys, "ps": ps},
                     "images": {k: (images[i], {"event_idx": event_idx[i], "timestamp": img_ts[i]}) for i, k in enumerate(keys)},
                     "attrs": {"sensor_resolution": np.array([H, W]), "num_events": n_ev, "num_imgs": n_img, "source": "hqf"}}
    sys.modules["h5py"].File = _FakeH5
    out = {"events/ts": ts, "events/xs": xs, "events/ys": ys, "events/ps": ps, "images/stack": images, "images/keys": np.array(keys),
           "images/event_idx": event_idx, "images/timestamp": img_ts, "attrs/sensor_resolution": np.array([H, W]),
           "attrs/num_events": np.array(n_ev), "attrs/num_imgs": np.array(n_img), "attrs/source": np.array("hqf")}
    cfgs = {"a": {"sequence_length": 4, "num_bins": 5, "dataset_name": "hqf"},
            "b": {"sequence_length": 5, "warm_up_length": 1, "num_bins": 3, "interpolate_bins": True, "output_additional_frame": True,
                  "output_additional_evs": True, "image_range": 1, "dataset_name": "hqf"}}
    for tag, cfg in cfgs.items():
        ds = ref_th5.TestH5Dataset("/fake/hqf_h5/bike_bay_hdr.h5", cfg)
        out[f"th5_{tag}__len"] = np.array(len(ds))
        out[f"th5_{tag}__samples"] = np.array(ds.samples)
        for i in range(len(ds)):
            s = ds[i]
            assert s["sequence_name"] == ["bike_bay_hdr"] * len(s["frame_idx"])
            out[f"th5_{tag}__{i}__frame"] = s["frame"].numpy()
            out[f"th5_{tag}__{i}__events"] = s["events"].numpy()
            out[f"th5_{tag}__{i}__meta"] = np.stack([s["real_begin_idx"].numpy(), s["frame_idx"].numpy()])
            out[f"th5_{tag}__{i}__source"] = np.array(int(s["data_source_idx"]))
    from data.dataset import DynamicH5Dataset
    for tb in (False, True):
        ds = DynamicH5Dataset(data_path="/fake/esim_h5/seq.h5", temporal_bilinear=tb)
        items = [ds[i] for i in range(len(ds))]
        tag = "bil" if tb else "nobi"
        out[f"cache_{tag}__frames"] = np.stack([it["frame"].numpy() for it in items]).astype(np.float32)          # esim_to_voxel.py:38-50
        out[f"cache_{tag}__flow_is_zero"] = np.array(all(float(it["flow"].abs().sum()) == 0 for it in items))
        out[f"cache_{tag}__events"] = np.stack([it["events"].numpy() for it in items]).astype(np.float32)
        out[f"cache_{tag}__timestamps"] = np.stack([it["timestamp"] for it in items]).astype(np.float32)
        out[f"cache_{tag}__dt"] = np.stack([it["dt"] for it in items]).astype(np.float32)
    save("g16_monash_sequence.npz", **out)


def g17_degrade_video():
    """The reference's degrade_video (data/v2v_datasets.py:413-486) in its three NumPy modes on a small clip, gray ([H,W,1]) and
    colour frames, after np.random.seed.  cv2 is absent: the only OpenCV call on these paths, cv2.flip(img, 1), is served by
    its definition (mirror the columns; a [H,W,1] input comes back [H,W], which the reference itself undoes at :467-469)."""
    cv2 = sys.modules["cv2"]
    cv2.flip = lambda img, code: (np.ascontiguousarray(img[:, ::-1, 0]) if img.ndim == 3 and img.shape[2] == 1 else np.ascontiguousarray(img[:, ::-1]))
    rng = np.random.default_rng(17)
    out = {}
    for chan in (1, 3):
        clip = rng.integers(0, 256, size=(7, 12, 16, chan), dtype=np.uint8)
        out[f"clip_c{chan}"] = clip
        for mode in ("dirtyshotcut", "hdr", "ldr"):
            for seed in (0, 1, 2):
                inst = object.__new__(ref_ds.WebvidDatasetV2)
                inst.video_degrade = mode
                np.random.seed(seed)
                got = inst.degrade_video([f.copy() for f in clip])
                out[f"{mode}_c{chan}_s{seed}"] = np.stack([np.asarray(g) for g in got])
    save("g17_degrade_video.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g4", "g5", "g6", "g7", "g8", "g9", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20", "g21", "g22", "g23", "g24"]
    fns = {"g1": g1_luts, "g2": g2_g3_esim_clean, "g4": g4_esim_noisy, "g5": g5_floor_divide,
           "g6": g6_imgs_to_voxels, "g7": g7_bilinear, "g8": g8_make_voxel, "g9": g9_v2e, "g11": g11_philox_fed, "g14": g14_v2e_native, "g15": g15_bgr_to_gray, "g16": g16_monash_sequence, "g19": g19_event_and_fps_loaders, "g20": g20_flow_and_cache_loaders, "g21": g21_esim_h5_dataset, "g17": g17_degrade_video, "g18": g18_unet_modules, "g12": g12_events_to_voxel_torch, "g22": g22_voxel_grid_lists, "g23": g23_events_to_image, "g24": g24_yaml_dataset_blocks,
           "g13": g13_normalize_batch_voxel}
    for w in which:
        fns[w]()
