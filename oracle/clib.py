"""ctypes view of oracle/libv2v_oracle.so (the scalar C restatement).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libv2v_oracle.so")
_lib = None

IN_U8, IN_F32 = 0, 1
RNG_NONE, RNG_PHILOX, RNG_REPLAY = 0, 1, 2
BIN_SUM, BIN_BILINEAR = 0, 1


class Replay(C.Structure):
    _fields_ = [("u_init", C.c_void_p), ("u_hot", C.c_void_p), ("g_hot", C.c_void_p), ("g_base", C.c_void_p)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "v2v_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libv2v_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oracle_floor_divide.restype = C.c_double
        _lib.oracle_floor_divide.argtypes = [C.c_double, C.c_double]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def floor_divide(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(np.broadcast_to(b, a.shape), dtype=np.float64)
    q = np.empty_like(a)
    lib().oracle_floor_divide_vec(_p(a), _p(b), _p(q), C.c_int64(a.size))
    return q


def philox4x32(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().oracle_philox4x32(c, k, o)
    return list(o)


def gauss16(w):
    """(g0, g1) of one 32-bit word through the native table-inversion generator (top 14 bits of the high half-word -> g0, of the low -> g1)."""
    o = (C.c_float * 2)()
    lib().oracle_gauss16(C.c_uint32(w), o)
    return np.float32(o[0]), np.float32(o[1])


def gauss16_many(words):
    w = np.ascontiguousarray(words, dtype=np.uint32)
    g0, g1 = np.empty(w.size, dtype=np.float32), np.empty(w.size, dtype=np.float32)
    lib().oracle_gauss16_many(_p(w), C.c_int64(w.size), _p(g0), _p(g1))
    return g0, g1


def philox_uniform_field(seed, clip_id, field, n_pix, stream=0):
    out = np.empty(n_pix, dtype=np.float64)
    lib().oracle_philox_uniform_field(C.c_uint64(seed), C.c_uint32(clip_id & 0xFFFFFFFF), C.c_uint32(field),
                                      C.c_uint32(stream), C.c_int64(n_pix), _p(out))
    return out


def noise_rounds():
    """Philox rounds of the per-time-step noise fields (7); every other native field uses 10."""
    return int(lib().oracle_noise_rounds())


def philox_gauss_field(seed, clip_id, field, n_pix, stream=0, comp=0, rounds=10):
    """Normal `comp` (0: first, 1: second) of every pixel's table-inversion deviate pair (the two halves of one word) in Philox block `field`."""
    out = np.empty(n_pix, dtype=np.float32)
    lib().oracle_philox_gauss_field(C.c_uint64(seed), C.c_uint32(clip_id & 0xFFFFFFFF), C.c_uint32(field),
                                    C.c_uint32(stream), C.c_int64(n_pix), C.c_int(comp), C.c_int(rounds), _p(out))
    return out


def philox_uniform16_field(seed, clip_id, field, n_pix, stream=1, low=0):
    """Native shot-noise uniforms (midpoint grid, 16 bits): high (ON) or low (OFF) half of every pixel's word in block `field`."""
    out = np.empty(n_pix, dtype=np.float32)
    lib().oracle_philox_uniform16_field(C.c_uint64(seed), C.c_uint32(clip_id & 0xFFFFFFFF), C.c_uint32(field), C.c_uint32(stream),
                                        C.c_int64(n_pix), C.c_int(low), _p(out))
    return out


def expf_det(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().oracle_expf_det_vec(_p(x), C.c_int64(x.size), _p(out))
    return out


def poisson_inv_f32(lam, u):
    """The native Poisson sampler (float32 inversion), elementwise: counts as int64."""
    lam = np.ascontiguousarray(lam, dtype=np.float32)
    u = np.ascontiguousarray(np.broadcast_to(u, lam.shape), dtype=np.float32)
    out = np.empty(lam.shape, dtype=np.int64)
    lib().oracle_poisson_inv_f32_vec(_p(lam), _p(u), C.c_int64(lam.size), _p(out))
    return out


def bgr_to_gray(img):
    """data/v2v_datasets.py:19-22 on a [...,3] uint8 stack, in NumPy's own evaluation order (golden G15)."""
    img = np.ascontiguousarray(img[..., :3], dtype=np.uint8)
    out = np.empty(img.shape[:-1], dtype=np.uint8)
    lib().oracle_bgr_to_gray(_p(img), C.c_int64(out.size), _p(out))
    return out


def esim_voxel(frames, params, luts, *, noise_external=False, rng_mode=RNG_PHILOX, seed=0, clip_id0=0,
               bin_mode=BIN_SUM, num_bins=5, frames_per_bin=1, replay=None, threads=None, check_integer=True, timing=None):
    """frames [B,N,H,W] uint8 or float32 (integer-valued); params [5] or [B,5].
    Returns (voxel float64 [B,L,Tb,H,W] | [B,Tb,H,W], totals int64 [B,2]).
    check_integer=False skips the (single-threaded NumPy) validation of float32 content -- bench.py's timed leg, whose clips the parity guard
    has already been through; timing: a dict that receives `call_s`, the seconds spent inside the library call alone."""
    frames = np.ascontiguousarray(frames)
    b, n, h, w = frames.shape
    k = n - 1
    in_dtype = {np.dtype(np.uint8): IN_U8, np.dtype(np.float32): IN_F32}[frames.dtype]
    if in_dtype == IN_F32 and check_integer:
        iv = frames.astype(np.int64)
        assert np.array_equal(iv, frames) and iv.min() >= 0 and iv.max() <= 255, "C oracle is LUT-only"
    params = np.ascontiguousarray(params, dtype=np.float64)
    stride = 0 if params.ndim == 1 else 5
    lut64 = np.ascontiguousarray(luts["lut64"], dtype=np.float64)
    lut32 = np.ascontiguousarray(luts["lut32"], dtype=np.float32)
    if bin_mode == BIN_SUM:
        assert k % (num_bins * frames_per_bin) == 0
        shape = (b, k // (num_bins * frames_per_bin), num_bins, h, w)
    else:
        shape = (b, num_bins, h, w)
    out = np.zeros(shape, dtype=np.float64)
    totals = np.zeros((b, 2), dtype=np.int64)
    L = lib()
    if replay is not None:
        assert b == 1 and rng_mode == RNG_REPLAY
        keep = [np.ascontiguousarray(x, dtype=np.float64) for x in replay]
        rp = Replay(*[x.ctypes.data for x in keep])
        rc = L.oracle_esim_voxel_clip(_p(frames), in_dtype, C.c_int64(n), C.c_int64(h * w), _p(lut64), _p(lut32),
                                      _p(params), int(noise_external), rng_mode, C.c_uint64(seed),
                                      C.c_uint32(clip_id0), C.byref(rp), bin_mode, num_bins, frames_per_bin,
                                      _p(out), _p(totals))
    else:
        if threads is not None:
            os.environ["OMP_NUM_THREADS"] = str(threads)
        import time
        t0 = time.perf_counter()
        rc = L.oracle_esim_voxel_batch(_p(frames), in_dtype, C.c_int64(b), C.c_int64(n), C.c_int64(h * w),
                                       _p(lut64), _p(lut32), _p(params), C.c_int64(stride), int(noise_external),
                                       rng_mode, C.c_uint64(seed), C.c_uint64(clip_id0), bin_mode, num_bins,
                                       frames_per_bin, _p(out), _p(totals))
        if timing is not None:
            timing["call_s"] = time.perf_counter() - t0
    if rc != 0:
        raise RuntimeError(f"oracle_esim_voxel rc={rc}")
    return out, totals


def make_voxel(ts_us, xs, ys, ps01, num_bins, h, w, interpolate):
    ts_us = np.ascontiguousarray(ts_us, dtype=np.int64)
    xs = np.ascontiguousarray(xs, dtype=np.int64)
    ys = np.ascontiguousarray(ys, dtype=np.int64)
    ps01 = np.ascontiguousarray(ps01, dtype=np.int8)
    out = np.zeros((num_bins, h, w), dtype=np.float64)
    rc = lib().oracle_make_voxel(_p(ts_us), _p(xs), _p(ys), _p(ps01), C.c_int64(ts_us.size), num_bins,
                                 C.c_int64(h), C.c_int64(w), int(interpolate), _p(out))
    if rc != 0:
        raise RuntimeError(f"oracle_make_voxel rc={rc}")
    return out


# ---------------------------------------------------------------- v2e model
V2E_MODELS = {"pn_related": 0, "spatial_independent": 1, "spatial_temporal_independent": 2}


class V2EParams(C.Structure):
    _fields_ = [("fps", C.c_double), ("threshold_model", C.c_int), ("thres_mean_mean", C.c_double),
                ("thres_mean_std", C.c_double), ("thres_diff_mean", C.c_double), ("thres_diff_std", C.c_double),
                ("cutoff_hz", C.c_double), ("leak_rate_hz", C.c_double), ("refractory_period_s", C.c_double),
                ("shot_noise_rate_hz", C.c_double), ("leak_jitter_fraction", C.c_double),
                ("noise_rate_cov_decades", C.c_double), ("uint8_wrap", C.c_int)]


class V2EReplay(C.Structure):
    _fields_ = [("pos_thres", C.c_void_p), ("neg_thres", C.c_void_p), ("thres_frame_stride", C.c_int64),
                ("noise_rate", C.c_void_p), ("leak_randn", C.c_void_p), ("shot_pos", C.c_void_p), ("shot_neg", C.c_void_p)]


def v2e_params(FPS, threshold_model, thres_mean_mean, thres_mean_std, thres_diff_mean, thres_diff_std, cutoff_hz,
               leak_rate_hz, refractory_period_s, shot_noise_rate_hz, leak_jitter_fraction, noise_rate_cov_decades,
               uint8_wrap=True):
    return V2EParams(float(FPS), V2E_MODELS[threshold_model], thres_mean_mean, thres_mean_std, thres_diff_mean,
                     thres_diff_std, cutoff_hz, leak_rate_hz, refractory_period_s, shot_noise_rate_hz,
                     leak_jitter_fraction, noise_rate_cov_decades, int(bool(uint8_wrap)))


def v2e_replay_arrays(record, k, hw):
    """Pack the fields recorded by oracle.v2v_oracle.v2e_video_to_voxel(record=...) for one clip."""
    pt = np.ascontiguousarray(np.stack(record["pos_thres"]).reshape(-1, hw), dtype=np.float64)
    nt = np.ascontiguousarray(np.stack(record["neg_thres"]).reshape(-1, hw), dtype=np.float64)
    temporal = pt.shape[0] > 1
    if temporal:                       # entry 0 is the _init draw (never used for events); frames 1..N-1 follow
        pt, nt = np.ascontiguousarray(pt[1:]), np.ascontiguousarray(nt[1:])
    arrs = {"pos_thres": pt, "neg_thres": nt, "stride": hw if temporal else 0,
            "noise_rate": np.ascontiguousarray(record["noise_rate"].reshape(hw), dtype=np.float32),
            "leak_randn": np.ascontiguousarray(np.stack(record["leak_randn"]).reshape(k, hw), dtype=np.float64) if record["leak_randn"] else None,
            "shot_pos": np.ascontiguousarray(np.stack(record["shot_pos"]).reshape(k, hw), dtype=np.int64),
            "shot_neg": np.ascontiguousarray(np.stack(record["shot_neg"]).reshape(k, hw), dtype=np.int64)}
    return arrs


def v2e_voxel(frames, params: V2EParams, luts, *, rng_mode=RNG_PHILOX, seed=0, clip_id0=0, bin_mode=BIN_SUM, num_bins=5,
              frames_per_bin=1, replay=None):
    """frames [B,N,H,W] uint8 / float32 (integer-valued) -> (float64 voxels, totals[B,2])."""
    frames = np.ascontiguousarray(frames)
    b, n, h, w = frames.shape
    k = n - 1
    in_dtype = {np.dtype(np.uint8): IN_U8, np.dtype(np.float32): IN_F32}[frames.dtype]
    lut = np.ascontiguousarray(luts["v2e32"], dtype=np.float32)
    shape = (b, k // (num_bins * frames_per_bin), num_bins, h, w) if bin_mode == BIN_SUM else (b, num_bins, h, w)
    out = np.zeros(shape, dtype=np.float64)
    totals = np.zeros((b, 2), dtype=np.int64)
    L = lib()
    if replay is not None:
        assert b == 1
        keep = replay
        rp = V2EReplay(keep["pos_thres"].ctypes.data, keep["neg_thres"].ctypes.data, keep["stride"],
                       keep["noise_rate"].ctypes.data,
                       keep["leak_randn"].ctypes.data if keep["leak_randn"] is not None else None,
                       keep["shot_pos"].ctypes.data, keep["shot_neg"].ctypes.data)
        rc = L.oracle_v2e_voxel_clip(_p(frames), in_dtype, C.c_int64(n), C.c_int64(h * w), _p(lut), C.byref(params),
                                     RNG_REPLAY, C.c_uint64(seed), C.c_uint32(clip_id0), C.byref(rp), bin_mode, num_bins,
                                     frames_per_bin, _p(out), _p(totals))
    else:
        rc = L.oracle_v2e_voxel_batch(_p(frames), in_dtype, C.c_int64(b), C.c_int64(n), C.c_int64(h * w), _p(lut),
                                      C.byref(params), C.c_uint64(seed), C.c_uint64(clip_id0), bin_mode, num_bins,
                                      frames_per_bin, _p(out), _p(totals))
    if rc != 0:
        raise RuntimeError(f"oracle_v2e_voxel rc={rc}")
    return out, totals
