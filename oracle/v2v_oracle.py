"""CPU restatement of the V2V video->voxel hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  Nothing under v2v_amd/ (the product) imports it; the product path fails
loudly when the HIP library is missing and never falls back to this code.

Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
against golden vectors captured from the imported reference in the build container
(tests/golden/make_goldens.py is the generating script; the reference itself never
travels).  Each function cites the reference file:line it restates.

NumPy is the arithmetic engine because the reference's arithmetic IS NumPy's
(float64 state, np.floor_divide == npy_divmod, legacy RandomState streams).  The
scalar C twin lives in oracle/v2v_oracle.c (same algorithm, one pixel at a time).
"""
from __future__ import annotations

import math
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_GOLDEN = os.path.join(os.path.dirname(_HERE), "tests", "golden")

# ----------------------------------------------------------------------------------------
# Log-intensity tables (reference: data/v2v_core_esim.py:3-4,33-34; data/v2v_core_v2e.py:108-137)
# ----------------------------------------------------------------------------------------

def esim_log_direct(video: np.ndarray) -> np.ndarray:
    """Literal op sequence of v2v_core_esim.py:33-34 (dtype follows the input, as NumPy does)."""
    lin = (video / 255) ** 2.2 * 255          # reverse gamma, v2v_core_esim.py:3-4
    return np.log(0.001 + lin / 255.0)        # v2v_core_esim.py:34


def v2e_linlog_direct(frame: np.ndarray) -> np.ndarray:
    """Effective formula of lin_log (v2v_core_v2e.py:123-137): everything before :135 is dead."""
    x = frame.astype(np.float64)
    return np.log(x / 255 + 0.01).astype(np.float32)


_LUTS = None


def load_luts() -> dict:
    """Committed golden G1: the reference's own outputs on the 256 integer intensities.

    lut64 = oracle on arange(256,uint8) (float64 path), lut32 = oracle on arange(256,float32)
    (float32 path), v2e32 = lin_log on arange(256).  NumPy's SIMD log/pow differ from libm by
    1 ulp on a few entries, and which SIMD kernel runs depends on the host CPU, so the table is
    data, not something to recompute on another machine.
    """
    global _LUTS
    if _LUTS is None:
        z = np.load(os.path.join(_GOLDEN, "g1_luts.npz"))
        _LUTS = {k: z[k] for k in z.files}
    return _LUTS


def esim_log_lut(video: np.ndarray) -> np.ndarray:
    """LUT form of esim_log_direct for integer-valued content in 0..255 (uint8 / float64 -> lut64,
    float32 container -> lut32).  Bitwise equal to the direct form in the build container."""
    luts = load_luts()
    idx = video.astype(np.int64)
    if not np.array_equal(idx, video) or idx.min() < 0 or idx.max() > 255:
        raise ValueError("LUT path needs integer-valued content in 0..255")
    return luts["lut32"][idx] if video.dtype == np.float32 else luts["lut64"][idx]


# ----------------------------------------------------------------------------------------
# np.floor_divide, restated as a scalar algorithm (third-party: numpy npy_divmod, numpy 2.2.6
# numpy/_core/src/npymath/npy_math_internal.h.src; call sites v2v_core_esim.py:51,54 and
# v2v_core_v2e.py:59-60)
# ----------------------------------------------------------------------------------------

def floor_divide_scalar(a: float, b: float) -> float:
    if b == 0.0:
        return a / b if a != 0.0 and not math.isnan(a) else float("nan")
    mod = math.fmod(a, b)
    div = (a - mod) / b
    if mod != 0.0:
        if (b < 0) != (mod < 0):
            mod += b
            div -= 1.0
    if div != 0.0:
        fl = math.floor(div)
        if div - fl > 0.5:
            fl += 1.0
        return float(fl)
    return math.copysign(0.0, a / b)


# ----------------------------------------------------------------------------------------
# RNG sources.  The reference draws from the global legacy np.random stream in a fixed order
# (v2v_core_esim.py:29,37,38,44).  Three interchangeable sources with .rand/.randn:
# ----------------------------------------------------------------------------------------

class GlobalNumpyRNG:
    """The reference's own source: global MT19937 legacy stream."""
    rand = staticmethod(lambda h, w: np.random.rand(h, w))
    randn = staticmethod(lambda h, w: np.random.randn(h, w))


class ReplayRNG:
    """Pre-drawn fields handed back in draw order (what the HIP replay mode consumes)."""

    def __init__(self, rand_fields, randn_fields):
        self._r = list(rand_fields)
        self._n = list(randn_fields)

    def rand(self, h, w):
        return self._r.pop(0)

    def randn(self, h, w):
        return self._n.pop(0)


def draw_replay_fields(n_frames: int, h: int, w: int):
    """Draw the ESIM fields from the *global* np.random stream in the reference's order
    (v2v_core_esim.py:29 rand, :37 rand, :38 randn, then one randn per pair :44)."""
    u_init = np.random.rand(h, w)
    u_hot = np.random.rand(h, w)
    g_hot = np.random.randn(h, w)
    g_base = np.stack([np.random.randn(h, w) for _ in range(n_frames - 1)]) if n_frames > 1 \
        else np.zeros((0, h, w))
    return u_init, u_hot, g_hot, g_base


# ---- counter-based Philox4x32-10 (Salmon et al., SC'11; Random123 KAT vectors in tests) ----
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Inputs broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32) for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


# Field ids of the device-native RNG mode (shared contract with v2v_amd/csrc/v2v_rng.h).
FIELD_POT_INIT, FIELD_HOT_MASK, FIELD_HOT_GAUSS, FIELD_BASE0 = 0, 1, 2, 3
STREAM_ESIM = 0


def philox_uniform53(seed: int, clip_id: int, field: int, n_pix: int, stream: int = STREAM_ESIM):
    """float64 uniforms in [0,1), numpy's 53-bit recipe ((a>>5)*2^26+(b>>6))/2^53 on Philox words.
    Pixel p uses words (2j,2j+1), j=p&1, of the block with counter (p>>1, field, clip_id, stream)."""
    p = np.arange(n_pix, dtype=np.uint32)
    w = philox4x32(p >> np.uint32(1), np.uint32(field), np.uint32(clip_id & 0xFFFFFFFF), np.uint32(stream),
                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    odd = (p & np.uint32(1)).astype(bool)
    a = np.where(odd, w[2], w[0]).astype(np.float64)
    b = np.where(odd, w[3], w[1])
    a = np.floor(a / 32.0)                      # a >> 5
    b = np.floor(b.astype(np.float64) / 64.0)   # b >> 6
    return (a * 67108864.0 + b) / 9007199254740992.0


def philox_gauss32(seed: int, clip_id: int, field: int, n_pix: int, stream: int = STREAM_ESIM, comp: int = 0, rounds: int = 10):
    """float32 standard normals from Philox words by direct table inversion (oracle/v2v_oracle.c: each 16-bit half of a word
    indexes the 8192-entry inverse-CDF table oracle/gauss_icdf.inc, sign from the top bit -- the same data the device holds).
    Pixel p: word p&3 of block (p>>2, field, clip_id, stream) -> one deviate pair (low half, high half); comp picks its member."""
    from oracle import clib  # local import: the C twin is optional for everything else
    return clib.philox_gauss_field(seed, clip_id, field, n_pix, stream, comp, rounds)


class PhiloxFieldRNG:
    """Adapter exposing the device-native fields through .rand/.randn in the reference's draw
    order, so the *reference itself* can be run on them (golden G11) and so can this oracle."""

    def __init__(self, seed: int, clip_id: int):
        self.seed, self.clip_id = seed, clip_id
        self._n_rand = 0
        self._n_randn = 0

    def rand(self, h, w):
        field = (FIELD_POT_INIT, FIELD_HOT_MASK)[self._n_rand]
        self._n_rand += 1
        return philox_uniform53(self.seed, self.clip_id, field, h * w).reshape(h, w)

    def randn(self, h, w):
        # draw 0: hot-pixel normals (first member of block 2); draw 1+k: base noise of frame pair k = member k&1 of
        # block 3 + (k>>1) -- one Philox block and one deviate pair (one word) per pixel serve two consecutive pairs
        from oracle import clib
        if self._n_randn == 0:
            field, comp, rounds = FIELD_HOT_GAUSS, 0, 10
        else:
            k = self._n_randn - 1
            field, comp, rounds = FIELD_BASE0 + (k >> 1), k & 1, clib.noise_rounds()     # per-step field: Philox4x32-7
        self._n_randn += 1
        return philox_gauss32(self.seed, self.clip_id, field, h * w, comp=comp, rounds=rounds).astype(np.float64).reshape(h, w)


# ----------------------------------------------------------------------------------------
# ESIM frame-pair simulator (reference: data/v2v_core_esim.py:26-69)
# ----------------------------------------------------------------------------------------

def esim_video_to_voxel(video, pos_thres=0.2, neg_thres=0.2, base_noise_std=0.1,
                        hot_pixel_fraction=0.001, hot_pixel_std=0.1, put_noise_external=False,
                        rng=GlobalNumpyRNG, use_lut=False, return_polarity=False):
    """[N,H,W] -> float64 [N-1,H,W] signed event counts per frame pair.

    Order of operations and of RNG draws exactly as v2v_core_esim.py:29-67:
      rand (potential init) ; log frames ; rand (hot mask) ; randn (hot noise) ;
      per pair: += diff ; randn (base noise, drawn even when std == 0) ; two separate noise adds ;
      ON = floor_divide where pot >= C+ ; OFF = floor_divide(-pot) where pot <= -C- ; reset.
    """
    n, h, w = video.shape
    potential = rng.rand(h, w) * (pos_thres + neg_thres) - neg_thres           # :29
    log_imgs = esim_log_lut(video) if use_lut else esim_log_direct(video)      # :33-34
    hot_mask = rng.rand(h, w) < hot_pixel_fraction                             # :37
    hot_noise = np.where(hot_mask, hot_pixel_std * rng.randn(h, w), 0)         # :38-39
    out, pol = [], []
    for i in range(n - 1):
        potential = potential + (log_imgs[i + 1] - log_imgs[i])                # :42-43 (diff in input precision)
        base_noise = base_noise_std * rng.randn(h, w)                          # :44
        if not put_noise_external:
            potential = potential + base_noise                                  # :48
            potential = potential + hot_noise                                   # :49
        with np.errstate(divide="ignore", invalid="ignore"):
            on = np.where(potential >= pos_thres, np.floor_divide(potential, pos_thres), 0)     # :51-52
            off = np.where(potential <= -neg_thres, np.floor_divide(-potential, neg_thres), 0)  # :54-55
        potential = potential - on * pos_thres                                 # :57
        potential = potential + off * neg_thres                                # :58
        vox = on - off                                                          # :60
        if put_noise_external:
            vox = vox + base_noise                                              # :64
            vox = vox + hot_noise                                               # :65
        out.append(vox)
        pol.append((on, off))
    out = np.array(out).reshape(n - 1, h, w)
    if return_polarity:
        on = np.array([p[0] for p in pol]).reshape(n - 1, h, w)
        off = np.array([p[1] for p in pol]).reshape(n - 1, h, w)
        return out, on, off
    return out


# ----------------------------------------------------------------------------------------
# Binning of per-pair counts into voxel grids
# ----------------------------------------------------------------------------------------

def bin_sum(counts, num_bins, frames_per_bin):
    """v2v_datasets.py:365,399-400: [K,H,W] -> [L,Tb,H,W] by summing frames_per_bin pairs."""
    k, h, w = counts.shape
    if k % (num_bins * frames_per_bin) != 0:
        raise AssertionError("(N-1) % (num_bins*frames_per_bin) != 0")
    l = k // (num_bins * frames_per_bin)
    return counts.reshape(l, num_bins, frames_per_bin, h, w).sum(axis=2)


def bilinear_weights(k_pairs: int, num_bins: int) -> np.ndarray:
    """Temporal-bilinear weights of utils/event_utils.py:715-719 for pseudo-events at ts=0..K-1
    (ts[0]=0, ts[-1]=K-1).  Returns float64 [Tb,K]: w[b,k] = max(0, 1-|k/(K-1)*(Tb-1) - b|)."""
    ts = np.arange(k_pairs, dtype=np.float64)
    dt = ts[-1] - ts[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        t_norm = (ts - ts[0]) / dt * (num_bins - 1)
    return np.stack([np.maximum(0.0, 1.0 - np.abs(t_norm - b)) for b in range(num_bins)])


def bin_bilinear(counts, num_bins):
    """SURVEY §8a composition: event_utils.events_to_voxel (:692-728) fed one pseudo-event per
    (k,y,x) with weight counts[k,y,x]; bincount accumulates in event (k-ascending) order in fp64."""
    k, h, w = counts.shape
    wts = bilinear_weights(k, num_bins)
    out = np.zeros((num_bins, h, w), dtype=np.float64)
    for b in range(num_bins):
        for kk in range(k):
            out[b] = out[b] + counts[kk] * wts[b, kk]       # product rounded, then added (no fma)
    return out


def imgs_to_voxels(imgs, num_bins, frames_per_bin, threshold_range=(0.05, 2),
                   max_thres_pos_neg_gap=1.5, base_noise_std_range=(0, 0.2),
                   hot_pixel_fraction_range=(0, 0.001), hot_pixel_std_range=(0, 0.2),
                   put_noise_external=False, scale_noise_strength=False,
                   use_fixed_thresholds=False, pos_thres=None, neg_thres=None, use_lut=False):
    """v2v_datasets.py:363-410 with its six scalar draws from the global np.random stream."""
    n = imgs.shape[0]
    assert (n - 1) % (num_bins * frames_per_bin) == 0                          # :365
    if not use_fixed_thresholds:
        thres_1 = np.random.uniform(*threshold_range)                          # :369
        gap = np.random.uniform(1, max_thres_pos_neg_gap)                      # :370
        thres_2 = thres_1 * gap
        if np.random.rand() > 0.5:                                             # :372
            pos_thres, neg_thres = thres_1, thres_2
        else:
            pos_thres, neg_thres = thres_2, thres_1
    base_noise_std = np.random.uniform(*base_noise_std_range)                  # :379
    hot_pixel_fraction = np.random.uniform(*hot_pixel_fraction_range)          # :380
    hot_pixel_std = np.random.uniform(*hot_pixel_std_range)                    # :381
    if scale_noise_strength and not put_noise_external:                        # :383-386
        base_noise_std = base_noise_std * pos_thres
        hot_pixel_std = hot_pixel_std * pos_thres
    counts = esim_video_to_voxel(imgs, pos_thres, neg_thres, base_noise_std, hot_pixel_fraction,
                                 hot_pixel_std, put_noise_external, use_lut=use_lut)
    params = {"pos_thres": pos_thres, "neg_thres": neg_thres, "base_noise_std": base_noise_std,
              "hot_pixel_fraction": hot_pixel_fraction, "hot_pixel_std": hot_pixel_std}
    return params, bin_sum(counts, num_bins, frames_per_bin)


def bgr_to_gray(img_stack):
    """v2v_datasets.py:19-22: weights applied to channels 0,1,2 in that order, truncating cast."""
    return np.dot(img_stack[..., :3], [0.5870, 0.1140, 0.2989]).astype(np.uint8)


# ----------------------------------------------------------------------------------------
# Event-list voxelisers
# ----------------------------------------------------------------------------------------

def events_to_voxel(xs, ys, ts, ps, num_bins, sensor_size):
    """utils/event_utils.py:692-728 (temporal_bilinear=True) + events_to_image :155-174.
    1-D inputs accepted (the reference needs ts, ps as [N,1] columns - SURVEY §4)."""
    xs = np.asarray(xs).reshape(-1)
    ys = np.asarray(ys).reshape(-1)
    ts = np.asarray(ts, dtype=np.float64).reshape(-1)
    ps = np.asarray(ps, dtype=np.float64).reshape(-1)
    h, w = sensor_size
    dt = ts[-1] - ts[0]
    t_norm = (ts - ts[0]) / dt * (num_bins - 1)
    flat = np.ravel_multi_index((ys, xs), (h, w))
    bins = []
    for b in range(num_bins):
        wt = ps * np.maximum(0.0, 1.0 - np.abs(t_norm - b))
        bins.append(np.bincount(flat, weights=wt, minlength=h * w).reshape(h, w))
    return np.stack(bins)


def make_voxel(evs, num_bins, h, w, interpolate_bins):
    """data/testh5.py:60-90 (== scripts/visualize_esim_sample.py:113-135)."""
    voxel = np.zeros((num_bins, h, w))
    ts, xs, ys, ps = evs
    if ts.shape[0] == 0:
        return voxel
    ps = ps.astype(np.int8) * 2 - 1                                            # {0,1} -> {-1,+1}
    ts = ((ts - ts[0]) * 1e6).astype(np.int64)
    if not interpolate_bins:
        t_per_bin = (ts[-1] + 0.001) / num_bins
        bin_idx = np.floor(ts / t_per_bin).astype(np.uint8)
        np.add.at(voxel, (bin_idx, ys, xs), ps)
    else:
        dt = ts[-1] - ts[0]
        t_norm = (ts - ts[0]) / (dt + 0.0001) * (num_bins - 1)
        for b in range(num_bins):
            np.add.at(voxel, (b, ys, xs), np.maximum(0, 1.0 - np.abs(t_norm - b)) * ps)
    return voxel


# ----------------------------------------------------------------------------------------
# Synthetic inputs; shared by tests and bench so CPU and GPU see the same clips
# ----------------------------------------------------------------------------------------

def synth_clip_s1(n=8, h=128, w=128, seed=1234, dtype=np.uint8):
    """S1: smooth random-walk video, integer-valued 0..255."""
    g = np.random.default_rng(seed)
    base = g.uniform(0, 255, size=(h, w))
    frames = []
    for _ in range(n):
        base = np.clip(base + g.normal(0, 12, size=(h, w)), 0, 255)
        frames.append(base.astype(np.uint8))
    return np.stack(frames).astype(dtype)


# ----------------------------------------------------------------------------------------
# v2e-derived DVS model (reference: data/v2v_core_v2e.py -- "deprecated" per its own header :1, BASELINE config 3)
# ----------------------------------------------------------------------------------------

V2E_MODELS = ("pn_related", "spatial_independent", "spatial_temporal_independent")


class V2ERng:
    """Random source of the v2e model.  Default: the global legacy np.random stream, like the reference
    (seeded by np.random.seed(seed) in its constructor, v2v_core_v2e.py:312-314)."""
    normal = staticmethod(lambda loc, scale, shape: np.random.normal(loc=loc, scale=scale, size=shape))
    randn = staticmethod(lambda h, w: np.random.randn(h, w))
    poisson = staticmethod(lambda lam: np.random.poisson(lam))


def v2e_video_to_voxel(video, FPS, threshold_model, thres_mean_mean, thres_mean_std, thres_diff_mean, thres_diff_std,
                       cutoff_hz, leak_rate_hz, refractory_period_s, shot_noise_rate_hz, leak_jitter_fraction,
                       noise_rate_cov_decades, seed=None, rng=V2ERng, use_lut=False, record=None):
    """video_to_voxel of v2v_core_v2e.py:556-581 around EventEmulator.generate_events (:401-553).

    Array dtypes are left to NumPy exactly as in the reference (they depend on the input dtype, on cutoff_hz and on
    leak_rate_hz: SURVEY §4).  Deviations from the literal reference, both forced by reference defects:
      * refractory_period_s > 0: the reference raises TypeError (np.clip without a_min, :534-537); the intended
        semantics min(count, int(dt/refractory)) is implemented.
      * threshold_model 'spatial_independent_temporal_changing' crashes in the reference (:423-426): rejected.
    `record` (dict) receives the random fields in draw order, for replay on the GPU."""
    if threshold_model not in V2E_MODELS:
        raise ValueError(f"unsupported threshold_model {threshold_model!r}")
    if seed is not None and rng is V2ERng:
        np.random.seed(seed)                                                   # :312-314
    pos_nominal = thres_mean_mean + thres_diff_mean / 2                        # :294-295
    neg_nominal = thres_mean_mean - thres_diff_mean / 2
    n, h, w = video.shape
    rec = record if record is not None else {}
    for key in ("pos_thres", "neg_thres", "leak_randn", "shot_pos", "shot_neg"):
        rec[key] = []

    def clip_thres(pt, nt):                                                    # change_pos_neg_thres :392-399
        pt = np.clip(pt, a_min=0.01, a_max=None)
        nt = np.clip(nt, a_min=0.01, a_max=None)
        return pt, nt, np.divide(pos_nominal, pt), np.divide(neg_nominal, nt)

    lp = base = pos_thres = neg_thres = pos_pre = neg_pre = noise_rate = None
    t_prev = 0.0
    out = []
    for i in range(n):
        frame = video[i]
        t_frame = i / FPS
        if threshold_model == "spatial_temporal_independent":                  # :417-421 (also runs on frame 0, then _init redraws)
            pt = rng.normal(thres_mean_mean, thres_mean_std, frame.shape)
            nt = rng.normal(thres_mean_mean, thres_mean_std, frame.shape)
            pos_thres, neg_thres, pos_pre, neg_pre = clip_thres(pt, nt)
        delta_time = t_frame - t_prev                                          # :440 (t_prev stays 0 through frame 1)
        log_new = (load_luts()["v2e32"][frame.astype(np.int64)] if use_lut else v2e_linlog_direct(frame))   # :445
        inten01 = None
        if cutoff_hz > 0 or shot_noise_rate_hz > 0:
            inten01 = (frame + 20) / 275.                                      # :184-190,455 (uint8 input wraps, as in the reference)
        if base is None:
            lp = log_new                                                       # :465
        if cutoff_hz > 0:                                                      # low_pass_filter :139-182
            tau = 1 / (math.pi * 2 * cutoff_hz)
            eps = inten01 * (delta_time / tau)
            eps = np.clip(eps, a_min=None, a_max=1)
            lp = (1 - eps) * lp + eps * log_new
        else:
            lp = log_new
        if base is None:                                                       # _init :317-349
            if threshold_model == "pn_related":
                pn_mean = rng.normal(thres_mean_mean, thres_mean_std, frame.shape)
                pn_diff = rng.normal(thres_diff_mean, thres_diff_std, frame.shape)
                pt, nt = pn_mean + (pn_diff / 2), pn_mean - (pn_diff / 2)
            else:
                pt = rng.normal(thres_mean_mean, thres_mean_std, frame.shape)
                nt = rng.normal(thres_mean_mean, thres_mean_std, frame.shape)
            pos_thres, neg_thres, pos_pre, neg_pre = clip_thres(pt, nt)
            noise_rate = rng.randn(h, w).astype(np.float32)
            noise_rate = np.exp(math.log(10) * noise_rate_cov_decades * noise_rate)
            rec["noise_rate"] = noise_rate
            rec["pos_thres"].append(pos_thres)
            rec["neg_thres"].append(neg_thres)
            base = lp                                                          # :476 (aliases lp; lp is rebound every frame)
            base = base.copy()
            continue
        if threshold_model == "spatial_temporal_independent":
            rec["pos_thres"].append(pos_thres)
            rec["neg_thres"].append(neg_thres)
        if leak_rate_hz > 0:                                                   # subtract_leak_current :192-211
            g = rng.randn(h, w)
            rec["leak_randn"].append(g)
            curr = leak_rate_hz * noise_rate * (1 - leak_jitter_fraction * g)
            base = base - delta_time * curr * pos_thres
        diff = lp - base                                                       # :502
        pos_frame = np.clip(diff, a_min=0, a_max=None)                         # compute_event_map :42-62
        neg_frame = np.clip(-diff, a_min=0, a_max=None)
        pos_ev = np.floor_divide(pos_frame, pos_thres)
        neg_ev = np.floor_divide(neg_frame, neg_thres)
        if shot_noise_rate_hz > 0:                                             # generate_shot_noise :65-105
            inten_factor = 1 - (1 - 0.25) * inten01
            pos_factor = inten_factor * pos_pre
            pos_pix = pos_factor / np.mean(pos_factor)
            neg_factor = inten_factor * neg_pre
            neg_pix = neg_factor / np.mean(neg_factor)
            f = (shot_noise_rate_hz / 2) * delta_time
            sp = rng.poisson(pos_pix * f)
            sn = rng.poisson(neg_pix * f)
            rec.setdefault("shot_lambda_pos", []).append(pos_pix * f)
            rec.setdefault("shot_lambda_neg", []).append(neg_pix * f)
        else:
            sp = np.zeros_like(pos_ev)
            sn = np.zeros_like(neg_ev)
        rec["shot_pos"].append(sp)
        rec["shot_neg"].append(sn)
        fpos = pos_ev + sp
        fneg = neg_ev + sn
        if refractory_period_s > 0:                                            # intended semantics of :534-537
            cap = int(delta_time / refractory_period_s)
            fpos = np.minimum(fpos, cap)
            fneg = np.minimum(fneg, cap)
        base += fpos * pos_thres                                               # :547 (in place: keeps base's dtype)
        base -= fneg * neg_thres                                               # :548
        t_prev = t_frame
        out.append(fpos - fneg)
    return np.array(out)
