/*
 * v2v_hip.h -- C ABI of libv2v_hip.so: the MI355X (gfx950) video -> event-count -> voxel-grid hot path.
 *
 * This is the drop-in boundary for the one accelerated path of HYLZ-2019/V2V (reference paths below are
 * relative to the reference checkout).  Every entry point replaces a Python/NumPy function of the
 * reference; a reference maintainer binds it with a ctypes stub (INTEGRATION.md shows each one).
 *
 * Conventions
 *   - All buffer arguments are DEVICE pointers owned by the caller (PyTorch tensors on the host side).
 *     The library never allocates, frees or retains caller-visible memory.
 *   - Every call is asynchronous on the caller's HIP stream (`stream` is a hipStream_t passed as void*;
 *     NULL = the default stream) and never synchronises.
 *   - Return value: 0 (V2V_OK) or a negative v2v_status.  No exceptions, no aborts.  A human-readable
 *     message for the last failure on the calling thread is available from v2v_last_error().
 *   - No global mutable state except the log-intensity tables (v2v_lut_set) and the per-thread error slot.
 *   - There is NO CPU implementation behind this ABI.  Without a GPU every compute entry point
 *     returns V2V_ERR_HIP.
 */
#ifndef V2V_HIP_H
#define V2V_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: every device-native (V2V_RNG_PHILOX) random stream differs from ABI 1 for the same (seed, clip id): Gaussians by
 *    table inversion (was Box-Muller), Philox4x32-7 for per-step fields, two time steps per Gaussian field id; the v2e
 *    shot-noise sampler and its fields changed again in round 3; v2v_v2e_workspace_bytes and v2v_postops_workspace_bytes
 *    grew; v2v_lut_set acts on the current device only (a single-process multi-GPU host calls it once per device).  The
 *    front-end's `gray_first` was RE-MAPPED: 1 now selects OpenCV 4.x's 15-bit BGR2GRAY weights (3735 / 19235 / 9798, >> 15);
 *    the 14-bit OpenCV 2.x/3.x form that ABI 1 ran for gray_first = 1 moved to gray_first = 2 -- an ABI-1 caller passing 1 gets
 *    different gray values on some colours, with no error.  A client must re-query workspace sizes and cannot replay ABI-1
 *    native noise.  Replay-mode (V2V_RNG_REPLAY) results are unchanged.
 * 3: v2v_esim_extras grew two trailing fields (stored_frames, frames_elems: the bounds of the frame_index gather).  Every function
 *    signature and every result of ABI 2 is unchanged; a caller that fills v2v_esim_extras must be recompiled (or zero the struct at
 *    its new size): an ABI-2 struct is 16 bytes shorter. */
#define V2V_ABI_VERSION 3

typedef enum v2v_status {
    V2V_OK = 0,
    V2V_ERR_NULL = -1,   /* a required pointer is NULL                                              */
    V2V_ERR_SHAPE = -2,  /* B,N,H,W out of range, N < 2, strides smaller than the extent            */
    V2V_ERR_BINS = -3,   /* SUM: (N-1) % (num_bins*frames_per_bin) != 0  (v2v_datasets.py:365)      */
    V2V_ERR_DTYPE = -4,  /* unsupported in/out dtype                                                */
    V2V_ERR_MODE = -5,   /* unknown rng / bin mode, or replay fields missing in replay mode         */
    V2V_ERR_ALIGN = -6,  /* pointer not aligned for its element type                                */
    V2V_ERR_HIP = -7,    /* HIP runtime error (launch failed, no device); see v2v_last_error()      */
    V2V_ERR_PARAM = -8   /* invalid scalar parameter                                                */
} v2v_status;

typedef enum v2v_dtype { V2V_U8 = 0, V2V_F32 = 1, V2V_F64 = 2, V2V_BF16 = 3 /* consumer-side entry points only */ } v2v_dtype;

/* Where the simulator's random fields come from (reference: global np.random, v2v_core_esim.py:29,37,38,44) */
typedef enum v2v_rng_mode {
    V2V_RNG_NONE = 0,   /* deterministic debug mode: u_init = 0.5, no hot pixels, all Gaussians 0      */
    V2V_RNG_PHILOX = 1, /* device-native counter RNG keyed by (seed, clip_id, field, pixel): results   */
                        /* do not depend on batch size or on how the batch is sharded over GPUs        */
    V2V_RNG_REPLAY = 2, /* caller supplies the fields (e.g. drawn from np.random in the reference's    */
                        /* order): bit-exact replay of a reference run                                 */
    V2V_RNG_PHILOX_FAST = 3 /* accepted as an alias of V2V_RNG_PHILOX.  The round-1 variant on the hardware         */
                        /* transcendental units (not reproducible on a CPU) is gone: the exact generator is as  */
                        /* fast.  Gaussians are drawn by TABLE INVERSION: each 16-bit half n of a Philox4x32    */
                        /* word is one deviate, sign = n >> 15, magnitude = table[(n >> 2) & 0x1FFF] = float32  */
                        /* Phi^-1(1/2 + (i + 1/2) / 2^14) (8192 entries, 2^14 distinct values, |g| <= 4.009:    */
                        /* tails end at 4 sigma; variance 0.99992).  Philox4x32-7 for per-time-step fields, -10 */
                        /* for per-clip fields.  ABI 2 changed every native stream (see V2V_ABI_VERSION).       */
} v2v_rng_mode;

typedef enum v2v_bin_mode {
    V2V_BIN_SUM = 0,      /* v2v_datasets.py:399-400: out[B,L,Tb,H,W], each bin = sum of frames_per_bin pairs.  */
                          /* num_bins = N-1, frames_per_bin = 1 gives EventEmulator.video_to_voxel's [N-1,H,W]. */
    V2V_BIN_BILINEAR = 1  /* utils/event_utils.py:692-728 weights on pseudo-events at ts = 0..K-1: out[B,Tb,H,W] */
} v2v_bin_mode;

#define V2V_FLAG_NOISE_EXTERNAL 0x1u /* put_noise_external (v2v_core_esim.py:46,62-65) */
#define V2V_FLAG_NO_NOISE 0x2u       /* caller guarantees base_noise_std == 0 and hot_pixel_fraction == 0 for every  */
                                     /* clip: selects the kernel without the noise adds (same results, fewer ops)  */
#define V2V_FLAG_SYMMETRIC 0x4u      /* caller guarantees pos_thres == neg_thres for every clip (EventEmulator's own */
                                     /* defaults): instances without the asymmetric loop at 4 waves per SIMD; same   */
                                     /* results; a clip that breaks the guarantee comes out as NaN.  A hint: ignored */
                                     /* where no such instance exists (float64 output, replay, external noise ...)   */
#define V2V_FLAG_MAP_4PX 0x8u        /* work-item mapping: by default 4 pixels per work-item when the layout allows  */
#define V2V_FLAG_MAP_1PX 0x10u       /* it AND the batch gives more than one such wave per SIMD, else 2 pixels, and 1 */
#define V2V_FLAG_MAP_2PX 0x20u       /* pixel below half a wave per SIMD; these three pin the choice (4PX / 2PX still */
                                     /* need the aligned layout).  Results do not depend on the mapping; tests use   */
                                     /* them to cover all three families of instances                                */

/* Replay fields, all float64 device arrays, one set per clip (clip-major). */
typedef struct v2v_esim_replay {
    const double *u_init; /* [B,H*W]     rand  #1: potential init        (v2v_core_esim.py:29) */
    const double *u_hot;  /* [B,H*W]     rand  #2: hot-pixel mask        (:37)                 */
    const double *g_hot;  /* [B,H*W]     randn #1: hot-pixel noise       (:38)                 */
    const double *g_base; /* [B,N-1,H*W] randn per frame pair            (:44)                 */
} v2v_esim_replay;

/* ---- library info / errors ---------------------------------------------------------------------- */
int v2v_version(void);                 /* V2V_ABI_VERSION the library was built with */
const char *v2v_last_error(void);      /* message of the last failure on this thread ("" if none) */
int v2v_device_count(void);            /* number of visible HIP devices, 0 if none (never fails) */

/* ---- log-intensity tables (replaces reverse_gamma_correction + np.log, v2v_core_esim.py:3-4,33-34) --
 * lut64[v] / lut32[v] = the reference's log image value for integer intensity v computed in float64 /
 * float32.  The library ships NumPy's values (golden G1); set lets a deployment re-pin them.
 * `which`: 0 = ESIM float64, 1 = ESIM float32, 2 = v2e lin_log float32.  Host pointers, 256 entries. */
int v2v_lut_get(int which, void *dst_host);
int v2v_lut_set(int which, const void *src_host);     /* the CURRENT device's table; call once per device a process drives */

/* ---- fused ESIM simulator + voxel binning ---------------------------------------------------------
 * Replaces EventEmulator.video_to_voxel (data/v2v_core_esim.py:26-69) fused with the binning of
 * WebvidDatasetV2.imgs_to_voxels (data/v2v_datasets.py:399-400) or the temporal-bilinear binning of
 * utils/event_utils.py:events_to_voxel (:692-728), for a whole batch of clips in one launch.
 *
 * frames        [B,N,H,W] grayscale (HWC with C=1), dtype V2V_U8 or V2V_F32; rows contiguous;
 *               clip_stride / frame_stride in ELEMENTS (>= N*H*W / >= H*W).
 * params        device float64: {pos_thres, neg_thres, base_noise_std, hot_pixel_fraction, hot_pixel_std}
 *               per clip; params_stride = 5 for [B,5], 0 to broadcast one set.  The table lives on the device, so it is checked
 *               there: a clip whose thresholds are outside [1e-9, 1e30] (the exact floor-divide needs |potential| / C < 2^40;
 *               zero, negative, NaN have no meaning in the reference either) or whose noise parameters are negative or not finite
 *               comes out as NaN planes (+ the flag word of `stats`); the other clips of the call are unaffected.
 * seed,clip_id0 Philox key and the GLOBAL id of clip 0 of this call (clip b uses clip_id0 + b).
 * replay        required iff rng_mode == V2V_RNG_REPLAY.
 * out_voxel     SUM: [B,L,Tb,H,W], L = (N-1)/(Tb*fpb); BILINEAR: [B,Tb,H,W]; dtype V2V_F32 or V2V_F64.
 * out_counts    optional int64 [B,2]: total ON / OFF events per clip (ADDED to the buffer: zero it first).
 */
int v2v_esim_voxel_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                       int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                       uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0,
                       const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                       void *out_voxel, int out_dtype, int64_t *out_counts, void *stream);

/* Same, with an optional per-clip RNG key: clip_keys = device uint64 [B,2] = {seed, clip id} of every clip (NULL ->
 * seed / clip_id0 + b).  Lets a collate step simulate independently-seeded samples in ONE launch with exactly the
 * result each sample gets when simulated alone (v2v_amd.datasets.SimulatingCollator). */
int v2v_esim_voxel_keyed_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                             int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                             uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0, const uint64_t *clip_keys,
                             const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                             void *out_voxel, int out_dtype, int64_t *out_counts, void *stream);

/* Same, writing the voxel planes into a PADDED layout: rows of out_row_pitch >= W elements, planes of out_plane_size >=
 * out_row_pitch*(H-1)+W elements -- e.g. the H,W -> multiples-of-16 padding of forward_sequence (model/train_utils.py:322-326)
 * written by the simulator itself instead of a later copy.  Only the H x W interior of every plane is written: the caller
 * zeroes the padding once (it is never touched again). */
int v2v_esim_voxel_padded_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                              int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                              uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0, const uint64_t *clip_keys,
                              const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                              void *out_voxel, int out_dtype, int64_t out_row_pitch, int64_t out_plane_size, int64_t *out_counts, void *stream);

/* Algorithmic HBM bytes of one v2v_esim_voxel_hip call (input read once + output written once);
 * the figure bench.py's roofline is computed from.  Returns a negative v2v_status on bad arguments. */
int64_t v2v_esim_voxel_bytes(int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W, int bin_mode,
                             int num_bins, int frames_per_bin, int out_dtype);

/* ---- synthetic clips (bench / tests input) --------------------------------------
 * Fills frames[B,N,H,W] (V2V_U8 or V2V_F32, integer-valued 0..255) with a per-clip low-frequency pattern
 * translating at a per-clip velocity in [-3,3] px/frame plus N(0,4) pixel noise; clip b is a pure function
 * of (seed, clip_id0 + b), independent of B. */
int v2v_synth_clips_hip(void *frames, int dtype, int64_t B, int64_t N, int64_t H, int64_t W, uint64_t seed,
                        uint64_t clip_id0, void *stream);

/* Segmented form: the voxel grids of consecutive image intervals in ONE launch -- what TestH5Dataset.__getitem__
 * (data/testh5.py:111-119) does with one make_voxel call per image.  seg_offsets = device int64 [n_segments+1]
 * ascending event offsets (the h5 `event_idx` attributes); grid f is built from events [seg[f], seg[f+1]) with that
 * interval's own first/last timestamp; out_voxel float64 [n_segments,num_bins,H,W]. */
int v2v_events_to_voxel_segmented_hip(const double *ts, const int64_t *xs, const int64_t *ys, const double *ps, int64_t n,
                                      const int64_t *seg_offsets, int64_t n_segments, int mode, int num_bins, int64_t H, int64_t W,
                                      double *out_voxel, uint64_t *dropped, void *stream);

/* ---- v2e-derived DVS model fused with voxel binning (BASELINE config 3) -----------------------------------
 * Replaces video_to_voxel (data/v2v_core_v2e.py:556-581) around EventEmulator.generate_events (:401-553): lin-log
 * table, intensity-dependent IIR low-pass, leak current, per-pixel random ON/OFF thresholds, Poisson shot noise,
 * refractory cap (intended semantics min(count, int(dt/refractory)); the reference's own line raises TypeError).
 * `params` is a HOST struct (the arguments of the reference's video_to_voxel, same for every clip of the call).
 * uint8_wrap = 1 reproduces the reference's uint8 wrap in (frame+20)/275 for uint8 input (v2v_core_v2e.py:184-190).
 * rng_mode: V2V_RNG_PHILOX (device-native fields; needs `workspace` of v2v_v2e_workspace_bytes() when
 * shot_noise_rate_hz > 0) or V2V_RNG_REPLAY (fields drawn by NumPy in the reference's order: bit-exact replay).
 * The native shot-noise sampler inverts the Poisson law up to 64 events per pixel, frame and polarity: shot_noise_rate_hz <= 32 fps
 * (16 expected events; the reference's default is 0.1), V2V_ERR_PARAM beyond -- REPLAY takes np.random.poisson's counts at any rate.
 * Output / binning / counts exactly as v2v_esim_voxel_hip. */
typedef enum v2v_v2e_threshold_model {
    V2V_V2E_PN_RELATED = 0,                  /* "pn_related"                    */
    V2V_V2E_SPATIAL_INDEPENDENT = 1,          /* "spatial_independent"           */
    V2V_V2E_SPATIAL_TEMPORAL_INDEPENDENT = 2  /* "spatial_temporal_independent"  */
} v2v_v2e_threshold_model;
typedef struct v2v_v2e_params {
    double fps;
    int threshold_model;
    double thres_mean_mean, thres_mean_std, thres_diff_mean, thres_diff_std;
    double cutoff_hz, leak_rate_hz, refractory_period_s, shot_noise_rate_hz, leak_jitter_fraction, noise_rate_cov_decades;
    int uint8_wrap;
} v2v_v2e_params;
typedef struct v2v_v2e_replay {   /* device pointers */
    const double *pos_thres, *neg_thres; /* clipped thresholds: [B,H*W] (stride 0) or [B,N-1,H*W] (temporal model)      */
    int64_t thres_frame_stride;          /* 0, or H*W for spatial_temporal_independent                                  */
    const float *noise_rate;             /* [B,H*W]  exp(ln10*cov*randn) as NumPy computed it (float32)                 */
    const double *leak_randn;            /* [B,N-1,H*W] or NULL when leak_rate_hz == 0                                  */
    const int64_t *shot_pos, *shot_neg;  /* [B,N-1,H*W] Poisson counts or NULL when shot_noise_rate_hz == 0             */
} v2v_v2e_replay;
int64_t v2v_v2e_workspace_bytes(int64_t B, int64_t N);
int v2v_v2e_voxel_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W, int64_t clip_stride,
                      int64_t frame_stride, const v2v_v2e_params *params, int rng_mode, uint64_t seed, uint64_t clip_id0,
                      const v2v_v2e_replay *replay, int bin_mode, int num_bins, int frames_per_bin, void *out_voxel,
                      int out_dtype, int64_t *out_counts, void *workspace, void *stream);

/* ---- event list -> voxel grid -------------------------------------------------------------------------
 * Replaces TestH5Dataset.make_voxel (data/testh5.py:60-90; same function at scripts/visualize_esim_sample.py:
 * 113-135) and events_to_voxel (utils/event_utils.py:692-728, temporal_bilinear=True).
 * ts float64 seconds (ascending), xs/ys int64 pixel coordinates, ps float64 (make_voxel modes: 0/1 polarity
 * flags, mapped to -1/+1 like the reference; BILINEAR mode: signed weights used as they are).
 * out_voxel float64 [num_bins,H,W] is zeroed by the call (an empty list gives zeros, testh5.py:63-64).
 * dropped   uint64 device word: number of events outside the sensor / bin range (NumPy would raise IndexError).
 * Interpolated sums use float64 atomics: per-event terms are NumPy's bits, summation order is not. */
typedef enum v2v_event_mode {
    V2V_EV_MAKE_VOXEL_DISCRETE = 0, /* testh5.py:71-74  */
    V2V_EV_MAKE_VOXEL_INTERP = 1,   /* testh5.py:75-81  */
    V2V_EV_BILINEAR = 2             /* event_utils.py:713-727 */
} v2v_event_mode;
int v2v_events_to_voxel_hip(const double *ts, const int64_t *xs, const int64_t *ys, const double *ps, int64_t n, int mode,
                            int num_bins, int64_t H, int64_t W, double *out_voxel, uint64_t *dropped, void *stream);

/* ---- decode-side front-end ("next" row) ----------------------------------------------------
 * Replaces the per-frame host loop of WebvidDatasetV2.read_video (data/v2v_datasets.py:191-224: [cvtColor BGR2GRAY]
 * -> crop -> cv2.resize INTER_LINEAR -> [flip] -> shake crop) and the pause-index gather + gray extraction of
 * __getitem__ (:311-316) for frames that are already decoded and on the device.
 * src [T,Hs,Ws,Cs] uint8 (Cs = 3 BGR or 1); frame_idx [N] int32 = decoded frame shown by each simulator frame;
 * shake_di/dj [T] int32 (>= 0) or NULL; out_gray [N,crop,crop] uint8 (simulator input); out_imgs [N,crop,crop,C]
 * uint8 or NULL.  gray_first: color_mode 'gray' (cvtColor BGR2GRAY before the resize) = 1 with the 15-bit weights of
 * OpenCV >= 4.0 (3735/19235/9798, +2^14, >> 15: what an unpinned opencv-python installs), 2 with the 14-bit weights of
 * OpenCV 2.x/3.x (1868/9617/4899, +2^13, >> 14); 0 with Cs == 3: 'gray_in_bgr_out' (resize BGR, gray = bgr_to_gray,
 * v2v_datasets.py:19-22).
 * PARITY UNPINNED against OpenCV (absent here): bit-exact against the restatement of the OpenCV 4.x 8-bit algorithms in
 * oracle/frontend_oracle.py. */
int v2v_frontend_hip(const uint8_t *src, int64_t T, int64_t Hs, int64_t Ws, int64_t Cs, int64_t min_i, int64_t min_j,
                     int64_t crop_before, int64_t need_h, int64_t need_w, int64_t crop, int flip, int gray_first,
                     const int32_t *frame_idx, int64_t N, const int32_t *shake_di, const int32_t *shake_dj, uint8_t *out_imgs,
                     uint8_t *out_gray, void *stream);

/* Batch form: src [B,T,Hs,Ws,Cs], frame_idx [B,N], clip_table device int32 [B,4] = {min_i, min_j, crop_before, flip}
 * per clip (no shake: the resize target is crop x crop); outputs [B,N,crop,crop(,C)].  One launch for the batch.
 * max_crop_before: upper bound of the table's crop_before column if the host knows it (sizes the LDS tile of the
 * gray-only path; clips above it still produce the same bytes through a slower path), 0 = bounded by the frame only.
 * The tables live on the device and are not validated by this call: whatever they hold is CLAMPED into bounds on the device (frame
 * numbers into [0, T), crop_before into [1, min(Hs, Ws)], the corner so that the rectangle lies inside the frame) -- a clamped crop,
 * never an out-of-bounds read; the Python wrappers validate host tables and raise.
 * `src` may also be DEVICE-ACCESSIBLE PAGE-LOCKED HOST memory (hipHostMalloc / torch pin_memory: mapped, same address on the device):
 * the kernel's staging loads then read each clip's crop rectangle straight over PCIe -- no copy of the whole frames (BASELINE config 4's
 * stream: 69 MB instead of 885 MB per 8 x 40 frames of 720p).  Every other pointer is device memory. */
int v2v_frontend_batch_hip(const uint8_t *src, int64_t B, int64_t T, int64_t Hs, int64_t Ws, int64_t Cs, const int32_t *clip_table,
                           int64_t max_crop_before, int64_t crop, int gray_first, const int32_t *frame_idx, int64_t N,
                           uint8_t *out_imgs, uint8_t *out_gray, void *stream);

/* ---- voxel post-ops of the consumer ("next" row) ------------------------------------------------
 * Replaces normalize_batch_voxel (model/train_utils.py:147-166: per-sample torch.kthvalue at int(0.01*M) / int(0.99*M),
 * clamp(min=1), where(v > 0, v/pos_max, v/neg_max)) fused with the zero padding of H,W to multiples of `pad_to`
 * (model/train_utils.py:322-326).  voxel float32 [B,planes,H,W] (planes = T*C); out float32 [B,planes,Hp,Wp].
 * The k-th values are exact (3-pass radix select); workspace of v2v_postops_workspace_bytes(B) needed iff normalize.
 * B <= 65535 samples per call for every post-op entry point (one grid row per sample; V2V_ERR_SHAPE beyond). */
int64_t v2v_postops_workspace_bytes(int64_t B);
int v2v_normalize_pad_hip(const float *voxel, int64_t B, int64_t planes, int64_t H, int64_t W, int normalize, int pad_to,
                          float *out, void *workspace, void *stream);

/* Same with (a) an input that is already padded -- planes of H_in x W_in >= H x W, pads zero, as v2v_esim_voxel_padded_hip
 * writes them; out may then alias voxel (in-place normalise) -- and (b) a choice of the k-th value method:
 *   V2V_NORM_NONE   pad only
 *   V2V_NORM_RADIX  exact 3-pass radix select, any float32 content (unpadded input only): 3 + 1 reads, 1 write
 *   V2V_NORM_COUNT  exact counting select for INTEGER-valued voxels with |v| <= 255 (the SUM-mode grids V2V trains on, without
 *                   external noise): 1 + 1 reads, 1 write; a sample holding any other value comes out as NaN (loud, no fallback) */
typedef enum v2v_norm_method { V2V_NORM_NONE = 0, V2V_NORM_RADIX = 1, V2V_NORM_COUNT = 2 } v2v_norm_method;
int v2v_normalize_pad_ex_hip(const float *voxel, int64_t B, int64_t planes, int64_t H, int64_t W, int64_t H_in, int64_t W_in, int method,
                             int pad_to, float *out, void *workspace, void *stream);

/* float32 twin: replaces events_to_voxel_torch (utils/event_utils.py:466-507; only caller data/dataset.py:328).
 * ts/ps float32, out float32 [num_bins,H,W]; discrete != 0 selects the `temporal_bilinear=False` branch (:502-505). */
int v2v_events_to_voxel_f32_hip(const float *ts, const int64_t *xs, const int64_t *ys, const float *ps, int64_t n, int discrete,
                                int num_bins, int64_t H, int64_t W, float *out_voxel, uint64_t *dropped, void *stream);

/* Segmented form of the float32 twin: every "between frames" voxel grid of a Monash-format sequence in ONE launch.
 * Replaces the per-index loop of scripts/esim_to_voxel.py:29-36 over DynamicH5Dataset.__getitem__ (data/dataset.py:176-199,
 * 419-427): events [seg[f], seg[f+1]) form grid f; ts float64 as stored in the file -- (ts - ts[seg[f]]).astype(float32) is
 * applied per interval on the device (:194); ps float32 already in {-1,+1} (:385); intervals with fewer than `min_events`
 * events stay zero (:189-190 uses 3).  out_voxel float32 [n_segments, num_bins, H, W]. */
int v2v_events_to_voxel_f32_segmented_hip(const double *ts, const int64_t *xs, const int64_t *ys, const float *ps, int64_t n,
                                          const int64_t *seg_offsets, int64_t n_segments, int min_events, int discrete, int num_bins,
                                          int64_t H, int64_t W, float *out_voxel, uint64_t *dropped, void *stream);

/* ---- consumer side, SURVEY §8f rank 4: the recurrent encoders' ConvLSTM step as one matrix-core kernel --------------------
 * Replaces ConvLSTM.forward (model/submodules.py:179-235): Gates = Conv2d(2C -> 4C, 3x3, pad 1) over cat(x, h_prev), chunk
 * into in/remember/out/cell gates (:218), sigmoid x3 + tanh (:221-226), cell = remember*c_prev + in*cell_gate (:229),
 * hidden = out*tanh(cell) (:230).  bf16 operands, fp32 accumulation and fp32 cell state (a precision choice of THIS kernel:
 * the tests state the tolerance against the fp32 module).  Activations are NHWC: x, h_prev, h_state [B,H,W,C] bf16;
 * c_prev, c_state [B,H,W,C] fp32; h_nchw (optional) the same hidden state as [B,C,H,W] in h_nchw_dtype (V2V_F32 or V2V_BF16: the
 * dtype the module's input had, :202-203) for the stock layers downstream.
 * h_prev == NULL and c_prev == NULL mean the zero state (prev_state=None, :196-209).  c_state may alias c_prev; h_state must
 * not alias h_prev.  tile_rows: pixels per workgroup tile, 64, 128 or 256; 0 = pick by image size.
 * Requirements (else V2V_ERR_SHAPE): C % 64 == 0, (H*W) % 4 == 0.  Any B*H*W: the last pixel tile of a launch may be partial (its rows
 * past the end read zeros and are not stored), so real-data frame sizes (180x240, 260x346 padded to multiples of 16) run at batch 1. */
int v2v_convlstm_packed_bytes(int64_t C, uint64_t *bytes);   /* size of the packed weight buffer: 4C*2C*9 bf16 */
/* gates_weight: the module's Gates.weight, fp32 [4C, 2C, 3, 3] on the device -> packed bf16 (once per weight update) */
int v2v_convlstm_pack_weights_hip(const float *gates_weight, int64_t C, void *packed, void *stream);
int v2v_convlstm_step_hip(const void *x, const void *h_prev, const float *c_prev, const void *packed, const float *gates_bias,
                          int64_t B, int64_t H, int64_t W, int64_t C, void *h_state, float *c_state, void *h_nchw, int h_nchw_dtype, int tile_rows,
                          void *stream);
/* The residual blocks of the same encoder (ResidualBlock.forward, model/submodules.py:143-177, norm=None as E2VID instantiates
 * it at model/unet.py:48): a 3x3, stride-1, pad-1 convolution on the matrix cores with the same tiles and pipeline as the
 * ConvLSTM step -- out = [relu]( conv(x) + bias [+ residual] ), x [B,H,W,Cin] / residual, out [B,H,W,Cout] bf16 NHWC,
 * weight = the module's conv weight fp32 [Cout,Cin,3,3] packed once by v2v_conv3x3_pack_weights_hip (Cout*Cin*9 bf16).
 * Two calls make a block: conv1 with relu, conv2 with residual = the block's input and relu.
 * Requirements (else V2V_ERR_SHAPE): Cin % 64 == 0, Cout % 256 == 0 (or 128 / 64 / 32), (H*W) % 4 == 0 (any B*H*W: a partial last
 * tile); tile_rows: 32, 64, 128 or 256 pixels per workgroup, the first two for Cout % 256 == 0 only; 0 = the largest tile that
 * still fills the CUs;
 * out must not alias x (neighbouring tiles read x); it may alias residual. */
int v2v_conv3x3_pack_weights_hip(const float *weight, int64_t Cin, int64_t Cout, void *packed, void *stream);   /* = conv_pack, ks 3 */
int v2v_conv3x3_nhwc_hip(const void *x, const void *packed, const float *bias, const void *residual, int relu, int64_t B, int64_t H,
                         int64_t W, int64_t Cin, int64_t Cout, void *out, int tile_rows, void *stream);

/* The general form: the encoder / decoder convolutions around those blocks (ConvLayer, model/submodules.py:10-50 as built at
 * model/unet.py:34-60, 83-87: 5x5, stride 2 in the encoders, stride 1 after the bilinear upsampling in the decoders, ReLU).
 * ks = 3 or 5 (pad ks/2), stride 1 or 2; x [B,Hin,Win,Cin] -> out [B,Hout,Wout,Cout] with Hout = (Hin-1)/stride + 1; Cin % 64 == 0;
 * Cout a multiple of 256, or 128 / 64 / 32; (Hout*Wout) % 4 == 0, any B*Hout*Wout; weight fp32 [Cout,Cin,ks,ks].
 * Cin = 32 with Cout 64 / 128 (the UNet's first encoder, model/unet.py:34-44 with base 32) packs two taps per 64-wide K chunk.
 * tile_rows 16 = halo tiles (a 16x16 pixel patch staged once per channel chunk with its halo; stride 1, H and W multiples of
 * 16, Cout 32 / 64, or 128 for 3x3): what 0 picks for the 5x5 decoders with 32 / 64 output channels. */
int64_t v2v_conv_packed_elems(int64_t Cin, int64_t Cout, int ks);      /* bf16 elements of the packed stream; -1: shape not taken */
int v2v_conv_pack_weights_hip(const float *weight, int64_t Cin, int64_t Cout, int ks, void *packed, void *stream);
int v2v_conv_nhwc_hip(const void *x, const void *packed, const float *bias, const void *residual, int relu, int64_t B, int64_t Hin,
                      int64_t Win, int64_t Cin, int64_t Cout, int ks, int stride, void *out, int tile_rows, void *stream);

/* The bilinear x2 upsampling in front of a decoder convolution (UpsampleConvLayer.forward, model/submodules.py:86-87:
 * f.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)) with the sum skip connection that feeds it
 * (skip_sum, model/model_util.py:14, applied at model/unet.py:304) folded in: out = up2(x [+ skip]); x / skip [B,H,W,C], out
 * [B,2H,2W,C], all bf16 NHWC; skip may be NULL; the sum is rounded to bf16 before the interpolation, as the stock bf16 add is.
 * C % 8 == 0, 16-byte aligned buffers, out must not alias an input. */
int v2v_upsample2x_nhwc_hip(const void *x, const void *skip, int64_t B, int64_t H, int64_t W, int64_t C, void *out, void *stream);

/* The head of the recurrent UNet: ConvLayer(num_bins, base_num_channels = 32, kernel_size 5, stride 1, padding 2, relu)
 * (model/unet.py:77-78 / :265-266, ConvLayer.forward model/submodules.py:25-33) -- few input channels (<= 8: the voxel bins),
 * 32 output channels.  x8 = the input as bf16 NHWC with the channels padded to 8 (v2v_to_nhwc8_bf16_hip makes it from a float32
 * [B,C,H,W] tensor of any element strides); weight fp32 [32,Cin,ks,ks] packed once (v2v_conv_head_packed_elems(ks) bf16
 * elements); out bf16 [B,H,W,32] = [relu](conv + bias).  ks 3 or 5 (pad ks/2), H and W multiples of 16. */
int64_t v2v_conv_head_packed_elems(int ks);
int v2v_conv_head_pack_weights_hip(const float *weight, int64_t Cin, int ks, void *packed, void *stream);
int v2v_to_nhwc8_bf16_hip(const float *src, int64_t stride_b, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t B, int64_t C,
                          int64_t H, int64_t W, void *dst, void *stream);
/* the same conversion with normalize_batch_voxel's scaling folded into the read (model/train_utils.py:162-166): scales float32 [B,2] =
 * {neg_max, pos_max} (v2v_voxel_scales_hip), x -> where(x > 0, x / pos_max, x / neg_max) before the bf16 rounding; NULL = no scaling */
int v2v_to_nhwc8_bf16_scaled_hip(const float *src, int64_t stride_b, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t B, int64_t C,
                                 int64_t H, int64_t W, const float *scales, void *dst, void *stream);
int v2v_conv_head_nhwc_hip(const void *x8, const void *packed, const float *bias, int relu, int64_t B, int64_t H, int64_t W, int ks, void *out, void *stream);

/* The prediction layer: ConvLayer(base_num_channels, out, kernel_size=1, activation=None) (model/unet.py:58-64) applied to
 * skip_sum(x, head) (model/unet.py:307): out[m][o] = bias[o] + sum_c w[o][c] * (x[m][c] + skip[m][c]); x / skip [M, C] bf16
 * (NHWC with M = B*H*W; skip may be NULL; the sum is rounded to bf16 first, the weights to bf16, as bf16 autocast does), weight
 * fp32 [Cout, C], out [M, Cout] fp32 or bf16 (out_dtype) -- for Cout = 1 that is the NCHW image.  C a power of two in 8..512,
 * Cout 1..3. */
int v2v_conv1x1_nhwc_hip(const void *x, const void *skip, const float *weight, const float *bias, int64_t M, int64_t C, int64_t Cout,
                         void *out, int out_dtype, void *stream);

/* fp32 or bf16 [B,C,H,W] (src_dtype V2V_F32 / V2V_BF16) -> bf16 [B,H,W,C] (relu != 0: through max(x,0), the activation in front of the recurrent block,
 * model/submodules.py:267-271 RecurrentConvLayer = ConvLayer(relu) -> ConvLSTM).  C % 64 == 0 and (H*W) % 64 == 0. */
int v2v_nchw_to_nhwc_bf16_hip(const void *src, int src_dtype, int64_t B, int64_t C, int64_t H, int64_t W, int relu, void *dst, void *stream);

/* normalize_batch_voxel's per-sample k-th values gathered BY THE SIMULATOR'S WRITER (model/train_utils.py:147-166):
 * v2v_esim_voxel_stats_hip = v2v_esim_voxel_padded_hip + `stats`, uint32 [B][V2V_VOXEL_STATS_WORDS] (zeroed here, then accumulated by
 * the launch): word 256 + v counts the voxels of clip b equal to the integer v in -255..255 except v = 0 (left 0: the reader derives
 * it from the element count), words 0 / 512 the voxels below -255 / above 255 (hot pixels), word 513 != 0 flags a clip whose planes
 * are not counts.  SUM mode, float32 grid, no V2V_FLAG_NOISE_EXTERNAL (anything else: V2V_ERR_MODE).
 * v2v_voxel_scales_hip: scales[b] = {neg_max, pos_max} = {clamp(-kthvalue(1 %), min=1), clamp(kthvalue(99 %), min=1)} (:153-160; ranks
 * int(0.01*M), int(0.99*M), 1-based, M = elems_per_sample = the sample's voxel count WITHOUT padding) from those statistics, exact;
 * NaN when a rank falls into an overflow word or the clip is flagged.
 * v2v_voxel_apply_scales_hip: out = where(voxel > 0, voxel / pos_max, voxel / neg_max) (:162-166; IEEE division) [+ zero padding of H, W to
 * multiples of pad_to]; voxel [B,planes,H_in,W_in] (interior H x W valid), out [B,planes,Hp,Wp]; out == voxel allowed when the layouts
 * agree; scales NULL = copy / pad only.  A consumer that scales while it reads (v2v_to_nhwc8_bf16_scaled_hip) needs no such pass. */
#define V2V_VOXEL_STATS_WORDS 516
/* Optional extras of one simulator launch (v2v_esim_voxel_ex_hip = v2v_esim_voxel_padded_hip + this; NULL fields are off):
 *   stats         as v2v_esim_voxel_stats_hip.
 *   frame_index   int32 [B, N]: simulator frame f of clip b is STORED frame frame_index[b*N + f] -- the reference's pause-index gather
 *                 (data/v2v_datasets.py:286-311: all_imgs = stack(raw_imgs[i] for i in img_idxes)) folded into the kernel's loads, so a
 *                 clip whose video pauses is stored (and crosses PCIe) once per DECODED frame; results are those of the gathered clip.
 *   clip_offsets  int64 [B]: clip b starts at element clip_offsets[b] of `frames` (clips of different stored lengths packed back to
 *                 back); `clip_stride` then only states the alignment every offset keeps (in elements, e.g. 16).  Comes with frame_index.
 *   stored_frames int32 [B] (with frame_index; NULL = the caller vouches for the rows): clip b holds stored_frames[b] frames.  The launch
 *                 checks every row while it stages it through LDS: a clip whose row names a frame outside [0, stored_frames[b]) is NOT
 *                 read -- its planes come out NaN and, with `stats`, its flag word (513) is set, so v2v_voxel_scales_hip hands out NaN
 *                 scales for it -- the same loud failure as a broken V2V_FLAG_SYMMETRIC promise; the other clips are unaffected.  (The
 *                 reference's gather, a Python list index at data/v2v_datasets.py:311, raises IndexError there.)
 *   frames_elems  int64 (with stored_frames; 0 = not stated): elements in `frames`.  A clip with clip_offsets[b] < 0 or
 *                 clip_offsets[b] + (stored_frames[b]-1)*frame_stride + H*W > frames_elems is poisoned the same way.
 * Indexed launches: uint8 clips, SUM bins, device-native noise, float32 grid, none of the NO_NOISE / NOISE_EXTERNAL / SYMMETRIC flags
 * (V2V_ERR_MODE otherwise).  The index rows live on the device, so their check cannot return a status code without a synchronisation:
 * it is reported through the data (NaN planes / the flag word), never by reading out of bounds. */
typedef struct v2v_esim_extras {
    uint32_t *stats;
    const int32_t *frame_index;
    const int64_t *clip_offsets;
    const int32_t *stored_frames;
    int64_t frames_elems;
} v2v_esim_extras;
int v2v_esim_voxel_ex_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                          int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                          uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0, const uint64_t *clip_keys,
                          const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                          void *out_voxel, int out_dtype, int64_t out_row_pitch, int64_t out_plane_size, int64_t *out_counts,
                          const v2v_esim_extras *extras, void *stream);
int v2v_esim_voxel_stats_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                             int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                             uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0, const uint64_t *clip_keys,
                             const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                             void *out_voxel, int out_dtype, int64_t out_row_pitch, int64_t out_plane_size, int64_t *out_counts, uint32_t *stats,
                             void *stream);
int v2v_voxel_scales_hip(const uint32_t *stats, int64_t B, int64_t elems_per_sample, float *scales, void *stream);
int v2v_voxel_apply_scales_hip(const float *voxel, int64_t B, int64_t planes, int64_t H, int64_t W, int64_t H_in, int64_t W_in, int pad_to,
                               const float *scales, float *out, void *stream);
/* The same scales by selection over a FINISHED tensor (what v2v_normalize_pad_ex_hip does before it scales): method V2V_NORM_RADIX (any
 * float32 content, unpadded input) or V2V_NORM_COUNT (integer-valued, padded input allowed); workspace of v2v_postops_workspace_bytes(B). */
int v2v_voxel_scales_select_hip(const float *voxel, int64_t B, int64_t planes, int64_t H, int64_t W, int64_t H_in, int64_t W_in, int method,
                                float *scales, void *workspace, void *stream);

/* The `frame` tensor of a batch from the uint8 clips already in HBM (WebvidDatasetV2.__getitem__, data/v2v_datasets.py:329-338, 352:
 * frame[l] = float(all_imgs[idx_l]).permute(2,0,1) / 255): out[b,l,c,y,x] = float(src[b, pick[l], y, x, c]) / 255.0f (IEEE division:
 * torch's CPU values bit for bit).  src uint8 [B, *, H, W, C] with element strides clip_stride / frame_stride; pick = L device int32
 * frame indices (NULL: frames 0..L-1); out float32 [B,L,C,H,W] contiguous.  C >= 1. */
int v2v_clip_frames_f32_hip(const void *src, int64_t clip_stride, int64_t frame_stride, const int32_t *pick, int64_t B, int64_t L, int64_t H,
                            int64_t W, int64_t C, float *out, void *stream);
/* the same for packed clips (v2v_esim_extras): clip b starts at clip_offsets[b] (NULL: b * clip_stride; with offsets clip_stride states
 * their common alignment) and picks frame pick[b * pick_stride + l] (pick_stride 0: one row of picks for every clip) */
int v2v_clip_frames_f32_ex_hip(const void *src, int64_t clip_stride, const int64_t *clip_offsets, int64_t frame_stride, const int32_t *pick,
                               int64_t pick_stride, int64_t B, int64_t L, int64_t H, int64_t W, int64_t C, float *out, void *stream);
/* the same with the gather's bounds (ABI 3, round 5; what the loader calls): stored_frames int32 [B] = frames clip b holds, src_elems =
 * bytes in `src` (0: not stated).  A picked frame outside [0, stored_frames[b]), or one that does not fit `src`, is not read: its output
 * frame is NaN -- like v2v_esim_extras.stored_frames for the simulator (the picks live on the device: no status code without a
 * synchronisation).  stored_frames NULL = v2v_clip_frames_f32_ex_hip (the caller vouches for the picks).  Any B: more than 65,535
 * output frames go out as several launches of whole clips on `stream`; L <= 65535. */
int v2v_clip_frames_f32_bounded_hip(const void *src, int64_t clip_stride, const int64_t *clip_offsets, int64_t frame_stride, const int32_t *pick,
                                    int64_t pick_stride, const int32_t *stored_frames, int64_t src_elems, int64_t B, int64_t L, int64_t H, int64_t W,
                                    int64_t C, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* V2V_HIP_H */
