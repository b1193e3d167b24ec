"""ctypes binding of libv2v_hip.so (C ABI in include/v2v_hip.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  -- imported FIRST so the process uses one HIP runtime (torch's libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("V2V_HIP_LIB") or os.path.join(_HERE, "libv2v_hip.so")   # override: kernel sweeps only

# enums of include/v2v_hip.h
U8, F32, F64, BF16 = 0, 1, 2, 3
RNG_NONE, RNG_PHILOX, RNG_REPLAY, RNG_PHILOX_FAST = 0, 1, 2, 3
BIN_SUM, BIN_BILINEAR = 0, 1
FLAG_NOISE_EXTERNAL = 0x1
FLAG_NO_NOISE = 0x2
FLAG_SYMMETRIC = 0x4
FLAG_MAP_4PX, FLAG_MAP_1PX, FLAG_MAP_2PX = 0x8, 0x10, 0x20
OK, ERR_NULL, ERR_SHAPE, ERR_BINS, ERR_DTYPE, ERR_MODE, ERR_ALIGN, ERR_HIP, ERR_PARAM = 0, -1, -2, -3, -4, -5, -6, -7, -8
ABI_VERSION = 3

EXPORTS = ("v2v_version", "v2v_last_error", "v2v_device_count", "v2v_lut_get", "v2v_lut_set",
           "v2v_esim_voxel_hip", "v2v_esim_voxel_keyed_hip", "v2v_esim_voxel_bytes", "v2v_synth_clips_hip", "v2v_events_to_voxel_hip", "v2v_events_to_voxel_segmented_hip",
           "v2v_v2e_voxel_hip", "v2v_v2e_workspace_bytes", "v2v_events_to_voxel_f32_hip", "v2v_events_to_voxel_f32_segmented_hip", "v2v_esim_voxel_padded_hip", "v2v_normalize_pad_ex_hip", "v2v_frontend_hip", "v2v_frontend_batch_hip",
           "v2v_normalize_pad_hip", "v2v_postops_workspace_bytes",
           "v2v_convlstm_packed_bytes", "v2v_convlstm_pack_weights_hip", "v2v_convlstm_step_hip", "v2v_nchw_to_nhwc_bf16_hip",
           "v2v_conv3x3_pack_weights_hip", "v2v_conv3x3_nhwc_hip", "v2v_conv_pack_weights_hip", "v2v_conv_nhwc_hip", "v2v_upsample2x_nhwc_hip", "v2v_conv1x1_nhwc_hip", "v2v_conv_packed_elems", "v2v_conv_head_packed_elems", "v2v_conv_head_pack_weights_hip",
           "v2v_to_nhwc8_bf16_hip", "v2v_conv_head_nhwc_hip", "v2v_clip_frames_f32_hip",
           "v2v_esim_voxel_stats_hip", "v2v_voxel_scales_hip", "v2v_voxel_apply_scales_hip", "v2v_voxel_scales_select_hip", "v2v_to_nhwc8_bf16_scaled_hip", "v2v_esim_voxel_ex_hip", "v2v_clip_frames_f32_ex_hip", "v2v_clip_frames_f32_bounded_hip")
EV_MAKE_VOXEL_DISCRETE, EV_MAKE_VOXEL_INTERP, EV_BILINEAR = 0, 1, 2
NORM_NONE, NORM_RADIX, NORM_COUNT = 0, 1, 2
VOXEL_STATS_WORDS = 516


class EsimReplay(C.Structure):
    _fields_ = [("u_init", C.c_void_p), ("u_hot", C.c_void_p), ("g_hot", C.c_void_p), ("g_base", C.c_void_p)]


class EsimExtras(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("frame_index", C.c_void_p), ("clip_offsets", C.c_void_p), ("stored_frames", C.c_void_p), ("frames_elems", C.c_int64)]


class V2EParams(C.Structure):
    _fields_ = [("fps", C.c_double), ("threshold_model", C.c_int), ("thres_mean_mean", C.c_double),
                ("thres_mean_std", C.c_double), ("thres_diff_mean", C.c_double), ("thres_diff_std", C.c_double),
                ("cutoff_hz", C.c_double), ("leak_rate_hz", C.c_double), ("refractory_period_s", C.c_double),
                ("shot_noise_rate_hz", C.c_double), ("leak_jitter_fraction", C.c_double),
                ("noise_rate_cov_decades", C.c_double), ("uint8_wrap", C.c_int)]


class V2EReplay(C.Structure):
    _fields_ = [("pos_thres", C.c_void_p), ("neg_thres", C.c_void_p), ("thres_frame_stride", C.c_int64),
                ("noise_rate", C.c_void_p), ("leak_randn", C.c_void_p), ("shot_pos", C.c_void_p), ("shot_neg", C.c_void_p)]


V2E_MODELS = {"pn_related": 0, "spatial_independent": 1, "spatial_temporal_independent": 2}


class V2VError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libv2v_hip error {code}: {msg}")
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C v2v_amd/csrc`).  v2v_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.v2v_version.restype = C.c_int
    L.v2v_last_error.restype = C.c_char_p
    L.v2v_device_count.restype = C.c_int
    L.v2v_lut_get.argtypes = [C.c_int, C.c_void_p]
    L.v2v_lut_set.argtypes = [C.c_int, C.c_void_p]
    L.v2v_esim_voxel_hip.restype = C.c_int
    L.v2v_esim_voxel_hip.argtypes = [
        C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,   # frames..frame_stride
        C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64,                        # params..clip_id0
        C.POINTER(EsimReplay), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.v2v_esim_voxel_keyed_hip.restype = C.c_int
    L.v2v_esim_voxel_keyed_hip.argtypes = [
        C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
        C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p,
        C.POINTER(EsimReplay), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.v2v_esim_voxel_padded_hip.restype = C.c_int
    L.v2v_esim_voxel_padded_hip.argtypes = [
        C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
        C.c_void_p, C.c_int64, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64, C.c_void_p,
        C.POINTER(EsimReplay), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.v2v_esim_voxel_stats_hip.restype = C.c_int
    L.v2v_esim_voxel_stats_hip.argtypes = L.v2v_esim_voxel_padded_hip.argtypes[:-1] + [C.c_void_p, C.c_void_p]
    L.v2v_clip_frames_f32_ex_hip.restype = C.c_int
    L.v2v_clip_frames_f32_ex_hip.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                             C.c_int64, C.c_void_p, C.c_void_p]
    L.v2v_clip_frames_f32_bounded_hip.restype = C.c_int
    L.v2v_clip_frames_f32_bounded_hip.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64] + [C.c_int64] * 5 + [C.c_void_p, C.c_void_p]
    L.v2v_esim_voxel_ex_hip.restype = C.c_int
    L.v2v_esim_voxel_ex_hip.argtypes = L.v2v_esim_voxel_padded_hip.argtypes[:-1] + [C.POINTER(EsimExtras), C.c_void_p]
    L.v2v_voxel_scales_hip.restype = C.c_int
    L.v2v_voxel_scales_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.v2v_voxel_apply_scales_hip.restype = C.c_int
    L.v2v_voxel_apply_scales_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p]
    L.v2v_voxel_scales_select_hip.restype = C.c_int
    L.v2v_voxel_scales_select_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p]
    L.v2v_esim_voxel_bytes.restype = C.c_int64
    L.v2v_esim_voxel_bytes.argtypes = [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
    L.v2v_synth_clips_hip.restype = C.c_int
    L.v2v_synth_clips_hip.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_uint64,
                                      C.c_uint64, C.c_void_p]
    L.v2v_events_to_voxel_hip.restype = C.c_int
    L.v2v_events_to_voxel_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                          C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.v2v_events_to_voxel_segmented_hip.restype = C.c_int
    L.v2v_events_to_voxel_segmented_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                                    C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                                    C.c_void_p]
    L.v2v_events_to_voxel_f32_segmented_hip.restype = C.c_int
    L.v2v_events_to_voxel_f32_segmented_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                                        C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.v2v_events_to_voxel_f32_hip.restype = C.c_int
    L.v2v_events_to_voxel_f32_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                              C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.v2v_frontend_hip.restype = C.c_int
    L.v2v_frontend_hip.argtypes = [C.c_void_p] + [C.c_int64] * 10 + [C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p,
                                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.v2v_frontend_batch_hip.restype = C.c_int
    L.v2v_frontend_batch_hip.argtypes = [C.c_void_p] + [C.c_int64] * 5 + [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                                                           C.c_void_p, C.c_void_p, C.c_void_p]
    L.v2v_postops_workspace_bytes.restype = C.c_int64
    L.v2v_postops_workspace_bytes.argtypes = [C.c_int64]
    L.v2v_normalize_pad_ex_hip.restype = C.c_int
    L.v2v_normalize_pad_ex_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
    L.v2v_normalize_pad_hip.restype = C.c_int
    L.v2v_normalize_pad_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
    L.v2v_convlstm_packed_bytes.restype = C.c_int
    L.v2v_convlstm_packed_bytes.argtypes = [C.c_int64, C.POINTER(C.c_uint64)]
    L.v2v_convlstm_pack_weights_hip.restype = C.c_int
    L.v2v_convlstm_pack_weights_hip.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.v2v_convlstm_step_hip.restype = C.c_int
    L.v2v_convlstm_step_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.v2v_conv3x3_pack_weights_hip.restype = C.c_int
    L.v2v_conv3x3_pack_weights_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.v2v_conv3x3_nhwc_hip.restype = C.c_int
    L.v2v_conv3x3_nhwc_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                       C.c_void_p, C.c_int, C.c_void_p]
    L.v2v_conv_pack_weights_hip.restype = C.c_int
    L.v2v_conv_pack_weights_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.v2v_conv_nhwc_hip.restype = C.c_int
    L.v2v_conv_nhwc_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                    C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.v2v_conv_packed_elems.restype = C.c_int64
    L.v2v_conv_packed_elems.argtypes = [C.c_int64, C.c_int64, C.c_int]
    L.v2v_conv_head_packed_elems.restype = C.c_int64
    L.v2v_conv_head_packed_elems.argtypes = [C.c_int]
    L.v2v_conv_head_pack_weights_hip.restype = C.c_int
    L.v2v_conv_head_pack_weights_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.v2v_to_nhwc8_bf16_hip.restype = C.c_int
    L.v2v_to_nhwc8_bf16_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.v2v_to_nhwc8_bf16_scaled_hip.restype = C.c_int
    L.v2v_to_nhwc8_bf16_scaled_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
    L.v2v_conv_head_nhwc_hip.restype = C.c_int
    L.v2v_conv_head_nhwc_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.v2v_conv1x1_nhwc_hip.restype = C.c_int
    L.v2v_conv1x1_nhwc_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
    L.v2v_upsample2x_nhwc_hip.restype = C.c_int
    L.v2v_upsample2x_nhwc_hip.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    L.v2v_nchw_to_nhwc_bf16_hip.restype = C.c_int
    L.v2v_nchw_to_nhwc_bf16_hip.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]
    L.v2v_clip_frames_f32_hip.restype = C.c_int
    L.v2v_clip_frames_f32_hip.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                          C.c_void_p, C.c_void_p]
    L.v2v_v2e_workspace_bytes.restype = C.c_int64
    L.v2v_v2e_workspace_bytes.argtypes = [C.c_int64, C.c_int64]
    L.v2v_v2e_voxel_hip.restype = C.c_int
    L.v2v_v2e_voxel_hip.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                    C.POINTER(V2EParams), C.c_int, C.c_uint64, C.c_uint64, C.POINTER(V2EReplay), C.c_int,
                                    C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    if L.v2v_version() != ABI_VERSION:
        raise ImportError(f"libv2v_hip.so ABI {L.v2v_version()} != binding ABI {ABI_VERSION}: rebuild")
    _lib = L
    return L


def check(rc: int):
    """Map a v2v_status to the exception the reference would raise at the same spot."""
    if rc == OK:
        return
    msg = lib().v2v_last_error().decode()
    if rc == ERR_BINS:
        raise AssertionError(msg)            # reference: `assert (N-1) % (num_bins*frames_per_bin) == 0`
    if rc in (ERR_SHAPE, ERR_DTYPE, ERR_MODE, ERR_PARAM, ERR_NULL, ERR_ALIGN):
        raise ValueError(msg)
    raise V2VError(rc, msg)


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("v2v_amd needs an AMD GPU (torch.cuda.is_available() is False); there is no CPU fallback")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
