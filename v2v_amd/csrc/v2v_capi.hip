// v2v_capi.hip -- the C ABI of libv2v_hip.so (include/v2v_hip.h).  Argument validation, kernel
// dispatch on (input dtype, vector width, bin mode, rng mode), error reporting.  Single translation unit:
// the kernels live in the headers included below.  gfx950 only; no CPU fallback behind any entry point.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/v2v_hip.h"
#include "v2v_args.hpp"
#include "v2v_rng.hpp"
#include "v2v_events.hpp"
#include "v2v_frontend.hpp"
#include "v2v_postops.hpp"
#include "v2v_synth.hpp"
#include "v2v_assemble.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    return fail(V2V_ERR_HIP, "%s: %s (%d)", what, hipGetErrorString(e), (int)e);
}

bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// SIMDs of the current device (4 per CU), cached per device; 1024 when the query fails
int simd_count()
{
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1024;
    int v = cached[dev].load(std::memory_order_relaxed);
    if (!v) {
        int cus = 0;
        v = (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? 4 * cus : 1024;
        cached[dev].store(v, std::memory_order_relaxed);       // every thread computes the same value
    }
    return v;
}

}  // namespace

extern "C" {

int v2v_version(void) { return V2V_ABI_VERSION; }

const char *v2v_last_error(void) { return g_err; }

int v2v_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int v2v_lut_get(int which, void *dst)
{
    if (!dst) return fail(V2V_ERR_NULL, "v2v_lut_get: dst is NULL");
    if (which < 0 || which > 2) return fail(V2V_ERR_PARAM, "v2v_lut_get: which=%d", which);
    if (which == 2) (void)v2v::lut_v2e_copy(dst, false); else if (which == 1) (void)v2v::lut_esim32_copy(dst, false); else (void)v2v::lut_esim64_copy(dst, false);
    return V2V_OK;
}

int v2v_lut_set(int which, const void *src)
{
    if (!src) return fail(V2V_ERR_NULL, "v2v_lut_set: src is NULL");
    if (which < 0 || which > 2) return fail(V2V_ERR_PARAM, "v2v_lut_set: which=%d", which);
    // The tables are per-device symbols and this call sets the one of the CURRENT device only: under one process per GPU a rank
    // must not create HIP contexts (VRAM + start-up cost) on the devices of its peers.  A process that drives several GPUs
    // calls it once per device (hipSetDevice first).
    void *p = const_cast<void *>(src);
    const hipError_t e = which == 2 ? v2v::lut_v2e_copy(p, true) : which == 1 ? v2v::lut_esim32_copy(p, true) : v2v::lut_esim64_copy(p, true);
    return e == hipSuccess ? V2V_OK : hip_fail(e, "v2v_lut_set");
}

int64_t v2v_esim_voxel_bytes(int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W, int bin_mode, int num_bins,
                             int frames_per_bin, int out_dtype)
{
    if (B < 0 || N < 2 || H < 1 || W < 1 || num_bins < 1 || frames_per_bin < 1) return V2V_ERR_SHAPE;
    const int64_t in_sz = in_dtype == V2V_U8 ? 1 : in_dtype == V2V_F32 ? 4 : 0;
    const int64_t out_sz = out_dtype == V2V_F32 ? 4 : out_dtype == V2V_F64 ? 8 : 0;
    if (!in_sz || !out_sz) return V2V_ERR_DTYPE;
    int64_t planes;
    if (bin_mode == V2V_BIN_SUM) {
        if ((N - 1) % ((int64_t)num_bins * frames_per_bin) != 0) return V2V_ERR_BINS;
        planes = (N - 1) / frames_per_bin;
    } else if (bin_mode == V2V_BIN_BILINEAR) {
        planes = num_bins;
    } else {
        return V2V_ERR_MODE;
    }
    return B * H * W * (N * in_sz + planes * out_sz);
}

int v2v_esim_voxel_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                       int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                       uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0,
                       const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                       void *out_voxel, int out_dtype, int64_t *out_counts, void *stream)
{
    return v2v_esim_voxel_keyed_hip(frames, in_dtype, B, N, H, W, clip_stride, frame_stride, params, params_stride, flags,
                                    rng_mode, seed, clip_id0, nullptr, replay, bin_mode, num_bins, frames_per_bin, out_voxel,
                                    out_dtype, out_counts, stream);
}

int v2v_esim_voxel_keyed_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                             int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                             uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0, const uint64_t *clip_keys,
                             const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                             void *out_voxel, int out_dtype, int64_t *out_counts, void *stream)
{
    return v2v_esim_voxel_padded_hip(frames, in_dtype, B, N, H, W, clip_stride, frame_stride, params, params_stride, flags, rng_mode, seed,
                                     clip_id0, clip_keys, replay, bin_mode, num_bins, frames_per_bin, out_voxel, out_dtype, W, H * W,
                                     out_counts, stream);
}

static int esim_launch_impl(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                            int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                            uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0, const uint64_t *clip_keys,
                            const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                            void *out_voxel, int out_dtype, int64_t out_row_pitch, int64_t out_plane_size, int64_t *out_counts, uint32_t *stats,
                            const int32_t *frame_index, const int64_t *clip_offsets, const int32_t *stored_frames, int64_t frames_elems, void *stream);

int v2v_esim_voxel_padded_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                              int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                              uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0, const uint64_t *clip_keys,
                              const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                              void *out_voxel, int out_dtype, int64_t out_row_pitch, int64_t out_plane_size, int64_t *out_counts, void *stream)
{
    return esim_launch_impl(frames, in_dtype, B, N, H, W, clip_stride, frame_stride, params, params_stride, flags, rng_mode, seed, clip_id0, clip_keys,
                            replay, bin_mode, num_bins, frames_per_bin, out_voxel, out_dtype, out_row_pitch, out_plane_size, out_counts, nullptr, nullptr, nullptr, nullptr, 0, stream);
}

int v2v_esim_voxel_ex_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                          int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                          uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0, const uint64_t *clip_keys,
                          const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                          void *out_voxel, int out_dtype, int64_t out_row_pitch, int64_t out_plane_size, int64_t *out_counts,
                          const v2v_esim_extras *extras, void *stream)
{
    if (!extras) return v2v_esim_voxel_padded_hip(frames, in_dtype, B, N, H, W, clip_stride, frame_stride, params, params_stride, flags, rng_mode, seed, clip_id0,
                                                  clip_keys, replay, bin_mode, num_bins, frames_per_bin, out_voxel, out_dtype, out_row_pitch, out_plane_size,
                                                  out_counts, stream);
    if ((extras->frame_index == nullptr) != (extras->clip_offsets == nullptr))
        return fail(V2V_ERR_NULL, "v2v_esim_voxel_ex_hip: frame_index and clip_offsets come together");
    if (extras->frame_index) {
        if (in_dtype != V2V_U8 || bin_mode != V2V_BIN_SUM || out_dtype != V2V_F32 || rng_mode == V2V_RNG_REPLAY || rng_mode == V2V_RNG_NONE ||
            (flags & (V2V_FLAG_NOISE_EXTERNAL | V2V_FLAG_NO_NOISE | V2V_FLAG_SYMMETRIC)))
            return fail(V2V_ERR_MODE, "indexed frames: uint8 clips, SUM bins, device noise, float32 grid, no NO_NOISE / NOISE_EXTERNAL / SYMMETRIC flag");
        if (!aligned(extras->frame_index, 4) || !aligned(extras->clip_offsets, 8) || !aligned(extras->stored_frames, 4))
            return fail(V2V_ERR_ALIGN, "frame_index / clip_offsets / stored_frames misaligned");
        if (extras->frames_elems < 0) return fail(V2V_ERR_SHAPE, "frames_elems must be >= 0 (0: not stated)");
    } else if (extras->stored_frames || extras->frames_elems) {
        return fail(V2V_ERR_NULL, "v2v_esim_voxel_ex_hip: stored_frames / frames_elems bound the frame_index gather and come with it");
    }
    if (extras->stats) {
        if (bin_mode != V2V_BIN_SUM || out_dtype != V2V_F32 || (flags & V2V_FLAG_NOISE_EXTERNAL))
            return fail(V2V_ERR_MODE, "voxel statistics are kept for SUM-mode float32 grids without external noise (integer counts)");
        if (!aligned(extras->stats, 4)) return fail(V2V_ERR_ALIGN, "stats must be 4-byte aligned");
        if (B > 0) {
            const hipError_t e = hipMemsetAsync(extras->stats, 0, sizeof(uint32_t) * (size_t)B * V2V_VOXEL_STATS_WORDS, static_cast<hipStream_t>(stream));
            if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(stats)");
        }
    }
    return esim_launch_impl(frames, in_dtype, B, N, H, W, clip_stride, frame_stride, params, params_stride, flags, rng_mode, seed, clip_id0, clip_keys,
                            replay, bin_mode, num_bins, frames_per_bin, out_voxel, out_dtype, out_row_pitch, out_plane_size, out_counts, extras->stats,
                            extras->frame_index, extras->clip_offsets, extras->stored_frames, extras->frames_elems, stream);
}

int v2v_esim_voxel_stats_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                             int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                             uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0, const uint64_t *clip_keys,
                             const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                             void *out_voxel, int out_dtype, int64_t out_row_pitch, int64_t out_plane_size, int64_t *out_counts, uint32_t *stats,
                             void *stream)
{
    if (!stats) return fail(V2V_ERR_NULL, "v2v_esim_voxel_stats_hip: stats is NULL");
    if (bin_mode != V2V_BIN_SUM || out_dtype != V2V_F32 || (flags & V2V_FLAG_NOISE_EXTERNAL))
        return fail(V2V_ERR_MODE, "voxel statistics are kept for SUM-mode float32 grids without external noise (integer counts)");
    if (!aligned(stats, 4)) return fail(V2V_ERR_ALIGN, "stats must be 4-byte aligned");
    if (B > 0) {
        const hipError_t e = hipMemsetAsync(stats, 0, sizeof(uint32_t) * (size_t)B * V2V_VOXEL_STATS_WORDS, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(stats)");
    }
    return esim_launch_impl(frames, in_dtype, B, N, H, W, clip_stride, frame_stride, params, params_stride, flags, rng_mode, seed, clip_id0, clip_keys,
                            replay, bin_mode, num_bins, frames_per_bin, out_voxel, out_dtype, out_row_pitch, out_plane_size, out_counts, stats, nullptr, nullptr, nullptr, 0, stream);
}

int v2v_voxel_scales_hip(const uint32_t *stats, int64_t B, int64_t elems_per_sample, float *scales, void *stream)
{
    if (!stats || !scales) return fail(V2V_ERR_NULL, "v2v_voxel_scales_hip: stats/scales is NULL");
    if (B < 0 || elems_per_sample < 1) return fail(V2V_ERR_SHAPE, "need B>=0, elems_per_sample>=1");
    // torch.kthvalue is 1-based: max_k = int(0.99*M), min_k = int(0.01*M) (model/train_utils.py:153-154)
    const int64_t max_k = (int64_t)(0.99 * (double)elems_per_sample), min_k = (int64_t)(0.01 * (double)elems_per_sample);
    if (min_k < 1 || max_k < 1) return fail(V2V_ERR_SHAPE, "k-th value undefined: fewer than 100 elements per sample (torch.kthvalue would raise)");
    if (B == 0) return V2V_OK;
    static_assert(v2v::kStatBins == v2v::kCntBins && v2v::kStatZero == v2v::kCntZero && V2V_VOXEL_STATS_WORDS == v2v::kStatWords, "one bin layout");
    hipLaunchKernelGGL(v2v::count_pick_kernel, dim3((unsigned)(B * 2)), dim3(64), 0, static_cast<hipStream_t>(stream), stats, (int64_t)v2v::kStatWords,
                       stats + v2v::kStatBad, (int64_t)v2v::kStatWords, B * 2, (uint64_t)0, (uint64_t)(min_k - 1), (uint64_t)(max_k - 1), 1,
                       (uint64_t)elems_per_sample, scales);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? V2V_OK : hip_fail(e, "count_pick_kernel launch");
}

int v2v_voxel_apply_scales_hip(const float *voxel, int64_t B, int64_t planes, int64_t H, int64_t W, int64_t H_in, int64_t W_in, int pad_to,
                               const float *scales, float *out, void *stream)
{
    if (!voxel || !out) return fail(V2V_ERR_NULL, "v2v_voxel_apply_scales_hip: voxel/out is NULL");
    if (B < 0 || planes < 1 || H < 1 || W < 1 || pad_to < 1 || H_in < H || W_in < W) return fail(V2V_ERR_SHAPE, "need B>=0, planes,H,W,pad_to>=1, H_in>=H, W_in>=W");
    if (B > 65535) return fail(V2V_ERR_SHAPE, "more than 65535 samples in one call (one grid row per sample): split the batch");
    if (!aligned(voxel, 4) || !aligned(out, 4) || (scales && !aligned(scales, 4))) return fail(V2V_ERR_ALIGN, "buffers misaligned");
    if (B == 0) return V2V_OK;
    const int Hp = (int)((H + pad_to - 1) / pad_to * pad_to), Wp = (int)((W + pad_to - 1) / pad_to * pad_to);
    if (out == voxel && (Hp != H_in || Wp != W_in)) return fail(V2V_ERR_SHAPE, "in-place scaling needs identical input and output layouts");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t per_out = planes * Hp * Wp;
    const bool rows4 = W % 4 == 0 && Wp % 4 == 0 && W_in % 4 == 0 && planes * Hp < ((int64_t)1 << 31) &&
                       ((reinterpret_cast<uintptr_t>(voxel) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    if (rows4) {
        const unsigned gr = (unsigned)std::min<int64_t>((planes * Hp + 3) / 4, 4096);
        hipLaunchKernelGGL(v2v::normalize_pad_rows_kernel, dim3(gr, (unsigned)B), dim3(256), 0, s, voxel, out, scales, (int)planes, (int)H, (int)W, Hp, Wp,
                           (int)H_in, (int)W_in);
    } else {
        const unsigned gx2 = (unsigned)std::min<int64_t>((per_out + 256 * 4 - 1) / (256 * 4), 2048);
        hipLaunchKernelGGL(v2v::normalize_pad_kernel, dim3(gx2, (unsigned)B), dim3(256), 0, s, voxel, out, scales, planes, (int)H, (int)W, Hp, Wp, (int)H_in,
                           (int)W_in);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? V2V_OK : hip_fail(e, "normalize_pad launch");
}

static int esim_launch_impl(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W,
                            int64_t clip_stride, int64_t frame_stride, const double *params, int64_t params_stride,
                            uint32_t flags, int rng_mode, uint64_t seed, uint64_t clip_id0, const uint64_t *clip_keys,
                            const v2v_esim_replay *replay, int bin_mode, int num_bins, int frames_per_bin,
                            void *out_voxel, int out_dtype, int64_t out_row_pitch, int64_t out_plane_size, int64_t *out_counts, uint32_t *stats,
                            const int32_t *frame_index, const int64_t *clip_offsets, const int32_t *stored_frames, int64_t frames_elems, void *stream)
{
    if (out_row_pitch < W || out_plane_size < out_row_pitch * (H - 1) + W) return fail(V2V_ERR_SHAPE, "out_row_pitch / out_plane_size smaller than the frame");
    if (!frames || !params || !out_voxel) return fail(V2V_ERR_NULL, "v2v_esim_voxel_hip: frames/params/out_voxel is NULL");
    if (B < 0 || N < 2 || H < 1 || W < 1) return fail(V2V_ERR_SHAPE, "need B>=0, N>=2, H,W>=1 (got B=%lld N=%lld H=%lld W=%lld)", (long long)B, (long long)N, (long long)H, (long long)W);
    const int64_t HW = H * W, K = N - 1;
    if (HW > (int64_t)1 << 30 || K > (1 << 20) || B > (int64_t)1 << 31) return fail(V2V_ERR_SHAPE, "H*W, N or B too large");
    if (frame_stride < HW || (!clip_offsets && B > 1 && clip_stride < (N - 1) * frame_stride + HW)) return fail(V2V_ERR_SHAPE, "strides smaller than the extent");
    if (clip_offsets && clip_stride < 1) return fail(V2V_ERR_SHAPE, "with clip_offsets, clip_stride states the alignment (in elements) every offset keeps");
    if (in_dtype != V2V_U8 && in_dtype != V2V_F32) return fail(V2V_ERR_DTYPE, "in_dtype must be V2V_U8 or V2V_F32");
    if (out_dtype != V2V_F32 && out_dtype != V2V_F64) return fail(V2V_ERR_DTYPE, "out_dtype must be V2V_F32 or V2V_F64");
    if (params_stride != 0 && params_stride < 5) return fail(V2V_ERR_PARAM, "params_stride must be 0 or >= 5");
    if (num_bins < 1 || frames_per_bin < 1) return fail(V2V_ERR_PARAM, "num_bins and frames_per_bin must be >= 1");
    if (rng_mode < V2V_RNG_NONE || rng_mode > V2V_RNG_PHILOX_FAST) return fail(V2V_ERR_MODE, "unknown rng_mode %d", rng_mode);
    if (rng_mode == V2V_RNG_PHILOX_FAST) rng_mode = V2V_RNG_PHILOX;             // alias (see the header)
    if (rng_mode == V2V_RNG_REPLAY && (!replay || !replay->u_init || !replay->u_hot || !replay->g_hot || !replay->g_base))
        return fail(V2V_ERR_MODE, "rng_mode REPLAY needs all four replay fields");
    if (bin_mode == V2V_BIN_SUM) {
        if (K % ((int64_t)num_bins * frames_per_bin) != 0)
            return fail(V2V_ERR_BINS, "(N-1)=%lld is not a multiple of num_bins*frames_per_bin=%d", (long long)K, num_bins * frames_per_bin);
    } else if (bin_mode == V2V_BIN_BILINEAR) {
        if (K < 2) return fail(V2V_ERR_BINS, "BILINEAR needs at least 2 frame pairs (dt = 0 otherwise)");
    } else {
        return fail(V2V_ERR_MODE, "unknown bin_mode %d", bin_mode);
    }
    const size_t in_sz = in_dtype == V2V_U8 ? 1 : 4, out_sz = out_dtype == V2V_F32 ? 4 : 8;
    if (!aligned(frames, in_sz) || !aligned(out_voxel, out_sz) || !aligned(params, 8) || (out_counts && !aligned(out_counts, 8)))
        return fail(V2V_ERR_ALIGN, "buffer not aligned to its element size");
    if (B == 0) return V2V_OK;

    // 4 (or 2) pixels per work-item when every row segment a lane touches is 16-byte (8-byte) fp32 / 4-byte (2-byte) u8 aligned
    auto layout_ok = [&](int64_t v) {
        // (with clip_offsets, clip_stride is the alignment every offset keeps: it counts for a single clip too)
        return (HW % v == 0) && (frame_stride % v == 0) && ((B == 1 && !clip_offsets) || clip_stride % v == 0) && aligned(frames, (size_t)v * in_sz) &&
               aligned(out_voxel, (size_t)v * 4) &&
               (out_row_pitch == W ? out_plane_size % v == 0 : (W % v == 0 && out_row_pitch % v == 0 && out_plane_size % v == 0));
    };
    const bool ok4 = layout_ok(4), ok2 = layout_ok(2) && (out_dtype == V2V_F32 || aligned(out_voxel, 16));
    // Small batches leave SIMDs idle at 4 pixels per work-item (the reference's training shape: 12 clips of 128x128 = 768 such
    // waves for 1024 SIMDs).  Fewer pixels per work-item = more waves; the 1-pixel instances share one Philox block between the
    // four lanes of a pixel quad (v2v_esim.hpp), so they pay no extra generator work for it.  Same box, 201x128x128 uint8 ->
    // 40x5 SUM bins, ms (4 / 2 / 1 pixels): 12 clips 0.095 / 0.093 / 0.064; 24 clips 0.136 / 0.128 / 0.113; 48 clips
    // 0.177 / 0.238 / 0.201 (tools/train_shape_time.py, profiles/r03/train_shape_time.json).  Hence: up to two 4-pixel waves per
    // SIMD -> 1 pixel (float32 frames: up to one; then 2 pixels up to two), above -> 4; 2 pixels also serve layouts whose rows are only 2-pixel aligned.  Results do not depend on the mapping.
    int vec = ok4 ? 4 : ok2 ? 2 : 1;
    const int64_t waves4 = B * ((HW + 1023) / 1024) * 4, simds = (int64_t)simd_count();
    if (flags & V2V_FLAG_MAP_1PX) vec = 1;
    else if (flags & V2V_FLAG_MAP_2PX) vec = ok2 ? 2 : 1;
    else if (!(flags & V2V_FLAG_MAP_4PX)) {
        if (waves4 <= simds || (in_dtype == V2V_U8 && waves4 <= 2 * simds)) vec = 1;
        else if (waves4 <= 2 * simds && ok2) vec = 2;             // float32 frames, 24 clips: 0.188 / 0.132 / 0.174 ms
    }
#ifdef V2V_FORCE_SCALAR_PATH       // kernel-tuning builds only: no environment lookups on the product's launch path
    vec = 1;
#endif

    v2v::EsimArgs a{};
    a.frames = frames;
    a.clip_stride = clip_stride;
    a.frame_stride = frame_stride;
    a.params = params;
    a.params_stride = params_stride;
    a.out = out_voxel;
    a.counts = reinterpret_cast<unsigned long long *>(out_counts);
    if (replay) { a.u_init = replay->u_init; a.u_hot = replay->u_hot; a.g_hot = replay->g_hot; a.g_base = replay->g_base; }
    a.seed = seed;
    a.clip_id0 = clip_id0;
    a.clip_keys = reinterpret_cast<const unsigned long long *>(clip_keys);
    a.HW = (int32_t)HW;
    a.K = (int32_t)K;
    a.Tb = num_bins;
    a.fpb = frames_per_bin;
    a.blocks_per_clip = (int32_t)((HW + (int64_t)v2v::kBlock * vec - 1) / ((int64_t)v2v::kBlock * vec));
    a.noise_external = (flags & V2V_FLAG_NOISE_EXTERNAL) ? 1u : 0u;
    a.W = (int32_t)W;
    a.out_pitch = out_row_pitch;
    a.out_plane = out_plane_size;
    a.stats = stats;
    a.frame_index = frame_index;
    a.clip_offsets = clip_offsets;
    a.stored_frames = stored_frames;
    a.frames_elems = frames_elems;
    const int64_t nblocks = B * a.blocks_per_clip;
    if (nblocks > 0x7FFFFFFF) return fail(V2V_ERR_SHAPE, "grid too large");
    const dim3 grid((unsigned)nblocks);
    const bool out64 = out_dtype == V2V_F64;
    const bool noise = !(flags & V2V_FLAG_NO_NOISE);
    a.sym_only = (flags & V2V_FLAG_SYMMETRIC) ? 1u : 0u;
    if (!noise && (flags & V2V_FLAG_NOISE_EXTERNAL)) return fail(V2V_ERR_PARAM, "V2V_FLAG_NO_NOISE and V2V_FLAG_NOISE_EXTERNAL are exclusive");
    if (!noise && rng_mode == V2V_RNG_REPLAY) return fail(V2V_ERR_PARAM, "V2V_FLAG_NO_NOISE is not available in replay mode");
    // dynamic LDS: the bilinear weights per frame pair and the segment starts per bin; static: log table, thresholds, and the
    // Gaussian table of the instances that draw device-native noise
    const size_t lds = bin_mode == V2V_BIN_BILINEAR ? (size_t)K * 2 * (out64 ? sizeof(double) : sizeof(float)) + (size_t)num_bins * sizeof(int)
                                                    : frame_index ? (size_t)(K + 1) * sizeof(int) : 0;
    const size_t lds_static = 256 * 8 + 32 + (bin_mode == V2V_BIN_SUM && !out64 ? 2064 : 0) /* the writer's histogram */ + (noise && rng_mode != V2V_RNG_REPLAY && rng_mode != V2V_RNG_NONE ? (size_t)v2v::kIcdfBytes : 0);
    if (lds + lds_static > 64 * 1024) return fail(V2V_ERR_SHAPE, "too many frame pairs for the LDS tables (%zu bytes > 64 KiB): split the clip", lds + lds_static);
    hipStream_t s = static_cast<hipStream_t>(stream);

    hipError_t e;
    e = in_dtype == V2V_U8 ? v2v::launch_esim_u8(vec, bin_mode, rng_mode, noise, out64, a, grid, lds, s)
                           : v2v::launch_esim_f32(vec, bin_mode, rng_mode, noise, out64, a, grid, lds, s);
    return e == hipSuccess ? V2V_OK : hip_fail(e, "esim_voxel_kernel launch");
}

int v2v_clip_frames_f32_hip(const void *src, int64_t clip_stride, int64_t frame_stride, const int32_t *pick, int64_t B, int64_t L, int64_t H,
                            int64_t W, int64_t C, float *out, void *stream)
{
    return v2v_clip_frames_f32_ex_hip(src, clip_stride, nullptr, frame_stride, pick, 0, B, L, H, W, C, out, stream);
}

int v2v_clip_frames_f32_ex_hip(const void *src, int64_t clip_stride, const int64_t *clip_offsets, int64_t frame_stride, const int32_t *pick,
                               int64_t pick_stride, int64_t B, int64_t L, int64_t H, int64_t W, int64_t C, float *out, void *stream)
{
    return v2v_clip_frames_f32_bounded_hip(src, clip_stride, clip_offsets, frame_stride, pick, pick_stride, nullptr, 0, B, L, H, W, C, out, stream);
}

int v2v_clip_frames_f32_bounded_hip(const void *src, int64_t clip_stride, const int64_t *clip_offsets, int64_t frame_stride, const int32_t *pick,
                                    int64_t pick_stride, const int32_t *stored_frames, int64_t src_elems, int64_t B, int64_t L, int64_t H, int64_t W,
                                    int64_t C, float *out, void *stream)
{
    if (!src || !out) return fail(V2V_ERR_NULL, "v2v_clip_frames_f32_hip: src/out is NULL");
    if (B < 0 || L < 1 || H < 1 || W < 1 || C < 1 || C > 4) return fail(V2V_ERR_SHAPE, "need B>=0, L,H,W>=1, 1<=C<=4");
    const int64_t HW = H * W;
    if (HW * C >= (int64_t)1 << 30 || L > 65535) return fail(V2V_ERR_SHAPE, "frame too large or more than 65535 picked frames per clip");
    if (frame_stride < HW * C || (!clip_offsets && B > 1 && clip_stride < frame_stride) || (clip_offsets && clip_stride < 1) || pick_stride < 0 ||
        (pick_stride != 0 && (!pick || pick_stride < L)))
        return fail(V2V_ERR_SHAPE, "strides smaller than the extent");
    if (!aligned(out, 4) || !aligned(clip_offsets, 8) || !aligned(pick, 4) || !aligned(stored_frames, 4))
        return fail(V2V_ERR_ALIGN, "out / clip_offsets / pick / stored_frames misaligned");
    if (src_elems < 0 || (!stored_frames && src_elems != 0)) return fail(V2V_ERR_SHAPE, "src_elems comes with stored_frames and is >= 0 (0: not stated)");
    if (B == 0) return V2V_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // with clip_offsets, clip_stride states the alignment (in bytes) every offset keeps
    const bool v4 = C == 1 && HW % 4 == 0 && frame_stride % 4 == 0 && ((B == 1 && !clip_offsets) || clip_stride % 4 == 0) && aligned(src, 4) && aligned(out, 16);
    // the grid's y dimension (one output frame each) holds 65,535: larger batches go out as several launches of whole clips
    const int64_t group = std::max<int64_t>(1, 65535 / L);
    for (int64_t b0 = 0; b0 < B; b0 += group) {
        const int64_t nb = std::min(group, B - b0);
        const uint8_t *sp = static_cast<const uint8_t *>(src) + (clip_offsets ? 0 : b0 * clip_stride);
        const int64_t *offs = clip_offsets ? clip_offsets + b0 : nullptr;
        const int32_t *pk = pick ? pick + b0 * pick_stride : nullptr;
        const v2v::ClipBounds cb{stored_frames ? stored_frames + b0 : nullptr, (src_elems > 0 && !clip_offsets) ? std::max<int64_t>(1, src_elems - b0 * clip_stride) : src_elems};
        float *o = out + b0 * L * HW * C;
        if (v4) {
            const int hw4 = (int)(HW / 4);
            const unsigned gx = (unsigned)std::min<int64_t>((hw4 + 255) / 256, 64);
            hipLaunchKernelGGL(v2v::clip_frames4_kernel, dim3(gx, (unsigned)(nb * L)), dim3(256), 0, s, sp, clip_stride, offs, frame_stride, pk, pick_stride,
                               (int)L, hw4, o, cb);
        } else {
            const int64_t n = HW * C;
            const unsigned gx = (unsigned)std::min<int64_t>((n + 255) / 256, 256);
            hipLaunchKernelGGL(v2v::clip_frames_kernel, dim3(gx, (unsigned)(nb * L)), dim3(256), 0, s, sp, clip_stride, offs, frame_stride, pk, pick_stride,
                               (int)L, (int)HW, (int)C, o, cb);
        }
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? V2V_OK : hip_fail(e, "clip_frames kernel launch");
}

int v2v_synth_clips_hip(void *frames, int dtype, int64_t B, int64_t N, int64_t H, int64_t W, uint64_t seed,
                        uint64_t clip_id0, void *stream)
{
    if (!frames) return fail(V2V_ERR_NULL, "v2v_synth_clips_hip: frames is NULL");
    if (B < 0 || N < 1 || H < 1 || W < 1 || W % 4 != 0) return fail(V2V_ERR_SHAPE, "need B>=0, N,H>=1 and W %% 4 == 0");
    if (dtype != V2V_U8 && dtype != V2V_F32) return fail(V2V_ERR_DTYPE, "dtype must be V2V_U8 or V2V_F32");
    if (!aligned(frames, dtype == V2V_F32 ? 16 : 4)) return fail(V2V_ERR_ALIGN, "frames must be 16-byte (f32) / 4-byte (u8) aligned");
    if (B == 0) return V2V_OK;
    v2v::SynthArgs a{};
    a.frames = frames;
    a.seed = seed;
    a.clip_id0 = clip_id0;
    a.N = (int32_t)N;
    a.H = (int32_t)H;
    a.W = (int32_t)W;
    a.is_f32 = dtype == V2V_F32;
    a.total_quads = B * N * H * W / 4;
    const int64_t nblocks = (a.total_quads + 255) / 256;
    if (nblocks > 0x7FFFFFFF) return fail(V2V_ERR_SHAPE, "grid too large");
    hipLaunchKernelGGL(v2v::synth_clips_kernel, dim3((unsigned)nblocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? V2V_OK : hip_fail(e, "synth_clips_kernel launch");
}

int64_t v2v_v2e_workspace_bytes(int64_t B, int64_t N)
{
    if (B < 0 || N < 2) return V2V_ERR_SHAPE;
    return B * (N - 1) * 4 * (int64_t)sizeof(int64_t);
}

int v2v_v2e_voxel_hip(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t H, int64_t W, int64_t clip_stride,
                      int64_t frame_stride, const v2v_v2e_params *params, int rng_mode, uint64_t seed, uint64_t clip_id0,
                      const v2v_v2e_replay *replay, int bin_mode, int num_bins, int frames_per_bin, void *out_voxel,
                      int out_dtype, int64_t *out_counts, void *workspace, void *stream)
{
    if (!frames || !params || !out_voxel) return fail(V2V_ERR_NULL, "v2v_v2e_voxel_hip: frames/params/out_voxel is NULL");
    if (B < 0 || N < 2 || H < 1 || W < 1) return fail(V2V_ERR_SHAPE, "need B>=0, N>=2, H,W>=1");
    const int64_t HW = H * W, K = N - 1;
    if (HW > (int64_t)1 << 30 || K > (1 << 20) || B > (int64_t)1 << 31) return fail(V2V_ERR_SHAPE, "H*W, N or B too large");
    if (frame_stride < HW || (B > 1 && clip_stride < (N - 1) * frame_stride + HW)) return fail(V2V_ERR_SHAPE, "strides smaller than the extent");
    if (in_dtype != V2V_U8 && in_dtype != V2V_F32) return fail(V2V_ERR_DTYPE, "in_dtype must be V2V_U8 or V2V_F32");
    if (out_dtype != V2V_F32 && out_dtype != V2V_F64) return fail(V2V_ERR_DTYPE, "out_dtype must be V2V_F32 or V2V_F64");
    if (num_bins < 1 || frames_per_bin < 1) return fail(V2V_ERR_PARAM, "num_bins and frames_per_bin must be >= 1");
    if (rng_mode != V2V_RNG_PHILOX && rng_mode != V2V_RNG_REPLAY) return fail(V2V_ERR_MODE, "v2e rng_mode must be PHILOX or REPLAY");
    if (params->threshold_model < V2V_V2E_PN_RELATED || params->threshold_model > V2V_V2E_SPATIAL_TEMPORAL_INDEPENDENT)
        return fail(V2V_ERR_MODE, "unsupported threshold_model %d (spatial_independent_temporal_changing crashes in the reference)", params->threshold_model);
    if (!(params->fps > 0)) return fail(V2V_ERR_PARAM, "fps must be > 0");
    const bool shot = params->shot_noise_rate_hz > 0, leak = params->leak_rate_hz > 0;
    const bool temporal = params->threshold_model == V2V_V2E_SPATIAL_TEMPORAL_INDEPENDENT;
    if (rng_mode == V2V_RNG_REPLAY) {
        if (!replay || !replay->pos_thres || !replay->neg_thres || !replay->noise_rate || (leak && !replay->leak_randn) ||
            (shot && (!replay->shot_pos || !replay->shot_neg)))
            return fail(V2V_ERR_MODE, "rng_mode REPLAY: missing replay field");
        if (temporal != (replay->thres_frame_stride != 0)) return fail(V2V_ERR_MODE, "thres_frame_stride must be H*W for the temporal model, 0 otherwise");
    }
    const bool presum = shot && rng_mode == V2V_RNG_PHILOX;
    if (presum && !workspace) return fail(V2V_ERR_NULL, "native shot noise needs a workspace of v2v_v2e_workspace_bytes()");
    // the native sampler inverts the Poisson distribution up to a count of 64 per pixel, frame and polarity: exact in distribution while the
    // expected count stays small against that (frame mean rate / (2 fps) <= 16; the reference's default is 0.1), silently truncated beyond --
    // refuse instead (np.random.poisson, i.e. REPLAY with host-drawn counts, has no such limit)
    if (presum && params->shot_noise_rate_hz > 32.0 * params->fps)
        return fail(V2V_ERR_PARAM, "native shot noise covers up to 16 expected noise events per pixel, frame and polarity (shot_noise_rate_hz <= 32 fps; got %g Hz at %g fps): "
                                   "use V2V_RNG_REPLAY with host-drawn counts beyond", params->shot_noise_rate_hz, params->fps);
    if (bin_mode == V2V_BIN_SUM) {
        if (K % ((int64_t)num_bins * frames_per_bin) != 0)
            return fail(V2V_ERR_BINS, "(N-1)=%lld is not a multiple of num_bins*frames_per_bin=%d", (long long)K, num_bins * frames_per_bin);
    } else if (bin_mode == V2V_BIN_BILINEAR) {
        if (K < 2) return fail(V2V_ERR_BINS, "BILINEAR needs at least 2 frame pairs");
    } else {
        return fail(V2V_ERR_MODE, "unknown bin_mode %d", bin_mode);
    }
    const size_t in_sz = in_dtype == V2V_U8 ? 1 : 4, out_sz = out_dtype == V2V_F32 ? 4 : 8;
    if (!aligned(frames, in_sz) || !aligned(out_voxel, out_sz) || (out_counts && !aligned(out_counts, 8)) || (workspace && !aligned(workspace, 8)))
        return fail(V2V_ERR_ALIGN, "buffer not aligned to its element size");
    if (B == 0) return V2V_OK;
    const bool vec4 = (HW % 4 == 0) && (frame_stride % 4 == 0) && (B == 1 || clip_stride % 4 == 0) &&
                      aligned(frames, 4 * in_sz) && aligned(out_voxel, 16);
    const int vec = vec4 ? 4 : 1;
    v2v::V2eArgs a{};
    a.frames = frames; a.clip_stride = clip_stride; a.frame_stride = frame_stride;
    a.out = out_voxel;
    a.counts = reinterpret_cast<unsigned long long *>(out_counts);
    a.shot_sums = presum ? static_cast<long long *>(workspace) : nullptr;
    if (replay) {
        a.r_pos_thres = replay->pos_thres; a.r_neg_thres = replay->neg_thres; a.r_thres_frame_stride = replay->thres_frame_stride;
        a.r_noise_rate = replay->noise_rate; a.r_leak_randn = replay->leak_randn;
        a.r_shot_pos = reinterpret_cast<const long long *>(replay->shot_pos);
        a.r_shot_neg = reinterpret_cast<const long long *>(replay->shot_neg);
    }
    a.seed = seed; a.clip_id0 = clip_id0;
    a.HW = (int32_t)HW; a.K = (int32_t)K; a.Tb = num_bins; a.fpb = frames_per_bin;
    a.blocks_per_clip = (int32_t)((HW + (int64_t)v2v::kBlock * vec - 1) / ((int64_t)v2v::kBlock * vec));
    a.pre_blocks_per_clip = (int32_t)((HW + (int64_t)v2v::kBlock * v2v::kPreGroups * vec - 1) / ((int64_t)v2v::kBlock * v2v::kPreGroups * vec));
    static_assert(sizeof(v2v::V2eParams) == sizeof(v2v_v2e_params), "v2e params layout");
    memcpy(&a.P, params, sizeof(a.P));
    const int64_t nblocks = B * a.blocks_per_clip;
    if (nblocks > 0x7FFFFFFF) return fail(V2V_ERR_SHAPE, "grid too large");
    const bool out64 = out_dtype == V2V_F64;
    const size_t lds = /* static: per-intensity records + the Gaussian table of the device-native instances */ (size_t)K * 32 /* per-frame constants */ + (bin_mode == V2V_BIN_BILINEAR ? (size_t)K * (2 * (out64 ? sizeof(double) : sizeof(float)) + sizeof(int)) : 0);
    const size_t lds_static = 256 * 16 + (rng_mode == V2V_RNG_PHILOX ? (size_t)v2v::kIcdfBytes : 0);
    if (lds + lds_static > 64 * 1024) return fail(V2V_ERR_SHAPE, "too many frame pairs for the LDS tables (%zu bytes > 64 KiB): split the clip", lds + lds_static);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (presum) {
        const hipError_t e0 = hipMemsetAsync(workspace, 0, (size_t)v2v_v2e_workspace_bytes(B, N), s);
        if (e0 != hipSuccess) return hip_fail(e0, "hipMemsetAsync(workspace)");
    }
    const dim3 grid((unsigned)nblocks);
    hipError_t e;
    e = v2v::launch_v2e(in_dtype == V2V_U8, vec4, bin_mode, rng_mode, out64, presum, a, grid, lds, s);
    return e == hipSuccess ? V2V_OK : hip_fail(e, "v2e kernel launch");
}

int v2v_events_to_voxel_hip(const double *ts, const int64_t *xs, const int64_t *ys, const double *ps, int64_t n, int mode,
                            int num_bins, int64_t H, int64_t W, double *out_voxel, uint64_t *dropped, void *stream)
{
    return v2v_events_to_voxel_segmented_hip(ts, xs, ys, ps, n, nullptr, 1, mode, num_bins, H, W, out_voxel, dropped, stream);
}

int v2v_events_to_voxel_segmented_hip(const double *ts, const int64_t *xs, const int64_t *ys, const double *ps, int64_t n,
                                      const int64_t *seg_offsets, int64_t n_segments, int mode, int num_bins, int64_t H, int64_t W,
                                      double *out_voxel, uint64_t *dropped, void *stream)
{
    if (!out_voxel || !dropped) return fail(V2V_ERR_NULL, "v2v_events_to_voxel: out_voxel/dropped is NULL");
    if (n < 0 || H < 1 || W < 1 || num_bins < 1 || num_bins > 255 || n_segments < 1) return fail(V2V_ERR_SHAPE, "need n>=0, H,W>=1, 1<=num_bins<=255, n_segments>=1");
    if (mode < V2V_EV_MAKE_VOXEL_DISCRETE || mode > V2V_EV_BILINEAR) return fail(V2V_ERR_MODE, "unknown event mode %d", mode);
    if (n > 0 && (!ts || !xs || !ys || !ps)) return fail(V2V_ERR_NULL, "v2v_events_to_voxel: event arrays are NULL");
    if (!aligned(out_voxel, 8) || !aligned(dropped, 8) || !aligned(ts, 8) || !aligned(xs, 8) || !aligned(ys, 8) || !aligned(ps, 8) ||
        (seg_offsets && !aligned(seg_offsets, 8)))
        return fail(V2V_ERR_ALIGN, "buffers must be 8-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(out_voxel, 0, sizeof(double) * (size_t)n_segments * num_bins * H * W, s);   // empty list -> zeros (testh5.py:63-64)
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(out_voxel)");
    e = hipMemsetAsync(dropped, 0, sizeof(uint64_t), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dropped)");
    if (n == 0) return V2V_OK;
    v2v::EventArgs a{};
    a.seg = seg_offsets; a.n_seg = n_segments;
    a.ts = ts; a.xs = xs; a.ys = ys; a.ps = ps; a.n = n; a.mode = mode; a.Tb = num_bins; a.H = H; a.W = W;
    a.out = out_voxel;
    a.dropped = reinterpret_cast<unsigned long long *>(dropped);
    const int64_t nblocks = (n + 255) / 256;
    if (nblocks > 0x7FFFFFFF) return fail(V2V_ERR_SHAPE, "too many events for one launch");
    hipLaunchKernelGGL(v2v::events_to_voxel_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, a);
    e = hipGetLastError();
    return e == hipSuccess ? V2V_OK : hip_fail(e, "events_to_voxel_kernel launch");
}

// Front-end dispatch: the LDS-tiled kernel for the training configuration (BGR source, gray output only, no shake),
// the per-pixel gather kernel otherwise.  LDS is sized for the largest crop rectangle the frame allows (the batch form
// keeps the rectangles on the device); tiles that do not fit fall back to global reads inside the kernel.
static hipError_t launch_frontend(const v2v::FrontendArgs &a_in, int64_t B, int64_t max_crop_before, hipStream_t s)
{
    v2v::FrontendArgs a = a_in;
    const int64_t frame_min = a.Hs < a.Ws ? a.Hs : a.Ws;
    a.cb_max = (int32_t)frame_min;            // what a device-resident table is clamped to (a crop cannot exceed the frame's short side)
#ifdef V2V_FRONTEND_FORCE_GATHER   // kernel-tuning builds only: always the gather kernel
    const bool force_gather = true;
#else
    const bool force_gather = false;
#endif
    // LDS-tiled kernel: BGR source, either color_mode 'gray' with gray output only, or 'gray_in_bgr_out' (resized BGR frames optional)
    const bool bgr_mode = !a.gray_first && a.Cs == 3;
    const bool tiled = !force_gather && a.Cs == 3 && (bgr_mode || !a.out_imgs) && (a.di != nullptr || (a.need_h == a.crop && a.need_w == a.crop)) &&
                       a.Hs <= 32767 && a.Ws <= 32767;                        // 16-bit source coordinates in the LDS coefficient tables
    if (tiled) {
        const int64_t cb_max = (max_crop_before > 0 && max_crop_before < frame_min) ? max_crop_before : frame_min;   // sizes the LDS tile; larger crops take the unstaged path
        const double s_max = (double)cb_max / (double)a.crop;
        // tile = (4 waves x rows_per_wave) rows x (64 lanes x cpl) columns; shrink until the worst-case source rectangle fits
        for (int cpl = a.crop > 128 ? 4 : 2; cpl >= 2; cpl -= 2) {
            const int64_t span_px = (int64_t)(64 * cpl * s_max) + 3;
            // BGR rows as they are (misalignment + span + tap reads); gray rows: one byte per source pixel in granules of 4 + the window reads
            const int64_t pitch = bgr_mode ? ((span_px * 3 + 12 + 15) / 16) * 16 + 16 : (((span_px + 3) / 4) * 4 + 12 + 15) / 16 * 16;
#ifndef V2V_FRONTEND_LDS_KB
#define V2V_FRONTEND_LDS_KB 48
#endif
            const int64_t budget = V2V_FRONTEND_LDS_KB * 1024 - v2v::tile_hdr_bytes(cpl);
            int rpw_cap = (int)((a.crop + 3) / 4);                        // no point in more rows per wave than the image has
            // rows per wave: gray rows are a third of the bytes, so twice the rows still leave >= 4 blocks per CU -- fewer source
            // rows are passed twice (tile borders) and the set-up is shared by more pixels (same box, config 4: 0.107 -> 0.092 ms)
#ifndef V2V_FRONTEND_RPW_CAP
#define V2V_FRONTEND_RPW_CAP (bgr_mode ? 4 : 8)
#endif
            if (rpw_cap > V2V_FRONTEND_RPW_CAP) rpw_cap = V2V_FRONTEND_RPW_CAP;
            for (int rpw = rpw_cap; rpw >= 1; rpw >>= 1) {
                const int64_t max_rows = (int64_t)(4 * rpw * s_max) + 3;
                if (max_rows * pitch > budget) continue;
                v2v::FrontendTileArgs ta{};
                ta.f = a;
                ta.pitch = (int32_t)pitch; ta.max_rows = (int32_t)max_rows; ta.rows_per_wave = rpw;
                ta.tiles_x = (a.crop + 64 * cpl - 1) / (64 * cpl);
                ta.tiles_y = (a.crop + 4 * rpw - 1) / (4 * rpw);
                // frames per block: the per-block set-up (extents, coefficient tables) is shared by the frames of a clip unless
                // the frames are shaken; as many as still leave 4 blocks for every CU slot (4 per CU)
#ifndef V2V_FRONTEND_FPB
#define V2V_FRONTEND_FPB 4
#endif
                int fpb = a.di ? 1 : V2V_FRONTEND_FPB;
#ifdef V2V_TUNING_KNOBS                                                           // tuning builds only (tools/frontend_time.py): never in the product launch path
                if (const char *e = getenv("V2V_FPB")) { if (!a.di && atoi(e) > 0) fpb = atoi(e); }
#endif
                while (fpb > 1 && (int64_t)ta.tiles_x * ta.tiles_y * ((a.N + fpb - 1) / fpb) * B < 4096) fpb >>= 1;
                ta.frames_per_block = fpb;
                const int64_t nblocks = (int64_t)ta.tiles_x * ta.tiles_y * ((a.N + fpb - 1) / fpb);
                if (nblocks > 0x7FFFFFFF) break;
                const size_t lds = (size_t)(v2v::tile_hdr_bytes(cpl) + max_rows * pitch);
                if (bgr_mode) {
                    if (cpl == 4) hipLaunchKernelGGL((v2v::frontend_tile_kernel<4, true>), dim3((unsigned)nblocks, (unsigned)B), dim3(256), lds, s, ta);
                    else hipLaunchKernelGGL((v2v::frontend_tile_kernel<2, true>), dim3((unsigned)nblocks, (unsigned)B), dim3(256), lds, s, ta);
                } else if (cpl == 4) hipLaunchKernelGGL(v2v::frontend_tile_kernel<4>, dim3((unsigned)nblocks, (unsigned)B), dim3(256), lds, s, ta);
                else hipLaunchKernelGGL(v2v::frontend_tile_kernel<2>, dim3((unsigned)nblocks, (unsigned)B), dim3(256), lds, s, ta);
                return hipGetLastError();
            }
        }
    }
    const int64_t nblocks = ((int64_t)a.N * a.crop * ((a.crop + v2v::kFrontPx - 1) / v2v::kFrontPx) + 255) / 256;
    if (nblocks > 0x7FFFFFFF) return hipErrorInvalidValue;
    hipLaunchKernelGGL(v2v::frontend_kernel, dim3((unsigned)nblocks, (unsigned)B), dim3(256), 0, s, a);
    return hipGetLastError();
}

int v2v_frontend_hip(const uint8_t *src, int64_t T, int64_t Hs, int64_t Ws, int64_t Cs, int64_t min_i, int64_t min_j,
                     int64_t crop_before, int64_t need_h, int64_t need_w, int64_t crop, int flip, int gray_first,
                     const int32_t *frame_idx, int64_t N, const int32_t *shake_di, const int32_t *shake_dj, uint8_t *out_imgs,
                     uint8_t *out_gray, void *stream)
{
    if (!src || !frame_idx || !out_gray) return fail(V2V_ERR_NULL, "v2v_frontend_hip: src/frame_idx/out_gray is NULL");
    if (T < 1 || N < 0 || Hs < 1 || Ws < 1 || (Cs != 1 && Cs != 3)) return fail(V2V_ERR_SHAPE, "need T>=1, N>=0, Hs,Ws>=1, Cs in {1,3}");
    if (crop_before < 1 || crop < 1 || need_h < crop || need_w < crop) return fail(V2V_ERR_SHAPE, "need crop_before>=1 and need_h,need_w >= crop >= 1");
    if (gray_first < 0 || gray_first > 2) return fail(V2V_ERR_MODE, "gray_first: 0 none, 1 BGR2GRAY as OpenCV >= 4.0 (15-bit), 2 as OpenCV 2.x/3.x (14-bit)");
    if (min_i < 0 || min_j < 0 || min_i + crop_before > Hs || min_j + crop_before > Ws) return fail(V2V_ERR_SHAPE, "crop rectangle outside the frame");
    if ((shake_di == nullptr) != (shake_dj == nullptr)) return fail(V2V_ERR_NULL, "shake_di and shake_dj must both be given or both be NULL");
    if (!shake_di && (need_h != crop || need_w != crop)) return fail(V2V_ERR_SHAPE, "need_h/need_w differ from crop but no shake offsets");
    if (!aligned(frame_idx, 4) || (shake_di && (!aligned(shake_di, 4) || !aligned(shake_dj, 4)))) return fail(V2V_ERR_ALIGN, "index arrays must be 4-byte aligned");
    if (N == 0) return V2V_OK;
    v2v::FrontendArgs a{};
    a.src = src; a.T = (int32_t)T; a.Hs = (int32_t)Hs; a.Ws = (int32_t)Ws; a.Cs = (int32_t)Cs;
    a.min_i = (int32_t)min_i; a.min_j = (int32_t)min_j; a.crop_before = (int32_t)crop_before;
    a.need_h = (int32_t)need_h; a.need_w = (int32_t)need_w; a.crop = (int32_t)crop;
    a.flip = flip ? 1 : 0; a.gray_first = gray_first;
    a.frame_idx = frame_idx; a.di = shake_di; a.dj = shake_dj; a.N = (int32_t)N;
    a.Cout = (gray_first || Cs == 1) ? 1 : 3;
    a.out_imgs = out_imgs; a.out_gray = out_gray;
    a.src_end = src + T * Hs * Ws * Cs;
    const int64_t total = N * crop * ((crop + v2v::kFrontPx - 1) / v2v::kFrontPx);
    if ((total + 255) / 256 > 0x7FFFFFFF) return fail(V2V_ERR_SHAPE, "grid too large");
    const hipError_t e = launch_frontend(a, 1, crop_before, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "frontend kernel launch");
}

int v2v_frontend_batch_hip(const uint8_t *src, int64_t B, int64_t T, int64_t Hs, int64_t Ws, int64_t Cs, const int32_t *clip_table,
                           int64_t max_crop_before, int64_t crop, int gray_first, const int32_t *frame_idx, int64_t N,
                           uint8_t *out_imgs, uint8_t *out_gray, void *stream)
{
    if (!src || !frame_idx || !out_gray || !clip_table) return fail(V2V_ERR_NULL, "v2v_frontend_batch_hip: src/clip_table/frame_idx/out_gray is NULL");
    if (B < 0 || T < 1 || N < 0 || Hs < 1 || Ws < 1 || (Cs != 1 && Cs != 3) || crop < 1) return fail(V2V_ERR_SHAPE, "need B>=0, T>=1, N>=0, Hs,Ws,crop>=1, Cs in {1,3}");
    if (!aligned(frame_idx, 4) || !aligned(clip_table, 4)) return fail(V2V_ERR_ALIGN, "index arrays must be 4-byte aligned");
    if (gray_first < 0 || gray_first > 2) return fail(V2V_ERR_MODE, "gray_first: 0 none, 1 BGR2GRAY as OpenCV >= 4.0 (15-bit), 2 as OpenCV 2.x/3.x (14-bit)");
    if (B == 0 || N == 0) return V2V_OK;
    v2v::FrontendArgs a{};
    a.src = src; a.T = (int32_t)T; a.Hs = (int32_t)Hs; a.Ws = (int32_t)Ws; a.Cs = (int32_t)Cs;
    a.need_h = a.need_w = a.crop = (int32_t)crop;
    a.gray_first = gray_first;
    a.frame_idx = frame_idx; a.N = (int32_t)N;
    a.Cout = (gray_first || Cs == 1) ? 1 : 3;
    a.out_imgs = out_imgs; a.out_gray = out_gray;
    a.src_end = src + B * T * Hs * Ws * Cs;
    a.clip_table = clip_table;                     // the crop rectangles are validated by the caller (device-resident table)
    const int64_t nblocks = (N * crop * ((crop + v2v::kFrontPx - 1) / v2v::kFrontPx) + 255) / 256;
    if (nblocks > 0x7FFFFFFF || B > 65535) return fail(V2V_ERR_SHAPE, "grid too large");
    const hipError_t e = launch_frontend(a, B, max_crop_before, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "frontend kernel (batch) launch");
}

int64_t v2v_postops_workspace_bytes(int64_t B)
{
    if (B < 0) return V2V_ERR_SHAPE;
    return B * 2 * ((int64_t)sizeof(v2v::SelectState) + (int64_t)v2v::kSelBins * (int64_t)sizeof(unsigned int)) + B * 16;
}

int v2v_normalize_pad_hip(const float *voxel, int64_t B, int64_t planes, int64_t H, int64_t W, int normalize, int pad_to,
                          float *out, void *workspace, void *stream)
{
    return v2v_normalize_pad_ex_hip(voxel, B, planes, H, W, H, W, normalize ? V2V_NORM_RADIX : V2V_NORM_NONE, pad_to, out, workspace, stream);
}

// the k-th value selection of normalize_batch_voxel -> scales_out[b] = {neg_max, pos_max}; workspace as v2v_postops_workspace_bytes(B)
static int select_scales(const float *voxel, int64_t B, int64_t planes, int64_t H, int64_t W, int64_t H_in, int64_t W_in, int method, void *workspace,
                         float *scales_out, hipStream_t s)
{
    const int64_t per_sample = planes * H * W;
    // torch.kthvalue is 1-based: max_k = int(0.99*M), min_k = int(0.01*M) (model/train_utils.py:153-154)
    const int64_t max_k = (int64_t)(0.99 * (double)per_sample), min_k = (int64_t)(0.01 * (double)per_sample);
    if (min_k < 1 || max_k < 1) return fail(V2V_ERR_SHAPE, "k-th value undefined: fewer than 100 elements per sample (torch.kthvalue would raise)");
    v2v::SelectState *st = static_cast<v2v::SelectState *>(workspace);
    if (method == V2V_NORM_RADIX) {
        unsigned int *hist = reinterpret_cast<unsigned int *>(st + B * 2);
        hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned int) * (size_t)B * 2 * v2v::kSelBins, s);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(hist)");
        hipLaunchKernelGGL(v2v::select_init_kernel, dim3((unsigned)((B * 2 + 255) / 256)), dim3(256), 0, s, st, B * 2,
                           (uint64_t)(min_k - 1), (uint64_t)(max_k - 1));
        const int64_t want = (2048 + B - 1) / B, most = (per_sample + 256 * 8 - 1) / (256 * 8);      // few, fat workgroups (see the counting path)
        const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, most));
        const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
        for (int p = 0; p < 3; ++p) {
            if (per_sample % 4 == 0 && (reinterpret_cast<uintptr_t>(voxel) & 15u) == 0)
                hipLaunchKernelGGL(v2v::select_hist4_kernel, dim3(gx, (unsigned)B), dim3(256), 0, s, voxel, per_sample, st, hist, shifts[p], bits[p]);
            else
                hipLaunchKernelGGL(v2v::select_hist_kernel, dim3(gx, (unsigned)B), dim3(256), 0, s, voxel, per_sample, st, hist, shifts[p], bits[p]);
            hipLaunchKernelGGL(v2v::select_pick_kernel, dim3((unsigned)(B * 2)), dim3(256), 0, s, st, hist, shifts[p], bits[p]);
        }
        hipLaunchKernelGGL(v2v::state_scales_kernel, dim3((unsigned)((B * 2 + 255) / 256)), dim3(256), 0, s, st, scales_out, B * 2);
    } else {
        static_assert(v2v::kCntBins <= 2 * v2v::kSelBins, "the counting histogram reuses the radix workspace");
        unsigned int *hist = reinterpret_cast<unsigned int *>(st + B * 2);
        unsigned int *bad = hist + (size_t)B * 2 * v2v::kSelBins;
        hipError_t e = hipMemsetAsync(hist, 0, sizeof(unsigned int) * ((size_t)B * 2 * v2v::kSelBins + (size_t)B), s);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(hist)");
        const int64_t per_in = planes * H_in * W_in;
        const unsigned gx = (unsigned)std::min<int64_t>((per_in + 256 * 8 - 1) / (256 * 8), 512);
        // 16-byte loads + wave-aggregated histogram updates when a sample is whole float4s (every layout the simulator writes)
        if (per_in % 4 == 0 && (reinterpret_cast<uintptr_t>(voxel) & 15u) == 0) {
            // few, fat workgroups: every workgroup ends with one global atomic per non-empty bin of ITS sample, and hundreds of them
            // on the same few words serialise at the L2 (9,600 workgroups: 78 us for 157 MB; ~2,048: see tools/postops_time.py)
            const int64_t want = (2048 + B - 1) / B, most = (per_in / 4 + 511) / 512;
            const unsigned gx4 = (unsigned)std::max<int64_t>(1, std::min<int64_t>(want, most));
            hipLaunchKernelGGL(v2v::count_hist4_kernel, dim3(gx4, (unsigned)B), dim3(256), 0, s, voxel, per_in, hist, bad);
        } else {
            hipLaunchKernelGGL(v2v::count_hist_kernel, dim3(gx, (unsigned)B), dim3(256), 0, s, voxel, per_in, hist, bad);
        }
        hipLaunchKernelGGL(v2v::count_pick_kernel, dim3((unsigned)(B * 2)), dim3(64), 0, s, hist, (int64_t)v2v::kCntBins, bad, (int64_t)1, B * 2,
                           (uint64_t)(per_in - per_sample), (uint64_t)(min_k - 1), (uint64_t)(max_k - 1), 0, (uint64_t)per_sample, scales_out);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? V2V_OK : hip_fail(e, "k-th value selection launch");
}

int v2v_voxel_scales_select_hip(const float *voxel, int64_t B, int64_t planes, int64_t H, int64_t W, int64_t H_in, int64_t W_in, int method,
                                float *scales, void *workspace, void *stream)
{
    if (!voxel || !scales || !workspace) return fail(V2V_ERR_NULL, "v2v_voxel_scales_select_hip: voxel/scales/workspace is NULL");
    if (B < 0 || planes < 1 || H < 1 || W < 1 || H_in < H || W_in < W) return fail(V2V_ERR_SHAPE, "need B>=0, planes,H,W>=1, H_in>=H, W_in>=W");
    if (B > 65535) return fail(V2V_ERR_SHAPE, "more than 65535 samples in one call (one grid row per sample): split the batch");
    if (method != V2V_NORM_RADIX && method != V2V_NORM_COUNT) return fail(V2V_ERR_MODE, "method must be V2V_NORM_RADIX or V2V_NORM_COUNT");
    if (method == V2V_NORM_RADIX && (H_in != H || W_in != W)) return fail(V2V_ERR_MODE, "the radix select reads unpadded input; use V2V_NORM_COUNT for padded input");
    if (!aligned(voxel, 4) || !aligned(scales, 4) || !aligned(workspace, 16)) return fail(V2V_ERR_ALIGN, "buffers misaligned");
    if (B == 0) return V2V_OK;
    return select_scales(voxel, B, planes, H, W, H_in, W_in, method, workspace, scales, static_cast<hipStream_t>(stream));
}

int v2v_normalize_pad_ex_hip(const float *voxel, int64_t B, int64_t planes, int64_t H, int64_t W, int64_t H_in, int64_t W_in, int method,
                             int pad_to, float *out, void *workspace, void *stream)
{
    if (!voxel || !out) return fail(V2V_ERR_NULL, "v2v_normalize_pad_hip: voxel/out is NULL");
    if (B < 0 || planes < 1 || H < 1 || W < 1 || pad_to < 1 || H_in < H || W_in < W) return fail(V2V_ERR_SHAPE, "need B>=0, planes,H,W,pad_to>=1, H_in>=H, W_in>=W");
    if (B > 65535) return fail(V2V_ERR_SHAPE, "more than 65535 samples in one call (one grid row per sample): split the batch");
    if (method < V2V_NORM_NONE || method > V2V_NORM_COUNT) return fail(V2V_ERR_MODE, "unknown normalisation method %d", method);
    const bool normalize = method != V2V_NORM_NONE;
    if (normalize && !workspace) return fail(V2V_ERR_NULL, "normalisation needs a workspace of v2v_postops_workspace_bytes(B)");
    if (method == V2V_NORM_RADIX && (H_in != H || W_in != W)) return fail(V2V_ERR_MODE, "the radix select reads unpadded input; use V2V_NORM_COUNT for padded input");
    if (!aligned(voxel, 4) || !aligned(out, 4) || (workspace && !aligned(workspace, 16))) return fail(V2V_ERR_ALIGN, "buffers misaligned");
    if (B == 0) return V2V_OK;
    const int Hp = (int)((H + pad_to - 1) / pad_to * pad_to), Wp = (int)((W + pad_to - 1) / pad_to * pad_to);
    if (out == voxel && (Hp != H_in || Wp != W_in)) return fail(V2V_ERR_SHAPE, "in-place normalisation needs identical input and output layouts");
    // workspace: [SelectState x 2B][histograms 2B x 2048 words][bad flags B words][scales 2B floats]
    float *scales = nullptr;
    if (normalize) {
        scales = reinterpret_cast<float *>(reinterpret_cast<unsigned int *>(static_cast<v2v::SelectState *>(workspace) + B * 2) + (size_t)B * 2 * v2v::kSelBins + (size_t)B);
        const int rc = select_scales(voxel, B, planes, H, W, H_in, W_in, method, workspace, scales, static_cast<hipStream_t>(stream));
        if (rc != V2V_OK) return rc;
    }
    return v2v_voxel_apply_scales_hip(voxel, B, planes, H, W, H_in, W_in, pad_to, scales, out, stream);
}

static int events_f32_launch(const float *ts, const double *ts64, const int64_t *xs, const int64_t *ys, const float *ps, int64_t n,
                             const int64_t *seg, int64_t n_seg, int min_events, int discrete, int num_bins, int64_t H, int64_t W,
                             float *out_voxel, uint64_t *dropped, void *stream, const char *who)
{
    if (!out_voxel || !dropped) return fail(V2V_ERR_NULL, "%s: out_voxel/dropped is NULL", who);
    if (n < 0 || H < 1 || W < 1 || num_bins < 1 || n_seg < 1) return fail(V2V_ERR_SHAPE, "need n>=0, H,W>=1, num_bins>=1, n_segments>=1");
    if (n > 0 && ((!ts && !ts64) || !xs || !ys || !ps)) return fail(V2V_ERR_NULL, "%s: event arrays are NULL", who);
    if (!aligned(out_voxel, 4) || !aligned(dropped, 8) || !aligned(ts, 4) || !aligned(ts64, 8) || !aligned(xs, 8) || !aligned(ys, 8) || !aligned(ps, 4) ||
        !aligned(seg, 8))
        return fail(V2V_ERR_ALIGN, "buffers not aligned to their element size");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(out_voxel, 0, sizeof(float) * (size_t)n_seg * num_bins * H * W, s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(out_voxel)");
    e = hipMemsetAsync(dropped, 0, sizeof(uint64_t), s);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(dropped)");
    if (n == 0) return V2V_OK;
    v2v::EventArgsF32 a{};
    a.ts = ts; a.ts64 = ts64; a.seg = seg; a.n_seg = n_seg; a.min_events = min_events;
    a.xs = xs; a.ys = ys; a.ps = ps; a.n = n; a.discrete = discrete ? 1 : 0; a.Tb = num_bins; a.H = H; a.W = W;
    a.out = out_voxel;
    a.dropped = reinterpret_cast<unsigned long long *>(dropped);
    const int64_t nblocks = (n + 255) / 256;
    if (nblocks > 0x7FFFFFFF) return fail(V2V_ERR_SHAPE, "too many events for one launch");
    hipLaunchKernelGGL(v2v::events_to_voxel_f32_kernel, dim3((unsigned)nblocks), dim3(256), 0, s, a);
    e = hipGetLastError();
    return e == hipSuccess ? V2V_OK : hip_fail(e, "events_to_voxel_f32_kernel launch");
}

int v2v_events_to_voxel_f32_hip(const float *ts, const int64_t *xs, const int64_t *ys, const float *ps, int64_t n, int discrete,
                                int num_bins, int64_t H, int64_t W, float *out_voxel, uint64_t *dropped, void *stream)
{
    return events_f32_launch(ts, nullptr, xs, ys, ps, n, nullptr, 1, 0, discrete, num_bins, H, W, out_voxel, dropped, stream,
                             "v2v_events_to_voxel_f32_hip");
}

int v2v_events_to_voxel_f32_segmented_hip(const double *ts, const int64_t *xs, const int64_t *ys, const float *ps, int64_t n,
                                          const int64_t *seg_offsets, int64_t n_segments, int min_events, int discrete, int num_bins,
                                          int64_t H, int64_t W, float *out_voxel, uint64_t *dropped, void *stream)
{
    if (!seg_offsets) return fail(V2V_ERR_NULL, "v2v_events_to_voxel_f32_segmented_hip: seg_offsets is NULL");
    return events_f32_launch(nullptr, ts, xs, ys, ps, n, seg_offsets, n_segments, min_events, discrete, num_bins, H, W, out_voxel, dropped,
                             stream, "v2v_events_to_voxel_f32_segmented_hip");
}

int v2v_convlstm_packed_bytes(int64_t C, uint64_t *bytes)
{
    if (!bytes) return fail(V2V_ERR_NULL, "v2v_convlstm_packed_bytes: bytes is NULL");
    if (C < 64 || C % 64 != 0) return fail(V2V_ERR_SHAPE, "ConvLSTM kernel needs C %% 64 == 0 (got %lld)", (long long)C);
    *bytes = (uint64_t)4 * C * 2 * C * 9 * 2;
    return V2V_OK;
}

int v2v_convlstm_pack_weights_hip(const float *gates_weight, int64_t C, void *packed, void *stream)
{
    if (!gates_weight || !packed) return fail(V2V_ERR_NULL, "v2v_convlstm_pack_weights_hip: gates_weight/packed is NULL");
    if (C < 64 || C % 64 != 0 || C > 4096) return fail(V2V_ERR_SHAPE, "ConvLSTM kernel needs C %% 64 == 0, C <= 4096 (got %lld)", (long long)C);
    if (!aligned(gates_weight, 4) || !aligned(packed, 16)) return fail(V2V_ERR_ALIGN, "gates_weight needs 4-byte, packed 16-byte alignment");
    const hipError_t e = v2v::launch_convlstm_pack(gates_weight, static_cast<uint16_t *>(packed), (int)C, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "convlstm_pack_kernel launch");
}

int v2v_convlstm_step_hip(const void *x, const void *h_prev, const float *c_prev, const void *packed, const float *gates_bias,
                          int64_t B, int64_t H, int64_t W, int64_t C, void *h_state, float *c_state, void *h_nchw, int h_nchw_dtype, int tile_rows, void *stream)
{
    if (h_nchw && h_nchw_dtype != V2V_F32 && h_nchw_dtype != V2V_BF16) return fail(V2V_ERR_DTYPE, "h_nchw_dtype must be V2V_F32 or V2V_BF16");
    if (!x || !packed || !gates_bias || !h_state || !c_state) return fail(V2V_ERR_NULL, "v2v_convlstm_step_hip: x/packed/gates_bias/h_state/c_state is NULL");
    if (B < 1 || H < 1 || W < 1 || C < 64 || C % 64 != 0 || C > 4096) return fail(V2V_ERR_SHAPE, "need B,H,W >= 1 and C %% 64 == 0, C <= 4096");
    if (tile_rows != 0 && tile_rows != 64 && tile_rows != 128 && tile_rows != 256) return fail(V2V_ERR_PARAM, "tile_rows must be 0 (auto), 64, 128 or 256");
    if ((H * W) % 4 != 0 || B * H * W * C > 0x7FFFFFFFLL)
        return fail(V2V_ERR_SHAPE, "ConvLSTM kernel needs (H*W) %% 4 == 0 and B*H*W*C < 2^31 (got %lldx%lldx%lldx%lld)",
                    (long long)B, (long long)H, (long long)W, (long long)C);
    if (h_state == h_prev || h_state == x) return fail(V2V_ERR_PARAM, "h_state must not alias h_prev or x (neighbouring tiles read them)");
    if (!aligned(x, 16) || !aligned(h_prev, 16) || !aligned(packed, 16) || !aligned(h_state, 2) || !aligned(c_prev, 4) || !aligned(c_state, 4) ||
        !aligned(gates_bias, 4) || !aligned(h_nchw, 16))
        return fail(V2V_ERR_ALIGN, "x/h_prev/packed/h_nchw need 16-byte alignment");
    v2v::ConvLstmArgs a{};
    a.x = static_cast<const uint16_t *>(x); a.h_prev = static_cast<const uint16_t *>(h_prev); a.c_prev = c_prev;
    a.wp = static_cast<const uint16_t *>(packed); a.bias = gates_bias;
    a.h_state = static_cast<uint16_t *>(h_state); a.c_state = c_state; a.h_nchw = h_nchw; a.h_nchw_bf16 = h_nchw_dtype == V2V_BF16;
    a.B = (int)B; a.H = (int)H; a.W = (int)W; a.C = (int)C;
    const hipError_t e = v2v::launch_convlstm_step(a, tile_rows, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "convlstm_step_kernel launch");
}

int64_t v2v_conv_packed_elems(int64_t Cin, int64_t Cout, int ks)
{
    if ((ks != 3 && ks != 5) || Cout < 1 || Cout > 4096 || v2v::conv_tile_cols((int)Cout) == 0) return -1;
    if (Cin == 32) return (Cout == 64 || Cout == 128) ? Cout * ((ks * ks + 1) / 2) * 64 : -1;      // two taps per 64-wide chunk
    if (Cin < 64 || Cin % 64 != 0 || Cin > 4096) return -1;
    return Cout * Cin * ks * ks;
}

int v2v_conv_pack_weights_hip(const float *weight, int64_t Cin, int64_t Cout, int ks, void *packed, void *stream)
{
    if (!weight || !packed) return fail(V2V_ERR_NULL, "v2v_conv_pack_weights_hip: weight/packed is NULL");
    if (v2v_conv_packed_elems(Cin, Cout, ks) < 0)
        return fail(V2V_ERR_SHAPE, "conv kernel needs Cin %% 64 == 0 (or Cin 32 with Cout 64 / 128), Cout in {32, 64, 128} or a multiple of 256, ks 3 or 5 (got %lld -> %lld, ks %d)",
                    (long long)Cin, (long long)Cout, ks);
    if (!aligned(weight, 4) || !aligned(packed, 16)) return fail(V2V_ERR_ALIGN, "weight needs 4-byte, packed 16-byte alignment");
    const hipError_t e = v2v::launch_conv_pack(weight, static_cast<uint16_t *>(packed), (int)Cin, (int)Cout, ks, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "conv_pack_kernel launch");
}

int v2v_conv_nhwc_hip(const void *x, const void *packed, const float *bias, const void *residual, int relu, int64_t B, int64_t Hin,
                      int64_t Win, int64_t Cin, int64_t Cout, int ks, int stride, void *out, int tile_rows, void *stream)
{
    if (!x || !packed || !bias || !out) return fail(V2V_ERR_NULL, "v2v_conv_nhwc_hip: x/packed/bias/out is NULL");
    if ((ks != 3 && ks != 5) || (stride != 1 && stride != 2)) return fail(V2V_ERR_PARAM, "ks must be 3 or 5, stride 1 or 2");
    if (B < 1 || Hin < 1 || Win < 1 || v2v_conv_packed_elems(Cin, Cout, ks) < 0)
        return fail(V2V_ERR_SHAPE, "need B,H,W >= 1, Cin %% 64 == 0 (or Cin 32 with Cout 64 / 128), Cout in {32, 64, 128} or a multiple of 256");
    const int64_t H = (Hin - 1) / stride + 1, W = (Win - 1) / stride + 1;          // output size for pad = ks / 2
    if (tile_rows != 0 && tile_rows != 128 && tile_rows != 256 && tile_rows != 16 && (Cout % 256 != 0 || (tile_rows != 32 && tile_rows != 64)))
        return fail(V2V_ERR_PARAM, "tile_rows must be 0 (auto), 128 or 256 (and 32 or 64 for Cout %% 256 == 0)");
    const bool halo_fits = Cin % 64 == 0 && Cout % 256 != 0 && stride == 1 && H % 16 == 0 && W % 16 == 0 && (ks == 3 || Cout <= 64);   // see launch_conv_nhwc
    const bool halo = halo_fits && (tile_rows == 16 || (tile_rows == 0 && ks == 5));
    if (tile_rows == 16 && !halo) return fail(V2V_ERR_PARAM, "tile_rows 16 (halo tiles) needs stride 1, H and W multiples of 16 and Cout 32 / 64 (128 for 3x3)");
    if ((H * W) % 4 != 0 || B * Hin * Win * Cin > 0x7FFFFFFFLL || B * H * W * Cout > 0x7FFFFFFFLL)
        return fail(V2V_ERR_SHAPE, "conv kernel needs (Hout*Wout) %% 4 == 0 and tensors below 2^31 elements");
    if (out == x) return fail(V2V_ERR_PARAM, "out must not alias x (neighbouring tiles read it)");
    if (!aligned(x, 16) || !aligned(packed, 16) || !aligned(out, 2) || !aligned(residual, 2) || !aligned(bias, 4))
        return fail(V2V_ERR_ALIGN, "x/packed need 16-byte alignment");
    v2v::ConvLstmArgs a{};
    a.x = static_cast<const uint16_t *>(x); a.wp = static_cast<const uint16_t *>(packed); a.bias = bias;
    a.residual = static_cast<const uint16_t *>(residual); a.out_nhwc = static_cast<uint16_t *>(out);
    a.n_cols = (int)Cout; a.relu = relu ? 1 : 0; a.ks = ks; a.stride = stride; a.Hin = (int)Hin; a.Win = (int)Win;
    a.B = (int)B; a.H = (int)H; a.W = (int)W; a.C = (int)Cin;
    const hipError_t e = v2v::launch_conv_nhwc(a, tile_rows, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "conv (convlstm_step_kernel, EPI = 1) launch");
}

int64_t v2v_conv_head_packed_elems(int ks) { return (ks == 3 || ks == 5) ? (int64_t)((ks * ks + 7) / 8) * 32 * 64 : -1; }

int v2v_conv_head_pack_weights_hip(const float *weight, int64_t Cin, int ks, void *packed, void *stream)
{
    if (!weight || !packed) return fail(V2V_ERR_NULL, "v2v_conv_head_pack_weights_hip: weight/packed is NULL");
    if (Cin < 1 || Cin > 8 || (ks != 3 && ks != 5)) return fail(V2V_ERR_SHAPE, "need 1 <= Cin <= 8 and ks 3 or 5 (32 output channels)");
    if (!aligned(packed, 16)) return fail(V2V_ERR_ALIGN, "packed needs 16-byte alignment");
    const hipError_t e = v2v::launch_conv_head_pack(weight, static_cast<uint16_t *>(packed), (int)Cin, ks, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "conv_head_pack_kernel launch");
}

int v2v_to_nhwc8_bf16_hip(const float *src, int64_t stride_b, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t B, int64_t C,
                          int64_t H, int64_t W, void *dst, void *stream)
{
    return v2v_to_nhwc8_bf16_scaled_hip(src, stride_b, stride_c, stride_h, stride_w, B, C, H, W, nullptr, dst, stream);
}

int v2v_to_nhwc8_bf16_scaled_hip(const float *src, int64_t stride_b, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t B, int64_t C,
                                 int64_t H, int64_t W, const float *scales, void *dst, void *stream)
{
    if (!src || !dst) return fail(V2V_ERR_NULL, "v2v_to_nhwc8_bf16_hip: src/dst is NULL");
    if (B < 1 || C < 1 || C > 8 || H < 1 || W < 1 || B * H * W > 0x7FFFFFFFLL) return fail(V2V_ERR_SHAPE, "need B,H,W >= 1, 1 <= C <= 8, B*H*W < 2^31");
    if (!aligned(dst, 16) || !aligned(src, 4) || (scales && !aligned(scales, 4))) return fail(V2V_ERR_ALIGN, "dst needs 16-byte alignment");
    const hipError_t e = v2v::launch_to_nhwc8_bf16(src, stride_b, stride_c, stride_h, stride_w, static_cast<uint16_t *>(dst), (int)B, (int)C, (int)H, (int)W,
                                                   scales, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "to_nhwc8_bf16_kernel launch");
}

int v2v_conv_head_nhwc_hip(const void *x8, const void *packed, const float *bias, int relu, int64_t B, int64_t H, int64_t W, int ks, void *out, void *stream)
{
    if (!x8 || !packed || !bias || !out) return fail(V2V_ERR_NULL, "v2v_conv_head_nhwc_hip: x8/packed/bias/out is NULL");
    if (ks != 3 && ks != 5) return fail(V2V_ERR_PARAM, "ks must be 3 or 5");
    if (B < 1 || H < 16 || W < 16 || H % 16 != 0 || W % 16 != 0 || B * H * W * 32 > 0x7FFFFFFFLL)
        return fail(V2V_ERR_SHAPE, "need B >= 1, H and W multiples of 16, output below 2^31 elements");
    if (!aligned(x8, 16) || !aligned(packed, 16) || !aligned(out, 2) || !aligned(bias, 4)) return fail(V2V_ERR_ALIGN, "x8/packed need 16-byte alignment");
    const hipError_t e = v2v::launch_conv_head(static_cast<const uint16_t *>(x8), static_cast<const uint16_t *>(packed), bias, static_cast<uint16_t *>(out),
                                               (int)B, (int)H, (int)W, ks, relu ? 1 : 0, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "conv_head_kernel launch");
}

int v2v_conv1x1_nhwc_hip(const void *x, const void *skip, const float *weight, const float *bias, int64_t M, int64_t C, int64_t Cout,
                         void *out, int out_dtype, void *stream)
{
    if (!x || !weight || !bias || !out) return fail(V2V_ERR_NULL, "v2v_conv1x1_nhwc_hip: x/weight/bias/out is NULL");
    if (out_dtype != V2V_F32 && out_dtype != V2V_BF16) return fail(V2V_ERR_DTYPE, "out_dtype must be V2V_F32 or V2V_BF16");
    if (M < 1 || C < 8 || C > 512 || (C & (C - 1)) != 0 || Cout < 1 || Cout > 3 || M * (C / 8) > 0x7FFFFFFFLL * 256)
        return fail(V2V_ERR_SHAPE, "need M >= 1, C a power of two in 8..512, Cout 1..3");
    if (!aligned(x, 16) || !aligned(skip, 16) || !aligned(out, out_dtype == V2V_F32 ? 4 : 2)) return fail(V2V_ERR_ALIGN, "x/skip need 16-byte alignment");
    const hipError_t e = v2v::launch_conv1x1_nhwc(static_cast<const uint16_t *>(x), static_cast<const uint16_t *>(skip), weight, bias, out,
                                                  out_dtype == V2V_BF16, M, (int)C, (int)Cout, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "conv1x1_nhwc_kernel launch");
}

int v2v_upsample2x_nhwc_hip(const void *x, const void *skip, int64_t B, int64_t H, int64_t W, int64_t C, void *out, void *stream)
{
    if (!x || !out) return fail(V2V_ERR_NULL, "v2v_upsample2x_nhwc_hip: x/out is NULL");
    if (B < 1 || H < 1 || W < 1 || C < 8 || C % 8 != 0 || B * H * W * C > (1LL << 36))
        return fail(V2V_ERR_SHAPE, "need B,H,W >= 1, C %% 8 == 0 and an input below 2^36 elements");
    if (!aligned(x, 16) || !aligned(out, 16) || !aligned(skip, 16)) return fail(V2V_ERR_ALIGN, "x/skip/out need 16-byte alignment");
    if (out == x || out == skip) return fail(V2V_ERR_PARAM, "out must not alias x or skip");
    const hipError_t e = v2v::launch_upsample2x_nhwc(static_cast<const uint16_t *>(x), static_cast<const uint16_t *>(skip), static_cast<uint16_t *>(out),
                                                     (int)B, (int)H, (int)W, (int)C, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "upsample2x_nhwc_bf16_kernel launch");
}

int v2v_conv3x3_pack_weights_hip(const float *weight, int64_t Cin, int64_t Cout, void *packed, void *stream)
{
    return v2v_conv_pack_weights_hip(weight, Cin, Cout, 3, packed, stream);
}

int v2v_conv3x3_nhwc_hip(const void *x, const void *packed, const float *bias, const void *residual, int relu, int64_t B, int64_t H,
                         int64_t W, int64_t Cin, int64_t Cout, void *out, int tile_rows, void *stream)
{
    return v2v_conv_nhwc_hip(x, packed, bias, residual, relu, B, H, W, Cin, Cout, 3, 1, out, tile_rows, stream);
}

int v2v_nchw_to_nhwc_bf16_hip(const void *src, int src_dtype, int64_t B, int64_t C, int64_t H, int64_t W, int relu, void *dst, void *stream)
{
    if (!src || !dst) return fail(V2V_ERR_NULL, "v2v_nchw_to_nhwc_bf16_hip: src/dst is NULL");
    if (src_dtype != V2V_F32 && src_dtype != V2V_BF16) return fail(V2V_ERR_DTYPE, "src_dtype must be V2V_F32 or V2V_BF16");
    if (B < 1 || H < 1 || W < 1 || C < 64 || C % 64 != 0 || (H * W) % 64 != 0 || B * (C / 64) * (H * W / 64) > 0x7FFFFFFFLL)
        return fail(V2V_ERR_SHAPE, "need B,H,W >= 1, C %% 64 == 0, (H*W) %% 64 == 0");
    if (!aligned(src, src_dtype == V2V_F32 ? 4 : 2) || !aligned(dst, 2)) return fail(V2V_ERR_ALIGN, "buffers misaligned");
    const hipError_t e = v2v::launch_nchw_to_nhwc_bf16(src, src_dtype == V2V_BF16, static_cast<uint16_t *>(dst), (int)B, (int)C, (int)(H * W),
                                                       relu ? 1 : 0, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? V2V_OK : hip_fail(e, "nchw_to_nhwc_bf16_kernel launch");
}

}  // extern "C"
