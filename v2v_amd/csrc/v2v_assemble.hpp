// v2v_assemble.hpp -- the `frame` tensor of a batch, built on the device from the uint8 clips the simulator reads (gfx950).
//
// Replaces the per-sample assembly of WebvidDatasetV2.__getitem__ (data/v2v_datasets.py:329-338, 352):
//     frame[l] = torch.tensor(all_imgs[idx_l]).float().permute(2, 0, 1) / 255          idx_l = (l+1)*frames_per_img (or l*..)
// for the whole batch: out[b, l, c, y, x] = float(src[b, pick[l], y, x, c]) / 255.0f.  The division is IEEE float32 division
// (the library is built without fast-math; v_div_scale / v_div_fmas / v_div_fixup), i.e. torch's CPU result bit for bit --
// tests/test_loader.py checks all 256 values.  HBM-bound: L*H*W*C bytes read, 4x that written, per clip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2v {

// Bounds of the gather (optional, v2v_clip_frames_f32_bounded_hip): clip b holds stored[b] frames and `src` holds src_elems bytes.  A frame
// number outside [0, stored[b]) or a frame that does not fit the buffer is NOT read: its output frame is NaN (workgroup-uniform test).
struct ClipBounds {
    const int32_t *stored;
    int64_t src_elems;
};
__device__ __forceinline__ bool clip_frame_ok(const ClipBounds &cb, int b, int f, int64_t clip_off, int64_t frame_stride, int64_t frame_bytes)
{
    if (!cb.stored) return true;
    const int32_t stored = cb.stored[b];
    if (stored <= 0 || (unsigned)f >= (unsigned)stored) return false;      // a negative (corrupted) count must not pass as a huge unsigned one
    const int64_t first = clip_off + (int64_t)f * frame_stride;
    return cb.src_elems <= 0 || (first >= 0 && first + frame_bytes <= cb.src_elems);
}

// C == 1, HW % 4 == 0, 4-byte aligned rows: a work-item converts 4 pixels (one dword in, 16 bytes out)
// pick_stride: elements between the pick rows of consecutive clips (0: one row for every clip); clip_offsets: per-clip start (or b * clip_stride)
__global__ void __launch_bounds__(256) clip_frames4_kernel(const uint8_t *src, int64_t clip_stride, const int64_t *clip_offsets, int64_t frame_stride,
                                                           const int32_t *pick, int64_t pick_stride, int L, int HW4, float *out, ClipBounds cb)
{
    const int bl = blockIdx.y;                                   // b * L + l
    const int b = bl / L, l = bl - b * L;
    const int f = pick ? pick[(int64_t)b * pick_stride + l] : l;
    const int64_t clip_off = clip_offsets ? clip_offsets[b] : (int64_t)b * clip_stride;
    float4 *o = reinterpret_cast<float4 *>(out + (int64_t)bl * HW4 * 4);
    if (!clip_frame_ok(cb, b, f, clip_off, frame_stride, (int64_t)HW4 * 4)) {
        const float nan = __builtin_nanf("");
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW4; i += gridDim.x * 256) o[i] = make_float4(nan, nan, nan, nan);
        return;
    }
    const uint32_t *s = reinterpret_cast<const uint32_t *>(src + clip_off + (int64_t)f * frame_stride);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW4; i += gridDim.x * 256) {
        const uint32_t v = __builtin_nontemporal_load(s + i);
        float4 r;
        r.x = (float)(v & 255u) / 255.0f;
        r.y = (float)((v >> 8) & 255u) / 255.0f;
        r.z = (float)((v >> 16) & 255u) / 255.0f;
        r.w = (float)(v >> 24) / 255.0f;
        o[i] = r;
    }
}

// any C (interleaved HWC source -> planar CHW output), any alignment: a work-item per (pixel, channel)
__global__ void __launch_bounds__(256) clip_frames_kernel(const uint8_t *src, int64_t clip_stride, const int64_t *clip_offsets, int64_t frame_stride,
                                                          const int32_t *pick, int64_t pick_stride, int L, int HW, int C, float *out, ClipBounds cb)
{
    const int bl = blockIdx.y;
    const int b = bl / L, l = bl - b * L;
    const int f = pick ? pick[(int64_t)b * pick_stride + l] : l;
    const int64_t clip_off = clip_offsets ? clip_offsets[b] : (int64_t)b * clip_stride;
    float *o = out + (int64_t)bl * HW * C;
    const int n = HW * C;
    if (!clip_frame_ok(cb, b, f, clip_off, frame_stride, (int64_t)n)) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) o[i] = __builtin_nanf("");
        return;
    }
    const uint8_t *s = src + clip_off + (int64_t)f * frame_stride;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int c = i / HW, p = i - c * HW;                   // output order: channel-major planes
        o[i] = (float)s[(int64_t)p * C + c] / 255.0f;
    }
}

}  // namespace v2v
