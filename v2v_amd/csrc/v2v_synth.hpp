// v2v_synth.hpp -- synthetic clips generated on the device (bench / tests input).
// Per clip: a low-frequency pattern (three random plane waves) translating at a per-clip velocity in
// [-3,3] px/frame, plus N(0,4) pixel noise, quantised to integers 0..255.  Clip b is a pure function of
// (seed, clip_id0 + b).  Not part of the reference; it only feeds the hot path with video-like input.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "v2v_rng.hpp"

namespace v2v {

struct SynthArgs {
    void *frames;
    uint64_t seed, clip_id0;
    int32_t N, H, W, is_f32;
    int64_t total_quads;   // B*N*H*W/4
};

__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }

__global__ void __launch_bounds__(256) synth_clips_kernel(const SynthArgs a)
{
    const int64_t quad = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (quad >= a.total_quads) return;
    const int32_t HW = a.H * a.W;
    const int64_t quads_per_frame = HW / 4;
    const int64_t fr = quad / quads_per_frame;
    const uint32_t p0 = (uint32_t)(quad - fr * quads_per_frame) * 4u;
    const int32_t n = (int32_t)(fr % a.N);
    const uint32_t clip = (uint32_t)(a.clip_id0 + (uint64_t)(fr / a.N));
    const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);

    // per-clip pattern parameters (uniform across the clip; recomputed per thread: ~3 Philox blocks)
    const u32x4 r0 = philox4x32_10(0u, 0xFFFF0000u, clip, kStreamSynth, k0, k1);
    const u32x4 r1 = philox4x32_10(1u, 0xFFFF0000u, clip, kStreamSynth, k0, k1);
    const u32x4 r2 = philox4x32_10(2u, 0xFFFF0000u, clip, kStreamSynth, k0, k1);
    const float vx = (u01(r0.x) - 0.5f) * 6.0f, vy = (u01(r0.y) - 0.5f) * 6.0f;
    const float two_pi = 6.28318530718f;
    const float inv = 1.0f / (float)(a.W > a.H ? a.W : a.H);
    const float fx0 = (0.5f + 2.5f * u01(r0.z)) * inv * two_pi, fy0 = (0.5f + 2.5f * u01(r0.w)) * inv * two_pi;
    const float fx1 = (0.5f + 4.5f * u01(r1.x)) * inv * two_pi, fy1 = -(0.5f + 4.5f * u01(r1.y)) * inv * two_pi;
    const float fx2 = (2.0f + 6.0f * u01(r1.z)) * inv * two_pi, fy2 = (2.0f + 6.0f * u01(r1.w)) * inv * two_pi;
    const float ph0 = u01(r2.x) * two_pi, ph1 = u01(r2.y) * two_pi, ph2 = u01(r2.z) * two_pi;
    const float mean = 64.0f + 128.0f * u01(r2.w);

    float g[4];
    {
        const u32x4 w = philox4x32_10(p0 >> 2, (uint32_t)n, clip, kStreamSynth, k0, k1);
        bm_pair(w.x, w.y, g[0], g[1]);
        bm_pair(w.z, w.w, g[2], g[3]);
    }
    const int32_t y = (int32_t)(p0 / (uint32_t)a.W);
    const int32_t x0 = (int32_t)(p0 - (uint32_t)y * (uint32_t)a.W);
    float out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // p0..p0+3 stay on one row when W % 4 == 0 (required by the launcher)
        const float xs = (float)(x0 + j) + vx * (float)n, ys = (float)y + vy * (float)n;
        float v = mean + 50.0f * __sinf(fx0 * xs + fy0 * ys + ph0) + 35.0f * __sinf(fx1 * xs + fy1 * ys + ph1) +
                  20.0f * __sinf(fx2 * xs + fy2 * ys + ph2) + 4.0f * g[j];
        v = rintf(v);
        out[j] = fminf(fmaxf(v, 0.0f), 255.0f);
    }
    if (a.is_f32) {
        reinterpret_cast<float4 *>(a.frames)[quad] = make_float4(out[0], out[1], out[2], out[3]);
    } else {
        const uint32_t w = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
        reinterpret_cast<uint32_t *>(a.frames)[quad] = w;
    }
}

}  // namespace v2v
