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

// Pixel noise of the synthetic clips: the round-1 float32 Box-Muller (two normals from two words), kept here so the
// bench/test inputs stay the same clips as in round 1; the simulators use the table-inversion generator of v2v_rng.hpp.  Cephes logf polynomial on [sqrt(.5), sqrt(2)],
// Cephes sinf/cosf kernels on [-pi/4, pi/4], exact-sign rotation by (q + 1/2)*pi/2.
__device__ __forceinline__ void synth_bm_pair(uint32_t a, uint32_t b, float &g0, float &g1)
{
    const float u1 = (float)((a >> 8) + 1u) * 5.9604644775390625e-08f;   // (0,1] on a 2^-24 grid
    const uint32_t bits = __float_as_uint(u1);
    int e = (int)(bits >> 23) - 127;
    float m = __uint_as_float((bits & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, f, -1.1514610310e-1f);
    p = __builtin_fmaf(p, f, 1.1676998740e-1f);
    p = __builtin_fmaf(p, f, -1.2420140846e-1f);
    p = __builtin_fmaf(p, f, 1.4249322787e-1f);
    p = __builtin_fmaf(p, f, -1.6668057665e-1f);
    p = __builtin_fmaf(p, f, 2.0000714765e-1f);
    p = __builtin_fmaf(p, f, -2.4999993993e-1f);
    p = __builtin_fmaf(p, f, 3.3333331174e-1f);
    float y = (p * f) * z;
    y = __builtin_fmaf(-0.5f, z, y);
    const float ln_m = f + y;
    const float ln_u = __builtin_fmaf((float)e, 0.693147182f, ln_m);
    const float t = -2.0f * ln_u;
    const float r = __builtin_sqrtf(t) * 0.707106769f;   // correctly-rounded sqrt (HIP default)

    const uint32_t q = b >> 30;
    const float yy = (float)((b >> 6) & 0x00FFFFFFu) * 5.9604644775390625e-08f - 0.5f;
    const float x = yy * 1.57079637f;
    const float zz = x * x;
    float s = -1.9515295891e-4f;
    s = __builtin_fmaf(s, zz, 8.3321608736e-3f);
    s = __builtin_fmaf(s, zz, -1.6666654611e-1f);
    s = __builtin_fmaf(s * zz, x, x);
    float c = 2.443315711809948e-5f;
    c = __builtin_fmaf(c, zz, -1.388731625493765e-3f);
    c = __builtin_fmaf(c, zz, 4.166664568298827e-2f);
    c = __builtin_fmaf(c * zz, zz, __builtin_fmaf(-0.5f, zz, 1.0f));
    const uint32_t sc = ((q == 1u) || (q == 2u)) ? 0x80000000u : 0u;
    const uint32_t ss = (q >= 2u) ? 0x80000000u : 0u;
    const float cc = __uint_as_float(__float_as_uint(c) ^ sc);
    const float cs = __uint_as_float(__float_as_uint(c) ^ ss);
    const float sc_s = __uint_as_float(__float_as_uint(s) ^ sc);
    const float ss_s = __uint_as_float(__float_as_uint(s) ^ ss);
    g0 = r * (cc - ss_s);
    g1 = r * (cs + sc_s);
}

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
        synth_bm_pair(w.x, w.y, g[0], g[1]);
        synth_bm_pair(w.z, w.w, g[2], g[3]);
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
