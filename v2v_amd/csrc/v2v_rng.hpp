// v2v_rng.hpp -- device-native random fields for the simulator (gfx950).
//
// The reference draws from NumPy's global MT19937 stream (data/v2v_core_esim.py:29,37,38,44), a
// sequential generator that cannot be replayed at HBM bandwidth.  The device-native mode replaces it
// with counter-based Philox4x32-10 keyed by (seed; pixel, field, clip_id, stream): any thread can
// produce any sample, so results are independent of launch geometry, batch size and GPU sharding.
// Field ids (contract shared with the CPU oracle, which restates this file independently):
//   0 potential-init uniform   1 hot-mask uniform   2 hot-pixel Gaussian   3+k base-noise Gaussian of pair k
// Uniforms are float64 on NumPy's 53-bit grid; Gaussians are float32 Box-Muller built only from
// IEEE-exact operations (+ - * fma sqrt, integer ops) so host (gcc) and device (hipcc) agree bit for bit.
// The whole library is compiled with -ffp-contract=off; every fused multiply-add below is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2v {

enum : uint32_t { kFieldPotInit = 0, kFieldHotMask = 1, kFieldHotGauss = 2, kFieldBase0 = 3 };
enum : uint32_t { kStreamEsim = 0, kStreamV2e = 1, kStreamSynth = 2 };

struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// NumPy legacy random_sample recipe on two 32-bit words: ((a>>5)*2^26 + (b>>6)) / 2^53
__device__ __forceinline__ double uniform53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// fp32 Box-Muller, two normals from two words.  Cephes logf polynomial on [sqrt(.5), sqrt(2)],
// Cephes sinf/cosf kernels on [-pi/4, pi/4], exact-sign rotation by (q + 1/2)*pi/2.
__device__ __forceinline__ void bm_pair(uint32_t a, uint32_t b, float &g0, float &g1)
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

// ---- per-pixel field accessors.  VEC consecutive pixels starting at p0 (p0 % VEC == 0). -------------
template <int VEC>
__device__ __forceinline__ void field_uniform53(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream,
                                                uint32_t p0, double (&u)[VEC])
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if constexpr (VEC == 1) {
        const u32x4 w = philox4x32_10(p0 >> 1, field, clip, stream, k0, k1);
        u[0] = (p0 & 1u) ? uniform53(w.z, w.w) : uniform53(w.x, w.y);
    } else {
#pragma unroll
        for (int j = 0; j < VEC; j += 2) {
            const u32x4 w = philox4x32_10((p0 + j) >> 1, field, clip, stream, k0, k1);
            u[j] = uniform53(w.x, w.y);
            u[j + 1] = uniform53(w.z, w.w);
        }
    }
}

template <int VEC>
__device__ __forceinline__ void field_gauss32(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream,
                                              uint32_t p0, float (&g)[VEC])
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if constexpr (VEC == 1) {
        const u32x4 w = philox4x32_10(p0 >> 2, field, clip, stream, k0, k1);
        float a, b;
        if ((p0 >> 1) & 1u) bm_pair(w.z, w.w, a, b); else bm_pair(w.x, w.y, a, b);
        g[0] = (p0 & 1u) ? b : a;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const u32x4 w = philox4x32_10((p0 + j) >> 2, field, clip, stream, k0, k1);
            bm_pair(w.x, w.y, g[j], g[j + 1]);
            bm_pair(w.z, w.w, g[j + 2], g[j + 3]);
        }
    }
}

// ---- fast (non bit-reproducible on a CPU) Gaussian field: V2V_RNG_PHILOX_FAST ----------------------------------
// Philox4x32-7 (the Random123 authors' minimum Crush-resistant round count) and Box-Muller on the hardware
// transcendental units (v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32, ~1 ulp, not IEEE-exact).  ~4x fewer VALU
// instructions per sample than the exact path; the fields are statistically equivalent but NOT the ones the CPU
// oracle generates, so parity for this mode is distributional (tests/test_hip_parity.py::test_fast_noise_*).
__device__ __forceinline__ u32x4 philox4x32_7(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        // one 32x32->64 multiply per product (v_mad_u64_u32) instead of a mul_hi + mul_lo pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

__device__ __forceinline__ void bm_pair_fast(uint32_t a, uint32_t b, float &g0, float &g1)
{
    const float u1 = (float)((a >> 8) + 1u) * 5.9604644775390625e-08f;            // (0,1]
    const float rev = (float)(b >> 8) * 5.9604644775390625e-08f;                   // [0,1) revolutions
    const float r = __builtin_amdgcn_sqrtf(-1.38629436f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), log2 based
    g0 = r * __builtin_amdgcn_cosf(rev);
    g1 = r * __builtin_amdgcn_sinf(rev);
}

template <int VEC>
__device__ __forceinline__ void field_gauss32_fast(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream,
                                                   uint32_t p0, float (&g)[VEC])
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if constexpr (VEC == 1) {
        const u32x4 w = philox4x32_7(p0 >> 2, field, clip, stream, k0, k1);
        float a, b;
        if ((p0 >> 1) & 1u) bm_pair_fast(w.z, w.w, a, b); else bm_pair_fast(w.x, w.y, a, b);
        g[0] = (p0 & 1u) ? b : a;
    } else {
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const u32x4 w = philox4x32_7((p0 + j) >> 2, field, clip, stream, k0, k1);
            bm_pair_fast(w.x, w.y, g[j], g[j + 1]);
            bm_pair_fast(w.z, w.w, g[j + 2], g[j + 3]);
        }
    }
}

}  // namespace v2v
