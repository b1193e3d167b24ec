// v2v_rng.hpp -- device-native random fields for the simulator (gfx950).
//
// The reference draws from NumPy's global MT19937 stream (data/v2v_core_esim.py:29,37,38,44), a
// sequential generator that cannot be replayed at HBM bandwidth.  The device-native mode replaces it
// with counter-based Philox4x32-10 keyed by (seed; pixel, field, clip_id, stream): any thread can
// produce any sample, so results are independent of launch geometry, batch size and GPU sharding.
// Field ids (contract shared with the CPU oracle, which restates this file independently):
//   0 potential-init uniform   1 hot-mask uniform   2 hot-pixel Gaussian   3+m base-noise Gaussians of pairs 2m, 2m+1
// Uniforms are float64 on NumPy's 53-bit grid.  Gaussians are float32 Box-Muller, ONE 32-bit Philox word per
// Box-Muller pair (16-bit radius x 16-bit angle on midpoint grids): word j of block (p>>2, field, clip, stream)
// belongs to pixel p = 4*(p>>2)+j, its first normal serves the even and its second the odd member of a pair of
// consecutive time steps, so one Philox block feeds 4 pixels x 2 steps.  The transform uses only IEEE-exact
// operations (+ - * fma, integer ops; no division, no sqrt, no library call) so host (gcc) and device (hipcc)
// agree bit for bit, and it is written on 2-vectors: the polynomials run on v_pk_fma_f32 / v_pk_mul_f32, the
// only form in which gfx950 reaches its full fp32 rate (tools/ubench/valu_rates.hip).
// The whole library is compiled with -ffp-contract=off; every fused multiply-add below is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2v {

enum : uint32_t { kFieldPotInit = 0, kFieldHotMask = 1, kFieldHotGauss = 2, kFieldBase0 = 3 };
enum : uint32_t { kStreamEsim = 0, kStreamV2e = 1, kStreamSynth = 2 };
// Philox rounds of the per-time-step noise fields (ESIM base noise, v2e leak jitter and shot uniforms): 7, the round count
// the Random123 authors state as Crush-resistant for Philox4x32 (Salmon et al., SC'11, table 2); the per-clip fields
// (potential init, hot pixels, thresholds, leak rates) are drawn once and keep the customary 10.  Worth 4 % of the
// noise-on launches (0.77 -> 0.74 ms ESIM, 2.06 -> 1.98 ms v2e, config 2/3).
#ifndef V2V_NOISE_ROUNDS
#define V2V_NOISE_ROUNDS 7
#endif
constexpr int kNoiseRounds = V2V_NOISE_ROUNDS;

struct u32x4 { uint32_t x, y, z, w; };

template <int ROUNDS>
__device__ __forceinline__ u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
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
__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
    return philox4x32<10>(c0, c1, c2, c3, k0, k1);
}

// NumPy legacy random_sample recipe on two 32-bit words: ((a>>5)*2^26 + (b>>6)) / 2^53
__device__ __forceinline__ double uniform53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// ---- float32 Box-Muller on 16+16 bits --------------------------------------------------------------------------
// word w: n = w >> 16 -> u = (n + 1/2) / 2^16 in (0,1), radius r = sqrt(-2 ln u) in [0.0039, 4.86];
//         a = w & 0xFFFF -> x = pi (a + 1/2) / 2^16 - pi/2 in (-pi/2, pi/2), angle 2x uniform on a 2^16 grid of (-pi, pi)
//         g0 = r cos 2x = r (1 - S^2),  g1 = r sin 2x = r S C   with S = sqrt2 sin x, C = sqrt2 cos x.
// -2 ln u = (16 - e) 2ln2 + L(f) for n + 1/2 = 2^e (1 + f): degree-7 minimax L (3.9e-7); sqrt by the integer
// seed + two tuned Newton steps on the reciprocal root (5.7e-7 relative); S, C degree 3/4 in x^2 (1.1e-6 / 6.6e-8).
// Coefficients: tools/fit_gauss16.py (minimax fits; checked exhaustively over the 2^16 x 2^16 inputs in
// tests/test_oracle_golden.py::test_gauss16_*).  The C oracle restates the same sequence with fmaf.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return f32x2{v, v}; }

// two words -> two Box-Muller pairs: (g0.x, g1.x) from w.x, (g0.y, g1.y) from w.y
__device__ __forceinline__ void gauss16_x2(u32x2 w, f32x2 &g0, f32x2 &g1)
{
    // radius
    const f32x2 xh = f32x2{(float)(w.x >> 16), (float)(w.y >> 16)} + pk_splat(0.5f);                // n + 1/2, exact
    const u32x2 xb = u32x2{__float_as_uint(xh.x), __float_as_uint(xh.y)};
    const f32x2 ef = f32x2{(float)((int)(xb.x >> 23) - 143), (float)((int)(xb.y >> 23) - 143)};     // e - 16 in [-17,-1]
    const f32x2 f = f32x2{__uint_as_float((xb.x & 0x007FFFFFu) | 0x3F800000u), __uint_as_float((xb.y & 0x007FFFFFu) | 0x3F800000u)} - pk_splat(1.0f);
    f32x2 L = pk_splat(-0x1.57869cp-6f);
    L = pk_fma(L, f, pk_splat(0x1.bb3e08p-4f));
    L = pk_fma(L, f, pk_splat(-0x1.10adbap-2f));
    L = pk_fma(L, f, pk_splat(0x1.cc4bd8p-2f));
    L = pk_fma(L, f, pk_splat(-0x1.4fa778p-1f));
    L = pk_fma(L, f, pk_splat(0x1.ff5d72p-1f));
    L = pk_fma(L, f, pk_splat(-0x1.fffc7ap+0f));
    L = pk_fma(L, f, pk_splat(-0x1.9cde6p-22f));
    const f32x2 t = pk_fma(ef, pk_splat(-0x1.62e43p+0f), L);                                         // -2 ln u  (> 0)
    const f32x2 th = t * pk_splat(0x1.007aa6p-1f);
    f32x2 y = f32x2{__uint_as_float(0x5f374000u - (__float_as_uint(t.x) >> 1)), __uint_as_float(0x5f374000u - (__float_as_uint(t.y) >> 1))};
    f32x2 p = y * y;
    f32x2 q = pk_fma(-th, p, pk_splat(0x1.804d8ep+0f));
    y = y * q;
    p = y * y;
    q = pk_fma(-th, p, pk_splat(0x1.803d52p+0f));
    const f32x2 r = (y * q) * t;                                                                     // sqrt(t)
    // angle
    const f32x2 x = pk_fma(f32x2{(float)(w.x & 0xFFFFu), (float)(w.y & 0xFFFFu)}, pk_splat(0x1.921fb6p-15f), pk_splat(-0x1.921e24p+0f));
    const f32x2 z = x * x;
    f32x2 S = pk_splat(-0x1.12b318p-12f);
    S = pk_fma(S, z, pk_splat(0x1.813e8ap-7f));
    S = pk_fma(S, z, pk_splat(-0x1.e2b092p-3f));
    S = pk_fma(S, z, pk_splat(0x1.6a09d4p+0f));
    const f32x2 s = x * S;
    f32x2 c = pk_splat(0x1.12ae8p-15f);
    c = pk_fma(c, z, pk_splat(-0x1.00cc2ap-9f));
    c = pk_fma(c, z, pk_splat(0x1.e2aebap-5f));
    c = pk_fma(c, z, pk_splat(-0x1.6a09bap-1f));
    c = pk_fma(c, z, pk_splat(0x1.6a09e6p+0f));
    const f32x2 t1 = r * s;
    g0 = pk_fma(-t1, s, r);
    g1 = t1 * c;
}

__device__ __forceinline__ void gauss16(uint32_t w, float &g0, float &g1)
{
    f32x2 a, b;
    gauss16_x2(u32x2{w, w}, a, b);
    g0 = a.x;
    g1 = b.x;
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

// Box-Muller pairs of VEC consecutive pixels starting at p0 (p0 % VEC == 0): ga = first, gb = second normal of
// every pixel's pair in block `field`.
template <int VEC, int ROUNDS = 10>
__device__ __forceinline__ void field_gauss_pairs(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream,
                                                  uint32_t p0, float (&ga)[VEC], float (&gb)[VEC])
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if constexpr (VEC == 1) {
        const u32x4 w = philox4x32<ROUNDS>(p0 >> 2, field, clip, stream, k0, k1);
        const uint32_t j = p0 & 3u;
        gauss16(j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w, ga[0], gb[0]);
    } else {
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const u32x4 w = philox4x32<ROUNDS>((p0 + j) >> 2, field, clip, stream, k0, k1);
            f32x2 a, b;
            gauss16_x2(u32x2{w.x, w.y}, a, b);
            ga[j] = a.x; ga[j + 1] = a.y; gb[j] = b.x; gb[j + 1] = b.y;
            gauss16_x2(u32x2{w.z, w.w}, a, b);
            ga[j + 2] = a.x; ga[j + 3] = a.y; gb[j + 2] = b.x; gb[j + 3] = b.y;
        }
    }
}

}  // namespace v2v
