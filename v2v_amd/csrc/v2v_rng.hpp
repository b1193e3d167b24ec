// v2v_rng.hpp -- device-native random fields for the simulator (gfx950).
//
// The reference draws from NumPy's global MT19937 stream (data/v2v_core_esim.py:29,37,38,44), a
// sequential generator that cannot be replayed at HBM bandwidth.  The device-native mode replaces it
// with counter-based Philox4x32-10 keyed by (seed; pixel, field, clip_id, stream): any thread can
// produce any sample, so results are independent of launch geometry, batch size and GPU sharding.
// Field ids (contract shared with the CPU oracle, which restates this file independently):
//   0 potential-init uniform   1 hot-mask uniform   2 hot-pixel Gaussian   3+m base-noise Gaussians of pairs 2m, 2m+1
// Uniforms are float64 on NumPy's 53-bit grid.  Gaussians are float32, TWO per 32-bit Philox word (one per 16-bit half, by
// direct table inversion of its top 14 bits): word j of block (p>>2, field, clip, stream) belongs to pixel p = 4*(p>>2)+j, its first deviate serves the
// even and its second the odd member of a pair of consecutive time steps, so one Philox block feeds 4 pixels x 2 steps.
// The transform is one table read: host (gcc) and device (hipcc) agree bit for bit.
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

// ---- float32 Gaussians by table inversion --------------------------------------------------------------------------------
// half-word n of a Philox word: sign = n >> 15, magnitude index i = (n >> 2) & 0x1FFF (13 bits; the two low bits are unused),
// i.e. the probability 1/2 + (i + 1/2) / 2^14 on a midpoint grid; deviate = Phi^-1 of it, read DIRECTLY from a table of 8192
// float32 values (tools/gen_gauss_icdf.py, 32 KB).  2^14 distinct values, |g| <= 4.009, variance 0.99992, 4th moment 2.997.
// Per deviate: one and (+ one shift for the high half) for the byte offset, ONE 4-byte LDS read, and + xor for the sign.
// History: the float32 Box-Muller this replaces (one word -> radius and angle, degree-7 log, two Newton steps,
// degree-3/4 sine and cosine polynomials) cost 18 packed instructions per pair and was the largest single item of the
// VALU-bound noise-on launch; a 16-bit variant (4096 x {intercept, slope}, one convert + fma per deviate) measured 6-8 % slower
// on the same box than this direct table for 1.5e-4 instead of 3.8e-5 of quantile spacing at the centre and tails ending at
// 4.0 instead of 4.2 sigma -- immaterial for sensor noise of a few percent of the contrast threshold.
// The table is data shared with the CPU oracle (oracle/gauss_icdf.inc holds the same text); kernels copy it to LDS once per
// workgroup, one-off per-pixel fields may read it from global memory.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 pk_splat(float v) { return f32x2{v, v}; }

constexpr int kIcdfEntries = 8192, kIcdfBytes = kIcdfEntries * 4;
static __device__ const float g_gauss_icdf[kIcdfEntries] = {
#include "v2v_gauss_icdf.inc"
};

// copy the table into LDS (256 work-items: 8 x 16 bytes each); the caller synchronises
__device__ __forceinline__ void icdf_to_lds(float *s_icdf)
{
    const float4 *src = reinterpret_cast<const float4 *>(g_gauss_icdf);
    float4 *dst = reinterpret_cast<float4 *>(s_icdf);
#pragma unroll
    for (int i = 0; i < kIcdfBytes / 16 / 256; ++i) dst[i * 256 + threadIdx.x] = src[i * 256 + threadIdx.x];
}

// one word -> two deviates: g0 from the high, g1 from the low half-word
__device__ __forceinline__ void icdf_pair(uint32_t w, const float *tab, float &g0, float &g1)
{
    const unsigned char *b = reinterpret_cast<const unsigned char *>(tab);
    const uint32_t t0 = *reinterpret_cast<const uint32_t *>(b + ((w >> 16) & 0x7FFCu));     // byte offset = 4 * ((n >> 2) & 0x1FFF)
    const uint32_t t1 = *reinterpret_cast<const uint32_t *>(b + (w & 0x7FFCu));
    g0 = __uint_as_float(t0 ^ (w & 0x80000000u));
    g1 = __uint_as_float(t1 ^ ((w << 16) & 0x80000000u));
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

// Gaussian pairs of VEC consecutive pixels starting at p0 (p0 % VEC == 0): ga = first, gb = second deviate of every pixel's
// word in block `field`; tab = the inverse-CDF table (LDS copy in the time loops, g_gauss_icdf for one-off fields).
template <int VEC, int ROUNDS = 10>
__device__ __forceinline__ void field_gauss_pairs(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream,
                                                  uint32_t p0, const float *tab, float (&ga)[VEC], float (&gb)[VEC])
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    if constexpr (VEC == 1) {
        const u32x4 w = philox4x32<ROUNDS>(p0 >> 2, field, clip, stream, k0, k1);
        const uint32_t j = p0 & 3u;
        icdf_pair(j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w, tab, ga[0], gb[0]);
    } else if constexpr (VEC == 2) {                       // p0 even: the lower or the upper word pair of the block
        const u32x4 w = philox4x32<ROUNDS>(p0 >> 2, field, clip, stream, k0, k1);
        const bool up = (p0 & 2u) != 0;
        icdf_pair(up ? w.z : w.x, tab, ga[0], gb[0]);
        icdf_pair(up ? w.w : w.y, tab, ga[1], gb[1]);
    } else {
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const u32x4 w = philox4x32<ROUNDS>((p0 + j) >> 2, field, clip, stream, k0, k1);
            icdf_pair(w.x, tab, ga[j], gb[j]);
            icdf_pair(w.y, tab, ga[j + 1], gb[j + 1]);
            icdf_pair(w.z, tab, ga[j + 2], gb[j + 2]);
            icdf_pair(w.w, tab, ga[j + 3], gb[j + 3]);
        }
    }
}

}  // namespace v2v
