// v2v_v2e.hpp -- v2e-derived DVS pixel model fused with voxel binning (gfx950).  BASELINE config 3.
//
// Replaces data/v2v_core_v2e.py: video_to_voxel (:556-581) around EventEmulator.generate_events (:401-553) with
// lin_log (:108-137, effective formula float32(log(x/255+0.01)) -> 256-entry table, golden G1),
// rescale_intensity_frame (:184-190), low_pass_filter (:139-182), subtract_leak_current (:192-211),
// compute_event_map (:42-62), generate_shot_noise (:65-105), _init / change_pos_neg_thres (:317-349, :392-399).
//
// Same mapping as the ESIM kernel: one work-item owns VEC adjacent pixels of one clip and streams the frames;
// per-pixel state (low-passed log intensity, memorised base, ON/OFF thresholds, leak rate) lives in registers.
// NumPy's dtype promotions are part of the reference's semantics and are reproduced explicitly:
//   lp32   = (cutoff_hz <= 0) || float32 input     low-passed frame is float32, else float64
//   base32 = lp32 && leak_rate_hz == 0              memorised frame is float32 (in-place += rounds to float32)
// Shot noise needs the per-frame mean of (intensity factor x threshold factor) over the whole clip frame:
// a pre-pass kernel accumulates it as an exact integer sum of products of two 2^20 fixed-point factors (order-independent,
// so the CPU oracle and any launch geometry agree bit for bit); native Poisson sampling is inversion from one Philox uniform with an exp(-lambda)
// built from IEEE-exact operations.  Replay mode takes NumPy-drawn fields instead (bit-exact reference replay).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "v2v_common.hpp"

namespace v2v {


enum { kV2ePnRelated = 0, kV2eSpatialIndependent = 1, kV2eSpatialTemporalIndependent = 2 };
// Philox blocks of the native fields: 0 static thresholds (pair = ON/OFF normals), 2 leak-rate normal; per frame i:
// kV2eFFrame0 + 8 i + {0 thresholds redrawn (temporal model), 3 shot uniforms}; per couple m of frame pairs: + 8 m + 2 leak jitter
enum : uint32_t { kV2eFThresA = 0, kV2eFNoiseRate = 2, kV2eFFrame0 = 16, kV2eFStride = 8 };



__device__ __forceinline__ double exp_neg_det(double lam)
{
    const double x = -lam;
    if (x < -745.0) return 0.0;
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(-k, 0.693147180369123816490e+00, x);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    const int ki = (int)k;
    if (ki < -1022) return 0.0;
    return p * __longlong_as_double((long long)(1023 + ki) << 52);
}

__device__ __forceinline__ double poisson_inv(double lam, double u)
{
    if (!(lam > 0.0)) return 0.0;
    double p = exp_neg_det(lam), s = p, x = 0.0;
    while (u > s && x < 1000.0) { x += 1.0; p = p * lam / x; s += p; }
    return x;
}

__device__ __forceinline__ float expf_det(float x);

// tail of the inversion (count > 3): entered with count 2 reached (p = its probability, s = the cumulative sum) and u > s
__device__ __forceinline__ float poisson_tail_f32(float lam, float u, float p, float s)
{
    float x = 2.0f;
    do { x += 1.0f; p = p * (lam / x); s = s + p; } while (u > s && x < 64.0f);
    return x;
}

// shot-noise uniforms of VEC pixels: ONE Philox block per 4 pixels and frame; word j -> pixel j, its high half the ON and
// its low half the OFF uniform, both on the midpoint grid (n + 1/2) / 2^16 in (0,1)
template <int VEC, int ROUNDS = 10>
__device__ __forceinline__ void field_uniform16x2(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream, uint32_t p0,
                                                  float (&ua)[VEC], float (&ub)[VEC])
{
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const u32x4 w = philox4x32<ROUNDS>(p0 >> 2, field, clip, stream, k0, k1);
    if constexpr (VEC == 1) {
        const uint32_t j = p0 & 3u;
        const uint32_t x = j == 0 ? w.x : j == 1 ? w.y : j == 2 ? w.z : w.w;
        ua[0] = ((float)(x >> 16) + 0.5f) * 1.52587890625e-05f;
        ub[0] = ((float)(x & 0xFFFFu) + 0.5f) * 1.52587890625e-05f;
    } else {
        const uint32_t x[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ua[j] = ((float)(x[j] >> 16) + 0.5f) * 1.52587890625e-05f;
            ub[j] = ((float)(x[j] & 0xFFFFu) + 0.5f) * 1.52587890625e-05f;
        }
    }
}

__device__ __forceinline__ float expf_det(float x)
{
    const bool under = x < -87.0f;                      // -> 0 (selected at the end: no branch in the callers' loops)
    x = under ? -87.0f : x;
    if (x > 88.0f) x = 88.0f;
    const float k = __builtin_rintf(x * 1.44269502f);
    float r = __builtin_fmaf(-k, 0.693359375f, x);
    r = __builtin_fmaf(-k, -2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    p = __builtin_fmaf(p, r2, r) + 1.0f;
    const float e = p * __uint_as_float((uint32_t)(127 + (int)k) << 23);
    return under ? 0.0f : e;
}

// ON/OFF thresholds of VEC pixels from two Gaussian fields (normal(loc,scale) = loc + scale*g), clipped at 0.01
template <int VEC>
__device__ __forceinline__ void v2e_native_thres(const V2eParams &P, uint64_t seed, uint32_t clip, uint32_t fa, uint32_t p0,
                                                 const float *tab, double (&pt)[VEC], double (&nt)[VEC])
{
    float ga[VEC], gb[VEC];                 // the two deviates of the pixel's word (block fa)
    field_gauss_pairs<VEC>(seed, clip, fa, kStreamV2e, p0, tab, ga, gb);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        double a, b;
        if (P.threshold_model == kV2ePnRelated) {
            const double sa = P.thres_mean_std * (double)ga[j], sb = P.thres_diff_std * (double)gb[j];
            const double mean = P.thres_mean_mean + sa, diff = P.thres_diff_mean + sb;
            a = mean + (diff / 2);
            b = mean - (diff / 2);
        } else {
            const double sa = P.thres_mean_std * (double)ga[j], sb = P.thres_mean_std * (double)gb[j];
            a = P.thres_mean_mean + sa;
            b = P.thres_mean_mean + sb;
        }
        pt[j] = a < 0.01 ? 0.01 : a;
        nt[j] = b < 0.01 ? 0.01 : b;
    }
}

template <int IN, int VEC>
__device__ __forceinline__ void v2e_pixels(const Raw<IN, VEC> &r, float (&x)[VEC])
{
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if constexpr (IN == kInU8) { if constexpr (VEC == 4) x[j] = (float)((r.v >> (8 * j)) & 0xFFu); else x[j] = (float)r.v; }
        else x[j] = raw_f32<VEC>(r, j);
    }
}

// intensity terms in the dtype NumPy gives them (float64 for uint8 input, float32 for float32 input):
//   inten01 = (x + 20)/275 (uint8 input wraps like the reference when uint8_wrap), fac = 1 - 0.75*inten01
template <int IN>
__device__ __forceinline__ void v2e_inten_direct(float x, int wrap, double &i01_64, float &i01_32, double &fac)
{
    if constexpr (IN == kInU8) {
        const uint32_t xi = (uint32_t)x;
        i01_64 = wrap ? (double)((xi + 20u) & 0xFFu) / 275. : ((double)xi + 20.0) / 275.;
        i01_32 = 0.0f;
        const double t = 0.75 * i01_64;
        fac = 1 - t;
    } else {
        i01_32 = (x + 20.0f) / 275.0f;
        i01_64 = 0.0;
        const float t = 0.75f * i01_32;
        fac = (double)(1.0f - t);
    }
}

// Both terms depend only on the 8-bit intensity: each workgroup tabulates them once in LDS with the expressions
// above (bitwise the same values), replacing a float64 division per pixel and frame by two LDS reads.
struct V2eIntenTables { double *i01_64; double *fac; float *i01_32; };

template <int IN>
__device__ __forceinline__ void v2e_inten(float x, int wrap, const V2eIntenTables &tb, double &i01_64, float &i01_32, double &fac)
{
    const uint32_t a = __float_as_uint(x + 8388608.0f) & 255u;
    if (__builtin_expect((float)a == x, 1)) {
        fac = tb.fac[a];
        if constexpr (IN == kInU8) { i01_64 = tb.i01_64[a]; i01_32 = 0.0f; }
        else { i01_32 = tb.i01_32[a]; i01_64 = 0.0; }
    } else {
        v2e_inten_direct<IN>(x, wrap, i01_64, i01_32, fac);
    }
}

template <int IN>
__device__ __forceinline__ V2eIntenTables v2e_build_tables(unsigned char *base, int wrap)
{
    V2eIntenTables tb;
    tb.i01_64 = reinterpret_cast<double *>(base);
    tb.fac = tb.i01_64 + 256;
    tb.i01_32 = reinterpret_cast<float *>(tb.fac + 256);
    double a64, f; float a32;
    v2e_inten_direct<IN>((float)threadIdx.x, wrap, a64, a32, f);
    tb.i01_64[threadIdx.x] = a64;
    tb.fac[threadIdx.x] = f;
    tb.i01_32[threadIdx.x] = a32;
    return tb;
}
constexpr int kV2eTableBytes = 256 * (8 + 8 + 4);

__device__ __forceinline__ float v2e_linlog(float x, const float *lut)
{
    const uint32_t a = __float_as_uint(x + 8388608.0f) & 255u;
    if (__builtin_expect((float)a == x, 1)) return lut[a];
    return (float)log((double)x / 255 + 0.01);          // non-integer content: within 1 ulp of NumPy's float64 log
}

// Wave-wide 64-bit integer sum on the VALU (DPP row shifts + four v_readlane per half): the LDS-routed __shfl_down version of
// this reduction made the pre-pass LDS-issue-bound (24 ds_bpermute per frame and wave).  Every lane returns the total.
template <int CTRL>
__device__ __forceinline__ long long dpp_mov_i64(long long v)
{
    const uint64_t b = (uint64_t)v;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return (long long)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ long long readlane_i64(long long v, int lane)
{
    const uint64_t b = (uint64_t)v;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), lane);
    return (long long)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ long long wave_sum_i64(long long v)
{
    v += dpp_mov_i64<0x111>(v);          // row_shr:1 (lanes shifted in from outside the 16-lane row read 0)
    v += dpp_mov_i64<0x112>(v);          // row_shr:2
    v += dpp_mov_i64<0x114>(v);          // row_shr:4
    v += dpp_mov_i64<0x118>(v);          // row_shr:8 -> lane 15 of every row holds the row total
    return (readlane_i64(v, 15) + readlane_i64(v, 31)) + (readlane_i64(v, 47) + readlane_i64(v, 63));
}

// ---- pre-pass: per (clip, frame) sums of the shot-noise factors (native mode only) -----------------------------------------
// generate_shot_noise (:86-100) normalises the Poisson rate by the frame mean of (intensity factor x threshold factor).  The
// native definition of that sum is an INTEGER one, so that it does not depend on summation order, launch geometry or the CPU
// oracle's loop order:   S = sum over pixels of  Q(intensity factor) * Q(nominal threshold / pixel threshold),
// Q(v) = rint(v * 2^20) clamped to +-(2^31 - 128) (NaN -> 0); mean = S / 2^40 / (H*W).  The intensity factor depends only on
// the 8-bit intensity (256-entry LDS table of Q values), the threshold factor only on the pixel: one v_mad_i64_i32 per pixel,
// frame and polarity.  A work-item owns kPreGroups x VEC pixels, so the two wave reductions per frame are shared by 16 pixels
// per lane.  Sums leave the wave split at bit 32 ({low 32 bits, arithmetic high part}, each its own 64-bit accumulator), which
// keeps every accumulator far from overflow for any frame size; the reader recombines them in float64 (one rounding).
// (A fused variant -- the blocks_per_clip workgroups of a clip as a team in one persistent cooperative launch: sums, team
// barrier on a device-scope counter, simulation with the re-read served by the Infinity Cache -- was built and measured in round
// 2: 2.58 ms against 2.04 ms for these two kernels on config 3.  At the simulator's 3 waves per SIMD the summing phase reads at
// half the rate of this kernel, and the team barrier turns the per-workgroup load balance of a plain launch into a max over
// 64 members.  It was removed; DESIGN.md section 4.3c keeps the numbers.)
constexpr double kShotQScale = 1048576.0;          // 2^20

__device__ __forceinline__ int32_t shot_quant(double v)
{
    double s = v * kShotQScale;
    s = s > 2147483520.0 ? 2147483520.0 : s;
    s = s < -2147483520.0 ? -2147483520.0 : s;
    return s == s ? (int32_t)__builtin_rint(s) : 0;
}
__device__ __forceinline__ long long mad_i64_i32(int32_t a, int32_t b, long long c) { return (long long)a * (long long)b + c; }

// shot_sums layout: [B, K, 4] = {ON low, ON high, OFF low, OFF high}
__device__ __forceinline__ double shot_sum_value(const long long *s4, int which)
{
    return (double)s4[2 * which + 1] * 4294967296.0 + (double)s4[2 * which];
}

template <int IN, int VEC, bool NT, int DEPTH>
__device__ __forceinline__ void v2e_presum_body(const V2eArgs &a, const int clip, const int blk, int32_t *s_q, unsigned long long *s_sum)
{
    {   // Q(intensity factor) per 8-bit intensity, in the dtype NumPy gives the factor (v2e_inten_direct)
        double i64, fac; float i32;
        v2e_inten_direct<IN>((float)threadIdx.x, a.P.uint8_wrap, i64, i32, fac);
        s_q[threadIdx.x] = shot_quant(fac);
    }
    for (int t = threadIdx.x; t < 4 * a.K; t += kBlock) s_sum[t] = 0ull;
    __syncthreads();
    const V2eParams &P = a.P;
    const uint32_t clip_id = (uint32_t)(a.clip_id0 + (uint64_t)clip);
    const double pos_nominal = P.thres_mean_mean + P.thres_diff_mean / 2, neg_nominal = P.thres_mean_mean - P.thres_diff_mean / 2;
    const bool temporal = P.threshold_model == kV2eSpatialTemporalIndependent;
    uint32_t p0[kPreGroups];
    bool active[kPreGroups];
    int64_t in_base[kPreGroups];
    int32_t cp[kPreGroups][VEC], cn[kPreGroups][VEC];
#pragma unroll
    for (int g = 0; g < kPreGroups; ++g) {
        p0[g] = (uint32_t)((blk * kPreGroups + g) * kBlock + threadIdx.x) * VEC;
        active[g] = p0[g] < (uint32_t)a.HW;
        in_base[g] = (int64_t)clip * a.clip_stride + (active[g] ? p0[g] : 0u);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { cp[g][j] = 0; cn[g][j] = 0; }
    }
    auto derive = [&](int g, uint32_t field) {      // Q(nominal / threshold) of group g's pixels; stays 0 for pixels outside the frame
        double pt[VEC], nt[VEC];
        v2e_native_thres<VEC>(P, a.seed, clip_id, field, p0[g], g_gauss_icdf, pt, nt);     // one-off per pixel: table from global memory
#pragma unroll
        for (int j = 0; j < VEC; ++j) { cp[g][j] = shot_quant(pos_nominal / pt[j]); cn[g][j] = shot_quant(neg_nominal / nt[j]); }
    };
#pragma unroll
    for (int g = 0; g < kPreGroups; ++g)
        if (active[g]) derive(g, kV2eFThresA);
    auto frame_sum = [&](int k, const Raw<IN, VEC> (&raw)[kPreGroups]) __attribute__((always_inline)) {
        long long sp = 0, sn = 0;
#pragma unroll
        for (int g = 0; g < kPreGroups; ++g) {
            if (temporal && active[g]) derive(g, kV2eFFrame0 + kV2eFStride * (uint32_t)(k + 1));
            float x[VEC];
            int32_t q[VEC];
            v2e_pixels<IN, VEC>(raw[g], x);
            uint32_t mismatch = 0;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                uint32_t idx;
                if constexpr (IN == kInU8) idx = VEC == 4 ? (raw[g].v >> (8 * j)) & 0xFFu : raw[g].v;
                else { idx = __float_as_uint(x[j] + 8388608.0f) & 255u; mismatch |= __float_as_uint((float)idx - x[j]); }
                q[j] = s_q[idx];
            }
            if constexpr (IN != kInU8) {
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(mismatch != 0) != 0, 0)) {     // some pixel is not an integer in 0..255
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        if ((float)(__float_as_uint(x[j] + 8388608.0f) & 255u) != x[j]) {
                            double i64, fac; float i32;
                            v2e_inten_direct<IN>(x[j], P.uint8_wrap, i64, i32, fac);
                            q[j] = shot_quant(fac);
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) { sp = mad_i64_i32(q[j], cp[g][j], sp); sn = mad_i64_i32(q[j], cn[g][j], sn); }
        }
        sp = wave_sum_i64(sp);
        sn = wave_sum_i64(sn);
        if ((threadIdx.x & 63) == 0) {                                   // LDS atomics: 4 waves per workgroup
            atomicAdd(&s_sum[4 * k], (unsigned long long)sp & 0xFFFFFFFFull);
            atomicAdd(&s_sum[4 * k + 1], (unsigned long long)(sp >> 32));
            atomicAdd(&s_sum[4 * k + 2], (unsigned long long)sn & 0xFFFFFFFFull);
            atomicAdd(&s_sum[4 * k + 3], (unsigned long long)(sn >> 32));
        }
    };
    {   // register ring of DEPTH frames x kPreGroups groups, reloaded right after use (clamped, unconditional loads)
        Raw<IN, VEC> ring[DEPTH][kPreGroups];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u)
#pragma unroll
            for (int g = 0; g < kPreGroups; ++g)
                ring[u][g] = load_raw<IN, VEC, NT>(a.frames, in_base[g] + (int64_t)(1 + u <= a.K ? 1 + u : a.K) * a.frame_stride);
        int k0 = 0;
        for (; k0 + DEPTH <= a.K; k0 += DEPTH) {
            static_for(std::make_integer_sequence<int, DEPTH>{}, [&](auto u_tag) {
                constexpr int u = decltype(u_tag)::value;
                Raw<IN, VEC> raw[kPreGroups];
                const int fn = k0 + u + 1 + DEPTH;
#pragma unroll
                for (int g = 0; g < kPreGroups; ++g) {
                    raw[g] = ring[u][g];
                    ring[u][g] = load_raw<IN, VEC, NT>(a.frames, in_base[g] + (int64_t)(fn <= a.K ? fn : a.K) * a.frame_stride);
                }
                frame_sum(k0 + u, raw);
            });
        }
        static_for(std::make_integer_sequence<int, DEPTH - 1>{}, [&](auto u_tag) {
            constexpr int u = decltype(u_tag)::value;
            if (k0 + u < a.K) frame_sum(k0 + u, ring[u]);
        });
    }
    // one global atomic per (workgroup, frame, accumulator): ~256 waves of a clip adding into the same address every frame
    // serialised at the memory side and cost more than the whole read of the clip
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * a.K; t += kBlock)
        atomicAdd(reinterpret_cast<unsigned long long *>(&a.shot_sums[(int64_t)clip * a.K * 4 + t]), s_sum[t]);
}

template <int IN, int VEC>
__global__ void __launch_bounds__(kBlock) v2e_shot_sum_kernel(const V2eArgs a)
{
    __shared__ int32_t s_q[256];
    extern __shared__ __align__(16) unsigned long long s_sum[];          // [K,4] workgroup partial sums
    const int clip = blockIdx.x / a.pre_blocks_per_clip;
    v2e_presum_body<IN, VEC, true, 2>(a, clip, blockIdx.x - clip * a.pre_blocks_per_clip, s_q, s_sum);
}

// ---- native shot-noise sampler (float32 inversion from one uniform; the CPU oracle restates it) ----------------------
// exp(-lam) on [0,1]: degree-6 minimax polynomial (1.5e-8; p(0) = 1 exactly, so lam = 0 never fires); above 1 the
// range-reduced expf_det.  Written on 2-vectors: the ON and OFF means of a pixel go through one packed chain.
__device__ __forceinline__ f32x2 exp_neg_small_x2(f32x2 lam)
{
    f32x2 e = pk_splat(0x1.be1ddep-11f);
    e = pk_fma(e, lam, pk_splat(-0x1.f60198p-8f));
    e = pk_fma(e, lam, pk_splat(0x1.51c0fcp-5f));
    e = pk_fma(e, lam, pk_splat(-0x1.5507a6p-3f));
    e = pk_fma(e, lam, pk_splat(0x1.fff9acp-2f));
    e = pk_fma(e, lam, pk_splat(-0x1.ffffcep-1f));
    e = pk_fma(e, lam, pk_splat(1.0f));
    return e;
}

// Per-frame constants every work-item of a clip shares (time step, low-pass step, refractory cap, shot-noise scales):
// tabulated once per workgroup in LDS -- they cost float64 divisions, which the time loop must not repeat.
struct __align__(16) V2eFrameConst { double dt, dt_tau, cap; float scale_p, scale_n; };      // 32 bytes
// Per-intensity terms (lin_log value, inten01, shot-noise intensity factor): ONE 16-byte LDS entry per 8-bit intensity,
// one ds_read_b128 per pixel and frame instead of three table reads with three address computations.
struct __align__(16) V2eIntenF32 { float logv, i01, fac, pad; };          // float32 input: inten01 is float32 (NumPy)
struct __align__(16) V2eIntenU8 { double i01; float logv, fac; };          // uint8 input: inten01 is float64

// FEAT < 0: model features are read from the parameters at run time (wave-uniform branches); FEAT >= 0: compile-time
// bit mask {1 low-pass, 2 leak, 4 shot noise, 16 per-frame thresholds} for the specialised instances (the refractory
// cap stays a run-time switch in every instance)
enum { kV2eLowpass = 1, kV2eLeak = 2, kV2eShot = 4, kV2eTemporal = 16 };

template <int IN, int VEC, int BIN, int RNG, bool OUT64, int FEAT>
__device__ __forceinline__ void v2e_main_body(const V2eArgs &a, const int clip, const int blk, unsigned char *s_raw)
{
    using acc_t = typename std::conditional<OUT64, double, float>::type;
    using inten_t = typename std::conditional<IN == kInU8, V2eIntenU8, V2eIntenF32>::type;
    static_assert(sizeof(V2eFrameConst) == 32 && sizeof(inten_t) == 16, "LDS record layout");
    constexpr bool PK = !OUT64 && BIN == kBinBilinear && VEC == 4;        // packed bilinear accumulation (see the ESIM kernel)
    // the Gaussian generator's table: a static LDS array of the device-native instances (compile-time address, see v2v_esim.hpp)
    float *s_icdf = nullptr;
    if constexpr (RNG == kRngPhilox) { __shared__ __align__(16) float s_icdf_static[kIcdfEntries]; s_icdf = s_icdf_static; icdf_to_lds(s_icdf); }
    __shared__ __align__(16) inten_t s_int[256];                                                // static as well: no base add per read
    V2eFrameConst *s_fc = reinterpret_cast<V2eFrameConst *>(s_raw);                             // [K]
    acc_t *s_wlo = reinterpret_cast<acc_t *>(s_fc + a.K);
    acc_t *s_whi = s_wlo + a.K;
    int *s_seg = reinterpret_cast<int *>(s_whi + a.K);
    const V2eParams &P = a.P;
    const bool lowpass = FEAT < 0 ? P.cutoff_hz > 0 : (FEAT & kV2eLowpass) != 0;
    const bool leak = FEAT < 0 ? P.leak_rate_hz > 0 : (FEAT & kV2eLeak) != 0;
    const bool shot = FEAT < 0 ? P.shot_noise_rate_hz > 0 : (FEAT & kV2eShot) != 0;
    const bool refractory = P.refractory_period_s > 0;   // always a run-time (wave-uniform) switch
    const bool temporal = FEAT < 0 ? P.threshold_model == kV2eSpatialTemporalIndependent : (FEAT & kV2eTemporal) != 0;
    // float32 input, low-pass on, feature-specialised instance: the IIR step's two intensity-dependent factors are TABULATED.
    // eps = min(inten01 * float32(dt / tau), 1) depends on the 8-bit intensity and on dt / tau only, and float32(dt / tau) is the
    // same bit pattern for every frame of a clip (dt = i/fps - (i-1)/fps wobbles in the last float64 ulp, which the float32 cast
    // absorbs: checked on the host, launch_v2e routes anything else to the run-time-feature kernel) -- so (1 - eps) and
    // eps * log_new, with the reference's own roundings, sit in the intensity record in place of log value and inten01:
    // lp = E1 * lp + E2, two instructions instead of six.
    constexpr bool LP_TAB = V2V_V2E_LP_TABLE && FEAT >= 0 && (FEAT & kV2eLowpass) != 0 && IN == kInF32;
    {   // ---- workgroup prologue: the two LDS tables
        double i64, fac; float i32;
        v2e_inten_direct<IN>((float)threadIdx.x, P.uint8_wrap, i64, i32, fac);
        inten_t e;
        e.logv = a.lut[threadIdx.x];
        e.fac = (float)fac;
        if constexpr (IN == kInU8) e.i01 = i64;
        else {
            e.i01 = i32; e.pad = 0.0f;
            if constexpr (LP_TAB) {
                const double dt1 = 1.0 / P.fps - 0.0 / P.fps;                                    // frame 1's time step; every frame's float32(dt / tau) equals it
                const float dtt = (float)(dt1 / (1 / (3.141592653589793 * 2 * P.cutoff_hz)));
                float eps = i32 * dtt;
                eps = __builtin_fminf(eps, 1.0f);
                const float lg = e.logv;
                e.logv = 1.0f - eps;                                                             // E1
                e.i01 = eps * lg;                                                                // E2
            }
        }
        s_int[threadIdx.x] = e;
    }
    const double tau = lowpass ? 1 / (3.141592653589793 * 2 * P.cutoff_hz) : 0.0;
    const double pos_nominal = P.thres_mean_mean + P.thres_diff_mean / 2, neg_nominal = P.thres_mean_mean - P.thres_diff_mean / 2;
    for (int k = threadIdx.x; k < a.K; k += kBlock) {
        const int i = k + 1;
        const double dt = (double)i / P.fps - (double)(i - 1) / P.fps;                  // t_frame - t_previous (:440)
        V2eFrameConst fc;
        fc.dt = dt;
        fc.dt_tau = lowpass ? dt / tau : 0.0;
        fc.scale_p = 0.0f; fc.scale_n = 0.0f;
        if (RNG == kRngPhilox && shot) {                                                // generate_shot_noise (:86-100)
            const long long *s4 = a.shot_sums + ((int64_t)clip * a.K + k) * 4;
            const double mean_p = (shot_sum_value(s4, 0) / 1099511627776.0) / (double)a.HW;      // / 2^40: two factors at 2^20
            const double mean_n = (shot_sum_value(s4, 1) / 1099511627776.0) / (double)a.HW;
            const double f = (P.shot_noise_rate_hz / 2) * dt;
            fc.scale_p = (float)(f / mean_p) * (float)pos_nominal;      // x nominal threshold: the per-pixel factor is 1/threshold
            fc.scale_n = (float)(f / mean_n) * (float)neg_nominal;
        }
        fc.cap = refractory ? (double)(int)(dt / P.refractory_period_s) : 0.0;
        s_fc[k] = fc;
        if constexpr (BIN == kBinBilinear) {
            const double t_norm = ((double)k - 0.0) / ((double)(a.K - 1) - 0.0) * (double)(a.Tb - 1);
            int b0 = (int)floor(t_norm);
            if (b0 > a.Tb - 2) b0 = a.Tb - 2;
            if (b0 < 0) b0 = 0;
            const double wl = 1.0 - fabs(t_norm - (double)b0), wh = 1.0 - fabs(t_norm - (double)(b0 + 1));
            s_wlo[k] = (acc_t)(wl > 0.0 ? wl : 0.0);
            s_whi[k] = (acc_t)(wh > 0.0 ? wh : 0.0);
            s_seg[k] = b0;
        }
    }
    __syncthreads();

    const uint32_t p0 = (uint32_t)(blk * kBlock + threadIdx.x) * VEC;
    if (p0 >= (uint32_t)a.HW) return;
    const uint32_t clip_id = (uint32_t)(a.clip_id0 + (uint64_t)clip);
    const bool lp32 = !lowpass || (IN == kInF32);
    const bool base32 = lp32 && !leak;
    const int64_t in_base = (int64_t)clip * a.clip_stride + p0;
    const int64_t pix_base = (int64_t)clip * a.HW + p0;

    // ---- frame 0: lp = base = lin_log(frame 0); thresholds; log-normal leak-rate factor (_init :317-349)
    double lp64[VEC], base64[VEC], pt[VEC], nt[VEC];
    float lp_f[VEC], base_f[VEC], nrate[VEC];
    {
        const Raw<IN, VEC> r0 = load_raw<IN, VEC>(a.frames, in_base);
        float x[VEC];
        v2e_pixels<IN, VEC>(r0, x);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { lp_f[j] = base_f[j] = v2e_linlog(x[j], a.lut); lp64[j] = base64[j] = (double)lp_f[j]; }
    }
    if constexpr (RNG == kRngPhilox) {
        v2e_native_thres<VEC>(P, a.seed, clip_id, kV2eFThresA, p0, s_icdf, pt, nt);
        float g[VEC], g_unused[VEC];
        field_gauss_pairs<VEC>(a.seed, clip_id, kV2eFNoiseRate, kStreamV2e, p0, s_icdf, g, g_unused);
        const float c = (float)(2.302585092994046 * P.noise_rate_cov_decades);
#pragma unroll
        for (int j = 0; j < VEC; ++j) nrate[j] = expf_det(c * g[j]);
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            // static thresholds [B,HW]; per-frame ones ([B,K,HW]) are fetched inside the time loop
            pt[j] = a.r_thres_frame_stride ? 1.0 : a.r_pos_thres[pix_base + j];
            nt[j] = a.r_thres_frame_stride ? 1.0 : a.r_neg_thres[pix_base + j];
            nrate[j] = a.r_noise_rate[pix_base + j];
        }
    }

    // per-pixel constants derived from the thresholds (recomputed per frame only for the temporal model): float32
    // reciprocals biased low by 2^-22 -- the quotient estimate of the exact floor-divide (never above the true quotient,
    // short by one with probability ~3e-7 x quotient -> rare fix-up path; quotients below ~3e6) and, times the nominal
    // threshold, the threshold factor of the shot-noise mean.  float32: two registers per pixel instead of six.
    float inv_p[VEC], inv_n[VEC];
    auto derive_thres = [&]() {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            inv_p[j] = (float)((1.0 / pt[j]) * 0x1.fffff8p-1);
            inv_n[j] = (float)((1.0 / nt[j]) * 0x1.fffff8p-1);
        }
    };
    // float -> uint32 with the hardware's saturation (negative and NaN -> 0, >= 2^32 -> 0xFFFFFFFF): v_cvt_u32_f32 IS the
    // clip(x, 0, inf) + floor of compute_event_map for the quotient estimate; a C cast of a negative float is undefined
    auto cvt_u32_sat = [](float t) -> uint32_t { uint32_t q; asm("v_cvt_u32_f32 %0, %1" : "=v"(q) : "v"(t)); return q; };
    derive_thres();

    float leak_cur[VEC];                                   // float32 product leak_rate_hz * noise_rate_array (:204); widening it once
                                                           // instead of every frame (4 more VGPRs) measured flat
#pragma unroll
    for (int j = 0; j < VEC; ++j) leak_cur[j] = (float)P.leak_rate_hz * nrate[j];

    acc_t acc_lo[VEC], acc_hi[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc_lo[j] = 0; acc_hi[j] = 0; }
    int cur_seg = 0, sub = 0, plane = 0;
    const int64_t planes_per_clip = (BIN == kBinSum) ? (a.K / a.fpb) : a.Tb;
    const int64_t out_base = (int64_t)clip * planes_per_clip * a.HW + p0;
    uint32_t n_on = 0, n_off = 0;
    const bool want_counts = a.counts != nullptr;
    float gleak_pend[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) gleak_pend[j] = 0.0f;

    // PAR (compile-time): k & 1 -- the leak-jitter normals come as deviate pairs (the two 16-bit halves of one Philox word, table inversion) shared by two consecutive frame pairs
    auto step = [&](auto par_tag, int k, const Raw<IN, VEC> &raw) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        const int i = k + 1;
        if constexpr (BIN == kBinBilinear) {
            const int seg = __builtin_amdgcn_readfirstlane(s_seg[k]);
            while (cur_seg < seg) {
                store_vec<VEC, acc_t>(a.out, out_base + (int64_t)cur_seg * a.HW, acc_lo);
#pragma unroll
                for (int j = 0; j < VEC; ++j) { acc_lo[j] = acc_hi[j]; acc_hi[j] = 0; }
                ++cur_seg;
            }
        }
        const V2eFrameConst fc = s_fc[k];                                               // wave-uniform LDS read
        const double dt = fc.dt, dt_tau = fc.dt_tau, cap = fc.cap;
        const float dt_tau32 = (float)dt_tau;
        if (temporal) {                                                                 // thresholds redrawn per frame (:417-421)
            if constexpr (RNG == kRngPhilox) v2e_native_thres<VEC>(P, a.seed, clip_id, kV2eFFrame0 + kV2eFStride * (uint32_t)i, p0, s_icdf, pt, nt);
            else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int64_t o = ((int64_t)clip * a.K + k) * a.HW + p0 + j;
                    pt[j] = a.r_pos_thres[o];
                    nt[j] = a.r_neg_thres[o];
                }
            }
            derive_thres();
        }
        float gleak[VEC];
        float u_sp[VEC], u_sn[VEC];
        if constexpr (RNG == kRngPhilox) {
            if (leak) {                     // one deviate pair (one Philox word) per pixel and couple of frame pairs (2m, 2m+1): block of couple m
                if constexpr (PAR == 0) field_gauss_pairs<VEC, kNoiseRounds>(a.seed, clip_id, kV2eFFrame0 + kV2eFStride * (uint32_t)(k >> 1) + 2u, kStreamV2e, p0, s_icdf, gleak, gleak_pend);
                else {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) gleak[j] = gleak_pend[j];
                }
            }
            if (shot) field_uniform16x2<VEC, kNoiseRounds>(a.seed, clip_id, kV2eFFrame0 + kV2eFStride * (uint32_t)i + 3u, kStreamV2e, p0, u_sp, u_sn);
        }
        const acc_t wl = BIN == kBinBilinear ? s_wlo[k] : (acc_t)1, wh = BIN == kBinBilinear ? s_whi[k] : (acc_t)0;

        // ---- lin_log (:445) + intensity terms: one 16-byte table entry per pixel, indexed by the 8-bit intensity; a wave
        // leaves the table path only when some float32 pixel is not an integer in 0..255
        float x[VEC], log_new[VEC], i01_32[VEC], fac32[VEC];
        double i01_64[VEC];
        v2e_pixels<IN, VEC>(raw, x);
        uint32_t mismatch = 0;                            // != 0 iff some float32 pixel is not an integer in 0..255 (or NaN)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            uint32_t idx;
            if constexpr (IN == kInU8) idx = VEC == 4 ? (raw.v >> (8 * j)) & 0xFFu : raw.v;
            else { idx = __float_as_uint(x[j] + 8388608.0f) & 255u; mismatch |= __float_as_uint((float)idx - x[j]); }
            const inten_t e = s_int[idx];
            log_new[j] = e.logv;
            fac32[j] = e.fac;
            if constexpr (IN == kInU8) { i01_64[j] = e.i01; i01_32[j] = 0.0f; }
            else { i01_32[j] = e.i01; i01_64[j] = 0.0; }
        }
        if constexpr (IN != kInU8) {
            if (__builtin_expect(__builtin_amdgcn_ballot_w64(mismatch != 0) != 0, 0)) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    if ((float)(__float_as_uint(x[j] + 8388608.0f) & 255u) != x[j]) {
                        double fac;
                        log_new[j] = v2e_linlog(x[j], a.lut);
                        v2e_inten_direct<IN>(x[j], P.uint8_wrap, i01_64[j], i01_32[j], fac);
                        fac32[j] = (float)fac;
                        if constexpr (LP_TAB) {                                                  // (E1, E2) of a pixel outside the table
                            float eps = i01_32[j] * dt_tau32;
                            eps = __builtin_fminf(eps, 1.0f);
                            const float lg = log_new[j];
                            log_new[j] = 1.0f - eps;
                            i01_32[j] = eps * lg;
                        }
                    }
                }
            }
        }

        // ---- low pass, leak, event map.  floor_divide(clip(+-diff, 0), thres): the quotient is ESTIMATED in float32 --
        // trunc(float(diff) * reciprocal); v_cvt_u32_f32 clips the negative side to 0.  The estimate can never EXCEED the true
        // quotient: reciprocal = fl32(fl64(1/thres) * (1 - 2^-22)), and the three float32 roundings on the way (the reciprocal's,
        // float(diff)'s, the product's: <= 2^-24 each, plus 2^-53 of the float64 division) leave the real value of the product at
        // most diff/thres * (1 - 2^-22)(1 + 2^-24)^3(1 + 2^-53) < diff/thres * (1 - 2^-24), and truncation only lowers it.  So only
        // the UPPER bound needs a run-time test: the exact (sign-exact fma) float64 residual diff - q*thres < thres.  An estimate
        // one short, a quotient beyond float32's integers or a non-finite operand fail it and send the wave down the exact
        // float64 path below (the lower bound 0 <= residual holds by the inequality above and is not tested).
        uint32_t qp[VEC], qn[VEC];
        unsigned long long fix = 0;                       // wave-level masks (SGPR pairs): no per-lane bool materialised
        auto cur_diff = [&](int j) -> double {            // lp - base in NumPy's dtypes (recomputed by the rare fix-up path)
            if (lp32 && base32) { const float d = lp_f[j] - base_f[j]; return (double)d; }
            return (lp32 ? (double)lp_f[j] : lp64[j]) - base64[j];
        };
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (lowpass) {                                                              // low_pass_filter (:139-182)
                if constexpr (LP_TAB) {
                    const float ta = log_new[j] * lp_f[j];                          // log_new = E1 = 1 - eps, i01_32 = E2 = eps * log value
                    lp_f[j] = ta + i01_32[j];
                } else if constexpr (IN == kInF32) {
                    float eps = i01_32[j] * dt_tau32;
                    eps = __builtin_fminf(eps, 1.0f);                               // np.minimum (:173); eps is never NaN here: one v_min_f32
                    const float ta = (1.0f - eps) * lp_f[j], tb2 = eps * log_new[j];
                    lp_f[j] = ta + tb2;
                } else {
                    double eps = i01_64[j] * dt_tau;
                    eps = eps > 1.0 ? 1.0 : eps;
                    const double ta = (1 - eps) * lp64[j], tb2 = eps * (double)log_new[j];
                    lp64[j] = ta + tb2;
                }
            } else {
                lp_f[j] = log_new[j];
            }
            if (leak) {                                                                 // subtract_leak_current (:192-211)
                double g;
                if constexpr (RNG == kRngPhilox) g = (double)gleak[j];
                else g = a.r_leak_randn[((int64_t)clip * a.K + k) * a.HW + p0 + j];
                const double jit = P.leak_jitter_fraction * g;
                const double curr = (double)leak_cur[j] * (1 - jit);
                const double dl = dt * curr * pt[j];
                base64[j] = base64[j] - dl;
            }
            const double diff = cur_diff(j);
            // compute_event_map (:42-62).  The residuals are taken against +-diff itself: for the side that is clipped
            // to zero they are negative (check passes), for a NaN difference they are NaN (check fails -> exact path)
            const float d32 = (float)diff;
            qp[j] = cvt_u32_sat(d32 * inv_p[j]);
            qn[j] = cvt_u32_sat(-d32 * inv_n[j]);
            const double fq = (double)qp[j], fn = (double)qn[j];
            const double rp = __builtin_fma(-fq, pt[j], diff), rn = __builtin_fma(-fn, nt[j], -diff);
            fix |= __ballot(!(rp < pt[j]));
            fix |= __ballot(!(rn < nt[j]));
        }

        int cnt_p[VEC], cnt_n[VEC];                       // native shot-noise counts (0 without shot noise / in replay mode)
#pragma unroll
        for (int j = 0; j < VEC; ++j) { cnt_p[j] = 0; cnt_n[j] = 0; }
        if (shot) {                                                                     // generate_shot_noise (:65-105)
            if constexpr (RNG == kRngPhilox) {
                // mean = (intensity factor / threshold) x (nominal threshold x rate/2 dt / frame mean) in float32; inversion
                // from the 16-bit uniforms: counts 0..3 from three thresholds p0, p0(1+l), p0(1+l+l^2/2) without a branch or
                // a division.  Wave-level rare paths (a mean above 1; a count above 3) recompute what they need, so the
                // common path keeps no per-pixel temporaries alive across them.
                const f32x2 sc = f32x2{fc.scale_p, fc.scale_n};
                auto mean_of = [&](int j) { return (pk_splat(fac32[j]) * f32x2{inv_p[j], inv_n[j]}) * sc; };
                auto p0_of = [&](f32x2 lam) {
                    f32x2 pz = exp_neg_small_x2(lam);
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(lam.x > 1.0f || lam.y > 1.0f) != 0, 0)) {
                        if (lam.x > 1.0f) pz.x = expf_det(-lam.x);
                        if (lam.y > 1.0f) pz.y = expf_det(-lam.y);
                    }
                    return pz;
                };
                unsigned long long more = 0;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const f32x2 lam = mean_of(j);
                    const f32x2 pz = p0_of(lam);
                    const f32x2 q1 = lam + pk_splat(1.0f);
                    const f32x2 q2 = pk_fma(lam * pk_splat(0.5f), lam, q1);
                    const f32x2 s1 = pz * q1, s2 = pz * q2;
                    // count = [u > p0] + [u > s1] + [u > s2]: the sign bit of (threshold - u) IS that comparison (an IEEE difference
                    // is negative exactly when u is larger), so three packed subtractions, shifts and one add3 replace six
                    // compare + select pairs
                    const f32x2 uu = f32x2{u_sp[j], u_sn[j]};
                    const f32x2 d0 = pz - uu, d1 = s1 - uu, d2 = s2 - uu;
                    cnt_p[j] = (int)((__float_as_uint(d0.x) >> 31) + (__float_as_uint(d1.x) >> 31) + (__float_as_uint(d2.x) >> 31));
                    cnt_n[j] = (int)((__float_as_uint(d0.y) >> 31) + (__float_as_uint(d1.y) >> 31) + (__float_as_uint(d2.y) >> 31));
                    more |= __ballot((int)__float_as_uint(d2.x) < 0) | __ballot((int)__float_as_uint(d2.y) < 0);
                }
                if (__builtin_expect(more != 0, 0)) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        f32x2 lam = mean_of(j);
                        asm volatile("" : "+v"(lam));      // opaque: a recomputation, not a value kept alive from the common path
                        const f32x2 pz = p0_of(lam);
                        const f32x2 hl = lam * pk_splat(0.5f);
                        const f32x2 s2 = pz * pk_fma(hl, lam, lam + pk_splat(1.0f));
                        if (u_sp[j] > s2.x) cnt_p[j] = (int)poisson_tail_f32(lam.x, u_sp[j], pz.x * (hl.x * lam.x), s2.x);
                        if (u_sn[j] > s2.y) cnt_n[j] = (int)poisson_tail_f32(lam.y, u_sn[j], pz.y * (hl.y * lam.y), s2.y);
                    }
                }
            }
        }

        // ---- totals.  Common case: signal quotient + shot count as INTEGERS, one convert each.  A failed residual check takes the
        // exact float64 evaluation (quotient one short; NaN / infinite operands keep NumPy's results: np.floor_divide's own terms)
        double fpos[VEC], fneg[VEC];
        if (__builtin_expect(fix == 0, 1)) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { fpos[j] = (double)(qp[j] + (uint32_t)cnt_p[j]); fneg[j] = (double)(qn[j] + (uint32_t)cnt_n[j]); }
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                double diff = cur_diff(j);
                asm volatile("" : "+v"(diff));             // opaque: recomputed here, not kept alive from the common path
                const double nd = -diff;
                const double pos_frame = diff > 0 ? diff : (diff == diff ? 0.0 : diff);
                const double neg_frame = nd > 0 ? nd : (nd == nd ? 0.0 : nd);
                fpos[j] = floor(pos_frame * (double)inv_p[j]);
                fneg[j] = floor(neg_frame * (double)inv_n[j]);
                { const double r = __builtin_fma(-fpos[j], pt[j], pos_frame); if (r >= pt[j]) fpos[j] += 1.0; else if (!(r < pt[j])) fpos[j] = pos_frame / pt[j]; }
                { const double r = __builtin_fma(-fneg[j], nt[j], neg_frame); if (r >= nt[j]) fneg[j] += 1.0; else if (!(r < nt[j])) fneg[j] = neg_frame / nt[j]; }
                fpos[j] = fpos[j] + (double)cnt_p[j];
                fneg[j] = fneg[j] + (double)cnt_n[j];
            }
        }
        if constexpr (RNG != kRngPhilox) {
            if (shot) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const int64_t o = ((int64_t)clip * a.K + k) * a.HW + p0 + j;
                    fpos[j] = fpos[j] + (double)a.r_shot_pos[o];
                    fneg[j] = fneg[j] + (double)a.r_shot_neg[o];
                }
            }
        }

        if (refractory) {                                                               // intended semantics of :534-537
            asm volatile("" ::: "memory");                                              // keep it a (scalar) branch, not eight selects
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                fpos[j] = fpos[j] > cap ? cap : fpos[j];
                fneg[j] = fneg[j] > cap ? cap : fneg[j];
            }
        }
        float vfs[PK ? VEC : 1];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (base32) {                                                               // in-place += on a float32 array (:547-548)
                const double up = fpos[j] * pt[j];
                base_f[j] = (float)((double)base_f[j] + up);
                const double dn = fneg[j] * nt[j];
                base_f[j] = (float)((double)base_f[j] - dn);
            } else {
                const double up = fpos[j] * pt[j];
                base64[j] = base64[j] + up;
                const double dn = fneg[j] * nt[j];
                base64[j] = base64[j] - dn;
            }
            const double vox = fpos[j] - fneg[j];
            if constexpr (OUT64) {
                if constexpr (BIN == kBinBilinear) {
                    const double cl = vox * wl, ch = vox * wh;
                    acc_lo[j] = acc_lo[j] + cl;
                    acc_hi[j] = acc_hi[j] + ch;
                } else {
                    acc_lo[j] = acc_lo[j] + vox;
                }
            } else {
                const float vf = (float)vox;
                if constexpr (PK) {
                    vfs[j] = vf;
                } else if constexpr (BIN == kBinBilinear) {
                    acc_lo[j] = __builtin_fmaf(vf, wl, acc_lo[j]);
                    acc_hi[j] = __builtin_fmaf(vf, wh, acc_hi[j]);
                } else {
                    acc_lo[j] = acc_lo[j] + vf;
                }
            }
        }
        if constexpr (PK) {
#pragma unroll
            for (int j = 0; j < VEC; j += 2) {
                const f32x2 v2 = f32x2{vfs[j], vfs[j + 1]};
                const f32x2 lo = pk_fma(v2, pk_splat((float)wl), f32x2{(float)acc_lo[j], (float)acc_lo[j + 1]});
                const f32x2 hi = pk_fma(v2, pk_splat((float)wh), f32x2{(float)acc_hi[j], (float)acc_hi[j + 1]});
                acc_lo[j] = lo.x; acc_lo[j + 1] = lo.y; acc_hi[j] = hi.x; acc_hi[j + 1] = hi.y;
            }
        }
        if (want_counts) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) { n_on += (uint32_t)fpos[j]; n_off += (uint32_t)fneg[j]; }
        }
        if constexpr (BIN == kBinSum) {
            if (++sub == a.fpb) {
                store_vec<VEC, acc_t>(a.out, out_base + (int64_t)plane * a.HW, acc_lo);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc_lo[j] = 0;
                sub = 0;
                ++plane;
            }
        }
    };

    // ---- time loop: register ring of kV2eDepth frames, reloaded right after use (clamped, unconditional loads), unrolled
    //      by an even factor so that the parity of k is a compile-time constant
    constexpr int kRing = V2V_V2E_DEPTH;
    static_assert(kRing % 2 == 0, "the v2e time loop must be unrolled by an even factor (leak-jitter pairs)");
    {
        Raw<IN, VEC> ring[kRing];
#pragma unroll
        for (int u = 0; u < kRing; ++u) {
            const int f = (1 + u <= a.K) ? 1 + u : a.K;
            ring[u] = load_raw<IN, VEC>(a.frames, in_base + (int64_t)f * a.frame_stride);
        }
        int k0 = 0;
        for (; k0 + kRing <= a.K; k0 += kRing) {
            static_for(std::make_integer_sequence<int, kRing>{}, [&](auto u_tag) {
                constexpr int u = decltype(u_tag)::value;
                const int k = k0 + u;
                step(std::integral_constant<int, (u & 1)>{}, k, ring[u]);
                const int fn = k + 1 + kRing;
                ring[u] = load_raw<IN, VEC>(a.frames, in_base + (int64_t)(fn <= a.K ? fn : a.K) * a.frame_stride);
            });
        }
        static_for(std::make_integer_sequence<int, kRing - 1>{}, [&](auto u_tag) {
            constexpr int u = decltype(u_tag)::value;
            if (k0 + u < a.K) step(std::integral_constant<int, (u & 1)>{}, k0 + u, ring[u]);
        });
    }
    if constexpr (BIN == kBinBilinear) {
        store_vec<VEC, acc_t>(a.out, out_base + (int64_t)cur_seg * a.HW, acc_lo);
        if (cur_seg + 1 < a.Tb) store_vec<VEC, acc_t>(a.out, out_base + (int64_t)(cur_seg + 1) * a.HW, acc_hi);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc_lo[j] = 0;
        for (int b = cur_seg + 2; b < a.Tb; ++b) store_vec<VEC, acc_t>(a.out, out_base + (int64_t)b * a.HW, acc_lo);
    }
    if (want_counts) {
        unsigned long long on = n_on, off = n_off;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const unsigned long long o1 = __shfl_down(on, s, 64), o2 = __shfl_down(off, s, 64);
            const int src = (int)(threadIdx.x & 63) + s;
            const bool src_active = src < 64 && (uint32_t)((blk * kBlock + (threadIdx.x & ~63) + src)) * VEC < (uint32_t)a.HW;
            if (src_active) { on += o1; off += o2; }
        }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&a.counts[2 * clip], on); atomicAdd(&a.counts[2 * clip + 1], off); }
    }
}

// The run-time-feature instances (FEAT < 0: float64 grids, replayed fields, 1 pixel per work-item, the per-frame threshold model,
// a cut-off whose float32(dt / tau) takes two values -- parity modes and rare configurations; launch_v2e routes everything a
// dataset draws to the specialised ones) are compiled for 2 waves per SIMD: the float32 / 4-pixel ones kept 2-12 spilled VGPRs at 3.
template <int IN, int VEC, int BIN, int RNG, bool OUT64, int FEAT = -1>
__global__ void __launch_bounds__(kBlock, (FEAT < 0 && IN == kInF32 && VEC == 4) ? 2 : V2V_V2E_MIN_WAVES) v2e_voxel_kernel(const V2eArgs a)
{
    extern __shared__ __align__(16) unsigned char s_raw[];
    const int clip = blockIdx.x / a.blocks_per_clip;
    v2e_main_body<IN, VEC, BIN, RNG, OUT64, FEAT>(a, clip, blockIdx.x - clip * a.blocks_per_clip, s_raw);
}

}  // namespace v2v
