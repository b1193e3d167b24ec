// v2v_esim.hpp -- fused ESIM frame-pair simulator + voxel binning for gfx950 (MI355X).
//
// Replaces, for a whole batch of clips in ONE launch:
//   data/v2v_core_esim.py:26-69   EventEmulator.video_to_voxel   (per-pixel potential, threshold crossing)
//   data/v2v_datasets.py:399-400  reshape(L,Tb,fpb,H,W).sum(2)   (V2V_BIN_SUM)
//   utils/event_utils.py:692-728  temporal-bilinear voxel bins    (V2V_BIN_BILINEAR, pseudo-events at ts=k)
//
// Mapping to the hardware (memory-bound: pointwise in space, sequential scan in time, zero reuse):
//   * one work-item owns VEC=4 horizontally adjacent pixels of one clip and streams the clip's N frames
//     through registers: per-pixel state (float64 potential, previous log value, hot-pixel noise, bin
//     accumulators) never leaves the register file, so HBM traffic is exactly "read every input byte
//     once, write every voxel byte once".
//   * a wave reads one contiguous 1 KiB segment per frame (global_load_dwordx4 per lane, fp32 input),
//     frames are software-prefetched U at a time into a register ping-pong so ~2U KiB per wave are in flight.
//   * the 256-entry log-intensity table (NumPy's bits, golden G1) lives in LDS; integer-valued input
//     costs one ds_read per sample instead of a float64 pow+log.
//   * np.floor_divide's exact result for a >= b > 0 is the true floor of the real quotient; it is
//     obtained without a division: q = floor(a * (1/b)), one exact fma residual, and a +-1 correction.
//   * per-clip ON/OFF totals: lane-local counters -> wave reduction -> one 64-bit atomic per wave.
// No MFMA (nothing here is a contraction), no cross-workgroup communication, no collectives.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "v2v_rng.hpp"

namespace v2v {

enum { kInU8 = 0, kInF32 = 1 };
enum { kRngNone = 0, kRngPhilox = 1, kRngReplay = 2 };
enum { kBinSum = 0, kBinBilinear = 1 };

struct EsimArgs {
    const void *frames;
    int64_t clip_stride, frame_stride;     // elements
    const double *params;
    int64_t params_stride;
    void *out;
    unsigned long long *counts;            // [B,2] or nullptr
    const double *u_init, *u_hot, *g_hot, *g_base;
    uint64_t seed, clip_id0;
    int32_t HW, K, Tb, fpb, blocks_per_clip;
    uint32_t noise_external, out_f64;
};

// Log-intensity tables in device memory (initialised with NumPy's bits, golden G1; re-pinnable through
// v2v_lut_set).  Each workgroup copies the one it needs into LDS.
#include "v2v_luts.inc"
__device__ double g_lut_esim64[256] = {V2V_LUT_ESIM64_VALUES};
__device__ float g_lut_esim32[256] = {V2V_LUT_ESIM32_VALUES};
__device__ float g_lut_v2e32[256] = {V2V_LUT_V2E32_VALUES};
static const double kLutEsim64[256] = {V2V_LUT_ESIM64_VALUES};
static const float kLutEsim32[256] = {V2V_LUT_ESIM32_VALUES};
static const float kLutV2e32[256] = {V2V_LUT_V2E32_VALUES};

constexpr int kBlock = 256;
constexpr int kPrefetch = 4;   // frames per register buffer (two buffers ping-pong)

// ------------------------------------------------------------------------------------------------
// raw input vectors
template <int IN, int VEC> struct Raw;
template <> struct Raw<kInF32, 4> { float4 v; };
template <> struct Raw<kInF32, 1> { float v; };
template <> struct Raw<kInU8, 4> { uint32_t v; };
template <> struct Raw<kInU8, 1> { uint8_t v; };

template <int IN, int VEC>
__device__ __forceinline__ Raw<IN, VEC> load_raw(const void *base, int64_t elem_off)
{
    Raw<IN, VEC> r;
    if constexpr (IN == kInF32 && VEC == 4) r.v = *reinterpret_cast<const float4 *>(static_cast<const float *>(base) + elem_off);
    else if constexpr (IN == kInF32) r.v = static_cast<const float *>(base)[elem_off];
    else if constexpr (VEC == 4) r.v = *reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(base) + elem_off);
    else r.v = static_cast<const uint8_t *>(base)[elem_off];
    return r;
}

// float32 container, value not an integer in 0..255: the reference's own float32 expression
// (v2v_core_esim.py:3-4,33-34 evaluated by NumPy in float32); device powf/logf are within 1-2 ulp of NumPy's.
__device__ __noinline__ float esim_log_generic_f32(float v)
{
    const float lin = powf(v / 255.0f, 2.2f) * 255.0f;
    return logf(0.001f + lin / 255.0f);
}

template <int IN> struct LutT { using type = double; };
template <> struct LutT<kInF32> { using type = float; };

template <int IN, int VEC>
__device__ __forceinline__ typename LutT<IN>::type pix_log(const Raw<IN, VEC> &r, int j,
                                                           const typename LutT<IN>::type *lut)
{
    if constexpr (IN == kInU8) {
        uint32_t idx;
        if constexpr (VEC == 4) idx = (r.v >> (8 * j)) & 0xFFu; else idx = r.v;
        return lut[idx];
    } else {
        float v;
        if constexpr (VEC == 4) v = (j == 0) ? r.v.x : (j == 1) ? r.v.y : (j == 2) ? r.v.z : r.v.w; else v = r.v;
        const int i = (int)v;
        if (__builtin_expect((float)i == v && (unsigned)i < 256u, 1)) return lut[i];
        return esim_log_generic_f32(v);
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec(void *out, uint32_t out_f64, int64_t off, const double (&v)[VEC])
{
    if (out_f64) {
        double *o = static_cast<double *>(out) + off;
        if constexpr (VEC == 4) {
            reinterpret_cast<double2 *>(o)[0] = make_double2(v[0], v[1]);
            reinterpret_cast<double2 *>(o)[1] = make_double2(v[2], v[3]);
        } else o[0] = v[0];
    } else {
        float *o = static_cast<float *>(out) + off;
        if constexpr (VEC == 4) *reinterpret_cast<float4 *>(o) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
        else o[0] = (float)v[0];
    }
}

// np.floor_divide(a, b) for a >= b > 0 (numpy npy_divmod): literal form, used only when the quotient is
// too large for the reciprocal estimate to be within +-1 (never for physical thresholds).
__device__ __noinline__ double floor_divide_slow(double a, double b)
{
    const double mod = fmod(a, b);
    const double div = (a - mod) / b;
    double fl = floor(div);
    if (div - fl > 0.5) fl += 1.0;
    return fl;
}

template <int IN, int VEC, int BIN, int RNG>
__global__ void __launch_bounds__(kBlock) esim_voxel_kernel(const EsimArgs a)
{
    using lut_t = typename LutT<IN>::type;
    extern __shared__ __align__(16) unsigned char s_raw[];
    lut_t *s_lut = reinterpret_cast<lut_t *>(s_raw);
    double *s_wlo = reinterpret_cast<double *>(s_raw + 256 * sizeof(lut_t));
    double *s_whi = s_wlo + a.K;
    int *s_seg = reinterpret_cast<int *>(s_whi + a.K);

    // ---- workgroup prologue: tables into LDS
    if constexpr (IN == kInU8) s_lut[threadIdx.x] = g_lut_esim64[threadIdx.x];
    else s_lut[threadIdx.x] = g_lut_esim32[threadIdx.x];
    if constexpr (BIN == kBinBilinear) {
        // weight of pair k for its two neighbouring bins: the float64 expression of event_utils.py:715-719
        for (int k = threadIdx.x; k < a.K; k += kBlock) {
            const double t_norm = ((double)k - 0.0) / ((double)(a.K - 1) - 0.0) * (double)(a.Tb - 1);
            int b0 = (int)floor(t_norm);
            if (b0 > a.Tb - 2) b0 = a.Tb - 2;
            if (b0 < 0) b0 = 0;
            double wl = 1.0 - fabs(t_norm - (double)b0);
            double wh = 1.0 - fabs(t_norm - (double)(b0 + 1));
            s_wlo[k] = wl > 0.0 ? wl : 0.0;
            s_whi[k] = wh > 0.0 ? wh : 0.0;
            s_seg[k] = b0;
        }
    }
    __syncthreads();

    const int clip = blockIdx.x / a.blocks_per_clip;
    const int blk = blockIdx.x - clip * a.blocks_per_clip;
    const uint32_t p0 = (uint32_t)(blk * kBlock + threadIdx.x) * VEC;
    if (p0 >= (uint32_t)a.HW) return;

    const double *pp = a.params + (int64_t)clip * a.params_stride;
    const double pos = pp[0], neg = pp[1], base_std = pp[2], hot_frac = pp[3], hot_std = pp[4];
    const double inv_pos = 1.0 / pos, inv_neg = 1.0 / neg;
    const uint32_t clip_id = (uint32_t)(a.clip_id0 + (uint64_t)clip);
    const bool ext = a.noise_external != 0;

    const int64_t esz = 1;
    const int64_t in_base = (int64_t)clip * a.clip_stride + p0;
    (void)esz;

    // ---- per-pixel state
    double pot[VEC], hot[VEC];
    {
        double u0[VEC];
        if constexpr (RNG == kRngPhilox) {
            field_uniform53<VEC>(a.seed, clip_id, kFieldPotInit, kStreamEsim, p0, u0);
        } else if constexpr (RNG == kRngReplay) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) u0[j] = a.u_init[(int64_t)clip * a.HW + p0 + j];
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) u0[j] = 0.5;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const double scaled = u0[j] * (pos + neg);
            pot[j] = scaled - neg;                                     // v2v_core_esim.py:29
            hot[j] = 0.0;
        }
        if constexpr (RNG == kRngPhilox) {
            if (hot_frac > 0.0) {                                      // uniform: skipping is exact (u >= 0)
                double u1[VEC];
                float gh[VEC];
                field_uniform53<VEC>(a.seed, clip_id, kFieldHotMask, kStreamEsim, p0, u1);
                field_gauss32<VEC>(a.seed, clip_id, kFieldHotGauss, kStreamEsim, p0, gh);
#pragma unroll
                for (int j = 0; j < VEC; ++j) hot[j] = (u1[j] < hot_frac) ? hot_std * (double)gh[j] : 0.0;   // :37-39
            }
        } else if constexpr (RNG == kRngReplay) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const int64_t o = (int64_t)clip * a.HW + p0 + j;
                hot[j] = (a.u_hot[o] < hot_frac) ? hot_std * a.g_hot[o] : 0.0;
            }
        }
    }

    lut_t lprev[VEC];
    {
        const Raw<IN, VEC> r0 = load_raw<IN, VEC>(a.frames, in_base);
#pragma unroll
        for (int j = 0; j < VEC; ++j) lprev[j] = pix_log<IN, VEC>(r0, j, s_lut);
    }

    // ---- binning state
    double acc_lo[VEC], acc_hi[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { acc_lo[j] = 0.0; acc_hi[j] = 0.0; }
    int cur_seg = 0;          // BILINEAR: bin index acc_lo belongs to
    int sub = 0, plane = 0;   // SUM: pairs accumulated into the current plane, plane index
    const int64_t planes_per_clip = (BIN == kBinSum) ? (a.K / a.fpb) : a.Tb;
    const int64_t out_base = (int64_t)clip * planes_per_clip * a.HW + p0;
    uint32_t n_on = 0, n_off = 0;
    const bool want_counts = a.counts != nullptr;

    auto step = [&](int k, const Raw<IN, VEC> &raw) {
        if constexpr (BIN == kBinBilinear) {
            const int seg = s_seg[k];
            while (cur_seg < seg) {                                    // wave-uniform
                store_vec<VEC>(a.out, a.out_f64, out_base + (int64_t)cur_seg * a.HW, acc_lo);
#pragma unroll
                for (int j = 0; j < VEC; ++j) { acc_lo[j] = acc_hi[j]; acc_hi[j] = 0.0; }
                ++cur_seg;
            }
        }
        double base[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) base[j] = 0.0;
        if constexpr (RNG == kRngPhilox) {
            if (base_std != 0.0) {                                     // uniform; 0*g adds nothing
                float g[VEC];
                field_gauss32<VEC>(a.seed, clip_id, kFieldBase0 + (uint32_t)k, kStreamEsim, p0, g);
#pragma unroll
                for (int j = 0; j < VEC; ++j) base[j] = base_std * (double)g[j];       // :44
            }
        } else if constexpr (RNG == kRngReplay) {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                base[j] = base_std * a.g_base[((int64_t)clip * a.K + k) * a.HW + p0 + j];
        }
        double wl = 1.0, wh = 0.0;
        if constexpr (BIN == kBinBilinear) { wl = s_wlo[k]; wh = s_whi[k]; }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const lut_t ln = pix_log<IN, VEC>(raw, j, s_lut);
            const lut_t d = ln - lprev[j];                             // difference in the input's precision (:42)
            lprev[j] = ln;
            double p = pot[j] + (double)d;                             // :43
            if constexpr (RNG != kRngNone) {
                if (!ext) { p = p + base[j]; p = p + hot[j]; }         // :48-49
            }
            double vox = 0.0;
            const bool neg_side = p < 0.0;
            const double mag = fabs(p);
            const double thr = neg_side ? neg : pos;
            if (mag >= thr) {                                          // p >= C+  or  p <= -C-   (:51-55)
                double q = floor(mag * (neg_side ? inv_neg : inv_pos));
                if (__builtin_expect(q < 1099511627776.0, 1)) {
                    const double r = __builtin_fma(-q, thr, mag);      // sign-exact residual
                    if (r < 0.0) q -= 1.0; else if (r >= thr) q += 1.0;
                } else {
                    q = floor_divide_slow(mag, thr);
                }
                const double qt = q * thr;
                const double m2 = mag - qt;                            // :57-58 (product rounded, then subtracted)
                p = neg_side ? -m2 : m2;
                vox = neg_side ? -q : q;
                if (want_counts) { if (neg_side) n_off += (uint32_t)q; else n_on += (uint32_t)q; }
            }
            pot[j] = p;
            if constexpr (RNG != kRngNone) {
                if (ext) { vox = vox + base[j]; vox = vox + hot[j]; }  // :64-65
            }
            if constexpr (BIN == kBinBilinear) {
                const double cl = vox * wl, ch = vox * wh;             // bincount adds ps*w (no fma)
                acc_lo[j] = acc_lo[j] + cl;
                acc_hi[j] = acc_hi[j] + ch;
            } else {
                acc_lo[j] = acc_lo[j] + vox;
            }
        }
        if constexpr (BIN == kBinSum) {
            if (++sub == a.fpb) {                                      // wave-uniform
                store_vec<VEC>(a.out, a.out_f64, out_base + (int64_t)plane * a.HW, acc_lo);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc_lo[j] = 0.0;
                sub = 0;
                ++plane;
            }
        }
    };

    // ---- time loop: register ping-pong, kPrefetch frames per buffer
    Raw<IN, VEC> bufA[kPrefetch], bufB[kPrefetch];
    auto load_chunk = [&](Raw<IN, VEC> (&buf)[kPrefetch], int f0) {
#pragma unroll
        for (int u = 0; u < kPrefetch; ++u)
            if (f0 + u <= a.K) buf[u] = load_raw<IN, VEC>(a.frames, in_base + (int64_t)(f0 + u) * a.frame_stride);
    };
    auto run_chunk = [&](const Raw<IN, VEC> (&buf)[kPrefetch], int k0) {
#pragma unroll
        for (int u = 0; u < kPrefetch; ++u)
            if (k0 + u < a.K) step(k0 + u, buf[u]);
    };
    load_chunk(bufA, 1);
    for (int k0 = 0; k0 < a.K; k0 += 2 * kPrefetch) {
        load_chunk(bufB, k0 + kPrefetch + 1);
        run_chunk(bufA, k0);
        if (k0 + kPrefetch < a.K) {
            load_chunk(bufA, k0 + 2 * kPrefetch + 1);
            run_chunk(bufB, k0 + kPrefetch);
        }
    }

    // ---- epilogue
    if constexpr (BIN == kBinBilinear) {
        store_vec<VEC>(a.out, a.out_f64, out_base + (int64_t)cur_seg * a.HW, acc_lo);
        if (cur_seg + 1 < a.Tb) store_vec<VEC>(a.out, a.out_f64, out_base + (int64_t)(cur_seg + 1) * a.HW, acc_hi);
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc_lo[j] = 0.0;
        for (int b = cur_seg + 2; b < a.Tb; ++b) store_vec<VEC>(a.out, a.out_f64, out_base + (int64_t)b * a.HW, acc_lo);
    }
    if (want_counts) {
        // wave64 reduction, then one atomic per wave and polarity (inactive tail lanes returned early,
        // so reduce with the active mask semantics of __shfl_down: missing lanes contribute their own value -> use ballot-safe loop)
        unsigned long long on = n_on, off = n_off;
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const unsigned long long o1 = __shfl_down(on, s, 64);
            const unsigned long long o2 = __shfl_down(off, s, 64);
            const int src = (int)(threadIdx.x & 63) + s;
            const bool src_active = src < 64 && (uint32_t)((blk * kBlock + (threadIdx.x & ~63) + src)) * VEC < (uint32_t)a.HW;
            if (src_active) { on += o1; off += o2; }
        }
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&a.counts[2 * clip], on);
            atomicAdd(&a.counts[2 * clip + 1], off);
        }
    }
}

}  // namespace v2v
