// v2v_postops.hpp -- voxel post-ops of the consumer side (gfx950).
//
// Replaces normalize_batch_voxel (model/train_utils.py:147-166: per-sample 1 % / 99 % k-th values via torch.kthvalue,
// clamp(min=1), where(v > 0, v/pos_max, v/neg_max)) and the zero padding of H,W to multiples of 16 in
// forward_sequence (model/train_utils.py:322-326; model/train_flow_utils.py:343-347), fused into one writer.
//
// Exact k-th smallest per sample without sorting: 3-pass radix select over order-preserving 32-bit keys
// (11 + 11 + 10 bits).  Each pass histograms the digit of the elements that still match the selected prefix
// (LDS histogram per workgroup, one global atomic per non-empty bin), a tiny kernel walks the 2048 bins to pick the
// bucket holding rank k, and the final pass leaves the exact float.  Two ranks per sample (1 % and 99 %) share every
// pass.  Traffic: 3 reads for the select + 1 read + 1 (padded) write for the normalise; streaming, HBM-bound.
//
// Integer-valued voxels (the SUM-mode grids V2V trains on, data/v2v_datasets.py:399-400, without external noise) take the
// COUNTING path instead: one pass histograms the values (|v| <= 255) per sample, a tiny kernel reads the two k-th values off
// the cumulative counts, and the normalise pass follows -- 1 read + 1 read + 1 write instead of 3 + 1 + 1, still exact.
// The histogram is NOT accumulated inside the simulator's epilogue: with frames_per_bin = 1 (every training config) the
// simulator stores a plane every time step and is VALU-issue-bound, so counting there (convert, clamp, LDS atomic per voxel,
// +~20 VALU instructions per 4-pixel step on ~190) costs more than this memory-bound pass over the finished tensor
// (157 MB at the training shape = 0.03 ms).  Both input and output may carry a padded row pitch / plane size, so the simulator
// can write straight into the x16-padded buffer (v2v_esim_voxel_padded_hip) and the normalise runs in place.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2v {

constexpr int kSelBins = 2048;

struct SelectState {          // per (sample, which) -- which: 0 = low rank (1 %), 1 = high rank (99 %)
    uint32_t prefix;          // key bits decided so far (left-aligned)
    uint32_t prefix_mask;     // mask of decided bits
    uint64_t rank;            // remaining 0-based rank inside the selected bucket
};

__device__ __forceinline__ uint32_t float_key(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // total order: more negative -> smaller key
}
__device__ __forceinline__ float key_float(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// hist[(sample*2 + which)*2048 + digit] += 1 for elements matching the current prefix of `which`
__global__ void __launch_bounds__(256) select_hist_kernel(const float *x, int64_t per_sample, const SelectState *st,
                                                          unsigned int *hist, int shift, int bits)
{
    __shared__ unsigned int lh[2 * kSelBins];
    const int sample = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * kSelBins; i += 256) lh[i] = 0;
    __syncthreads();
    const SelectState s0 = st[sample * 2], s1 = st[sample * 2 + 1];
    const uint32_t dmask = (1u << bits) - 1u;
    const float *xs = x + (int64_t)sample * per_sample;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_sample; i += (int64_t)gridDim.x * 256) {
        const uint32_t k = float_key(xs[i]);
        const uint32_t d = (k >> shift) & dmask;
        if ((k & s0.prefix_mask) == s0.prefix) atomicAdd(&lh[d], 1u);
        if ((k & s1.prefix_mask) == s1.prefix) atomicAdd(&lh[kSelBins + d], 1u);
    }
    __syncthreads();
    unsigned int *gh = hist + (int64_t)sample * 2 * kSelBins;
    for (int i = threadIdx.x; i < 2 * kSelBins; i += 256)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// The same pass with 16-byte loads and the wave's LEADING digit peeled by ballot before the LDS adds: a voxel grid is mostly one
// value (zero), i.e. one digit per pass for most of a wave's 64 lanes -- 64 same-address LDS atomics serialise, one add of the
// population count by one lane does not; lanes with another digit add as before.  per_sample % 4 == 0, 16-byte aligned samples.
__global__ void __launch_bounds__(256) select_hist4_kernel(const float *x, int64_t per_sample, const SelectState *st,
                                                           unsigned int *hist, int shift, int bits)
{
    __shared__ unsigned int lh[2 * kSelBins];
    const int sample = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * kSelBins; i += 256) lh[i] = 0;
    __syncthreads();
    const SelectState s0 = st[sample * 2], s1 = st[sample * 2 + 1];
    const uint32_t dmask = (1u << bits) - 1u;
    const float4 *xs = reinterpret_cast<const float4 *>(x + (int64_t)sample * per_sample);
    const int64_t n4 = per_sample >> 2;
    const int lane = threadIdx.x & 63;
    auto add = [&](bool match, uint32_t d, unsigned int *h) __attribute__((always_inline)) {
        const unsigned long long m = __ballot(match);
        if (m == 0) return;                                                   // wave-uniform
        const int src = __builtin_ctzll(m);
        const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)d, src);
        const unsigned long long same = __ballot(match && d == lead);
        if (lane == src) atomicAdd(&h[lead], (unsigned int)__popcll(same));
        if (match && d != lead) atomicAdd(&h[d], 1u);
    };
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n4; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + threadIdx.x;
        const bool in = i < n4;
        const float4 v4 = in ? xs[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t k = float_key(vv[e]);
            const uint32_t d = (k >> shift) & dmask;
            add(in && (k & s0.prefix_mask) == s0.prefix, d, lh);
            add(in && (k & s1.prefix_mask) == s1.prefix, d, lh + kSelBins);
        }
    }
    __syncthreads();
    unsigned int *gh = hist + (int64_t)sample * 2 * kSelBins;
    for (int i = threadIdx.x; i < 2 * kSelBins; i += 256)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
}

// one workgroup per (sample, which): find the bin holding the remaining rank, extend the prefix, clear the histogram
__global__ void __launch_bounds__(256) select_pick_kernel(SelectState *st, unsigned int *hist, int shift, int bits)
{
    // a work-item sums its `per` bins, the workgroup scans the 256 sums (wave prefix by shuffles + the four wave totals), and the
    // ONE work-item whose range holds the rank walks its own bins (round 2: work-item 0 walked the 256 partial sums alone, 17 us)
    __shared__ uint64_t wtot[4];
    const int sw = blockIdx.x;                                  // sample*2 + which
    unsigned int *h = hist + (int64_t)sw * kSelBins;
    const int nb = 1 << bits, per = (nb + 255) / 256;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t local = 0;
    for (int j = 0; j < per; ++j) { const int b = threadIdx.x * per + j; if (b < nb) local += h[b]; }
    // the state is read by EVERY work-item before the barrier below and stored by the owner after it: no load can see the
    // owner's update (read after the barrier, a late wave could take the reduced rank for its own and extend the prefix twice)
    SelectState s = st[sw];
    uint64_t incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    for (int w = 0; w < wave; ++w) incl += wtot[w];
    const uint64_t excl = incl - local;
    // the owner: rank inside [excl, incl); a rank beyond the total (cannot happen: rank < matching elements) falls to the last work-item
    const bool own = s.rank >= excl && (s.rank < incl || threadIdx.x == 255);
    if (own) {
        uint64_t r = s.rank - excl;
        int b = threadIdx.x * per;
        while (b < nb - 1 && r >= h[b]) { r -= h[b]; ++b; }
        s.prefix |= (uint32_t)b << shift;
        s.prefix_mask |= ((1u << bits) - 1u) << shift;
        s.rank = r;
        st[sw] = s;
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kSelBins; b += 256) h[b] = 0;
}

__global__ void select_init_kernel(SelectState *st, int64_t n_sw, uint64_t rank_lo, uint64_t rank_hi)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_sw) { st[i].prefix = 0; st[i].prefix_mask = 0; st[i].rank = (i & 1) ? rank_hi : rank_lo; }
}

// out[b, p, 0:H, 0:W] = normalised voxel, zero elsewhere (padded to Hp x Wp)
// scales[sample] = {neg_max, pos_max} = {clamp(-kth(1 %), min=1), clamp(kth(99 %), min=1)} (model/train_utils.py:157-160); nullptr: no scaling
__global__ void state_scales_kernel(const SelectState *st, float *scales, int64_t n_sw)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sw) return;
    const float k = key_float(st[i].prefix), m = (i & 1) ? k : -k;
    scales[i] = m < 1.0f ? 1.0f : m;                                          // torch.clamp(..., min=1): a NaN stays a NaN
}

__global__ void __launch_bounds__(256) normalize_pad_kernel(const float *x, float *out, const float *scales,
                                                           int64_t planes, int H, int W, int Hp, int Wp, int Hin, int Win)
{
    const int sample = blockIdx.y;
    const bool normalize = scales != nullptr;
    float pos_max = 1.0f, neg_max = 1.0f;
    if (normalize) { neg_max = scales[sample * 2]; pos_max = scales[sample * 2 + 1]; }
    const int64_t per_out = planes * Hp * Wp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per_out; i += (int64_t)gridDim.x * 256) {
        const int xw = (int)(i % Wp);
        const int64_t r = i / Wp;
        const int yh = (int)(r % Hp);
        const int64_t p = r / Hp;
        float v = 0.0f;
        if (xw < W && yh < H) {
            v = x[((int64_t)sample * planes + p) * Hin * Win + (int64_t)yh * Win + xw];     // input planes may be padded too (Hin x Win >= H x W)
            if (normalize) v = v > 0.0f ? v / pos_max : v / neg_max;            // torch.where(voxel > 0, ...)
        }
        out[(int64_t)sample * per_out + i] = v;
    }
}

// The same pass for rows that are whole float4s on both sides (W, Wp, Win multiples of 4, 16-byte aligned tensors: every layout
// the simulator writes): a wave walks output rows, a lane moves 16 bytes.  No 64-bit division per element (the kernel above pays
// two), one 32-bit division per row; the float32 division of the reference's voxel / pos_max stays an IEEE division.
// Round 3: 143 -> 76 us average over the three shapes of tools/postops_time.py (5.9 TB/s).
__global__ void __launch_bounds__(256) normalize_pad_rows_kernel(const float *x, float *out, const float *scales,
                                                                int planes, int H, int W, int Hp, int Wp, int Hin, int Win)
{
    const int sample = blockIdx.y;
    const bool normalize = scales != nullptr;
    float pos_max = 1.0f, neg_max = 1.0f;
    if (normalize) { neg_max = scales[sample * 2]; pos_max = scales[sample * 2 + 1]; }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t rows = (uint32_t)planes * (uint32_t)Hp;                    // < 2^31: checked by the launcher
    const float *xs = x + (int64_t)sample * planes * Hin * Win;
    float *os = out + (int64_t)sample * rows * Wp;
    for (uint32_t r = blockIdx.x * 4u + (uint32_t)wave; r < rows; r += gridDim.x * 4u) {
        const uint32_t p = r / (uint32_t)Hp, yh = r - p * (uint32_t)Hp;
        const float *xrow = xs + ((int64_t)p * Hin + yh) * Win;
        float *orow = os + (int64_t)r * Wp;
        const bool live = yh < (uint32_t)H;
        for (int x4 = lane * 4; x4 < Wp; x4 += 256) {
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (live && x4 < W) {                                              // W % 4 == 0: a float4 is inside or outside
                v = *reinterpret_cast<const float4 *>(xrow + x4);
                if (normalize) {
                    v.x = v.x > 0.0f ? v.x / pos_max : v.x / neg_max;           // torch.where(voxel > 0, ...)
                    v.y = v.y > 0.0f ? v.y / pos_max : v.y / neg_max;
                    v.z = v.z > 0.0f ? v.z / pos_max : v.z / neg_max;
                    v.w = v.w > 0.0f ? v.w / pos_max : v.w / neg_max;
                }
            }
            *reinterpret_cast<float4 *>(orow + x4) = v;
        }
    }
}

// ---- counting path (integer-valued voxels) ----------------------------------------------------------------------------
// Bins for the integers -255..255 plus one OVERFLOW bin per side (everything below -255 / above 255): hot pixels (<= 0.1 % of the
// pixels, hot_pixel_std up to 10 in the training configuration = hundreds of events per frame) land there, far outside the 1 % the
// k-th values cut off on each side.  A rank that does fall into an overflow bin has no exact answer here: that sample's k-th value
// is NaN (loud), as for a sample holding a non-integer.
constexpr int kCntMax = 255, kCntZero = kCntMax + 1, kCntBins = 2 * kCntMax + 3;
__device__ __forceinline__ int cnt_bin(int iv) { return (iv < -kCntMax - 1 ? -kCntMax - 1 : iv > kCntMax + 1 ? kCntMax + 1 : iv) + kCntZero; }

// hist[sample][bin(v)] += 1 over the sample's (possibly padded) planes; bad[sample] != 0 if a value is not an integer.
// Zeros (the bulk of a voxel grid) are counted per wave with one ballot instead of 64 same-address LDS atomics.
__global__ void __launch_bounds__(256) count_hist_kernel(const float *x, int64_t per_sample, unsigned int *hist, unsigned int *bad)
{
    __shared__ unsigned int lh[kCntBins + 1];
    const int sample = blockIdx.y;
    for (int i = threadIdx.x; i <= kCntBins; i += 256) lh[i] = 0;
    __syncthreads();
    const float *xs = x + (int64_t)sample * per_sample;
    unsigned int zeros = 0;
    bool any_bad = false;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < per_sample; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + threadIdx.x;
        const bool in = i < per_sample;
        const float v = in ? xs[i] : 0.0f;
        const int iv = (int)v;
        const bool ok = (float)iv == v;
        any_bad |= in && !ok;
        const unsigned long long zmask = __ballot(in && v == 0.0f);
        if ((threadIdx.x & 63) == 0) zeros += (unsigned int)__popcll(zmask);
        if (in && ok && iv != 0) atomicAdd(&lh[cnt_bin(iv)], 1u);
    }
    if ((threadIdx.x & 63) == 0 && zeros) atomicAdd(&lh[kCntZero], zeros);
    if (any_bad) lh[kCntBins] = 1u;                             // benign race: every writer stores 1
    __syncthreads();
    unsigned int *gh = hist + (int64_t)sample * kCntBins;
    for (int i = threadIdx.x; i < kCntBins; i += 256)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
    if (threadIdx.x == 0 && lh[kCntBins]) atomicExch(&bad[sample], 1u);
}

// The same histogram with 16-byte loads; the three values that fill a voxel grid (0 and +-1) are counted per wave with one ballot each
// (64 lanes adding to the same LDS word serialise), everything else goes to the LDS word of its value.  Measured and not kept:
// peeling EVERY distinct value of the wave by ballot -- exact and atomic-free, but 13 distinct values (a uniform test pattern)
// cost 13 rounds per element: 236 us against 124 for the scalar kernel above.  per_sample % 4 == 0.
__global__ void __launch_bounds__(256) count_hist4_kernel(const float *x, int64_t per_sample, unsigned int *hist, unsigned int *bad)
{
    __shared__ unsigned int lh[kCntBins + 1];
    const int sample = blockIdx.y;
    for (int i = threadIdx.x; i <= kCntBins; i += 256) lh[i] = 0;
    __syncthreads();
    const float4 *xs = reinterpret_cast<const float4 *>(x + (int64_t)sample * per_sample);
    const int64_t n4 = per_sample >> 2;
    unsigned int n0 = 0, np = 0, nm = 0;                          // wave totals of 0 / +1 / -1, kept by every lane (wave-uniform adds)
    bool any_bad = false;
    for (int64_t i0 = (int64_t)blockIdx.x * 512; i0 < n4; i0 += (int64_t)gridDim.x * 512) {          // two 16-byte loads in flight per lane
        const int64_t ia = i0 + threadIdx.x, ib = ia + 256;
        const bool ina = ia < n4, inb = ib < n4;
        const float4 va = ina ? xs[ia] : make_float4(0.0f, 0.0f, 0.0f, 0.0f), vb = inb ? xs[ib] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float vv[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool in = e < 4 ? ina : inb;
            const float v = vv[e];
            const int iv = (int)v;
            const bool ok = (float)iv == v;
            any_bad |= in && !ok;
            n0 += (unsigned int)__popcll(__ballot(in && v == 0.0f));
            np += (unsigned int)__popcll(__ballot(in && v == 1.0f));
            nm += (unsigned int)__popcll(__ballot(in && v == -1.0f));
            if (in && ok && (iv > 1 || iv < -1)) atomicAdd(&lh[cnt_bin(iv)], 1u);
        }
    }
    if ((threadIdx.x & 63) == 0) {
        if (n0) atomicAdd(&lh[kCntZero], n0);
        if (np) atomicAdd(&lh[kCntZero + 1], np);
        if (nm) atomicAdd(&lh[kCntZero - 1], nm);
    }
    if (any_bad) lh[kCntBins] = 1u;                             // benign race: every writer stores 1
    __syncthreads();
    unsigned int *gh = hist + (int64_t)sample * kCntBins;
    for (int i = threadIdx.x; i < kCntBins; i += 256)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
    if (threadIdx.x == 0 && lh[kCntBins]) atomicExch(&bad[sample], 1u);
}

// One 64-lane workgroup per (sample, side): a lane sums 9 bins, the wave's inclusive prefix finds the lane that holds the rank, that
// lane walks its bins (round 2 walked the bins of every (sample, side) in ONE work-item: 511 dependent global loads, 44-210 us).
// The zero bin either holds the counted zeros incl. the n_pad padding zeros (count_hist*: removed here) or -- derive_zero, the
// histogram the SIMULATOR's writer accumulates (it counts non-zero values only) -- is whatever the other bins leave of n_elems.
constexpr int kCntPerLane = (kCntBins + 63) / 64;
__global__ void __launch_bounds__(64) count_pick_kernel(const unsigned int *hist, int64_t hist_stride, const unsigned int *bad, int64_t bad_stride,
                                                      int64_t n_sw, uint64_t n_pad, uint64_t rank_lo, uint64_t rank_hi, int derive_zero,
                                                      uint64_t n_elems, float *scales)
{
    const int64_t sw = blockIdx.x;
    if (sw >= n_sw) return;
    const int64_t sample = sw >> 1;
    const unsigned int *h = hist + sample * hist_stride;
    const uint64_t rank = (sw & 1) ? rank_hi : rank_lo;
    const int lane = threadIdx.x;
    uint64_t c[kCntPerLane], mine = 0, nonzero = 0;
#pragma unroll
    for (int j = 0; j < kCntPerLane; ++j) {
        const int b = lane * kCntPerLane + j;
        c[j] = b < kCntBins ? ((b == kCntZero) ? (derive_zero ? 0 : (uint64_t)h[b] - n_pad) : (uint64_t)h[b]) : 0;
        mine += c[j];
    }
    if (derive_zero) {                                          // wave total of the non-zero bins -> the zero bin's count
        nonzero = mine;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) nonzero += __shfl_xor(nonzero, d);
        if (lane == kCntZero / kCntPerLane) {
            const uint64_t z = n_elems > nonzero ? n_elems - nonzero : 0;
            c[kCntZero % kCntPerLane] = z;
            mine += z;
        }
    }
    uint64_t incl = mine;                                       // inclusive prefix sum over the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    // first lane whose inclusive count exceeds the rank; a rank beyond the total (cannot happen: rank < elements) ends in the last bin
    const unsigned long long over = __ballot(rank < incl);
    const int owner = over ? __builtin_ctzll(over) : 63;
    if (lane != owner) return;
    uint64_t r = rank - (incl - mine);
    int b = lane * kCntPerLane;
    for (int j = 0; j < kCntPerLane; ++j) {
        if (b >= kCntBins - 1 || r < c[j]) break;
        r -= c[j];
        ++b;
    }
    if (b > kCntBins - 1) b = kCntBins - 1;
    const bool exact = !bad[sample * bad_stride] && b > 0 && b < kCntBins - 1;  // an overflow bin has no single value
    const float kth = exact ? (float)(b - kCntZero) : __uint_as_float(0x7FC00000u);
    const float m = (sw & 1) ? kth : -kth;                      // (neg_max, pos_max) = clamp(-kth(1 %), min=1), clamp(kth(99 %), min=1)
    scales[sw] = m < 1.0f ? 1.0f : m;                           // a NaN stays a NaN (m < 1 is false)
}

}  // namespace v2v
