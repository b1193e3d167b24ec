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

// one workgroup per (sample, which): find the bin holding the remaining rank, extend the prefix, clear the histogram
__global__ void __launch_bounds__(256) select_pick_kernel(SelectState *st, unsigned int *hist, int shift, int bits)
{
    __shared__ unsigned int part[256];
    const int sw = blockIdx.x;                                  // sample*2 + which
    unsigned int *h = hist + (int64_t)sw * kSelBins;
    const int nb = 1 << bits, per = (nb + 255) / 256;
    unsigned int local = 0;
    for (int j = 0; j < per; ++j) { const int b = threadIdx.x * per + j; if (b < nb) local += h[b]; }
    part[threadIdx.x] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        SelectState s = st[sw];
        uint64_t r = s.rank;
        int t = 0;
        while (t < 255 && r >= part[t]) { r -= part[t]; ++t; }
        int b = t * per;
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
__global__ void __launch_bounds__(256) normalize_pad_kernel(const float *x, float *out, const SelectState *st, int normalize,
                                                           int64_t planes, int H, int W, int Hp, int Wp, int Hin, int Win)
{
    const int sample = blockIdx.y;
    float pos_max = 1.0f, neg_max = 1.0f;
    if (normalize) {
        const float hi = key_float(st[sample * 2 + 1].prefix), lo = -key_float(st[sample * 2].prefix);
        pos_max = hi < 1.0f ? 1.0f : hi;                                      // torch.clamp(kth(0.99), min=1): a NaN stays a NaN
        neg_max = lo < 1.0f ? 1.0f : lo;                                      // torch.clamp(-kth(0.01), min=1)
    }
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

// ---- counting path (integer-valued voxels, |v| <= kCntMax) ------------------------------------------------------------
constexpr int kCntMax = 255, kCntBins = 2 * kCntMax + 1;

// hist[sample][v + kCntMax] += 1 over the sample's (possibly padded) planes; bad[sample] != 0 if a value is not an integer
// in range.  Zeros (the bulk of a voxel grid) are counted per wave with one ballot instead of 64 same-address LDS atomics.
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
        const bool ok = (float)iv == v && iv >= -kCntMax && iv <= kCntMax;
        any_bad |= in && !ok;
        const unsigned long long zmask = __ballot(in && v == 0.0f);
        if ((threadIdx.x & 63) == 0) zeros += (unsigned int)__popcll(zmask);
        if (in && ok && iv != 0) atomicAdd(&lh[iv + kCntMax], 1u);
    }
    if ((threadIdx.x & 63) == 0 && zeros) atomicAdd(&lh[kCntMax], zeros);
    if (any_bad) lh[kCntBins] = 1u;                             // benign race: every writer stores 1
    __syncthreads();
    unsigned int *gh = hist + (int64_t)sample * kCntBins;
    for (int i = threadIdx.x; i < kCntBins; i += 256)
        if (lh[i]) atomicAdd(&gh[i], lh[i]);
    if (threadIdx.x == 0 && lh[kCntBins]) atomicExch(&bad[sample], 1u);
}

// one thread per (sample, which): walk the cumulative counts to the bin holding 0-based rank k; pad zeros (n_pad per sample,
// counted with the data) are removed from bin 0 first.  A sample with a bad value gets NaN k-th values (its output is NaN).
__global__ void count_pick_kernel(SelectState *st, const unsigned int *hist, const unsigned int *bad, int64_t n_sw, uint64_t n_pad,
                                  uint64_t rank_lo, uint64_t rank_hi)
{
    const int64_t sw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sw >= n_sw) return;
    const int64_t sample = sw >> 1;
    const unsigned int *h = hist + sample * kCntBins;
    uint64_t r = (sw & 1) ? rank_hi : rank_lo;
    int b = 0;
    for (; b < kCntBins - 1; ++b) {
        const uint64_t c = (b == kCntMax) ? (uint64_t)h[b] - n_pad : (uint64_t)h[b];
        if (r < c) break;
        r -= c;
    }
    SelectState s;
    s.prefix = bad[sample] ? float_key(__uint_as_float(0x7FC00000u)) : float_key((float)(b - kCntMax));
    s.prefix_mask = 0xFFFFFFFFu;
    s.rank = 0;
    st[sw] = s;
}

}  // namespace v2v
