// v2v_convlstm.hpp -- the recurrent UNet of the consumer side on the matrix cores (gfx950 MFMA).
//
// Kernels in this file:  convlstm_step_kernel (the ConvLSTM step, below; its EPI = 1 instances are the plain convolutions: residual
// blocks, stride-2 encoders, wide decoders; TPC = 2 = 32 input channels) | conv_halo_kernel (5x5 decoders with 32 / 64 columns) |
// conv_head_kernel (voxel bins -> 32 channels) | conv1x1_nhwc_kernel (prediction layer) | upsample2x_nhwc_bf16_kernel |
// layout / packing helpers (nchw_to_nhwc_bf16, to_nhwc8_bf16, *_pack_kernel).  Everything below this paragraph describes the step.
//
// Replaces one ConvLSTM.forward of the reference's recurrent encoders (model/submodules.py:179-235):
//     gates = Conv2d(2C -> 4C, 3x3, pad 1)(cat(x, h_prev));  i, r, o, g = gates.chunk(4, 1)
//     c = sigmoid(r) * c_prev + sigmoid(i) * tanh(g);  h = sigmoid(o) * tanh(c)
// as ONE kernel: the 3x3 convolution is an implicit GEMM on the bf16 matrix cores
//     M = B*H*W pixels,  N = 4C gate columns,  K = 9 taps * 2C channels          (fp32 accumulation)
// and the four gate activations + the cell/hidden update run on the accumulators in registers -- the [B,4C,H,W] gate tensor,
// the cat() copy and the five elementwise passes of the stock graph never touch HBM.
//
// Layouts.  Activations are NHWC (channel-minor) bf16 so that one pixel's 64 channels of one tap are one 128-byte line:
//   x, h_prev, h_state : [B,H,W,C] bf16        c_prev, c_state : [B,H,W,C] fp32        h_nchw (optional) : [B,C,H,W] fp32
// Weights are packed once per module (v2v_convlstm_pack_weights_hip) into the order the kernel streams them:
//   wp[col tile t = C/64][chunk ck = tap*(2C/64) + cc][column n = 0..255][k = 0..63] bf16,
//   column n = wn*128 + gate*32 + c32  <->  output channel gate*C + t*64 + wn*32 + c32,   k <-> input channel cc*64 + k
// i.e. every (tile, chunk) is one contiguous 32 KB block and a wave's four B fragments are the four gates of ITS 32 channels.
//
// Tiling.  A workgroup (WM x 2 waves) owns 32 MF WM consecutive pixels x 64 hidden channels (x 4 gates = 256 columns); wave (wm, wn)
// owns 32 MF pixels x 32 channels x 4 gates = MF x 4 accumulators of v_mfma_f32_32x32x16_bf16 (128 VGPRs).  K is walked in chunks
// of 64 (one tap, 64 channels): the A tile (pixels x 64 k) and the B tile (256 cols x 64 k, 32 KB) of chunk k+1 are
// brought into the other LDS buffer by global_load_lds_dwordx4 (no VGPR staging) while the MFMAs run on chunk k; out-of-image
// taps read a 128-byte zero line instead.  LDS rows are 128 B; the 16-byte slot index is XORed with (row >> 1) & 7 -- applied
// on the SOURCE address of the LDS-DMA (its LDS destination is lane-linear) and on the ds_read_b128 address -- which makes
// every 16-lane group of a fragment read hit 16 distinct (row parity, slot) pairs = all 64 banks once.
// The launcher picks the largest tile that still gives every CU a workgroup (256 -> 128 -> 64 pixels).  Where a wave's cycles
// go (cycle stamps of a -DV2V_CL_TIMING build: docs/experiments/convlstm_ablation_and_timing_switches.patch re-adds them; 256-pixel tile): 7-16 % in the vmcnt wait, 16-17 % in the
// barrier, the rest in LDS-DMA issue + fragment reads + MFMA -- the 8 LDS-DMA instructions a wave issues per chunk cost about
// as much issue time as half of its 32 MFMAs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "v2v_args.hpp"

namespace v2v {

constexpr int kClBK = 64, kClCh = 64, kClBN = 4 * kClCh;
constexpr int kClBBytes = kClBN * kClBK * 2;
// MF = 32-pixel accumulator blocks per wave (1 or 2), WM = wave rows (2 or 4; x 2 wave columns), STAGES = LDS buffers: the
// workgroup tile is 32*MF*WM pixels.  Shipped for the step: <1,2,2> 64 pixels (4 waves, 80 KB of LDS, two workgroups per CU; as two K
// groups when a CU gets one), <1,4,3> 128 pixels (8 waves, 144 KB, three stages with counted vmcnt), <1,8,2> 256 pixels (16 waves, 128 KB)
constexpr int cl_lds_bytes(int mf, int wm, int stages = 2, int bn = kClBN) { return stages * (32 * mf * wm * kClBK * 2 + bn * kClBK * 2); }

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 cl_bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float cl_f32x16;


__device__ __attribute__((aligned(128))) unsigned char g_cl_zero_line[128];      // zero-initialised: the padding source

// float -> bf16, round to nearest even, NaN stays NaN: gfx950's v_cvt_pk_bf16_f32 (two values per instruction)
typedef __bf16 cl_hwbf16x2 __attribute__((ext_vector_type(2)));
typedef float cl_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cl_pack_bf16(float lo, float hi)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(cl_f32x2{lo, hi}, cl_hwbf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) { return (uint16_t)cl_pack_bf16(f, f); }
// gate activations on the hardware transcendentals: ONE v_exp_f32 (2^x, the log2(e) folded into the multiply) and ONE v_rcp_f32 (1 ulp)
// per activation.  Round 2 wrote __frcp_rn(1 + __expf(-v)): the IEEE-rounded reciprocal expands to a 10-instruction division
// (v_div_scale x 2, v_rcp, 4 fma, v_div_fmas, v_div_fixup) and __expf adds a denormal-range rescue -- 80 divisions per wave made the
// gate epilogue ~9 us of a 49 us tile.  Saturation is exact: exp2(+inf) = inf -> rcp = 0, exp2(-inf) = 0 -> rcp(1) = 1.
__device__ __forceinline__ float cl_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
__device__ __forceinline__ float cl_tanh(float v) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -2.8853900817779268f)) - 1.0f; }

__device__ __forceinline__ void cl_glds16(const void *src, unsigned char *lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

// plain-convolution epilogue: accumulator g of the wave holds output channels col0 + 32 g .. + 31 (col0 = this lane's first)
template <int MF, int NF>
__device__ __forceinline__ void cl_epilogue_conv(const ConvLstmArgs &a, cl_f32x16 (&acc)[MF][NF], int64_t mw0, int col0, int fh, int64_t M)
{
    // one 64-bit element index per lane; every element adds a wave-uniform 32-bit offset (pixel row x column count) to it
    const uint32_t N = (uint32_t)a.n_cols;
    const int64_t lane0 = (mw0 + 4 * fh) * (int64_t)N + col0;
    uint16_t *const obase = a.out_nhwc + lane0;
    const uint16_t *const rbase = a.residual ? a.residual + lane0 : nullptr;
#pragma unroll
    for (int g = 0; g < NF; ++g) {
        const float bias = a.bias[col0 + g * 32];
#pragma unroll
        for (int i = 0; i < MF; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // the last tile of a launch may reach past the M = B*H*W pixels (M % 4 == 0, so 4 consecutive rows are in or out together)
                if (mw0 + 4 * fh + i * 32 + 8 * (r >> 2) >= M) continue;
                const uint32_t eo = (uint32_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * N + (uint32_t)(g * 32);
                float v = acc[i][g][r] + bias;
                if (rbase) v += __uint_as_float((uint32_t)rbase[eo] << 16);
                if (a.relu) v = v > 0.0f ? v : (v != v ? v : 0.0f);
                obase[eo] = f32_to_bf16_rne(v);
            }
        }
    }
}

// EPI = 0: ConvLSTM step (two inputs x|h, 4C gate columns, gate/cell epilogue).  EPI = 1: plain 3x3 convolution of x with
// n_cols output channels (a multiple of 256), epilogue bias (+ residual) (+ ReLU) -> bf16 NHWC: the residual blocks of the same
// encoder (model/submodules.py:143-177) on the same tiles and pipeline.
// WN x NF: wave columns x 32-column B fragments per wave = the tile's columns (2 x 4 = 256 for the gates; 1 x {4,2,1} = 128 / 64 /
// 32 output channels for the narrower plain convolutions).  EPI = 1 also takes the tap count (ks x ks, pad ks/2) and the stride
// from the arguments (5x5 and stride-2 encoder / decoder convolutions of model/unet.py).
// TPC = 2 (EPI = 1, Cin = 32: the first encoder of the UNet): TWO taps per 64-wide K chunk, k = 32 idx + c <-> tap 2 ck + idx,
// channel c -- a lane's 16-byte LDS-DMA source belongs to the tap its (swizzled) slot falls in; an odd tap count leaves the last
// half chunk to zero weights and the zero line.
// KS = 2 (EPI = 1, three stages): the workgroup is TWO wave groups that walk alternate K chunks of the SAME tile (each with its own
// LDS stages), i.e. twice the waves, LDS-DMA in flight and MFMA issue per CU for the small layers that cannot fill the chip with
// more tiles; group 1's accumulators meet group 0's through LDS before the epilogue.
template <int MF, int WM, int STAGES = 2, int EPI = 0, int WN = 2, int NF = 4, int TPC = 1, int KS = 1>
__global__ void __launch_bounds__(64 * WM * WN * KS, KS == 2 || WM * WN == 16 ? 1 : (STAGES == 2 && MF == 1) || WM * WN == 8 || NF < 4 ? 2 : 1) convlstm_step_kernel(const ConvLstmArgs a)
{
    static_assert(KS == 1 || KS == 2, "one or two K groups");
    static_assert(EPI == 1 || NF == 4, "the gate epilogue needs the four gates of a channel in one wave");
    static_assert(TPC == 1 || (TPC == 2 && EPI == 1), "two taps per chunk: plain convolution of 32 input channels");
    constexpr int kBN = WN * NF * 32, kBBytes = kBN * kClBK * 2;
    constexpr int kClBM = 32 * MF * WM, kClABytes = kClBM * kClBK * 2, kClStage = kClABytes + kBBytes;
    constexpr int NS = WM * WN;                                   // waves (all of them stage)
    extern __shared__ __attribute__((aligned(128))) unsigned char cl_lds_all[];
    // wave: inside its K group; kgrp wave-uniform in an SGPR (it feeds the chunk -> tap arithmetic), the constant 0 without the split
    const int lane = threadIdx.x & 63, wave = KS == 1 ? (int)(threadIdx.x >> 6) : (int)(threadIdx.x >> 6) % NS;
    const int kgrp = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) / NS);
    unsigned char *const cl_lds = cl_lds_all + kgrp * (STAGES * kClStage);
    const int wm = wave % WM, wn = wave / WM;
    const int C = a.C, HW = a.H * a.W;
    const int ks = EPI == 0 ? 3 : a.ks, pad = ks >> 1, n_taps = ks * ks, stride = EPI == 0 ? 1 : a.stride;
    const int Hin = EPI == 0 ? a.H : a.Hin, Win = EPI == 0 ? a.W : a.Win;
    constexpr int kStepCh = WN * 32;                              // hidden channels per column tile of the step (64, or 32 with one wave column)
    const int n_ct = EPI == 0 ? C / kStepCh : a.n_cols / kBN;     // column tiles
    const int ct = blockIdx.x % n_ct;
    const int64_t m0 = (int64_t)(blockIdx.x / n_ct) * kClBM;      // first pixel of the tile (flattened b,y,x)
    const int64_t M = (int64_t)a.B * HW;                          // pixels of the launch: the LAST tile may reach past them (any B*H*W with H*W % 4 == 0)
    const int cc_x = TPC == 2 ? 1 : C / kClBK, cc_all = EPI == 0 ? 2 * cc_x : cc_x;
    const int cc_eff = (EPI == 0 && a.h_prev) ? cc_all : cc_x;    // zero state: skip h's chunks
    const int n_chunks = TPC == 2 ? (n_taps + 1) / 2 : n_taps * cc_eff;

    // ---- staging plan: wave w issues A pieces NA*w.. (8 rows each) and B pieces NB*w.. per chunk ----------------------
    const int srow = lane >> 3, sslot = lane & 7;                 // this lane's (row in piece, LDS slot)
    constexpr int NA = kClBM / 8 / NS;                            // A pieces per wave
    static_assert(NA >= 1 && (kBN / 8) % NS == 0, "every wave stages whole pieces");
    int ay[NA], ax[NA];
    int64_t apix[NA];
    uint32_t aswz[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int row = (wave * NA + j) * 8 + srow;
        const int64_t m = m0 + row;
        const int p = (int)(m % HW), bimg = (int)(m / HW);
        ay[j] = m < M ? (p / a.W) * stride : -(1 << 28);          // the tap centre in INPUT coordinates; a row past the last pixel: every tap out of the image (zero line)
        ax[j] = (p % a.W) * stride;
        apix[j] = (((int64_t)bimg * Hin + ay[j]) * Win + ax[j]) * C;    // element offset of the centre pixel's channel vector
        aswz[j] = (uint32_t)((sslot ^ ((row >> 1) & 7)) * 8);     // source channel offset inside the 64-channel chunk
        if constexpr (TPC == 2) aswz[j] = (aswz[j] & 24u) | (aswz[j] >> 5 << 31);   // channel offset in the tap's 32 | which tap (bit 31)
    }
    constexpr int NB = kBN / 8 / NS;                              // B pieces per wave
    uint32_t boff[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int row = (wave * NB + j) * 8 + srow;
        boff[j] = (uint32_t)(row * kClBK + (sslot ^ ((row >> 1) & 7)) * 8);
    }
    // a column tile narrower than the packed one (EPI = 1, 128-column instances on 256-column packing): sub-tile `ct % per`
    const int pcols = EPI == 0 ? kClBN : a.pack_cols ? a.pack_cols : kBN, per = pcols / kBN;   // the step's weights are packed per 64 channels
    const uint16_t *wtile = a.wp + (int64_t)(ct / per) * (TPC == 2 ? n_chunks : n_taps * cc_all) * (pcols * kClBK) + (ct % per) * (kBN * kClBK);

    // LDS-DMA of chunk ck into buffer buf (part / nparts: a subset of the pieces, j % nparts == part).  The main loop issues
    // the whole chunk in front of the first k-step: spreading the pieces over the four k-steps was measured 10-15 % slower on
    // the same box with two buffers AND with three (every piece issued between MFMAs stalls the wave's instruction stream for
    // its turn in the address unit)
    auto stage = [&](int ck, int buf, int part, int nparts) __attribute__((always_inline)) {
        if constexpr (TPC == 2) {
            const int t0 = 2 * ck, t1 = 2 * ck + 1;
            const int dy0 = t0 / ks - pad, dx0 = t0 % ks - pad, dy1 = t1 / ks - pad, dx1 = t1 % ks - pad;
            const bool live1 = t1 < n_taps;
            const int sh0 = (dy0 * Win + dx0) * C, sh1 = (dy1 * Win + dx1) * C;          // tensors are below 2^31 elements
            unsigned char *abase = cl_lds + buf * kClStage, *bbase = abase + kClABytes;
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                if (j % nparts != part) continue;
                const bool second = (int32_t)aswz[j] < 0;
                const int dy = second ? dy1 : dy0, dx = second ? dx1 : dx0;
                const bool in = (!second || live1) && (unsigned)(ay[j] + dy) < (unsigned)Hin && (unsigned)(ax[j] + dx) < (unsigned)Win;
                const void *g = in ? (const void *)(a.x + apix[j] + (second ? sh1 : sh0) + (aswz[j] & 24u)) : (const void *)g_cl_zero_line;
                cl_glds16(g, abase + (wave * NA + j) * 1024);
            }
            const uint16_t *wchunk = wtile + (int64_t)ck * (pcols * kClBK);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j % nparts != part) continue;
                cl_glds16(wchunk + boff[j], bbase + (wave * NB + j) * 1024);
            }
            return;
        }
        const int tap = ck / cc_eff, cc = ck - tap * cc_eff;
        const int dy = tap / ks - pad, dx = tap % ks - pad;
        const uint16_t *src = cc < cc_x ? a.x : a.h_prev;
        const int c0 = (cc < cc_x ? cc : cc - cc_x) * kClBK;
        unsigned char *abase = cl_lds + buf * kClStage, *bbase = abase + kClABytes;
        const int64_t shift = ((int64_t)dy * Win + dx) * C + c0;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if (j % nparts != part) continue;
            const bool in = (unsigned)(ay[j] + dy) < (unsigned)Hin && (unsigned)(ax[j] + dx) < (unsigned)Win;
            const void *g = in ? (const void *)(src + apix[j] + shift + aswz[j]) : (const void *)g_cl_zero_line;
            cl_glds16(g, abase + (wave * NA + j) * 1024);
        }
        const uint16_t *wchunk = wtile + (int64_t)(tap * cc_all + cc) * (pcols * kClBK);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j % nparts != part) continue;
            cl_glds16(wchunk + boff[j], bbase + (wave * NB + j) * 1024);
        }
    };

    // ---- fragment read plan ------------------------------------------------------------------------------------------------
    const int fr = lane & 31, fh = lane >> 5;
    const uint32_t fsw = (uint32_t)((fr >> 1) & 7);
    uint32_t koff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) koff[s] = ((uint32_t)(2 * s + fh) ^ fsw) << 4;
    const uint32_t a_row = (uint32_t)((wm * 32 * MF + fr) * 128), b_row = (uint32_t)(kClABytes + (wn * 32 * NF + fr) * 128);

    cl_f32x16 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int g = 0; g < NF; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][g][r] = 0.0f;

    auto k_steps = [&](const unsigned char *base, auto &&before_step) __attribute__((always_inline)) {
        // fragments of k-step s+1 are read before the MFMAs of k-step s are issued (two register sets: +1..3 %)
        cl_bf16x8 af[2][MF], bf[2][NF];
        auto load = [&](int s, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < MF; ++i) af[slot][i] = *reinterpret_cast<const cl_bf16x8 *>(base + a_row + i * (32 * 128) + koff[s]);
#pragma unroll
            for (int g = 0; g < NF; ++g) bf[slot][g] = *reinterpret_cast<const cl_bf16x8 *>(base + b_row + g * (32 * 128) + koff[s]);
        };
        before_step(0);
        load(0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) { before_step(s + 1); load(s + 1, (s + 1) & 1); }
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int g = 0; g < NF; ++g) acc[i][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][i], bf[s & 1][g], acc[i][g], 0, 0, 0);
        }
    };
    // 64- and 128-pixel tiles (MF = 1): the previous cell state of this lane's outputs is fetched while the LAST chunk's MFMAs
    // run, so the epilogue does not start with a round trip to HBM (-1...-4 % on the same box; the 256-pixel tile has no
    // registers left for it: 104 spilled, +30 %)
    constexpr bool kCpre = EPI == 0 && MF == 1;
    float cpre[kCpre ? MF : 1][16];
    auto prefetch_c = [&]() __attribute__((always_inline)) {
        if constexpr (kCpre) {
            const int ch = ct * kStepCh + wn * 32 + fr;
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t mrow = m0 + wm * 32 * MF + i * 32 + (r >> 2) * 8 + fh * 4 + (r & 3);
                    cpre[i][r] = (a.c_prev && mrow < M) ? a.c_prev[mrow * C + ch] : 0.0f;
                }
        }
    };
    if constexpr (STAGES == 2) {
        // two LDS buffers: the whole next chunk is issued in front of the first k-step, drained (vmcnt 0) before the next barrier
        // (K group g walks chunks g, g + KS, ...: the same number of iterations and barriers for every wave)
        const int n_it = (n_chunks + KS - 1) / KS;
        if (kgrp < n_chunks) stage(kgrp, 0, 0, 1);
        for (int it = 0; it < n_it; ++it) {
            const int ck = it * KS + kgrp;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // chunk ck has landed; everyone is done with the other buffer
            const bool more = ck + KS < n_chunks;
            if (it + 1 == n_it) prefetch_c();
            if (ck < n_chunks)
                k_steps(cl_lds + (it & 1) * kClStage, [&](int s) __attribute__((always_inline)) {
                    if (more && s == 0) stage(ck + KS, (it + 1) & 1, 0, 1);
                });
        }
    } else {
        // three LDS buffers: chunk ck+2 is issued in front of the first k-step of chunk ck and stays in flight ACROSS the next
        // barrier -- the wait in front of a barrier is counted (the NA+NB pieces of the youngest chunk may be outstanding:
        // LDS-DMA completes in order) and the barrier is a raw s_barrier (a __syncthreads() would drain vmcnt)
        constexpr int kPieces = NA + NB;
        // K group g walks chunks g, g + KS, ...; every wave runs the same number of iterations (and barriers)
        const int n_it = (n_chunks + KS - 1) / KS;
        auto chunk_of = [&](int it) __attribute__((always_inline)) { return it * KS + kgrp; };
        if (chunk_of(0) < n_chunks) stage(chunk_of(0), 0, 0, 1);
        if (chunk_of(1) < n_chunks) stage(chunk_of(1), 1, 0, 1);
        int buf = 0, buf2 = 2;
        for (int it = 0; it < n_it; ++it) {
            if (chunk_of(it + 1) < n_chunks) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPieces) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                        // chunk `it` has landed for every wave; buffer buf2 is free
            asm volatile("" ::: "memory");
            const bool more2 = chunk_of(it + 2) < n_chunks;
            if (it + 1 == n_it) prefetch_c();
            if (chunk_of(it) < n_chunks)
                k_steps(cl_lds + buf * kClStage, [&](int s) __attribute__((always_inline)) {
                    if (more2 && s == 0) stage(chunk_of(it + 2), buf2, 0, 1);
                });
            buf = buf == 2 ? 0 : buf + 1;
            buf2 = buf2 == 2 ? 0 : buf2 + 1;
        }
    }
    if constexpr (KS == 2) {                                      // group 1's partial sums -> group 0 (through the stage buffers, now idle)
        __syncthreads();
        float *red = reinterpret_cast<float *>(cl_lds_all) + (size_t)wave * (MF * NF * 16 * 64) + lane;
        if (kgrp == 1) {
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int g = 0; g < NF; ++g)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((i * NF + g) * 16 + r) * 64] = acc[i][g][r];
        }
        __syncthreads();
        if (kgrp == 1) return;
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int g = 0; g < NF; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][g][r] += red[((i * NF + g) * 16 + r) * 64];
    }
    if constexpr (EPI == 1) {
        cl_epilogue_conv<MF, NF>(a, acc, m0 + wm * 32 * MF, ct * kBN + wn * 32 * NF + fr, fh, M);
        return;
    }
    if constexpr (EPI == 0) {
    // ---- epilogue: gates -> cell / hidden, straight from the accumulators ---------------------------------------------------
    // accumulator element r of lane l: column (channel) l & 31, row (pixel) (r & 3) + 8 (r >> 2) + 4 (l >> 5)
    const int ch = ct * kStepCh + wn * 32 + fr;
    const float b_i = a.bias[ch], b_r = a.bias[C + ch], b_o = a.bias[2 * C + ch], b_g = a.bias[3 * C + ch];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float hv[4];
            const int64_t mq = m0 + wm * 32 * MF + i * 32 + q * 8 + fh * 4;       // first of 4 consecutive pixels
            if (mq >= M) continue;                                               // past the last pixel (partial last tile)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = q * 4 + e;
                const int64_t idx = (mq + e) * C + ch;
                const float gi = cl_sigmoid(acc[i][0][r] + b_i), gr = cl_sigmoid(acc[i][1][r] + b_r);
                const float go = cl_sigmoid(acc[i][2][r] + b_o), gg = cl_tanh(acc[i][3][r] + b_g);
                float cp;
                if constexpr (kCpre) cp = cpre[i][r];
                else cp = a.c_prev ? a.c_prev[idx] : 0.0f;
                const float cn = gr * cp + gi * gg;
                const float hn = go * cl_tanh(cn);
                a.c_state[idx] = cn;
                a.h_state[idx] = f32_to_bf16_rne(hn);
                hv[e] = hn;
            }
            if (a.h_nchw) {
                const int64_t b = mq / HW, p = mq - b * HW;                  // HW % 4 == 0: the 4 pixels share an image
                const int64_t o = (b * C + ch) * HW + p;
                if (a.h_nchw_bf16) {
                    const uint32_t lo = cl_pack_bf16(hv[0], hv[1]), hi = cl_pack_bf16(hv[2], hv[3]);
                    *reinterpret_cast<uint2 *>(static_cast<uint16_t *>(a.h_nchw) + o) = make_uint2(lo, hi);
                } else {
                    *reinterpret_cast<float4 *>(static_cast<float *>(a.h_nchw) + o) = make_float4(hv[0], hv[1], hv[2], hv[3]);
                }
            }
        }
    }
    }
}

// fp32 or bf16 NCHW -> bf16 NHWC (optionally through a ReLU): the layout change between the stock convolution upstream and the
// fused step.  One workgroup moves 64 pixels x 64 channels through LDS so that both sides are full-line accesses.
__device__ __forceinline__ float cl_load_f32(const float *p, int64_t i) { return p[i]; }
__device__ __forceinline__ float cl_load_f32(const uint16_t *p, int64_t i) { return __uint_as_float((uint32_t)p[i] << 16); }

template <typename SRC>
__global__ void __launch_bounds__(256) nchw_to_nhwc_bf16_kernel(const SRC *src, uint16_t *dst, int C, int HW, int relu)
{
    __shared__ float t[64][65];
    const int n_ct = C / 64, n_pt = HW / 64;
    const int ctile = blockIdx.x % n_ct;
    const int ptile = (blockIdx.x / n_ct) % n_pt;
    const int b = blockIdx.x / (n_ct * n_pt);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = ty * 16 + k;
        float v = cl_load_f32(src, ((int64_t)b * C + ctile * 64 + c) * HW + ptile * 64 + tx);
        if (relu) v = v > 0.0f ? v : (v != v ? v : 0.0f);                       // torch.relu keeps NaN
        t[c][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int p = ty * 16 + k;
        dst[((int64_t)b * HW + ptile * 64 + p) * C + ctile * 64 + tx] = f32_to_bf16_rne(t[tx][p]);
    }
}

// out = bilinear_x2(x [+ skip]) on NHWC bf16 (f.interpolate(scale_factor=2, mode='bilinear', align_corners=False) of
// UpsampleConvLayer.forward, model/submodules.py:86-87, with the sum skip of model/unet.py:304 in front).  One work-item per INPUT
// pixel and 8 channels (16 bytes): it reads the clamped 3x3 neighbourhood once and writes the 2x2 output quad.  For scale 2 the
// source index (dst + 0.5) / 2 - 0.5 (clamped at 0) gives output 2k the neighbours (k-1, k) with weights (0.25, 0.75) and output
// 2k+1 the neighbours (k, k+1) with (0.75, 0.25), edge neighbours clamped -- torch's upsample_bilinear2d weights and its order
// (rows first, then columns: h0 (w0 v00 + w1 v01) + h1 (w0 v10 + w1 v11)) in float; x + skip rounds to bf16 first, as the
// stock bf16 add does.  HBM-bound: 2 bytes written per output element, a quarter (half with skip) of that read.
__device__ __forceinline__ void cl_unpack8(const uint4 v, float (&f)[8])
{
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
}

// One work-item walks `rs` consecutive input rows of one (image, column, 8-channel group): it keeps the horizontally filtered
// rows iy - 1, iy, iy + 1 (left / right output column) in registers and loads only row iy + 1 per step -- 3 (+3 skip) 16-byte
// loads per input pixel instead of 9 (+9) when every pixel gathers its own 3 x 3 neighbourhood.  Same expressions in the
// same order as before (horizontal 0.25 / 0.75 first, then vertical), so the output is bit-identical for any rs.
// (Round 5: with packed float32 instructions -- the compiler's SLP pass had paired these blends into v_pk_mul_f32 / v_pk_add_f32 -- this
// kernel returned wrong values in lanes 48-63 whenever a matrix-core kernel of another stream shared its CU; the whole library is now
// built without packed float32 instructions, see the Makefile.)
__global__ void __launch_bounds__(256) upsample2x_nhwc_bf16_kernel(const uint16_t *x, const uint16_t *skip, uint16_t *out, int B, int H, int W, int C, int rs)
{
    const uint32_t c8n = (uint32_t)C >> 3, segs = (uint32_t)(H + rs - 1) / (uint32_t)rs;
    const uint32_t n = (uint32_t)B * segs * (uint32_t)W * c8n;   // < 2^31: checked by the launcher (32-bit divisions below)
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    uint32_t r = i / c8n;
    const int c8 = (int)(i - r * c8n);
    const uint32_t r2 = r / (uint32_t)W;
    const int ix = (int)(r - r2 * (uint32_t)W);
    const int b = (int)(r2 / segs), seg = (int)(r2 - (uint32_t)b * segs);
    const int xs[3] = {ix > 0 ? ix - 1 : 0, ix, ix < W - 1 ? ix + 1 : ix};
    float L[3][8], R[3][8];                                       // rows iy - 1, iy, iy + 1: the left (2 ix) and right (2 ix + 1) output column
    auto hrow = [&](int y, float (&l)[8], float (&rr)[8]) __attribute__((always_inline)) {
        float v[3][8];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int64_t o = (((int64_t)b * H + y) * W + xs[k]) * C + c8 * 8;
            cl_unpack8(*reinterpret_cast<const uint4 *>(x + o), v[k]);
            if (skip) {
                float sk[8];
                cl_unpack8(*reinterpret_cast<const uint4 *>(skip + o), sk);
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const uint32_t pk = cl_pack_bf16(v[k][e] + sk[e], v[k][e + 1] + sk[e + 1]);
                    v[k][e] = __uint_as_float(pk << 16);
                    v[k][e + 1] = __uint_as_float(pk & 0xFFFF0000u);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float mid = 0.75f * v[1][e];
            l[e] = 0.25f * v[0][e] + mid;
            rr[e] = mid + 0.25f * v[2][e];
        }
    };
    auto put = [&](int oy, int ox, const float (&top)[8], float wt, const float (&bot)[8], float wb) __attribute__((always_inline)) {
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 8; e += 2) w[e >> 1] = cl_pack_bf16(wt * top[e] + wb * bot[e], wt * top[e + 1] + wb * bot[e + 1]);
        *reinterpret_cast<uint4 *>(out + ((((int64_t)b * 2 * H + oy) * 2 * W + ox) * C + c8 * 8)) = make_uint4(w[0], w[1], w[2], w[3]);
    };
    const int y0 = seg * rs, y1 = min(y0 + rs, H);
    hrow(y0 > 0 ? y0 - 1 : 0, L[0], R[0]);
    hrow(y0, L[1], R[1]);
    for (int iy = y0; iy < y1; ++iy) {
        if (iy < H - 1) hrow(iy + 1, L[2], R[2]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { L[2][e] = L[1][e]; R[2][e] = R[1][e]; }
        }
        put(2 * iy, 2 * ix, L[0], 0.25f, L[1], 0.75f);
        put(2 * iy, 2 * ix + 1, R[0], 0.25f, R[1], 0.75f);
        put(2 * iy + 1, 2 * ix, L[1], 0.75f, L[2], 0.25f);
        put(2 * iy + 1, 2 * ix + 1, R[1], 0.75f, R[2], 0.25f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { L[0][e] = L[1][e]; R[0][e] = R[1][e]; L[1][e] = L[2][e]; R[1][e] = R[2][e]; }
    }
}

// ---- stride-1 convolutions with few output channels (the decoders: 5x5, 32 / 64 / 128 columns): HALO tiles ------------------
// With one column tile of NF <= 4 fragments the kernel above is LDS-DMA-bound: every tap re-stages the 256 activations of the
// tile for 8 NF MFMAs per wave.  Here the workgroup owns a 16 x 16 pixel patch of one image and stages, once per 64-channel
// chunk, the patch WITH its halo ((16 + 2 pad)^2 pixels, out-of-image ones from the zero line) -- all ks^2 taps then read
// their A fragments from it at shifted rows.  Weights stream through LDS in groups of `tps` taps (two buffers); one barrier
// per group; one patch buffer, restaged between channel chunks, so that two workgroups fit a CU.  4 waves x 64 pixels (MF = 2: fragment row f = image rows 2f, 2f + 1 of the patch) x 32 NF columns.
// LDS rows are 128 B with the slot XOR (row >> 1) & 7 as above, row = pixel index inside the halo patch.
// Same packed weights as the kernel above ([tap * cc_all + cc][column][k]).
__host__ __device__ constexpr int conv_halo_pw(int ks) { return 16 + 2 * (ks >> 1); }
__host__ __device__ constexpr int conv_halo_pieces(int ks) { return (conv_halo_pw(ks) * conv_halo_pw(ks) + 7) / 8; }

template <int NF>
__global__ void __launch_bounds__(256, 2) conv_halo_kernel(const ConvLstmArgs a, int tps)
{
    constexpr int kBN = NF * 32;
    extern __shared__ __attribute__((aligned(128))) unsigned char cl_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;            // wave = wm: fragment rows 2 wm, 2 wm + 1
    const int C = a.C, H = a.H, W = a.W, ks = a.ks, pad = ks >> 1, n_taps = ks * ks;
    const int PW = 16 + 2 * pad, NP = PW * PW, n_pa = (NP + 7) >> 3;
    const int a_bytes = n_pa * 1024, b_bytes = tps * kBN * 128;
    unsigned char *const a_lds = cl_lds, *const b_lds = cl_lds + a_bytes;    // ONE patch buffer (two workgroups per CU), two weight-group buffers
    const int tiles_x = W >> 4, tiles_y = H >> 4;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, bimg = blockIdx.x / (tiles_x * tiles_y);
    const int cc_x = C / kClBK, n_groups = (n_taps + tps - 1) / tps, n_chunks = cc_x * n_groups;
    const int srow = lane >> 3, sslot = lane & 7;

    // ---- A staging plan: wave w stages patch pieces w, w + 4, ... (8 patch pixels each) ----------------------------------
    constexpr int kMaxPa = 13;                                            // ceil(50 / 4)
    uint32_t a_off[kMaxPa];
    uint32_t a_in = 0;
#pragma unroll
    for (int j = 0; j < kMaxPa; ++j) {
        const int pa = wave + 4 * j, hr = pa * 8 + srow;
        const int hy = hr / PW, hx = hr - hy * PW;
        const int iy = ty * 16 + hy - pad, ix = tx * 16 + hx - pad;
        const bool in = pa < n_pa && hr < NP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        a_off[j] = in ? (uint32_t)((((int64_t)bimg * H + iy) * W + ix) * C) + (uint32_t)((sslot ^ ((hr >> 1) & 7)) * 8) : 0u;
        a_in |= in ? (1u << j) : 0u;
    }
    auto stage_a = [&](int cc) __attribute__((always_inline)) {
        unsigned char *dst = a_lds;
        const uint16_t *src = a.x + cc * kClBK;
#pragma unroll
        for (int j = 0; j < kMaxPa; ++j) {
            const int pa = wave + 4 * j;
            if (pa < n_pa) {
                const void *g = (a_in >> j) & 1u ? (const void *)(src + a_off[j]) : (const void *)g_cl_zero_line;
                cl_glds16(g, dst + pa * 1024);
            }
        }
    };
    // weights of chunk (cc, tap group g) -> buffer `buf`: pieces of 8 columns, tap-major
    auto stage_b = [&](int ck, int buf) __attribute__((always_inline)) {
        const int cc = ck / n_groups, g = ck - cc * n_groups;
        const int tap0 = g * tps, nt = min(tps, n_taps - tap0), n_pb = nt * (kBN / 8);
        unsigned char *dst = b_lds + buf * b_bytes;
        for (int pb = wave; pb < n_pb; pb += 4) {
            const int tig = pb / (kBN / 8), brow0 = (pb - tig * (kBN / 8)) * 8, brow = brow0 + srow;
            const uint16_t *wsrc = a.wp + (int64_t)((tap0 + tig) * cc_x + cc) * (kBN * kClBK) + brow * kClBK + (sslot ^ ((brow >> 1) & 7)) * 8;
            cl_glds16(wsrc, dst + (tig * kBN + brow0) * 128);
        }
    };

    // ---- fragment read plan ----------------------------------------------------------------------------------------------
    const int fr = lane & 31, fh = lane >> 5;
    int hr_base[2];                                                       // patch pixel of this lane's A row for tap (-pad, -pad)
#pragma unroll
    for (int i = 0; i < 2; ++i) hr_base[i] = ((wave * 2 + i) * 2 + (fr >> 4)) * PW + (fr & 15);
    const uint32_t b_sw = (uint32_t)((fr >> 1) & 7);

    cl_f32x16 acc[2][NF];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < NF; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][g][r] = 0.0f;

    stage_a(0);
    stage_b(0, 0);
    for (int ck = 0; ck < n_chunks; ++ck) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                  // chunk ck has landed; everyone is done with the other buffers
        const int cc = ck / n_groups, g = ck - cc * n_groups;
        if (ck + 1 < n_chunks) stage_b(ck + 1, (ck + 1) & 1);
        const unsigned char *ab = a_lds, *bb = b_lds + (ck & 1) * b_bytes;
        const int tap0 = g * tps, nt = min(tps, n_taps - tap0);
        // 32 columns: fragments of tap t + 1 are read before the MFMAs of tap t are issued (two register sets)
        cl_bf16x8 af0[2][4], bf0[NF][4], af1[2][4], bf1[NF][4];
        auto load = [&](int tig, cl_bf16x8 (&af)[2][4], cl_bf16x8 (&bf)[NF][4]) __attribute__((always_inline)) {
            const int tap = tap0 + tig, dy = tap / ks, dx = tap - dy * ks;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hr = hr_base[i] + dy * PW + dx;
                const uint32_t sw = (uint32_t)((hr >> 1) & 7);
#pragma unroll
                for (int s = 0; s < 4; ++s) af[i][s] = *reinterpret_cast<const cl_bf16x8 *>(ab + hr * 128 + (((uint32_t)(2 * s + fh) ^ sw) << 4));
            }
#pragma unroll
            for (int q = 0; q < NF; ++q)
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    bf[q][s] = *reinterpret_cast<const cl_bf16x8 *>(bb + (tig * kBN + q * 32 + fr) * 128 + (((uint32_t)(2 * s + fh) ^ b_sw) << 4));
        };
        auto mma = [&](const cl_bf16x8 (&af)[2][4], const cl_bf16x8 (&bf)[NF][4]) __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < NF; ++q) acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][s], bf[q][s], acc[i][q], 0, 0, 0);
        };
        if constexpr (NF == 1) {                                          // wider column tiles run one tap per group: nothing to prefetch
            load(0, af0, bf0);
            int tig = 0;
            for (; tig + 1 < nt; tig += 2) {
                load(tig + 1, af1, bf1);
                mma(af0, bf0);
                if (tig + 2 < nt) load(tig + 2, af0, bf0);
                mma(af1, bf1);
            }
            if (tig < nt) mma(af0, bf0);
        } else {
            for (int tig = 0; tig < nt; ++tig) { load(tig, af0, bf0); mma(af0, bf0); }
        }
        if (g == n_groups - 1 && cc + 1 < cc_x) {                        // restage the patch between two channel chunks (the CU's other workgroup covers the wait)
            __syncthreads();
            stage_a(cc + 1);
        }
    }

    // ---- epilogue: bias (+ residual) (+ ReLU) -> bf16 NHWC ---------------------------------------------------------------------
    // element (i, r) of lane (fr, fh): patch row (wave * 2 + i) * 2 + (r >> 3), patch column (r & 3) + 8 ((r >> 2) & 1) + 4 fh.
    // One 64-bit element index per lane; every element adds a scalar (rows) and a compile-time (columns) 32-bit offset to it
    // (written as one 64-bit index expression per element this was ~23 vector instructions per element, a third of a tile's matrix time)
    constexpr int N = kBN;                                               // the halo tile IS the layer's column count
    const int64_t lane0 = ((((int64_t)bimg * H + ty * 16 + wave * 4) * W) + tx * 16 + 4 * fh) * N + fr;
    const uint32_t row_pitch = (uint32_t)W * (uint32_t)N;
    uint16_t *const obase = a.out_nhwc + lane0;
    const uint16_t *const rbase = a.residual ? a.residual + lane0 : nullptr;
#pragma unroll
    for (int q = 0; q < NF; ++q) {
        const float bias = a.bias[q * 32 + fr];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t eo = (uint32_t)(i * 2 + (r >> 3)) * row_pitch + (uint32_t)(((r & 3) + 8 * ((r >> 2) & 1)) * N + q * 32);
                float v = acc[i][q][r] + bias;
                if (rbase) v += __uint_as_float((uint32_t)rbase[eo] << 16);
                if (a.relu) v = v > 0.0f ? v : (v != v ? v : 0.0f);
                obase[eo] = f32_to_bf16_rne(v);
            }
        }
    }
}

// The prediction layer (ConvLayer(base, out, kernel_size 1, activation None), model/unet.py:58-64, applied to skip_sum(x, head) at
// :307): out[m][o] = bias[o] + sum_c bf16(w[o][c]) * bf16(x[m][c] + skip[m][c]) on NHWC bf16 -- HBM-bound (C x 2 (x 4 with the
// skip) bytes read per pixel, COUT x 2 written).  C / 8 lanes share a pixel (16 bytes each, power of two <= 64), partial sums
// meet by xor-shuffles; fp32 accumulation, one rounding of the result (bf16 or fp32 output).
template <int COUT>
__global__ void __launch_bounds__(256) conv1x1_nhwc_kernel(const uint16_t *x, const uint16_t *skip, const float *w, const float *bias,
                                                           void *out, int out_bf16, int64_t M, int C)
{
    const int lpp = C >> 3;                                       // lanes per pixel
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t m = i / lpp;
    const int g = (int)(i - m * lpp);
    float part[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) part[o] = 0.0f;
    if (m < M) {
        float v[8];
        cl_unpack8(*reinterpret_cast<const uint4 *>(x + m * C + g * 8), v);
        if (skip) {
            float s[8];
            cl_unpack8(*reinterpret_cast<const uint4 *>(skip + m * C + g * 8), s);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const uint32_t pk = cl_pack_bf16(v[e] + s[e], v[e + 1] + s[e + 1]);
                v[e] = __uint_as_float(pk << 16);
                v[e + 1] = __uint_as_float(pk & 0xFFFF0000u);
            }
        }
#pragma unroll
        for (int o = 0; o < COUT; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float wv = __uint_as_float((uint32_t)f32_to_bf16_rne(w[o * C + g * 8 + e]) << 16);
                part[o] = __builtin_fmaf(wv, v[e], part[o]);
            }
    }
    for (int d = 1; d < lpp; d <<= 1)
#pragma unroll
        for (int o = 0; o < COUT; ++o) part[o] += __shfl_xor(part[o], d);
    if (m < M && g == 0) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) {
            const float r = part[o] + bias[o];
            if (out_bf16) static_cast<uint16_t *>(out)[m * COUT + o] = f32_to_bf16_rne(r);
            else static_cast<float *>(out)[m * COUT + o] = r;
        }
    }
}

// ---- the head of the UNet: few input channels (the voxel bins, <= 8), 32 output channels, stride 1 (ConvLayer, model/unet.py:77) ----
// Input NHWC with the channels padded to 8 (one 16-byte slot per pixel).  K is packed TAP-major: k = 8 slot + c of chunk ck is
// channel c of tap 8 ck + slot (25 taps -> 4 chunks of 64, the 7 spare slots are zero weights and read a zero slot), so a lane's
// A fragment of one k-step is ONE 16-byte LDS read of the halo patch at its tap's pixel -- no im2col copy anywhere.  A 16 x 16
// pixel patch per workgroup (4 waves x 64 pixels), patch + halo (6.4 KB) and ALL weights (ceil(taps / 8) x 32 x 128 B) staged once:
// one barrier in the kernel.  Output bias (+ ReLU) -> bf16 NHWC [B,H,W,32].  HBM-bound on the output write.
__global__ void __launch_bounds__(256) conv_head_kernel(const uint16_t *x8, const uint16_t *wp, const float *bias, uint16_t *out, int B, int H, int W, int ks, int relu)
{
    extern __shared__ __attribute__((aligned(128))) unsigned char cl_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pad = ks >> 1, n_taps = ks * ks, n_ck = (n_taps + 7) >> 3;
    const int PW = 16 + 2 * pad, NP = PW * PW;
    unsigned char *const patch = cl_lds;                              // NP pixels x 16 B, then one zero slot
    unsigned char *const wl = cl_lds + ((NP * 16 + 16 + 127) & ~127);  // n_ck x 32 columns x 128 B, slot-swizzled rows
    const int tiles_x = W >> 4, tiles_y = H >> 4;
    const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, bimg = blockIdx.x / (tiles_x * tiles_y);
    for (int i = threadIdx.x; i <= NP; i += 256) {                    // patch (+ the zero slot at index NP)
        const int hy = i / PW, hx = i - hy * PW;
        const int iy = ty * 16 + hy - pad, ix = tx * 16 + hx - pad;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (i < NP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            v = *reinterpret_cast<const uint4 *>(x8 + (((int64_t)bimg * H + iy) * W + ix) * 8);
        *reinterpret_cast<uint4 *>(patch + i * 16) = v;
    }
    for (int i = threadIdx.x; i < n_ck * 32 * 8; i += 256) {          // weights: 16-byte slots, XOR swizzle on the slot as everywhere
        const int row = i >> 3, slot = i & 7, col = row & 31;
        *reinterpret_cast<uint4 *>(wl + row * 128 + ((slot ^ ((col >> 1) & 7)) << 4)) = *reinterpret_cast<const uint4 *>(wp + (int64_t)row * 64 + slot * 8);
    }
    __syncthreads();
    const int fr = lane & 31, fh = lane >> 5;
    int hr_base[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) hr_base[i] = ((wave * 2 + i) * 2 + (fr >> 4)) * PW + (fr & 15);
    const uint32_t b_sw = (uint32_t)((fr >> 1) & 7);
    cl_f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int ck = 0; ck < n_ck; ++ck) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int tap = ck * 8 + 2 * s + fh;                      // this lane's tap in k-step s
            const int dy = tap / ks, dx = tap - dy * ks;
            const bool live = tap < n_taps;
            const cl_bf16x8 bf = *reinterpret_cast<const cl_bf16x8 *>(wl + (ck * 32 + fr) * 128 + (((uint32_t)(2 * s + fh) ^ b_sw) << 4));
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int hr = live ? hr_base[i] + dy * PW + dx : NP;
                const cl_bf16x8 af = *reinterpret_cast<const cl_bf16x8 *>(patch + hr * 16);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[i], 0, 0, 0);
            }
        }
    }
    const float bv = bias[fr];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
            const int py = (wave * 2 + i) * 2 + (row >> 4), px = row & 15;
            float v = acc[i][r] + bv;
            if (relu) v = v > 0.0f ? v : (v != v ? v : 0.0f);
            out[((((int64_t)bimg * H + ty * 16 + py) * W) + tx * 16 + px) * 32 + fr] = f32_to_bf16_rne(v);
        }
    }
}

// nn.Conv2d weight fp32 [32, Cin <= 8, ks, ks] -> [chunk][32 columns][64 k], k = 8 slot + c <-> tap 8 chunk + slot, channel c (zero beyond)
__global__ void __launch_bounds__(256) conv_head_pack_kernel(const float *w, uint16_t *wp, int Cin, int ks)
{
    const int n_taps = ks * ks, n_ck = (n_taps + 7) >> 3;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_ck * 32 * 64) return;
    const int k = i & 63, col = (i >> 6) & 31, ck = i >> 11;
    const int tap = ck * 8 + (k >> 3), c = k & 7;
    wp[i] = (tap < n_taps && c < Cin) ? f32_to_bf16_rne(w[((int64_t)col * Cin + c) * n_taps + tap]) : (uint16_t)0;
}

// float32 [B, C <= 8, H, W] with arbitrary element strides -> bf16 [B, H, W, 8], channels C..7 zero (the head's input layout)
// scales (optional, [B,2] = {neg_max, pos_max} of normalize_batch_voxel, model/train_utils.py:157-166): the voxels are normalised WHILE
// they are read -- where(v > 0, v / pos_max, v / neg_max) in IEEE float32 division, then rounded to bf16 -- so a normalised copy of the
// event tensor never exists
__global__ void __launch_bounds__(256) to_nhwc8_bf16_kernel(const float *src, int64_t sb, int64_t sc, int64_t sh, int64_t sw, uint16_t *dst, int B, int C, int H, int W,
                                                            const float *scales)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = c < C ? src[b * sb + c * sc + y * sh + x * sw] : 0.0f;
    if (scales) {
        const float neg_max = scales[2 * b], pos_max = scales[2 * b + 1];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = v[c] > 0.0f ? v[c] / pos_max : v[c] / neg_max;
    }
    *reinterpret_cast<uint4 *>(dst + i * 8) = make_uint4(cl_pack_bf16(v[0], v[1]), cl_pack_bf16(v[2], v[3]), cl_pack_bf16(v[4], v[5]), cl_pack_bf16(v[6], v[7]));
}

// Cin = 32 (two taps per chunk): wp[col tile][chunk ck = 0 .. ceil(taps / 2) - 1][column][k], k = 32 idx + c <-> tap 2 ck + idx, channel c
__global__ void __launch_bounds__(256) conv_pack32_kernel(const float *w, uint16_t *wp, int Cout, int ks, int bn)
{
    const int taps = ks * ks, n_ck = (taps + 1) / 2;
    const int64_t n = (int64_t)Cout * n_ck * kClBK;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t r = i;
    const int k = (int)(r % kClBK); r /= kClBK;
    const int col = (int)(r % bn); r /= bn;
    const int ck = (int)(r % n_ck); r /= n_ck;
    const int ct = (int)r, tap = 2 * ck + (k >> 5), c = k & 31;
    wp[i] = tap < taps ? f32_to_bf16_rne(w[((int64_t)(ct * bn + col) * 32 + c) * taps + tap]) : (uint16_t)0;
}

// [4C, 2C, 3, 3] fp32 (nn.Conv2d weight of ConvLSTM.Gates) -> packed bf16 (layout at the top of this file); one thread per element
__global__ void __launch_bounds__(256) convlstm_pack_kernel(const float *w, uint16_t *wp, int C)
{
    const int64_t n = (int64_t)4 * C * 2 * C * 9;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int cc_all = 2 * C / kClBK;
    int64_t r = i;
    const int k = (int)(r % kClBK); r /= kClBK;
    const int col = (int)(r % kClBN); r /= kClBN;
    const int ck = (int)(r % (9 * cc_all)); r /= 9 * cc_all;
    const int ct = (int)r;
    const int tap = ck / cc_all, cc = ck % cc_all;
    const int wn = col >> 7, gate = (col >> 5) & 3, c32 = col & 31;
    const int oc = gate * C + ct * kClCh + wn * 32 + c32, ic = cc * kClBK + k;
    wp[i] = f32_to_bf16_rne(w[((int64_t)oc * 2 * C + ic) * 9 + tap]);
}

// [Cout, Cin, ks, ks] fp32 (nn.Conv2d weight) -> packed bf16 for the EPI = 1 instances, bn = the instance's tile columns:
//   wp[col tile t = Cout/bn][chunk ck = tap*(Cin/64) + cc][column n = 0..bn-1][k = 0..63],  column n <-> output channel t*bn + n
__global__ void __launch_bounds__(256) conv_pack_kernel(const float *w, uint16_t *wp, int Cin, int Cout, int ks, int bn)
{
    const int taps = ks * ks;
    const int64_t n = (int64_t)Cout * Cin * taps;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int cc_all = Cin / kClBK;
    int64_t r = i;
    const int k = (int)(r % kClBK); r /= kClBK;
    const int col = (int)(r % bn); r /= bn;
    const int ck = (int)(r % (taps * cc_all)); r /= taps * cc_all;
    const int ct = (int)r;
    const int tap = ck / cc_all, cc = ck % cc_all;
    wp[i] = f32_to_bf16_rne(w[((int64_t)(ct * bn + col) * Cin + cc * kClBK + k) * taps + tap]);
}

}  // namespace v2v
