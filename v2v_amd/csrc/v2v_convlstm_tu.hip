// v2v_convlstm_tu.hip -- translation unit of the fused ConvLSTM step: launchers.
#include <atomic>

#include "v2v_convlstm.hpp"
#include "v2v_args.hpp"

namespace v2v {

namespace {
template <int MF, int WM, int STAGES = 2, int EPI = 0, int WN = 2, int NF = 4, int TPC = 1, int KS = 1>
hipError_t launch_step_t(const ConvLstmArgs &a, hipStream_t s)
{
    // 80-128 KB of dynamic LDS is above the 64 KB a kernel gets by default: raise the limit once per device (kept out of the
    // launch path so that a step captures into a hipGraph as a bare kernel node)
    constexpr int lds = KS * cl_lds_bytes(MF, WM, STAGES, WN * NF * 32);
    static std::atomic<bool> raised[64];              // idempotent attribute call: a benign repeat, but no data race
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    if (dev < 0 || dev >= 64 || !raised[dev].load(std::memory_order_acquire)) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&convlstm_step_kernel<MF, WM, STAGES, EPI, WN, NF, TPC, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) raised[dev].store(true, std::memory_order_release);
    }
    // ceil: the last pixel tile may be partial (rows past B*H*W read the zero line and are not stored)
    const int64_t tiles = (((int64_t)a.B * a.H * a.W + 32 * MF * WM - 1) / (32 * MF * WM)) * (EPI == 0 ? a.C / (WN * 32) : a.n_cols / (WN * NF * 32));
    hipLaunchKernelGGL((convlstm_step_kernel<MF, WM, STAGES, EPI, WN, NF, TPC, KS>), dim3((unsigned)tiles), dim3(64 * WM * WN * KS), lds, s, a);
    return hipGetLastError();
}
}  // namespace

// tile_rows: pixels per workgroup tile (64, 128 or 256); 0 = auto: the largest tile that still gives every CU a workgroup.
// Same box, 8 clips (ms per step; tools/ab_convlstm.sh):        64 px    128 px   256 px
//   64 ch @128^2 / 128 ch @64^2 / 256 ch @32^2 (256^2 input)     0.119/0.100/0.094   0.114/0.096/0.089   0.101/0.083/0.126
//   64 ch @64^2 / 128 ch @32^2 / 256 ch @16^2 (128^2 input)      0.034/0.042/0.071   0.033/0.048/0.080   0.049/0.074/0.127
hipError_t launch_convlstm_step(const ConvLstmArgs &a, int tile_rows, hipStream_t s)
{
    if (tile_rows == 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int64_t m = (int64_t)a.B * a.H * a.W, ct = a.C / kClCh;
        tile_rows = (m / 256 * ct >= cus) ? 256 : (m / 128 * ct >= cus) ? 128 : 64;      // (a partial last tile is fine: any B*H*W)
        // when even 64-pixel tiles give a CU at most one workgroup: that workgroup as two K groups (KS = 2; same box, 8 clips:
        // 128 ch @32^2 0.039 -> 0.033 ms, 256 ch @16^2 0.067 -> 0.053 ms; slower where more tiles exist: 0.115 -> 0.144 ms)
        if (tile_rows == 64 && m / 64 * ct <= cus) tile_rows = 65;
    }
    // (round 4, measured and not kept: tiles of ONE wave column -- 32 hidden channels x 4 gates -- so that two workgroups share a CU and
    // one's epilogue runs under the other's main loop: launch_step_t<1, 4, 2, 0, 1, 4>, 128 px x 32 ch, 64 KB: 0.107 / 0.101 / 0.098 ms
    // against 0.088 / 0.078 / 0.095 for the tiles below at 64@128^2 / 128@64^2 / 256@32^2, same box -- twice the LDS-DMA pieces per MFMA
    // cost more than the overlap buys; the kernel template still takes WN = 1 for the step)
    if (tile_rows == 65) return launch_step_t<1, 2, 2, 0, 2, 4, 1, 2>(a, s);   // 64 px as two K groups of 4 waves (160 KB, internal code)
    // 256 px as 16 waves of 32 px x 128 columns (4 per SIMD, 114 VGPRs) instead of 8 of 64 x 128: bit-identical, -2...-4 % same box
    if (tile_rows == 256) return launch_step_t<1, 8, 2>(a, s);
    return tile_rows == 128 ? launch_step_t<1, 4, 3>(a, s) : launch_step_t<1, 2, 2>(a, s);
}

// plain convolution (EPI = 1): Cout % 256 == 0 takes the 256-column tiles of the gate kernel (three pixel-tile sizes), Cout =
// 128 / 64 / 32 a 256-pixel tile of 4 waves with 4 / 2 / 1 column fragments per wave
int conv_tile_cols(int Cout) { return Cout % 256 == 0 ? 256 : (Cout == 128 || Cout == 64 || Cout == 32) ? Cout : 0; }

hipError_t launch_conv_nhwc(const ConvLstmArgs &a, int tile_rows, hipStream_t s)
{
    const int64_t m = (int64_t)a.B * a.H * a.W;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // halo tiles (conv_halo_kernel): the 16 x 16 patch with its halo staged once per 64-channel chunk.  One patch buffer + two
    // weight-group buffers must leave room for TWO workgroups per CU (<= 80 KB): 8 clips of 256^2, same box, 5x5: 64 -> 32
    // columns 0.117 -> 0.085 ms (3 taps per group), 128 -> 64 columns 0.079 -> 0.071 ms (1 tap); 256 -> 128 columns does not fit
    // twice and measured 0.126-0.137 against 0.090 ms on the 128-pixel tile; with one workgroup per CU (two patch buffers or
    // 5-tap groups) the same kernel is slower than the tiles above; an 8-row patch (30 KB, so that 128 columns fit twice) measured
    // 0.109 against 0.090 ms as well.  tile_rows 16 forces it (tests), 0 takes it where measured.
    const int halo_a = conv_halo_pieces(a.ks) * 1024;
    const int halo_tps = a.n_cols % 256 != 0 ? (80 * 1024 - halo_a) / (2 * a.n_cols * 128) : 0;
    const bool halo_ok = a.C % 64 == 0 && a.n_cols % 256 != 0 && a.stride == 1 && a.H % 16 == 0 && a.W % 16 == 0 && halo_tps >= 1;
    if (halo_ok && (tile_rows == 16 || (tile_rows == 0 && a.ks == 5 && a.n_cols <= 64))) {
        const int nf = a.n_cols / 32;
        const int tps = halo_tps < a.ks ? halo_tps : a.ks;
        const int lds = halo_a + 2 * tps * a.n_cols * 128;
        const void *fn = nf == 4 ? (const void *)&conv_halo_kernel<4> : nf == 2 ? (const void *)&conv_halo_kernel<2> : (const void *)&conv_halo_kernel<1>;
        static std::atomic<bool> raised[3][64];                              // once per instance and device (not in a captured launch path)
        const int inst = nf == 4 ? 2 : nf - 1;
        if (dev < 0 || dev >= 64 || !raised[inst][dev].load(std::memory_order_acquire)) {
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) raised[inst][dev].store(true, std::memory_order_release);
        }
        const unsigned tiles = (unsigned)(a.B * (a.H / 16) * (a.W / 16));
        if (nf == 4) hipLaunchKernelGGL(conv_halo_kernel<4>, dim3(tiles), dim3(256), lds, s, a, tps);
        else if (nf == 2) hipLaunchKernelGGL(conv_halo_kernel<2>, dim3(tiles), dim3(256), lds, s, a, tps);
        else hipLaunchKernelGGL(conv_halo_kernel<1>, dim3(tiles), dim3(256), lds, s, a, tps);
        return hipGetLastError();
    }
    if (a.C == 32) {                                                 // two taps per chunk (the UNet's first encoder: 32 -> 64, 5x5, stride 2)
        if (tile_rows == 0) tile_rows = (m / 256 >= cus) ? 256 : 128;
        if (a.n_cols == 64) return tile_rows == 256 ? launch_step_t<2, 4, 2, 1, 1, 2, 2>(a, s) : tile_rows == 128 ? launch_step_t<1, 4, 3, 1, 1, 2, 2>(a, s) : hipErrorInvalidValue;
        if (a.n_cols == 128) return tile_rows == 256 ? launch_step_t<2, 4, 2, 1, 1, 4, 2>(a, s) : tile_rows == 128 ? launch_step_t<1, 4, 3, 1, 1, 4, 2>(a, s) : hipErrorInvalidValue;
        return hipErrorInvalidValue;
    }
    if (a.n_cols % 256 != 0) {
        // one column tile of 4 / 2 / 1 B fragments per wave: 256 pixels (8 fragment rows of 4 waves x MF 2), or 128 pixels on
        // three stages when 256-pixel tiles leave CUs idle (8 clips at 64^2, same box: 5x5 256 -> 128 121 -> 88 us, 5x5
        // stride-2 64 -> 128 41 -> 30 us).  Measured and not kept for 32 columns: 512- and 256-pixel tiles of MF 4 (+26 / +41 %)
        if (tile_rows == 0) tile_rows = (m / 256 >= cus) ? 256 : 128;
        if (tile_rows == 256) {
            if (a.n_cols == 128) return launch_step_t<2, 4, 2, 1, 1, 4>(a, s);
            if (a.n_cols == 64) return launch_step_t<2, 4, 2, 1, 1, 2>(a, s);
            if (a.n_cols == 32) return launch_step_t<2, 4, 2, 1, 1, 1>(a, s);
        } else if (tile_rows == 128) {
            // 128 columns: two K groups of 4 waves on two stages each (128 KB) instead of 4 waves on three: same box, 8 clips at
            // 64^2, 5x5 256 -> 128 87.7 -> 72.4 us, stride-2 64 -> 128 29.4 -> 26.1 us
            if (a.n_cols == 128) return launch_step_t<1, 4, 2, 1, 1, 4, 1, 2>(a, s);
            if (a.n_cols == 64) return launch_step_t<1, 4, 3, 1, 1, 2>(a, s);
            if (a.n_cols == 32) return launch_step_t<1, 4, 3, 1, 1, 1>(a, s);
        }
        return hipErrorInvalidValue;
    }
    if (tile_rows == 0) {
        const int64_t ct = a.n_cols / kClBN;
        // 32-pixel tiles when even 64-pixel ones leave CUs idle (8 clips at 32^2, same box: residual-block convolution 36.7 ->
        // 30.0 us, 5x5 stride-2 128 -> 256 48.0 -> 39.1 us; a third stage on the 64-pixel tile measured no gain)
        // ... and better still 64 px x 128 columns, i.e. half a packed column tile per workgroup (4 waves of 32 x 64, three
        // stages): a third fewer LDS-DMA pieces per MFMA than 32 px x 256 columns (30.1 -> 27.4 us, 39.2 -> 36.0 us), and with the
        // workgroup as two K groups of 4 waves (KS = 2: 8 waves, 144 KB) 31.0 -> 24.3 us and 39.7 -> 30.2 us on the same box
        tile_rows = (m / 256 * ct >= cus) ? 256 : (m / 128 * ct >= cus) ? 128 : (m / 64 * ct < cus) ? 66 : 64;
    }
    if (tile_rows == 256) return launch_step_t<2, 4, 2, 1>(a, s);
    if (tile_rows == 32) return launch_step_t<1, 1, 3, 1, 4, 2>(a, s); // 4 waves of 32 px x 64 columns, three stages
    if (tile_rows == 66) {                                             // 64 px x HALF a packed column tile (internal code, see above)
        ConvLstmArgs b = a;
        b.pack_cols = kClBN;
        return launch_step_t<1, 2, 3, 1, 2, 2, 1, 2>(b, s);            // ... as two K groups of 4 waves (see the kernel's KS)
    }
    return tile_rows == 128 ? launch_step_t<1, 4, 3, 1>(a, s) : launch_step_t<1, 2, 2, 1>(a, s);
}

hipError_t launch_conv_pack(const float *w, uint16_t *wp, int Cin, int Cout, int ks, hipStream_t s)
{
    if (Cin == 32) {
        const int64_t n32 = (int64_t)Cout * ((ks * ks + 1) / 2) * kClBK;
        hipLaunchKernelGGL(conv_pack32_kernel, dim3((unsigned)((n32 + 255) / 256)), dim3(256), 0, s, w, wp, Cout, ks, conv_tile_cols(Cout));
        return hipGetLastError();
    }
    const int64_t n = (int64_t)Cout * Cin * ks * ks;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, wp, Cin, Cout, ks, conv_tile_cols(Cout));
    return hipGetLastError();
}

hipError_t launch_convlstm_pack(const float *w, uint16_t *wp, int C, hipStream_t s)
{
    const int64_t n = (int64_t)4 * C * 2 * C * 9;
    hipLaunchKernelGGL(convlstm_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, wp, C);
    return hipGetLastError();
}

hipError_t launch_nchw_to_nhwc_bf16(const void *src, bool src_bf16, uint16_t *dst, int B, int C, int HW, int relu, hipStream_t s)
{
    const int64_t blocks = (int64_t)B * (C / 64) * (HW / 64);
    if (src_bf16)
        hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint16_t *>(src), dst, C, HW, relu);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float *>(src), dst, C, HW, relu);
    return hipGetLastError();
}

hipError_t launch_conv_head(const uint16_t *x8, const uint16_t *wp, const float *bias, uint16_t *out, int B, int H, int W, int ks, int relu, hipStream_t s)
{
    const int pw = 16 + 2 * (ks >> 1), lds = ((pw * pw * 16 + 16 + 127) & ~127) + ((ks * ks + 7) / 8) * 32 * 128;
    hipLaunchKernelGGL(conv_head_kernel, dim3((unsigned)(B * (H / 16) * (W / 16))), dim3(256), lds, s, x8, wp, bias, out, B, H, W, ks, relu);
    return hipGetLastError();
}

hipError_t launch_conv_head_pack(const float *w, uint16_t *wp, int Cin, int ks, hipStream_t s)
{
    const int n = ((ks * ks + 7) / 8) * 32 * 64;
    hipLaunchKernelGGL(conv_head_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, wp, Cin, ks);
    return hipGetLastError();
}

hipError_t launch_to_nhwc8_bf16(const float *src, int64_t sb, int64_t sc, int64_t sh, int64_t sw, uint16_t *dst, int B, int C, int H, int W, const float *scales, hipStream_t s)
{
    const int64_t n = (int64_t)B * H * W;
    hipLaunchKernelGGL(to_nhwc8_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, sb, sc, sh, sw, dst, B, C, H, W, scales);
    return hipGetLastError();
}

hipError_t launch_conv1x1_nhwc(const uint16_t *x, const uint16_t *skip, const float *w, const float *bias, void *out, int out_bf16, int64_t M,
                               int C, int Cout, hipStream_t s)
{
    const int64_t n = M * (C / 8);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    switch (Cout) {
    case 1: hipLaunchKernelGGL(conv1x1_nhwc_kernel<1>, grid, block, 0, s, x, skip, w, bias, out, out_bf16, M, C); break;
    case 2: hipLaunchKernelGGL(conv1x1_nhwc_kernel<2>, grid, block, 0, s, x, skip, w, bias, out, out_bf16, M, C); break;
    case 3: hipLaunchKernelGGL(conv1x1_nhwc_kernel<3>, grid, block, 0, s, x, skip, w, bias, out, out_bf16, M, C); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_upsample2x_nhwc(const uint16_t *x, const uint16_t *skip, uint16_t *out, int B, int H, int W, int C, hipStream_t s)
{
    // one work-item per (image, segment of rs rows, column, 8 channels).  8 clips, same box, us per launch (events) for
    // rs = 1 / 2 / 4 / 8: 256 ch @32^2 17.1 / 18.0 / 17.9 / 23.8, 128 ch @64^2 23.0 / 20.7 / 20.0 / 22.6, 64 ch @128^2
    // 38.1 / 30.0 / 25.6 / 30.8 -> 4 rows where that leaves >= 2048 waves, else 1 (tools/upsample_time.py)
    const int64_t cols = (int64_t)B * W * (C / 8);
    int rs = 4;
#ifdef V2V_TUNING_KNOBS                                                           // tuning builds only (tools/upsample_time.py): never in the product launch path
    if (const char *e = getenv("V2V_UP_RS")) rs = atoi(e) > 0 ? atoi(e) : rs;
    else
#endif
    if (cols * ((H + rs - 1) / rs) < 2048 * 64) rs = 1;
    const int64_t n = cols * ((H + rs - 1) / rs);
    if (n >= (int64_t)1 << 31) return hipErrorInvalidValue;
    hipLaunchKernelGGL(upsample2x_nhwc_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, skip, out, B, H, W, C, rs);
    return hipGetLastError();
}

}  // namespace v2v
