// v2v_frontend.hpp -- decode-side front-end on the GPU (gfx950).
//
// Replaces, for frames that are already decoded and resident in HBM, the per-frame host loop of
// data/v2v_datasets.py:191-224 ([cv2.cvtColor BGR2GRAY] -> crop -> cv2.resize(INTER_LINEAR) -> [cv2.flip] -> shake
// crop) and the pause-index gather + gray extraction of :311-316, producing the simulator's uint8 input directly.
// One work-item per output pixel; every output byte depends on <= 4 source pixels -> bandwidth-trivial gather
// kernel (bytes: N*crop^2*C out, <= 4x that in, mostly L2 hits).
// OpenCV is absent from the reference tree and from this image: the arithmetic below restates OpenCV's published
// 8-bit algorithm (see oracle/frontend_oracle.py) and is checked bit-exactly against that restatement only --
// PARITY UNPINNED against cv2 itself.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2v {

struct FrontendArgs {
    const uint8_t *src;          // [T,Hs,Ws,Cs] decoded frames (Cs = 3 BGR or 1 gray)
    int32_t T, Hs, Ws, Cs;
    int32_t min_i, min_j, crop_before;      // crop rectangle in the source frame (square)
    int32_t need_h, need_w, crop;           // resize target (crop + shake extent), final crop size
    int32_t flip, gray_first;               // gray_first: cvtColor BGR2GRAY before the resize (color_mode 'gray')
    const int32_t *frame_idx;               // [N] decoded-frame index of every simulator frame (pause schedule)
    const int32_t *di, *dj;                 // [T] shake offsets (already shifted to >= 0) or nullptr
    int32_t N, Cout;                        // Cout = 1 (gray_first or Cs == 1) or 3
    uint8_t *out_imgs;                      // [N,crop,crop,Cout] or nullptr
    uint8_t *out_gray;                      // [N,crop,crop]
    // batch form: blockIdx.y = clip; every pointer above advances by one clip; per-clip crop parameters
    const int32_t *clip_table;              // [B,4] {min_i, min_j, crop_before, flip} or nullptr (scalars above)
    const uint8_t *src_end;                 // one past the last source byte (bounds the 8-byte neighbour loads)
};

struct Coef { int s0, s1, a0, a1; };

// OpenCV's source coordinate and 11-bit weights for destination index d; `scale` = 1.0 / ((double)dsize / ssize)
__device__ __forceinline__ Coef resize_coef(int d, int ssize, double scale)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = f - (float)s;
    if (s < 0) { f = 0.0f; s = 0; }
    if (s >= ssize - 1) { f = 0.0f; s = ssize - 1; }
    Coef c;
    c.s0 = s;
    c.s1 = s + 1 < ssize ? s + 1 : ssize - 1;
    c.a0 = (int)__builtin_rintf((1.0f - f) * 2048.0f);
    c.a1 = (int)__builtin_rintf(f * 2048.0f);
    return c;
}

__device__ __forceinline__ int bgr2gray_cv(const uint8_t *p)
{
    return ((int)p[0] * 1868 + (int)p[1] * 9617 + (int)p[2] * 4899 + (1 << 13)) >> 14;
}

constexpr int kFrontPx = 4;    // horizontally adjacent output pixels per work-item (one packed 4-byte gray store)

__global__ void __launch_bounds__(256) frontend_kernel(const FrontendArgs a_in)
{
    FrontendArgs a = a_in;
    const int qpr = (a.crop + kFrontPx - 1) / kFrontPx;          // work-items per output row
    const int64_t per_frame = (int64_t)a.crop * a.crop;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)a.N * a.crop * qpr) return;
    {
        const int64_t clip = blockIdx.y;
        a.src += clip * a.T * a.Hs * a.Ws * a.Cs;
        a.frame_idx += clip * a.N;
        if (a.di) { a.di += clip * a.T; a.dj += clip * a.T; }
        if (a.out_imgs) a.out_imgs += clip * a.N * per_frame * a.Cout;
        a.out_gray += clip * a.N * per_frame;
        if (a.clip_table) {
            const int32_t *t4 = a.clip_table + clip * 4;
            a.min_i = t4[0]; a.min_j = t4[1]; a.crop_before = t4[2]; a.flip = t4[3];
        }
    }
    const int n = (int)(gid / ((int64_t)a.crop * qpr));
    const int rem = (int)(gid - (int64_t)n * a.crop * qpr);
    const int y = rem / qpr, x0 = (rem - y * qpr) * kFrontPx;
    const int t = a.frame_idx[n];
    const int Y = y + (a.di ? a.di[t] : 0);
    const int dj = a.dj ? a.dj[t] : 0;
    const uint8_t *frame = a.src + ((int64_t)t * a.Hs * a.Ws + (int64_t)a.min_i * a.Ws + a.min_j) * a.Cs;
    const bool gray3 = a.gray_first && a.Cs == 3;
    auto src_px = [&](const uint8_t *row, int sx, int c) -> int {
        const uint8_t *p = row + (int64_t)sx * a.Cs;
        return gray3 ? bgr2gray_cv(p) : (int)p[c];
    };
    const bool area2 = (a.crop_before == 2 * a.need_w) && (a.crop_before == 2 * a.need_h);
    const double scale_x = 1.0 / ((double)a.need_w / (double)a.crop_before);
    const double scale_y = 1.0 / ((double)a.need_h / (double)a.crop_before);
    Coef cy{};
    if (!area2) cy = resize_coef(Y, a.crop_before, scale_y);
    const uint8_t *row0 = frame + (int64_t)(area2 ? 2 * Y : cy.s0) * a.Ws * a.Cs;
    const uint8_t *row1 = frame + (int64_t)(area2 ? 2 * Y + 1 : cy.s1) * a.Ws * a.Cs;
    uint32_t packed = 0;
    const int64_t obase = (int64_t)n * per_frame + (int64_t)y * a.crop + x0;
#pragma unroll
    for (int j = 0; j < kFrontPx; ++j) {
        const int x = x0 + j;
        if (x >= a.crop) break;
        int X = x + dj;
        if (a.flip) X = a.need_w - 1 - X;
        int vals[3];
        if (area2) {
            for (int c = 0; c < a.Cout; ++c)
                vals[c] = (src_px(row0, 2 * X, c) + src_px(row0, 2 * X + 1, c) + src_px(row1, 2 * X, c) + src_px(row1, 2 * X + 1, c) + 2) >> 2;
        } else {
            const Coef cx = resize_coef(X, a.crop_before, scale_x);
            const uint8_t *p0 = row0 + (int64_t)cx.s0 * 3, *p1 = row1 + (int64_t)cx.s0 * 3;
            if (gray3 && cx.s1 == cx.s0 + 1 && p0 + 8 <= a.src_end && p1 + 8 <= a.src_end) {
                // both horizontal neighbours are 6 contiguous bytes (B,G,R,B,G,R): one unaligned 8-byte load per row
                // instead of six byte loads (the texture-address unit, not bandwidth, bounds this kernel)
                uint64_t w0, w1;
                __builtin_memcpy(&w0, p0, 8);
                __builtin_memcpy(&w1, p1, 8);
                auto g2 = [](uint64_t w, int sh) -> int {
                    return ((int)((w >> sh) & 255) * 1868 + (int)((w >> (sh + 8)) & 255) * 9617 + (int)((w >> (sh + 16)) & 255) * 4899 + (1 << 13)) >> 14;
                };
                const int r0 = g2(w0, 0) * cx.a0 + g2(w0, 24) * cx.a1;
                const int r1 = g2(w1, 0) * cx.a0 + g2(w1, 24) * cx.a1;
                const int v = (((cy.a0 * (r0 >> 4)) >> 16) + ((cy.a1 * (r1 >> 4)) >> 16) + 2) >> 2;
                vals[0] = v < 0 ? 0 : (v > 255 ? 255 : v);
            } else
            for (int c = 0; c < a.Cout; ++c) {
                const int r0 = src_px(row0, cx.s0, c) * cx.a0 + src_px(row0, cx.s1, c) * cx.a1;
                const int r1 = src_px(row1, cx.s0, c) * cx.a0 + src_px(row1, cx.s1, c) * cx.a1;
                const int v = (((cy.a0 * (r0 >> 4)) >> 16) + ((cy.a1 * (r1 >> 4)) >> 16) + 2) >> 2;
                vals[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
            }
        }
        if (a.out_imgs) for (int c = 0; c < a.Cout; ++c) a.out_imgs[(obase + j) * a.Cout + c] = (uint8_t)vals[c];
        int gray = vals[0];
        if (a.Cout == 3) {                                   // bgr_to_gray (v2v_datasets.py:19-22): float64, truncating cast
            const double s01 = (double)vals[0] * 0.5870 + (double)vals[1] * 0.1140;
            gray = (int)(uint8_t)(s01 + (double)vals[2] * 0.2989);
        }
        packed |= (uint32_t)(gray & 0xFF) << (8 * j);
    }
    if ((a.crop % kFrontPx) == 0 && (reinterpret_cast<uintptr_t>(a.out_gray) % 4) == 0) {
        *reinterpret_cast<uint32_t *>(a.out_gray + obase) = packed;
    } else {
        for (int j = 0; j < kFrontPx && x0 + j < a.crop; ++j) a.out_gray[obase + j] = (uint8_t)(packed >> (8 * j));
    }
}

}  // namespace v2v
