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

// index part of resize_coef only: the clamped left/top tap of destination index d (its right/bottom tap is min(+1, ssize-1))
__device__ __forceinline__ int resize_src_lo(int d, int ssize, double scale)
{
    const float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    if (s < 0) s = 0;
    if (s >= ssize - 1) s = ssize - 1;
    return s;
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

// ---- LDS-tiled variant for the training configuration (BGR source, color_mode 'gray', no shake, gray output only) ---
// The gather kernel above issues eight unaligned 8-byte loads and ~350 VALU instructions (float64 coordinate maths)
// per four output pixels and runs at a fifth of the source bytes' streaming time.  Here a workgroup owns a tile of
// kTileRows x kTileCols output pixels of one frame: (1) 128+8 lanes compute the OpenCV column/row coefficients of the
// tile once into LDS, (2) the source rectangle the tile touches is copied with 16-byte loads into LDS, (3) every lane
// blends its 4 adjacent outputs from LDS (three aligned dword reads + v_alignbyte per tap row) and stores one dword.
// Same fixed-point arithmetic, bit for bit; clips whose rectangle does not fit the LDS budget (or that take OpenCV's
// 2x2 area shortcut) fall back to per-pixel global reads inside the same kernel.
constexpr int kTileRows = 8, kTileCols = 128;
struct ColC { uint16_t s0, single; int16_t a0, a1; };   // source column of the left tap, s1 == s0 (border clamp), weights
struct RowC { int16_t s0, s1, a0, a1; };
constexpr int kTileHdrBytes = kTileCols * 8 + kTileRows * 8;

struct FrontendTileArgs {
    FrontendArgs f;
    int32_t pitch, max_rows;      // LDS row pitch (bytes, multiple of 16) and row capacity
    int32_t tiles_x, tiles_y;
};

__device__ __forceinline__ int blend_cv(int g00, int g01, int g10, int g11, int xa0, int xa1, int ya0, int ya1)
{
    const int r0 = g00 * xa0 + g01 * xa1;
    const int r1 = g10 * xa0 + g11 * xa1;
    const int v = (((ya0 * (r0 >> 4)) >> 16) + ((ya1 * (r1 >> 4)) >> 16) + 2) >> 2;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// bytes [o, o+6) of an LDS row (o = byte offset from the 4-byte-aligned row start) -> gray of the two BGR pixels
__device__ __forceinline__ void lds_gray_pair(const unsigned char *row, uint32_t o, int &g0, int &g1)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(row + (o & ~3u));
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
    const uint32_t sh = o & 3u;
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);       // bytes o .. o+3
    const uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, sh);       // bytes o+4 .. o+7
    g0 = ((int)(lo & 255u) * 1868 + (int)((lo >> 8) & 255u) * 9617 + (int)((lo >> 16) & 255u) * 4899 + (1 << 13)) >> 14;
    g1 = ((int)(lo >> 24) * 1868 + (int)(hi & 255u) * 9617 + (int)((hi >> 8) & 255u) * 4899 + (1 << 13)) >> 14;
}

__global__ void __launch_bounds__(256) frontend_tile_kernel(const FrontendTileArgs ta)
{
    extern __shared__ __align__(16) unsigned char s_mem[];
    ColC *s_col = reinterpret_cast<ColC *>(s_mem);
    RowC *s_row = reinterpret_cast<RowC *>(s_mem + kTileCols * 8);
    unsigned char *s_rows = s_mem + kTileHdrBytes;
    const FrontendArgs &a = ta.f;
    const int clip = blockIdx.y;
    const int tiles = ta.tiles_x * ta.tiles_y;
    const int n = blockIdx.x / tiles;
    const int tile = blockIdx.x - n * tiles;
    const int ty = tile / ta.tiles_x, tx = tile - ty * ta.tiles_x;
    const int y0 = ty * kTileRows, x0 = tx * kTileCols;
    const int ncol = min(kTileCols, a.crop - x0), nrow = min(kTileRows, a.crop - y0);
    int min_i = a.min_i, min_j = a.min_j, cb = a.crop_before, flip = a.flip;
    if (a.clip_table) { const int32_t *t4 = a.clip_table + (int64_t)clip * 4; min_i = t4[0]; min_j = t4[1]; cb = t4[2]; flip = t4[3]; }
    const int t = a.frame_idx[(int64_t)clip * a.N + n];
    const uint8_t *frame = a.src + (((int64_t)clip * a.T + t) * a.Hs * a.Ws + (int64_t)min_i * a.Ws + min_j) * 3;
    uint8_t *out = a.out_gray + ((int64_t)clip * a.N + n) * a.crop * a.crop;
    const bool area2 = cb == 2 * a.crop;
    const double scale = 1.0 / ((double)a.crop / (double)cb);
    const int tid = threadIdx.x;

    // (1) extent of the source rectangle: every lane evaluates the four corner coordinates itself (index part only), so
    //     the copy below can be issued before any LDS traffic or barrier
    const int Xa = flip ? a.crop - 1 - (x0 + ncol - 1) : x0, Xb = flip ? a.crop - 1 - x0 : x0 + ncol - 1;
    const int sx_lo = resize_src_lo(Xa, cb, scale), sy_lo = resize_src_lo(y0, cb, scale);
    const int sx_hi = min(resize_src_lo(Xb, cb, scale) + 1, cb - 1), sy_hi = min(resize_src_lo(y0 + nrow - 1, cb, scale) + 1, cb - 1);
    const int rows = sy_hi - sy_lo + 1;
    const int span = (sx_hi - sx_lo + 1) * 3;
    const int nch = (span + 12 + 15) >> 4;               // 16-byte chunks per row: misalignment (<= 3) + span + the 12-byte tap reads
    const bool staged = !area2 && rows <= ta.max_rows && nch * 16 <= ta.pitch && cb <= 32767;

    // (2) source rectangle -> LDS (row r at s_rows + r*pitch, starting at the 4-byte-aligned address below its first
    //     byte); the tile's column / row coefficients are computed while those loads are in flight
    if (staged) {
        for (int idx = tid; idx < rows * nch; idx += 256) {
            const int r = idx / nch, c = idx - r * nch;
            const uint8_t *g = frame + ((int64_t)(sy_lo + r) * a.Ws + sx_lo) * 3;
            g -= reinterpret_cast<uintptr_t>(g) & 3u;
            g += c * 16;
            uint32_t v[4] = {0u, 0u, 0u, 0u};
            if (g + 16 <= a.src_end) __builtin_memcpy(v, g, 16);
            else for (int b = 0; b < 16 && g + b < a.src_end; ++b) reinterpret_cast<unsigned char *>(v)[b] = g[b];
            uint32_t *d = reinterpret_cast<uint32_t *>(s_rows + (size_t)r * ta.pitch + c * 16);
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
    }
    if (tid < ncol) {
        int X = x0 + tid;
        if (flip) X = a.crop - 1 - X;
        const Coef c = resize_coef(X, cb, scale);
        s_col[tid] = ColC{(uint16_t)c.s0, (uint16_t)(c.s1 == c.s0), (int16_t)c.a0, (int16_t)c.a1};
    } else if (tid >= kTileCols && tid < kTileCols + nrow) {
        const Coef c = resize_coef(y0 + tid - kTileCols, cb, scale);
        s_row[tid - kTileCols] = RowC{(int16_t)c.s0, (int16_t)c.s1, (int16_t)c.a0, (int16_t)c.a1};
    }
    __syncthreads();

    // (3) blend: lane -> one row, 4 adjacent columns
    const int r = tid >> 5, cg = (tid & 31) * 4;
    if (r >= nrow || cg >= ncol) return;
    const RowC rc = s_row[r];
    const int y = y0 + r;
    uint32_t packed = 0;
    if (staged) {
        const uint8_t *g0p = frame + ((int64_t)rc.s0 * a.Ws + sx_lo) * 3, *g1p = frame + ((int64_t)rc.s1 * a.Ws + sx_lo) * 3;
        const uint32_t m0 = (uint32_t)(reinterpret_cast<uintptr_t>(g0p) & 3u), m1 = (uint32_t)(reinterpret_cast<uintptr_t>(g1p) & 3u);
        const unsigned char *l0 = s_rows + (size_t)(rc.s0 - sy_lo) * ta.pitch, *l1 = s_rows + (size_t)(rc.s1 - sy_lo) * ta.pitch;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (cg + j >= ncol) break;
            const ColC cc = s_col[cg + j];
            const uint32_t off = (uint32_t)(cc.s0 - sx_lo) * 3u;
            int g00, g01, g10, g11;
            lds_gray_pair(l0, off + m0, g00, g01);
            lds_gray_pair(l1, off + m1, g10, g11);
            if (cc.single) { g01 = g00; g11 = g10; }
            packed |= (uint32_t)blend_cv(g00, g01, g10, g11, cc.a0, cc.a1, rc.a0, rc.a1) << (8 * j);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (cg + j >= ncol) break;
            int X = x0 + cg + j;
            if (flip) X = a.crop - 1 - X;
            int v;
            if (area2) {
                const uint8_t *p0 = frame + ((int64_t)(2 * y) * a.Ws + 2 * X) * 3, *p1 = p0 + (int64_t)a.Ws * 3;
                v = (bgr2gray_cv(p0) + bgr2gray_cv(p0 + 3) + bgr2gray_cv(p1) + bgr2gray_cv(p1 + 3) + 2) >> 2;
            } else {
                const ColC cc = s_col[cg + j];
                const int s1 = cc.single ? cc.s0 : cc.s0 + 1;
                const uint8_t *p0 = frame + (int64_t)rc.s0 * a.Ws * 3, *p1 = frame + (int64_t)rc.s1 * a.Ws * 3;
                v = blend_cv(bgr2gray_cv(p0 + cc.s0 * 3), bgr2gray_cv(p0 + s1 * 3), bgr2gray_cv(p1 + cc.s0 * 3), bgr2gray_cv(p1 + s1 * 3),
                             cc.a0, cc.a1, rc.a0, rc.a1);
            }
            packed |= (uint32_t)v << (8 * j);
        }
    }
    uint8_t *o = out + (int64_t)y * a.crop + x0 + cg;
    if (cg + 4 <= ncol && (reinterpret_cast<uintptr_t>(o) & 3u) == 0) *reinterpret_cast<uint32_t *>(o) = packed;
    else for (int j = 0; j < 4 && cg + j < ncol; ++j) o[j] = (uint8_t)(packed >> (8 * j));
}

}  // namespace v2v
