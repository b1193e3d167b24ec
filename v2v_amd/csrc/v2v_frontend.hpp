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
    int32_t flip, gray_first;               // gray_first: cvtColor BGR2GRAY before the resize (color_mode 'gray'): 1 = OpenCV >= 4.0's 15-bit form, 2 = the 14-bit form of 2.x / 3.x
    const int32_t *frame_idx;               // [N] decoded-frame index of every simulator frame (pause schedule)
    const int32_t *di, *dj;                 // [T] shake offsets (already shifted to >= 0) or nullptr
    int32_t N, Cout;                        // Cout = 1 (gray_first or Cs == 1) or 3
    uint8_t *out_imgs;                      // [N,crop,crop,Cout] or nullptr
    uint8_t *out_gray;                      // [N,crop,crop]
    // batch form: blockIdx.y = clip; every pointer above advances by one clip; per-clip crop parameters
    const int32_t *clip_table;              // [B,4] {min_i, min_j, crop_before, flip} or nullptr (scalars above)
    const uint8_t *src_end;                 // one past the last source byte (bounds the 8-byte neighbour loads)
    int32_t cb_max;                         // the frame's short side: a device-resident table's crop_before is clamped to it
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

// cv2.cvtColor(BGR2GRAY) on 8-bit data, in the two fixed-point forms OpenCV has shipped (wave-uniform choice, scalar registers):
//   gray_first == 1  OpenCV >= 4.0 (color_rgb.simd.hpp, RGB2Gray<uchar>): 15-bit weights B 3735, G 19235, R 9798, (+2^14) >> 15
//                    -- what an unpinned `opencv-python` (requirements.txt:11) installs; the default
//   gray_first == 2  OpenCV 2.x / 3.x (color.cpp): 14-bit weights 1868 / 9617 / 4899, (+2^13) >> 14
struct GrayCoef {
    int cb, cg, cr, half, shift;
    uint32_t lo, hi;              // the weights split into bytes for v_dot4_u32_u8: w = hi * 256 + lo per channel
    uint32_t lo16, hi16, half16;  // the same sum scaled by 2^(16 - shift): gray = byte 2 of it (every scaled weight still splits into two bytes)
};
__device__ __forceinline__ GrayCoef gray_coef(int mode)
{
    GrayCoef k;
    const bool v4 = mode != 2;
    k.cb = v4 ? 3735 : 1868; k.cg = v4 ? 19235 : 9617; k.cr = v4 ? 9798 : 4899;
    k.shift = v4 ? 15 : 14; k.half = 1 << (k.shift - 1);
    k.lo = (uint32_t)(k.cb & 255) | ((uint32_t)(k.cg & 255) << 8) | ((uint32_t)(k.cr & 255) << 16);
    k.hi = (uint32_t)(k.cb >> 8) | ((uint32_t)(k.cg >> 8) << 8) | ((uint32_t)(k.cr >> 8) << 16);
    const int up = 16 - k.shift, b16 = k.cb << up, g16 = k.cg << up, r16 = k.cr << up;          // 7470 / 38470 / 19596 (cv4), x4 for cv3
    k.lo16 = (uint32_t)(b16 & 255) | ((uint32_t)(g16 & 255) << 8) | ((uint32_t)(r16 & 255) << 16);
    k.hi16 = (uint32_t)(b16 >> 8) | ((uint32_t)(g16 >> 8) << 8) | ((uint32_t)(r16 >> 8) << 16);
    k.half16 = (uint32_t)k.half << up;
    return k;
}
__device__ __forceinline__ int bgr2gray_cv(const uint8_t *p, const GrayCoef &k)
{
    return ((int)p[0] * k.cb + (int)p[1] * k.cg + (int)p[2] * k.cr + k.half) >> k.shift;
}

constexpr int kFrontPx = 4;    // horizontally adjacent output pixels per work-item (one packed 4-byte gray store)

__global__ void __launch_bounds__(256) frontend_kernel(const FrontendArgs a_in)
{
    FrontendArgs a = a_in;
    const GrayCoef gk = gray_coef(a.gray_first);
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
            a.crop_before = min(max(a.crop_before, 1), a.cb_max);                 // device-resident table: clamped into the frame
            a.min_i = min(max(a.min_i, 0), a.Hs - a.crop_before);
            a.min_j = min(max(a.min_j, 0), a.Ws - a.crop_before);
        }
    }
    const int n = (int)(gid / ((int64_t)a.crop * qpr));
    const int rem = (int)(gid - (int64_t)n * a.crop * qpr);
    const int y = rem / qpr, x0 = (rem - y * qpr) * kFrontPx;
    const int t = min(max(a.frame_idx[n], 0), a.T - 1);
    const int Y = y + (a.di ? a.di[t] : 0);
    const int dj = a.dj ? a.dj[t] : 0;
    const uint8_t *frame = a.src + ((int64_t)t * a.Hs * a.Ws + (int64_t)a.min_i * a.Ws + a.min_j) * a.Cs;
    const bool gray3 = a.gray_first && a.Cs == 3;
    auto src_px = [&](const uint8_t *row, int sx, int c) -> int {
        const uint8_t *p = row + (int64_t)sx * a.Cs;
        return gray3 ? bgr2gray_cv(p, gk) : (int)p[c];
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
                auto g2 = [&gk](uint64_t w, int sh) -> int {
                    return ((int)((w >> sh) & 255) * gk.cb + (int)((w >> (sh + 8)) & 255) * gk.cg + (int)((w >> (sh + 16)) & 255) * gk.cr + gk.half) >> gk.shift;
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
        if (a.Cout == 3) {
            // bgr_to_gray (v2v_datasets.py:19-22): np.dot on the reference's [N,H,W,3] stack accumulates the three channels
            // with sequential float64 FMAs -- fma(r, w2, fma(g, w1, b*w0)) -- then truncates; bit-exact on all 2^24 colours
            // (golden G15, tests/test_frontend.py)
            const double s = __builtin_fma((double)vals[1], 0.1140, (double)vals[0] * 0.5870);
            gray = (int)(uint8_t)__builtin_fma((double)vals[2], 0.2989, s);
        }
        packed |= (uint32_t)(gray & 0xFF) << (8 * j);
    }
    if ((a.crop % kFrontPx) == 0 && (reinterpret_cast<uintptr_t>(a.out_gray) % 4) == 0) {
        *reinterpret_cast<uint32_t *>(a.out_gray + obase) = packed;
    } else {
        for (int j = 0; j < kFrontPx && x0 + j < a.crop; ++j) a.out_gray[obase + j] = (uint8_t)(packed >> (8 * j));
    }
}

// ---- LDS-tiled variant for the training configuration (BGR source, color_mode 'gray', gray output only; shake offsets per frame) ---
// The gather kernel above spends ~350 VALU instructions (float64 coordinate maths, four gray conversions) and eight
// unaligned 8-byte loads per four output pixels.  Here a workgroup owns a tile of (4 x rows_per_wave) x (64 x CPL) output
// pixels of one frame:
//  (1) every lane evaluates the four corner source coordinates (index part only), so that
//  (2) the copy of the source rectangle into LDS (16-byte loads, one wave per source row) is issued before any barrier;
//      the tile's column / row coefficients (OpenCV's fixed-point weights) are computed into LDS while the loads fly;
//  (3) a wave owns rows_per_wave adjacent output rows, a lane CPL adjacent columns.  The wave walks the SOURCE rows it
//      needs once, in order: the horizontal pass of a source row (BGR->gray of both taps from LDS -- three aligned
//      dword reads + v_alignbyte -- and the 11-bit blend) is computed once and reused by every output row that taps it
//      (OpenCV's own hresize/vresize order: ~scale horizontal passes per output row instead of 2); row bookkeeping is
//      wave-uniform (scalar branches).
// Same fixed-point arithmetic, bit for bit; clips whose rectangle does not fit the LDS budget (or that take OpenCV's
// 2x2 area shortcut) fall back to per-pixel global reads inside the same kernel.
constexpr int kTileMaxRows = 64;                          // output rows per tile <= 4 waves x 16
struct ColC { uint16_t s0, single; int16_t a0, a1; };   // source column of the left tap, s1 == s0 (border clamp), weights
struct RowC { int16_t s0, s1, a0, a1; };
__host__ __device__ constexpr int tile_hdr_bytes(int cpl) { return 64 * cpl * 8 + kTileMaxRows * 8; }

struct FrontendTileArgs {
    FrontendArgs f;
    int32_t pitch, max_rows;      // LDS row pitch (bytes, multiple of 16) and row capacity
    int32_t rows_per_wave;        // output rows per wave (tile = 4 x rows_per_wave rows)
    int32_t tiles_x, tiles_y;
    int32_t frames_per_block;     // consecutive output frames walked by one block (1 with shake)
};

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

// 16 bytes starting at g, zero beyond `end` (never inlined: it would bloat the copy loop it is the rare exit of)
__device__ __noinline__ u32x4v load16_clipped(const uint8_t *g, const uint8_t *end)
{
    unsigned char b[16];
    for (int i = 0; i < 16; ++i) b[i] = (g + i < end) ? g[i] : (unsigned char)0;
    u32x4v v;
    __builtin_memcpy(&v, b, 16);
    return v;
}

__device__ __forceinline__ int vblend_cv(int h0, int h1, int ya0, int ya1)
{
    // operands: h < 2^15 (255 * 2048 >> 4), 0 <= ya <= 2048 -> 24-bit multiplies (v_mul_i32_i24)
    const int v = ((__mul24(ya0, h0) >> 16) + (__mul24(ya1, h1) >> 16) + 2) >> 2;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// the same blend on h16 = 16 * (h >> 4)'s operand form: h16 = (horizontal sum) & ~15 < 2^24 and yb = ya << 12 <= 2^23 are unsigned
// 24-bit operands whose 48-bit product is (ya * h) << 16, so v_mul_hi_u32_u24 IS (ya * h) >> 16 -- no shifts around the multiplies
__device__ __forceinline__ int vblend16_cv(int h16_0, int h16_1, uint32_t yb0, uint32_t yb1)
{
    auto mulhi24 = [](uint32_t x, uint32_t y) { return (uint32_t)(((uint64_t)(x & 0xFFFFFFu) * (uint64_t)(y & 0xFFFFFFu)) >> 32); };
    const int v = (int)(mulhi24((uint32_t)h16_0, yb0) + mulhi24((uint32_t)h16_1, yb1) + 2u) >> 2;
    return v > 255 ? 255 : v;
}

// cv2's BGR2GRAY of the pixel in the three low bytes of p (byte 3 is ignored): (B*wb + G*wg + R*wr + half) >> shift with
// the weights split into bytes for v_dot4_u32_u8 (e.g. 19235 = 75*256 + 35): the same integer sum, 4 instructions
// instead of three extracts + three multiply-adds
__device__ __forceinline__ int bgr2gray_dot4(uint32_t p, const GrayCoef &k)
{
    const uint32_t lo = __builtin_amdgcn_udot4(p, k.lo, (uint32_t)k.half, false);
    const uint32_t hi = __builtin_amdgcn_udot4(p, k.hi, 0u, false);
    return (int)((lo + (hi << 8)) >> k.shift);
}

// bytes [o, o+6) of an LDS row (o = byte offset from the 4-byte-aligned row start) -> gray of the two BGR pixels
__device__ __forceinline__ void lds_gray_pair(const unsigned char *row, uint32_t o, const GrayCoef &k, int &g0, int &g1)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(row + (o & ~3u));
    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2];
    const uint32_t sh = o & 3u;
    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh);       // bytes o .. o+3
    const uint32_t hi = __builtin_amdgcn_alignbyte(d2, d1, sh);       // bytes o+4 .. o+7
    g0 = bgr2gray_dot4(lo, k);
    g1 = bgr2gray_dot4(__builtin_amdgcn_alignbyte(hi, lo, 3), k);        // bytes o+3 .. o+6
}

// BGR = false: color_mode 'gray' (cvtColor at the taps, gray clip out).  BGR = true: color_mode 'gray_in_bgr_out' (round 3): the three
// channels are resized as they are -- per source row one horizontal pass per channel (the two taps' bytes picked out of the
// aligned LDS dwords with v_perm, one v_dot2 with the 11-bit weights), one vertical blend per channel --, the resized BGR frame
// goes to out_imgs (when asked for) and its bgr_to_gray (data/v2v_datasets.py:19-22: float64 fma chain, golden G15) to out_gray.
template <int CPL, bool BGR = false>
__global__ void __launch_bounds__(256) frontend_tile_kernel(const FrontendTileArgs ta)
{
    constexpr int NC = BGR ? 3 : 1;                       // channels carried through the two passes
    constexpr int kCols = 64 * CPL;
    extern __shared__ __align__(16) unsigned char s_mem[];
    ColC *s_col = reinterpret_cast<ColC *>(s_mem);
    RowC *s_row = reinterpret_cast<RowC *>(s_mem + kCols * 8);
    unsigned char *s_rows = s_mem + tile_hdr_bytes(CPL);
    const FrontendArgs &a = ta.f;
    const GrayCoef gk = gray_coef(a.gray_first);
    const int clip = blockIdx.y;
    const int tiles = ta.tiles_x * ta.tiles_y;
    // a block walks ta.frames_per_block consecutive output frames of its tile: without shake the extents and the coefficient
    // tables (a third of the kernel's vector instructions: float64 divisions, floors and conversions per work-item) are the
    // same for every frame of a clip and are computed once
    const int ngrp = blockIdx.x / tiles;
    const int tile = blockIdx.x - ngrp * tiles;
    const int n_first = ngrp * ta.frames_per_block, n_last = min(n_first + ta.frames_per_block, (int)a.N) - 1;
    const int ty = tile / ta.tiles_x, tx = tile - ty * ta.tiles_x;
    const int tile_rows = 4 * ta.rows_per_wave;
    const int y0 = ty * tile_rows, x0 = tx * kCols;
    const int ncol = min(kCols, a.crop - x0), nrow = min(tile_rows, a.crop - y0);
    int min_i = a.min_i, min_j = a.min_j, cb = a.crop_before, flip = a.flip;
    if (a.clip_table) { const int32_t *t4 = a.clip_table + (int64_t)clip * 4; min_i = t4[0]; min_j = t4[1]; cb = t4[2]; flip = t4[3]; }
    // a table that lives on the device was not seen by the host: keep every read inside the clip's frames whatever it holds (no effect
    // on a valid table; an invalid one gives a clamped crop, never an out-of-bounds access; a rectangle larger than the LDS tile was
    // sized for takes the unstaged path below)
    cb = min(max(cb, 1), a.cb_max);
    min_i = min(max(min_i, 0), a.Hs - cb);
    min_j = min(max(min_j, 0), a.Ws - cb);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave-uniform: steers scalar loops
    for (int n = n_first; n <= n_last; ++n) {
    if (n > n_first) __syncthreads();                    // everyone is done with the previous frame's rows (and tables, with shake)
    const int t = min(max(a.frame_idx[(int64_t)clip * a.N + n], 0), a.T - 1);     // clamped like the crop rectangle
    const uint8_t *frame = a.src + (((int64_t)clip * a.T + t) * a.Hs * a.Ws + (int64_t)min_i * a.Ws + min_j) * 3;
    uint8_t *out = a.out_gray + ((int64_t)clip * a.N + n) * a.crop * a.crop;
    uint8_t *out3 = (BGR && a.out_imgs) ? a.out_imgs + ((int64_t)clip * a.N + n) * a.crop * a.crop * 3 : nullptr;
    // shake (data/v2v_datasets.py:217-224): the clip is resized to need_h x need_w = crop + the largest offset, flipped, and frame t
    // is cut out at (di[t], dj[t]); without shake need == crop and both offsets are 0.  The batch form has no shake.
    const int di = a.di ? a.di[(int64_t)clip * a.T + t] : 0, dj = a.dj ? a.dj[(int64_t)clip * a.T + t] : 0;
    const int need_w = a.need_w, need_h = a.need_h;
    const bool area2 = cb == 2 * need_w && cb == 2 * need_h;
    const double scale = 1.0 / ((double)need_w / (double)cb), scale_y = 1.0 / ((double)need_h / (double)cb);
    auto col_of = [&](int x) { const int X = x + dj; return flip ? need_w - 1 - X : X; };       // column in the resized image

    // (1) extent of the source rectangle
    const int Xa = flip ? col_of(x0 + ncol - 1) : col_of(x0), Xb = flip ? col_of(x0) : col_of(x0 + ncol - 1);
    const int sx_lo = resize_src_lo(Xa, cb, scale), sy_lo = resize_src_lo(y0 + di, cb, scale_y);
    const int sx_hi = min(resize_src_lo(Xb, cb, scale) + 1, cb - 1), sy_hi = min(resize_src_lo(y0 + di + nrow - 1, cb, scale_y) + 1, cb - 1);
    const int rows = sy_hi - sy_lo + 1;
    const int span_px = sx_hi - sx_lo + 1;
    const int span = span_px * 3;
    // BGR instance: rows are staged as they are, nch 16-byte chunks each (misalignment <= 3 + span + the 12-byte tap reads).
    // Gray instance: rows are staged AS GRAY, a granule of 4 source pixels (12 bytes) -> one dword; the horizontal pass reads a
    // 12-byte window at the lane's first tap (+ 8 bytes of slack in the pitch)
    const int nch = BGR ? (span + 12 + 15) >> 4 : (span_px + 3) >> 2;
    const bool staged = !area2 && rows <= ta.max_rows && (BGR ? nch * 16 : nch * 4 + 12) <= ta.pitch && cb <= 32767;

    // (2) source rectangle -> LDS.  A wave takes FOUR source rows at a time, 16 lanes each (lane >> 4 = row, lane & 15 = chunk
    // column, chunks 16 apart): a row of 70-100 chunks fills 16-lane groups to ~90 % where 64-lane groups of one row were 57 %
    // full -- and the conversion below runs for every lane of a wave whether it holds a chunk or not.  The kStageK loads of a
    // lane are ALL issued before the first LDS write (as load -> write per chunk the compiler waits for every load in turn).
    if (staged) {
        constexpr int kStageK = 4;
        constexpr uint32_t kChunk = BGR ? 16u : 12u;         // gray: the granule's 12 bytes + misalignment <= 3 are inside the 16 loaded
        constexpr uint32_t kOut = BGR ? 16u : 4u;
        const int sub = lane >> 4, cl = lane & 15;
        // (scalar) the tile's last row ends inside the source buffer: no per-load end test (only the last rows of the last frame fail)
        const bool inside = frame + ((int64_t)sy_hi * a.Ws + sx_lo) * 3 + (size_t)nch * kChunk + 16 <= a.src_end;
        for (int rg = wave * 4; rg < rows; rg += 16) {
            const int r = rg + sub;
            const bool rok = r < rows;
            const uint8_t *grow = frame + ((int64_t)(sy_lo + (rok ? r : rows - 1)) * a.Ws + sx_lo) * 3;
            const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(grow) & 3u);
            grow -= mis;                                                     // the 4-byte-aligned address below the row's first byte
            unsigned char *lrow = s_rows + (size_t)r * ta.pitch;
            for (int c0 = 0; c0 < nch; c0 += 16 * kStageK) {
                u32x4v v[kStageK];
#pragma unroll
                for (int k = 0; k < kStageK; ++k) {
                    const int c = c0 + cl + 16 * k;
                    if (rok && c < nch) {
                        const uint8_t *g = grow + (uint32_t)c * kChunk;
                        if (__builtin_expect(inside, 1)) __builtin_memcpy(&v[k], g, 16);
                        else v[k] = load16_clipped(g, a.src_end);             // last bytes of the whole source buffer
                    }
                }
#pragma unroll
                for (int k = 0; k < kStageK; ++k) {
                    const int c = c0 + cl + 16 * k;
                    if (rok && c < nch) {
                        if constexpr (BGR) {
                            *reinterpret_cast<u32x4v *>(lrow + (uint32_t)c * kOut) = v[k];
                        } else {
                            // every source pixel is converted ONCE (cv2 converts the frame before it resizes): the granule's 12
                            // bytes B G R B | G R B G | R B G R -> four gray bytes.  With the weights scaled to a shift of 16 the
                            // gray value is byte 2 of the sum: v_perm_b32(hi, lo, sel) packs the four of them (selector 2 = byte 2
                            // of lo, 6 = byte 2 of hi, 0x0c = zero)
                            const u32x4v w = v[k];
                            const uint32_t d0 = __builtin_amdgcn_alignbyte(w.y, w.x, mis), d1 = __builtin_amdgcn_alignbyte(w.z, w.y, mis),
                                           d2 = __builtin_amdgcn_alignbyte(w.w, w.z, mis);
                            auto sum16 = [&](uint32_t px) { return __builtin_amdgcn_udot4(px, gk.lo16, gk.half16, false) + (__builtin_amdgcn_udot4(px, gk.hi16, 0u, false) << 8); };
                            const uint32_t s0 = sum16(d0), s1 = sum16(__builtin_amdgcn_alignbyte(d1, d0, 3)),
                                           s2 = sum16(__builtin_amdgcn_alignbyte(d2, d1, 2)), s3 = sum16(d2 >> 8);
                            const uint32_t g01 = __builtin_amdgcn_perm(s1, s0, 0x0c0c0602u), g23 = __builtin_amdgcn_perm(s3, s2, 0x06020c0cu);
                            *reinterpret_cast<uint32_t *>(lrow + (uint32_t)c * kOut) = g01 | g23;
                        }
                    }
                }
            }
        }
    }
    const bool tables = n == n_first || a.di != nullptr;              // the tables depend on the frame only through the shake offsets
    if (tables && tid < ncol) {
        const Coef c = resize_coef(col_of(x0 + tid), cb, scale);
        // right tap clamped onto the left one (last source column): g*a0 + g*a1 == g*(a0+a1) + anything*0, so the
        // blend needs no special case
        const bool single = c.s1 == c.s0;
        s_col[tid] = ColC{(uint16_t)c.s0, (uint16_t)single, (int16_t)(single ? c.a0 + c.a1 : c.a0), (int16_t)(single ? 0 : c.a1)};
    }
    if (tables && tid < nrow) {
        const Coef c = resize_coef(y0 + di + tid, cb, scale_y);
        s_row[tid] = RowC{(int16_t)c.s0, (int16_t)c.s1, (int16_t)c.a0, (int16_t)c.a1};
    }
    __syncthreads();

    // (3) a wave: rows_per_wave adjacent output rows; a lane: CPL adjacent columns
    const int yw0 = wave * ta.rows_per_wave, yw1 = min(yw0 + ta.rows_per_wave, nrow) - 1;      // tile-relative, inclusive
    const int cg = lane * CPL;
    if (yw0 > yw1 || cg >= ncol) continue;
    auto store_row = [&](int y, uint32_t packed) {
        uint8_t *o = out + (int64_t)(y0 + y) * a.crop + x0 + cg;
        if (cg + CPL <= ncol && (reinterpret_cast<uintptr_t>(o) & (CPL - 1)) == 0) {
            if constexpr (CPL == 4) *reinterpret_cast<uint32_t *>(o) = packed;
            else *reinterpret_cast<uint16_t *>(o) = (uint16_t)packed;
        } else {
            for (int j = 0; j < CPL && cg + j < ncol; ++j) o[j] = (uint8_t)(packed >> (8 * j));
        }
    };
    // BGR: the CPL resized pixels of one output row -> out_imgs (3 bytes each) and their bgr_to_gray -> out_gray
    auto store_row_bgr = [&](int y, const int (&v)[CPL][3]) {
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const double sgr = __builtin_fma((double)v[j][1], 0.1140, (double)v[j][0] * 0.5870);     // the gather kernel's expression (G15)
            packed |= (uint32_t)(uint8_t)__builtin_fma((double)v[j][2], 0.2989, sgr) << (8 * j);
        }
        store_row(y, packed);
        if (out3) {
            uint8_t *o = out3 + ((int64_t)(y0 + y) * a.crop + x0 + cg) * 3;
            if (CPL == 4 && cg + CPL <= ncol && (reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
                uint32_t *o32 = reinterpret_cast<uint32_t *>(o);
                o32[0] = (uint32_t)v[0][0] | ((uint32_t)v[0][1] << 8) | ((uint32_t)v[0][2] << 16) | ((uint32_t)v[1][0] << 24);
                o32[1] = (uint32_t)v[1][1] | ((uint32_t)v[1][2] << 8) | ((uint32_t)v[2][0] << 16) | ((uint32_t)v[2][1] << 24);
                o32[2] = (uint32_t)v[2][2] | ((uint32_t)v[3][0] << 8) | ((uint32_t)v[3][1] << 16) | ((uint32_t)v[3][2] << 24);
            } else {
                for (int j = 0; j < CPL && cg + j < ncol; ++j)
                    for (int c = 0; c < 3; ++c) o[3 * j + c] = (uint8_t)v[j][c];
            }
        }
    };
    if (staged) {
        ColC cc[CPL];
        uint32_t off[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            cc[j] = s_col[min(cg + j, ncol - 1)];
            off[j] = (uint32_t)(cc[j].s0 - sx_lo) * (BGR ? 3u : 1u);     // byte of the left tap in a staged row
        }
        // gray rows: the lane's CPL left taps lie within 7 bytes of the lowest one when the scale is <= 2 (taps of neighbouring
        // columns are <= 2 apart): one 8-byte window per source row, each column picks its two taps out of it with v_perm
        // (selector bytes 0-7 address the window, 0x0c = zero) as {tap0 | tap1 << 16} for one v_dot2 with the 11-bit weights
        const uint32_t win = BGR ? 0u : min(off[0], off[CPL - 1]);
        const bool wide = !BGR && cb > 2 * need_w;                       // taps further apart: byte reads
        uint32_t sel[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            const uint32_t rel = (off[j] - win) & 7u;
            sel[j] = rel | 0x0c000c00u | (((rel + 1u) & 7u) << 16);
        }
        const uint32_t frame_lo = (uint32_t)reinterpret_cast<uintptr_t>(frame);
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        // horizontal pass of source row sr -> h (the value cv2 keeps between its two passes: (g0*a0 + g1*a1) >> 4)
        auto hpass = [&](int sr, int (&h)[CPL][NC]) __attribute__((always_inline)) {
            const unsigned char *lrow = s_rows + (size_t)(sr - sy_lo) * ta.pitch;
            [[maybe_unused]] const uint32_t mis = (frame_lo + (uint32_t)((sr * a.Ws + sx_lo) * 3)) & 3u;   // low address bits: wrap-safe
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                if constexpr (BGR) {
                    // bytes o .. o+5 = B0 G0 R0 B1 G1 R1 of the two taps: lo = bytes o..o+3, hi = bytes o+4..o+7
                    const uint32_t o = off[j] + mis;
                    const uint32_t *w = reinterpret_cast<const uint32_t *>(lrow + (o & ~3u));
                    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], sh = o & 3u;
                    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
                    const u16x2 wa = u16x2{(unsigned short)cc[j].a0, (unsigned short)cc[j].a1};
                    // v_perm_b32(hi, lo, sel): selector bytes 0-3 address lo, 4-7 address hi, 0x0c = zero -> {tap0 | tap1 << 16} per channel
                    const uint32_t pb = __builtin_amdgcn_perm(hi, lo, 0x0c030c00u), pg = __builtin_amdgcn_perm(hi, lo, 0x0c040c01u),
                                   pr = __builtin_amdgcn_perm(hi, lo, 0x0c050c02u);
                    h[j][0] = (int)(__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pb), wa, 0u, false) & ~15u);
                    h[j][NC > 1 ? 1 : 0] = (int)(__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pg), wa, 0u, false) & ~15u);
                    h[j][NC > 2 ? 2 : 0] = (int)(__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, pr), wa, 0u, false) & ~15u);
                }
            }
            if constexpr (!BGR) {
                if (wide) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) h[j][0] = ((int)lrow[off[j]] * cc[j].a0 + (int)lrow[off[j] + 1] * cc[j].a1) & ~15;
                } else {
                    const uint32_t *w = reinterpret_cast<const uint32_t *>(lrow + (win & ~3u));
                    const uint32_t d0 = w[0], d1 = w[1], d2 = w[2], sh = win & 3u;
                    const uint32_t lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
#pragma unroll
                    for (int j = 0; j < CPL; ++j) {
                        const u16x2 wa = u16x2{(unsigned short)cc[j].a0, (unsigned short)cc[j].a1};
                        h[j][0] = (int)(__builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, __builtin_amdgcn_perm(hi, lo, sel[j])), wa, 0u, false) & ~15u);
                    }
                }
            }
        };
        // the two most recent source rows' passes (ids idA < idB): an output row's taps are rs0 and rs1 in {rs0, rs0 + 1}, rows
        // only move down, so a tap is either cached or newer than both; rows between two output rows' taps (down-scaling by more
        // than 2) are never touched.  Everything that steers this loop is wave-uniform and lives in scalar registers.
        int idA = -1, idB = -1;
        int hA[CPL][NC], hB[CPL][NC];
#pragma unroll
        for (int j = 0; j < CPL; ++j)
#pragma unroll
            for (int c = 0; c < NC; ++c) hA[j][c] = hB[j][c] = 0;
        auto need = [&](int sr) __attribute__((always_inline)) {
            if (sr != idA && sr != idB) {
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#pragma unroll
                    for (int c = 0; c < NC; ++c) hA[j][c] = hB[j][c];
                idA = idB;
                hpass(sr, hB);
                idB = sr;
            }
        };
        for (int y = yw0; y <= yw1; ++y) {
            const RowC rc = s_row[y];
            const int rs0 = __builtin_amdgcn_readfirstlane((int)rc.s0), rs1 = __builtin_amdgcn_readfirstlane((int)rc.s1);
            const int ya0 = __builtin_amdgcn_readfirstlane((int)rc.a0), ya1 = __builtin_amdgcn_readfirstlane((int)rc.a1);
            need(rs0);
            need(rs1);
            const uint32_t yb0 = (uint32_t)ya0 << 12, yb1 = (uint32_t)ya1 << 12;
            // (scalar) branch on which cached pass is which tap: the usual case is top = older, bottom = newer
            auto emit = [&](const int (&top)[CPL][NC], const int (&bot)[CPL][NC]) __attribute__((always_inline)) {
                if constexpr (BGR) {
                    int v[CPL][3];
#pragma unroll
                    for (int j = 0; j < CPL; ++j)
#pragma unroll
                        for (int c = 0; c < 3; ++c) v[j][c] = vblend16_cv(top[j][c % NC], bot[j][c % NC], yb0, yb1);
                    store_row_bgr(y, v);
                } else {
                    uint32_t packed = 0;
#pragma unroll
                    for (int j = 0; j < CPL; ++j) packed |= (uint32_t)vblend16_cv(top[j][0], bot[j][0], yb0, yb1) << (8 * j);
                    store_row(y, packed);
                }
            };
            if (rs0 != idB) emit(hA, hB);                               // rs0 = idA, rs1 = idB
            else if (rs1 == idB) emit(hB, hB);                          // both taps on the newest row (clamped at the border)
            else emit(hB, hA);                                           // cannot happen for rows that only move down; kept for completeness
        }
    } else {
        for (int y = yw0; y <= yw1; ++y) {
            const RowC rc = s_row[y];
            if constexpr (BGR) {                                        // per-pixel global reads, one channel at a time
                int v3[CPL][3];
                for (int j = 0; j < CPL; ++j) {
                    for (int c = 0; c < 3; ++c) v3[j][c] = 0;
                    if (cg + j >= ncol) continue;
                    const int X = col_of(x0 + cg + j);
                    for (int c = 0; c < 3; ++c) {
                        if (area2) {
                            const uint8_t *p0 = frame + ((int64_t)(2 * (y0 + di + y)) * a.Ws + 2 * X) * 3 + c, *p1 = p0 + (int64_t)a.Ws * 3;
                            v3[j][c] = ((int)p0[0] + (int)p0[3] + (int)p1[0] + (int)p1[3] + 2) >> 2;
                        } else {
                            const ColC c1 = s_col[cg + j];
                            const int s1 = c1.single ? c1.s0 : c1.s0 + 1;
                            const uint8_t *p0 = frame + (int64_t)rc.s0 * a.Ws * 3 + c, *p1 = frame + (int64_t)rc.s1 * a.Ws * 3 + c;
                            const int h0 = ((int)p0[c1.s0 * 3] * c1.a0 + (int)p0[s1 * 3] * c1.a1) >> 4;
                            const int h1 = ((int)p1[c1.s0 * 3] * c1.a0 + (int)p1[s1 * 3] * c1.a1) >> 4;
                            v3[j][c] = vblend_cv(h0, h1, rc.a0, rc.a1);
                        }
                    }
                }
                store_row_bgr(y, v3);
                continue;
            }
            uint32_t packed = 0;
            for (int j = 0; j < CPL; ++j) {
                if (cg + j >= ncol) break;
                const int X = col_of(x0 + cg + j);
                int v;
                if (area2) {
                    const uint8_t *p0 = frame + ((int64_t)(2 * (y0 + di + y)) * a.Ws + 2 * X) * 3, *p1 = p0 + (int64_t)a.Ws * 3;
                    v = (bgr2gray_cv(p0, gk) + bgr2gray_cv(p0 + 3, gk) + bgr2gray_cv(p1, gk) + bgr2gray_cv(p1 + 3, gk) + 2) >> 2;
                } else {
                    const ColC c1 = s_col[cg + j];
                    const int s1 = c1.single ? c1.s0 : c1.s0 + 1;      // (a1 is 0 for a clamped tap: either pixel gives the same sum)
                    const uint8_t *p0 = frame + (int64_t)rc.s0 * a.Ws * 3, *p1 = frame + (int64_t)rc.s1 * a.Ws * 3;
                    const int h0 = (bgr2gray_cv(p0 + c1.s0 * 3, gk) * c1.a0 + bgr2gray_cv(p0 + s1 * 3, gk) * c1.a1) >> 4;
                    const int h1 = (bgr2gray_cv(p1 + c1.s0 * 3, gk) * c1.a0 + bgr2gray_cv(p1 + s1 * 3, gk) * c1.a1) >> 4;
                    v = vblend_cv(h0, h1, rc.a0, rc.a1);
                }
                packed |= (uint32_t)v << (8 * j);
            }
            store_row(y, packed);
        }
    }
    }   // frames of this block
}

}  // namespace v2v
