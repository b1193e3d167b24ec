// v2v_args.hpp -- kernel argument structs, shared enums and the launcher entry points of the three translation units
// (v2v_esim_{u8,f32}_tu.hip, v2v_v2e_tu.hip, v2v_v2e_spec_{u8,f32}_tu.hip, v2v_capi.hip).  The heavy kernels live in their own TUs so they build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2v {

enum { kInU8 = 0, kInF32 = 1 };
enum { kRngNone = 0, kRngPhilox = 1, kRngReplay = 2 };
enum { kBinSum = 0, kBinBilinear = 1 };
constexpr int kBlock = 256;
constexpr int kPreGroups = 4;              // VEC-pixel groups per work-item of the v2e frame-sum pre-pass

struct EsimArgs {
    const void *frames;
    int64_t clip_stride, frame_stride;     // elements
    const double *params;
    int64_t params_stride;
    void *out;
    unsigned long long *counts;            // [B,2] or nullptr
    const double *u_init, *u_hot, *g_hot, *g_base;
    uint64_t seed, clip_id0;
    const unsigned long long *clip_keys;   // optional [B,2] per-clip {seed, clip id}: overrides seed / clip_id0 + b
    int32_t HW, K, Tb, fpb, blocks_per_clip;
    uint32_t noise_external;
    uint32_t sym_only;                     // V2V_FLAG_SYMMETRIC: every clip has C+ == C- (instances without the asymmetric loop)
    int32_t W;                             // row length (for the padded output layout)
    int64_t out_pitch, out_plane;          // output row pitch / plane size in elements (W, H*W when unpadded)
    unsigned int *stats;                   // optional [B, kStatWords] per-clip value histogram of the voxels written (SUM mode, float32 grid)
    // optional (FIDX instances): simulator frame f of clip b is STORED frame frame_index[b * (K + 1) + f] (the reference's pause-index
    // gather, data/v2v_datasets.py:286-311, folded into the loads: paused frames are stored once), and clip b starts at element
    // clip_offsets[b] of `frames` instead of b * clip_stride (clips of different stored lengths packed back to back)
    const int32_t *frame_index;
    const int64_t *clip_offsets;
    // bounds of the gather (optional): clip b holds stored_frames[b] frames; `frames` holds frames_elems elements in all (0 = not stated).
    // A row that names a frame outside [0, stored_frames[b]) or a clip that does not fit the buffer poisons the clip (NaN planes,
    // kStatBad) instead of reading out of bounds -- checked once per workgroup while the row is staged through LDS.
    const int32_t *stored_frames;
    int64_t frames_elems;
};

// Per-clip statistics the simulator's writer accumulates for the consumer's normalize_batch_voxel (model/train_utils.py:147-166):
// word kStatZero + v counts the voxels equal to the integer v in -255..255 -- EXCEPT v = 0, which is left at 0 (the reader derives
// it from the element count) --, words 0 and kStatBins - 1 the voxels below -255 / above 255, word kStatBad != 0 flags a clip
// whose planes are not counts (NaN-poisoned).  Same bin layout as the counting select of v2v_postops.hpp.
constexpr int kStatMax = 255, kStatZero = kStatMax + 1, kStatBins = 2 * kStatMax + 3, kStatBad = kStatBins, kStatWords = 516;

struct V2eParams {            // mirrors v2v_v2e_params (include/v2v_hip.h)
    double fps;
    int threshold_model;
    double thres_mean_mean, thres_mean_std, thres_diff_mean, thres_diff_std;
    double cutoff_hz, leak_rate_hz, refractory_period_s, shot_noise_rate_hz, leak_jitter_fraction, noise_rate_cov_decades;
    int uint8_wrap;
};

struct V2eArgs {
    const void *frames;
    int64_t clip_stride, frame_stride;
    void *out;
    unsigned long long *counts;
    long long *shot_sums;                    // [B,K,4] integer sums {ON lo, ON hi, OFF lo, OFF hi} (native shot noise) or nullptr
    const float *lut;                        // 256-entry lin_log table in device memory (set by launch_v2e)
    const double *r_pos_thres, *r_neg_thres; // replay
    int64_t r_thres_frame_stride;
    const float *r_noise_rate;
    const double *r_leak_randn;
    const long long *r_shot_pos, *r_shot_neg;
    uint64_t seed, clip_id0;
    int32_t HW, K, Tb, fpb, blocks_per_clip;
    int32_t pre_blocks_per_clip;             // workgroups per clip of the frame-sum pre-pass (16 pixels per work-item)
    V2eParams P;
};

// defined in v2v_esim_{u8,f32}_tu.hip / v2v_v2e_tu.hip
hipError_t launch_esim_u8(int vec, int bin, int rng, bool noise, bool out64, const EsimArgs &a, dim3 grid, size_t lds, hipStream_t s);
hipError_t launch_esim_f32(int vec, int bin, int rng, bool noise, bool out64, const EsimArgs &a, dim3 grid, size_t lds, hipStream_t s);
hipError_t launch_v2e(bool in_u8, bool vec4, int bin, int rng, bool out64, bool presum, const V2eArgs &a, dim3 grid, size_t lds,
                      hipStream_t s);
// specialised v2e instances (v2v_v2e_spec_{u8,f32}_tu.hip): 4 pixels per work-item, float32 grid, device RNG, static
// thresholds; `feat` = compile-time feature mask {1 low-pass, 2 leak, 4 shot noise}
hipError_t launch_v2e_spec_u8(int bin, int feat, const V2eArgs &a, dim3 grid, size_t lds, hipStream_t s);
hipError_t launch_v2e_spec_f32(int bin, int feat, const V2eArgs &a, dim3 grid, size_t lds, hipStream_t s);
// log-intensity tables: which = 0 ESIM float64, 1 ESIM float32 (the esim TUs), 2 v2e float32 (v2e TU).
// to_device: copy host -> device symbol; else device symbol -> host (falls back to the built-in table without a device).
hipError_t lut_esim64_copy(void *host, bool to_device);
hipError_t lut_esim32_copy(void *host, bool to_device);
hipError_t lut_v2e_copy(void *host, bool to_device);

// fused ConvLSTM step (v2v_convlstm_tu.hip; kernel + argument struct in v2v_convlstm.hpp)
struct ConvLstmArgs {
    const uint16_t *x, *h_prev;            // bf16 NHWC; h_prev may be null (zero state: its half of K is skipped)
    const float *c_prev;                   // fp32 NHWC or null (zero)
    const uint16_t *wp;                    // packed weights
    const float *bias;                     // [4C] in the module's order (gate-major)
    uint16_t *h_state;                     // bf16 NHWC
    float *c_state;                        // fp32 NHWC
    void *h_nchw;                          // optional NCHW copy of h for the layers downstream
    int32_t h_nchw_bf16;                   // its dtype: 0 fp32, 1 bf16
    int32_t B, H, W, C;
    // plain 3x3 convolution (EPI = 1 instances of the same kernel): x [B,H,W,C] -> out_nhwc [B,H,W,n_cols], bias + optional
    // residual [B,H,W,n_cols] + optional ReLU; h_prev / c_prev / h_state / c_state / h_nchw unused
    const uint16_t *residual;
    uint16_t *out_nhwc;
    int32_t n_cols, relu;
    int32_t ks, stride, Hin, Win;          // taps per side (3 or 5, pad ks/2), stride (1 or 2), input size (H, W = the OUTPUT size)
    int32_t pack_cols;                     // columns per PACKED weight tile when the instance's tile is narrower (0: the same)
};
hipError_t launch_convlstm_step(const ConvLstmArgs &a, int tile_rows, hipStream_t s);   // tile_rows: 0 auto, 64, 128 or 256
hipError_t launch_convlstm_pack(const float *w, uint16_t *wp, int C, hipStream_t s);
hipError_t launch_conv_nhwc(const ConvLstmArgs &a, int tile_rows, hipStream_t s);
int conv_tile_cols(int Cout);             // columns per tile of the instance launch_conv_nhwc takes for Cout (0: unsupported)
hipError_t launch_conv_head(const uint16_t *x8, const uint16_t *wp, const float *bias, uint16_t *out, int B, int H, int W, int ks, int relu, hipStream_t s);
hipError_t launch_conv_head_pack(const float *w, uint16_t *wp, int Cin, int ks, hipStream_t s);
hipError_t launch_to_nhwc8_bf16(const float *src, int64_t sb, int64_t sc, int64_t sh, int64_t sw, uint16_t *dst, int B, int C, int H, int W, const float *scales, hipStream_t s);
hipError_t launch_conv1x1_nhwc(const uint16_t *x, const uint16_t *skip, const float *w, const float *bias, void *out, int out_bf16, int64_t M,
                               int C, int Cout, hipStream_t s);
hipError_t launch_upsample2x_nhwc(const uint16_t *x, const uint16_t *skip, uint16_t *out, int B, int H, int W, int C, hipStream_t s);
hipError_t launch_conv_pack(const float *w, uint16_t *wp, int Cin, int Cout, int ks, hipStream_t s);
hipError_t launch_nchw_to_nhwc_bf16(const void *src, bool src_bf16, uint16_t *dst, int B, int C, int HW, int relu, hipStream_t s);

}  // namespace v2v
