// v2v_v2e_tu.hip -- translation unit of the v2e DVS kernels: template instantiations + dispatch.
#include <cstring>

#include "../../include/v2v_hip.h"
#include "v2v_v2e.hpp"
#include "v2v_luts.inc"

namespace v2v {

__device__ float g_lut_v2e32[256] = {V2V_LUT_V2E32_VALUES};    // lin_log table (golden G1); replaceable via v2v_lut_set
static const float kLutV2e32[256] = {V2V_LUT_V2E32_VALUES};

namespace {

// the pre-pass covers kPreGroups x VEC pixels per work-item: its own grid (clips = main grid / blocks_per_clip)
dim3 pre_grid(const V2eArgs &a, dim3 grid) { return dim3((grid.x / (unsigned)a.blocks_per_clip) * (unsigned)a.pre_blocks_per_clip); }

template <int IN, int VEC, int BIN, int RNG>
hipError_t launch_v2e_out(bool out64, const V2eArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
    if (out64) hipLaunchKernelGGL((v2e_voxel_kernel<IN, VEC, BIN, RNG, true>), grid, dim3(kBlock), lds, s, a);
    else hipLaunchKernelGGL((v2e_voxel_kernel<IN, VEC, BIN, RNG, false>), grid, dim3(kBlock), lds, s, a);
    return hipGetLastError();
}

template <int IN, int VEC>
hipError_t launch_v2e_t(int bin, int rng, bool out64, bool presum, const V2eArgs &a, dim3 grid, size_t lds, hipStream_t s)
{
#ifdef V2V_SWEEP_MINIMAL
    return hipErrorInvalidValue;
#endif
    if (presum) {
        hipLaunchKernelGGL((v2e_shot_sum_kernel<IN, VEC>), pre_grid(a, grid), dim3(kBlock), (size_t)a.K * 32, s, a);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (bin == V2V_BIN_SUM)
        return rng == V2V_RNG_PHILOX ? launch_v2e_out<IN, VEC, kBinSum, kRngPhilox>(out64, a, grid, lds, s)
                                     : launch_v2e_out<IN, VEC, kBinSum, kRngReplay>(out64, a, grid, lds, s);
    return rng == V2V_RNG_PHILOX ? launch_v2e_out<IN, VEC, kBinBilinear, kRngPhilox>(out64, a, grid, lds, s)
                                 : launch_v2e_out<IN, VEC, kBinBilinear, kRngReplay>(out64, a, grid, lds, s);
}

}  // namespace

hipError_t launch_v2e(bool in_u8, bool vec4, int bin, int rng, bool out64, bool presum, const V2eArgs &a_in, dim3 grid, size_t lds,
                      hipStream_t s)
{
    V2eArgs a = a_in;
    void *lut = nullptr;
    const hipError_t e = hipGetSymbolAddress(&lut, HIP_SYMBOL(g_lut_v2e32));      // address on the current device
    if (e != hipSuccess) return e;
    a.lut = static_cast<const float *>(lut);
    const V2eParams &P = a.P;
    bool spec = vec4 && !out64 && rng == V2V_RNG_PHILOX && P.threshold_model != kV2eSpatialTemporalIndependent;
    if (spec && !in_u8 && P.cutoff_hz > 0) {
        // the specialised float32 instances tabulate the low-pass factors for ONE float32(dt / tau): true for every frame rate and
        // cut-off tried (the float64 wobble of i/fps - (i-1)/fps disappears in the cast), but checked, not assumed
        const double tau = 1 / (3.141592653589793 * 2 * P.cutoff_hz);
        const float first = (float)((1.0 / P.fps - 0.0 / P.fps) / tau);
        for (int i = 2; i <= a.K && spec; ++i) spec = (float)(((double)i / P.fps - (double)(i - 1) / P.fps) / tau) == first;
    }
    if (spec) {
        const int feat = (P.cutoff_hz > 0 ? kV2eLowpass : 0) | (P.leak_rate_hz > 0 ? kV2eLeak : 0) | (P.shot_noise_rate_hz > 0 ? kV2eShot : 0);
        if (presum) {
            if (in_u8) hipLaunchKernelGGL((v2e_shot_sum_kernel<kInU8, 4>), pre_grid(a, grid), dim3(kBlock), (size_t)a.K * 32, s, a);
            else hipLaunchKernelGGL((v2e_shot_sum_kernel<kInF32, 4>), pre_grid(a, grid), dim3(kBlock), (size_t)a.K * 32, s, a);
            const hipError_t e1 = hipGetLastError();
            if (e1 != hipSuccess) return e1;
        }
        return in_u8 ? launch_v2e_spec_u8(bin, feat, a, grid, lds, s) : launch_v2e_spec_f32(bin, feat, a, grid, lds, s);
    }
    if (in_u8) return vec4 ? launch_v2e_t<kInU8, 4>(bin, rng, out64, presum, a, grid, lds, s) : launch_v2e_t<kInU8, 1>(bin, rng, out64, presum, a, grid, lds, s);
    return vec4 ? launch_v2e_t<kInF32, 4>(bin, rng, out64, presum, a, grid, lds, s) : launch_v2e_t<kInF32, 1>(bin, rng, out64, presum, a, grid, lds, s);
}

hipError_t lut_v2e_copy(void *host, bool to_device)
{
    if (to_device) return hipMemcpyToSymbol(HIP_SYMBOL(g_lut_v2e32), host, sizeof(float) * 256);
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_lut_v2e32), sizeof(float) * 256) != hipSuccess) {
        (void)hipGetLastError();
        memcpy(host, kLutV2e32, sizeof(float) * 256);
    }
    return hipSuccess;
}

}  // namespace v2v
