// v2v_v2e_tu.hip -- translation unit of the v2e DVS kernels: template instantiations + dispatch.
#include <cstring>

#include "../../include/v2v_hip.h"
#include "v2v_v2e.hpp"

namespace v2v {
namespace {

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
        hipLaunchKernelGGL((v2e_shot_sum_kernel<IN, VEC>), grid, dim3(kBlock), 0, s, a);
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

hipError_t launch_v2e(bool in_u8, bool vec4, int bin, int rng, bool out64, bool presum, const V2eArgs &a, dim3 grid, size_t lds,
                      hipStream_t s)
{
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
