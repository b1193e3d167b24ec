// rng_rates.hip -- cost of the device-native random fields per 4-pixel work-item and time step (gfx950).
// Every kernel runs the generator of v2v_amd/csrc/v2v_rng.hpp in a loop, 4 waves per SIMD, and reports nanoseconds per
// wave and "step" (= what one work-item of the fused kernels needs for 4 pixels and one frame pair).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../v2v_amd/csrc/v2v_rng.hpp"
using namespace v2v;
constexpr int kIters = 512;

template <int ROUNDS> __global__ void __launch_bounds__(256) k_philox(float *out, uint32_t seed)
{
    uint32_t acc = 0;
    for (int i = 0; i < kIters; ++i) {
        const u32x4 w = philox4x32<ROUNDS>(threadIdx.x + blockIdx.x * 256u, (uint32_t)i, 7u, 0u, seed, 99u);
        acc ^= w.x ^ w.y ^ w.z ^ w.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc;
}
// one Philox block + four Box-Muller pairs: 4 pixels x 2 steps
template <int ROUNDS> __global__ void __launch_bounds__(256) k_pairs(float *out, uint32_t seed)
{
    float acc = 0;
    for (int i = 0; i < kIters; ++i) {
        float a[4], b[4];
        field_gauss_pairs<4, ROUNDS>(seed, 7u, (uint32_t)i, 0u, (threadIdx.x + blockIdx.x * 256u) * 4u, a, b);
        acc += (a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3]);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_pairs_fast(float *out, uint32_t seed)
{
    float acc = 0;
    for (int i = 0; i < kIters; ++i) {
        float a[4], b[4];
        field_gauss_pairs_fast<4>(seed, 7u, (uint32_t)i, 0u, (threadIdx.x + blockIdx.x * 256u) * 4u, a, b);
        acc += (a[0] + a[1]) + (a[2] + a[3]) + (b[0] + b[1]) + (b[2] + b[3]);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// transform only (words from a cheap LCG): isolates gauss16_x2
__global__ void __launch_bounds__(256) k_transform(float *out, uint32_t seed)
{
    float acc = 0;
    uint32_t s0 = seed + threadIdx.x * 2654435761u, s1 = s0 ^ 0x9E3779B9u;
    for (int i = 0; i < kIters * 2; ++i) {
        s0 = s0 * 1664525u + 1013904223u; s1 = s1 * 22695477u + 1u;
        f32x2 a, b;
        gauss16_x2(u32x2{s0, s1}, a, b);
        acc += (a.x + a.y) + (b.x + b.y);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename K> static float run(K k, float *out, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<blocks, 256>>>(out, 12345u); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<<<blocks, 256>>>(out, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main()
{
    float *out; const int blocks = 256 * 4;            // 4 waves per SIMD
    hipMalloc(&out, sizeof(float) * blocks * 256);
    const double waves_per_simd = 4.0;
    auto rep = [&](const char *name, float ms, double steps_per_iter, int iters) {
        printf("%-44s %8.3f ms  %7.1f ns per wave and 4-pixel step\n", name, ms, ms * 1e6 / (waves_per_simd * iters * steps_per_iter));
    };
    rep("philox4x32-10 block (per call)", run(k_philox<10>, out, blocks), 1, kIters);
    rep("philox4x32-7 block (per call)", run(k_philox<7>, out, blocks), 1, kIters);
    rep("gauss16_x2 transform only (2 words = 1 step)", run(k_transform, out, blocks), 1, kIters * 2);
    rep("philox-10 + 4 pairs (serves 2 steps)", run(k_pairs<10>, out, blocks), 2, kIters);
    rep("philox-7 + 4 pairs (serves 2 steps)", run(k_pairs<7>, out, blocks), 2, kIters);
    rep("fast: philox-7 + hw transcendentals", run(k_pairs_fast, out, blocks), 2, kIters);
    return 0;
}
