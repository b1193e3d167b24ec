// pk_cohazard.hip -- an ATTEMPT at a stand-alone reproducer of round 5's co-scheduling finding: in the library, kernels with
// packed float32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) returned wrong values while a matrix-core kernel (the
// ConvLSTM step, a rocBLAS bf16 GEMM) ran on ANOTHER stream -- reproducible at will with tools/pk_cohazard_probe.py on a build WITH packed
// instructions.  RESULT OF THIS FILE: the synthetic victims below (register chains; load-fed blends like the upsampling kernel's) beside
// the synthetic disturber (v_mfma_f32_32x32x16_bf16 on 128 AGPRs, fragments out of 128 KB of LDS) do NOT show it: 0 differing launches of
// 40 each.  Whatever triggers it needs more of the real kernels than this; kept so that the next attempt starts from here.
//
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o pk_cohazard tools/ubench/pk_cohazard.hip && ./pk_cohazard
//
// For each victim: run alone -> reference; then `rounds` times beside the disturber on a second stream; count outputs that differ from
// the reference and print which lanes / which half of the pair.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

constexpr int kIters = 400;

template <bool SGPR_OPERANDS>
__global__ void __launch_bounds__(256) victim_pk(float *out, float k25, float k75, int iters)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    f32x2 a = {1.0f + 0.001f * (float)(i & 1023), 2.0f - 0.002f * (float)(i & 511)};
    f32x2 b = {0.5f + 0.003f * (float)(i & 255), 1.5f + 0.001f * (float)(i & 127)};
    float q = k25, t = k75;
    if (!SGPR_OPERANDS) { asm volatile("" : "+v"(q)); asm volatile("" : "+v"(t)); }   // weights in VGPRs; else wave-uniform kernel arguments (SGPRs)
    for (int it = 0; it < iters; ++it) {
        const f32x2 m = a * f32x2{t, t};
        const f32x2 l = b * f32x2{q, q} + m;
        const f32x2 r = m + f32x2{b.y, b.x} * f32x2{q, q};       // swizzled second operand (op_sel)
        a = l * f32x2{0.999f, 1.001f} + f32x2{0.001f, -0.001f};
        b = r * f32x2{1.0005f, 0.9995f};
    }
    out[2 * i] = a.x + b.x;
    out[2 * i + 1] = a.y + b.y;
}

__global__ void __launch_bounds__(256) victim_scalar(float *out, float k25, float k75, int iters)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float ax = 1.0f + 0.001f * (float)(i & 1023), ay = 2.0f - 0.002f * (float)(i & 511);
    float bx = 0.5f + 0.003f * (float)(i & 255), by = 1.5f + 0.001f * (float)(i & 127);
    float q = k25, t = k75;
    asm volatile("" : "+v"(q));
    asm volatile("" : "+v"(t));
    for (int it = 0; it < iters; ++it) {
        const float mx = ax * t, my = ay * t;
        const float lx = bx * q + mx, ly = by * q + my;
        const float rx = mx + by * q, ry = my + bx * q;
        ax = lx * 0.999f + 0.001f;
        ay = ly * 1.001f + -0.001f;
        bx = rx * 1.0005f;
        by = ry * 0.9995f;
    }
    out[2 * i] = ax + bx;
    out[2 * i + 1] = ay + by;
}

// closer to the library's upsampling kernel: every iteration LOADS 16 bytes of bf16 data, unpacks them with shifts / masks, blends with packed
// float32 instructions and repacks -- the packed operands come straight out of VMEM returns
__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8])
{
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xFFFF0000u); }
}
template <bool PACKED>
__global__ void __launch_bounds__(256) victim_loads(const uint4 *src, float *out, int n_src, int iters)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        float a[8], b[8], c[8];
        unpack8(src[(i + it * 977) % n_src], a);
        unpack8(src[(i + it * 977 + 1) % n_src], b);
        unpack8(src[(i + it * 977 + 64) % n_src], c);
        for (int e = 0; e < 8; e += 2) {
            if (PACKED) {
                const f32x2 mid = f32x2{b[e], b[e + 1]} * f32x2{0.75f, 0.75f};
                const f32x2 l = f32x2{a[e], a[e + 1]} * f32x2{0.25f, 0.25f} + mid;
                const f32x2 r = mid + f32x2{c[e], c[e + 1]} * f32x2{0.25f, 0.25f};
                const f32x2 s2 = f32x2{acc[e], acc[e + 1]} * f32x2{0.5f, 0.5f} + (l * f32x2{0.75f, 0.75f} + r * f32x2{0.25f, 0.25f});
                acc[e] = s2.x;
                acc[e + 1] = s2.y;
            } else {
                for (int h = 0; h < 2; ++h) {
                    const float mid = b[e + h] * 0.75f;
                    const float l = a[e + h] * 0.25f + mid, r = mid + c[e + h] * 0.25f;
                    acc[e + h] = acc[e + h] * 0.5f + (l * 0.75f + r * 0.25f);
                }
            }
        }
    }
    for (int e = 0; e < 2; ++e) out[2 * i + e] = acc[e] + acc[e + 2] + acc[e + 4] + acc[e + 6];
}

__global__ void __launch_bounds__(256, 1) disturber_mfma(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                     // 128 KB: one workgroup per CU, like the library's ConvLSTM tiles
    bf16x8 *tile = reinterpret_cast<bf16x8 *>(lds);
    for (int i = threadIdx.x; i < 8192; i += 256) { bf16x8 v; for (int e = 0; e < 8; ++e) v[e] = (__bf16)(0.001f * (float)((i + e) & 63)); tile[i] = v; }
    __syncthreads();
    f32x16 acc[8];
    for (int g = 0; g < 8; ++g)
        for (int r = 0; r < 16; ++r) acc[g][r] = (float)(threadIdx.x + g + r) * 1e-3f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.01f * (float)((threadIdx.x + e) & 15)); b[e] = (__bf16)(0.02f * (float)((threadIdx.x * 3 + e) & 7)); }
    for (int it = 0; it < iters; ++it) {
        a = tile[(threadIdx.x * 5 + it * 64) & 8191];                                      // fragments out of LDS, as a GEMM main loop reads them
        b = tile[(threadIdx.x * 3 + it * 32 + 4096) & 8191];
#pragma unroll
        for (int g = 0; g < 8; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[g], 0, 0, 0);
    }
    float s = 0.0f;
    for (int g = 0; g < 8; ++g)
        for (int r = 0; r < 16; ++r) s += acc[g][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    const int blocks = 4096, n = blocks * 256 * 2, rounds = 40;
    float *d_out, *d_dist;
    CHECK(hipMalloc(&d_out, n * sizeof(float)));
    CHECK(hipMalloc(&d_dist, 1024 * 256 * sizeof(float)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&disturber_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    hipStream_t s1, s2;
    CHECK(hipStreamCreate(&s1));
    CHECK(hipStreamCreate(&s2));
    const int n_src = 1 << 20;
    uint4 *d_src;
    CHECK(hipMalloc(&d_src, n_src * sizeof(uint4)));
    {
        std::vector<unsigned> h(n_src * 4);
        unsigned st = 12345u;
        for (auto &v : h) { st = st * 1664525u + 1013904223u; v = ((st >> 9) & 0x007F007Fu) | 0x3F803F80u; }   // pairs of bf16 values in [1, 2)
        CHECK(hipMemcpy(d_src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    std::vector<float> ref(n), got(n), ref_scalar(n);
    auto launch = [&](int which) {
        if (which == 3) hipLaunchKernelGGL(victim_loads<true>, dim3(blocks), dim3(256), 0, s1, d_src, d_out, n_src, 40);
        else if (which == 4) hipLaunchKernelGGL(victim_loads<false>, dim3(blocks), dim3(256), 0, s1, d_src, d_out, n_src, 40);
        else if (which == 0) hipLaunchKernelGGL(victim_pk<false>, dim3(blocks), dim3(256), 0, s1, d_out, 0.25f, 0.75f, kIters);
        else if (which == 1) hipLaunchKernelGGL(victim_pk<true>, dim3(blocks), dim3(256), 0, s1, d_out, 0.25f, 0.75f, kIters);
        else hipLaunchKernelGGL(victim_scalar, dim3(blocks), dim3(256), 0, s1, d_out, 0.25f, 0.75f, kIters);
    };
    const char *names[5] = {"victim_pk (VGPR operands)", "victim_pk (SGPR operands)", "victim_scalar", "victim_loads (packed)", "victim_loads (scalar)"};
    for (int which = 0; which < 5; ++which) {
        launch(which);
        CHECK(hipStreamSynchronize(s1));
        CHECK(hipMemcpy(ref.data(), d_out, n * sizeof(float), hipMemcpyDeviceToHost));
        if (which == 2) ref_scalar = ref;
        int alone_bad = 0;
        for (int r = 0; r < 5; ++r) {
            launch(which);
            CHECK(hipStreamSynchronize(s1));
            CHECK(hipMemcpy(got.data(), d_out, n * sizeof(float), hipMemcpyDeviceToHost));
            alone_bad += memcmp(got.data(), ref.data(), n * sizeof(float)) != 0;
        }
        long bad_launches = 0, bad_elems = 0, lane_hist[4] = {0, 0, 0, 0}, half_hist[2] = {0, 0};
        for (int r = 0; r < rounds; ++r) {
            hipLaunchKernelGGL(disturber_mfma, dim3(1024), dim3(256), 128 * 1024, s2, d_dist, 3000);
            launch(which);
            CHECK(hipStreamSynchronize(s1));
            CHECK(hipStreamSynchronize(s2));
            CHECK(hipMemcpy(got.data(), d_out, n * sizeof(float), hipMemcpyDeviceToHost));
            long b = 0;
            for (int i = 0; i < n; ++i)
                if (memcmp(&got[i], &ref[i], 4) != 0) { ++b; ++lane_hist[((i / 2) & 63) / 16]; ++half_hist[i & 1]; }
            bad_launches += b != 0;
            bad_elems += b;
        }
        printf("%-28s alone: %d/5 launches differ | beside the MFMA kernel: %ld/%d launches differ, %ld elements; by lane quarter [0-15 16-31 32-47 48-63] = [%ld %ld %ld %ld], low / high half of the pair = %ld / %ld\n",
               names[which], alone_bad, bad_launches, rounds, bad_elems, lane_hist[0], lane_hist[1], lane_hist[2], lane_hist[3], half_hist[0], half_hist[1]);
    }
    // the packed and the scalar victim compute the same IEEE arithmetic
    launch(0);
    CHECK(hipStreamSynchronize(s1));
    CHECK(hipMemcpy(got.data(), d_out, n * sizeof(float), hipMemcpyDeviceToHost));
    printf("packed (alone) == scalar (alone): %s\n", memcmp(got.data(), ref_scalar.data(), n * sizeof(float)) == 0 ? "yes" : "NO");
    return 0;
}
