// stream_pattern.hip -- what HBM rate does the fused kernel's ACCESS PATTERN allow, with the arithmetic removed?
// Same mapping as esim_voxel_kernel (one work-item = 4 adjacent pixels of one clip, N frames streamed at stride H*W,
// 5 output planes of 16 B per lane), but each sample costs one add.  Compared with a linear read of the same bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int B = 256, N = 32, H = 256, W = 256, HW = H * W, TB = 5;

template <int DEPTH, bool NT, int REMAP = 0, bool NTS = false, bool SPREAD = false>
__global__ void __launch_bounds__(256) pattern_kernel(const float *__restrict__ in, float *__restrict__ out)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int blocks_per_clip = HW / 1024;
    int bid = blockIdx.x;
    if (REMAP == 1) {            // blocks b, b+8, b+16.. share an XCD: give each XCD a contiguous range of chunks
        const int per_xcd = gridDim.x / 8;
        bid = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    } else if (REMAP == 2) {     // same, but contiguous within a clip only (8 consecutive-chunk streams per clip)
        const int g = blockIdx.x / blocks_per_clip, r = blockIdx.x % blocks_per_clip;
        bid = g * blocks_per_clip + (r % 8) * (blocks_per_clip / 8) + r / 8;
    }
    const int clip = bid / blocks_per_clip, blk = bid % blocks_per_clip;
    const int p0 = (blk * 256 + threadIdx.x) * 4;
    const float *base = in + (size_t)clip * N * HW + p0;
    f32x4 acc = {0, 0, 0, 0};
    f32x4 ring[DEPTH];
    auto ld = [&](int f) { const f32x4 *p = reinterpret_cast<const f32x4 *>(base + (size_t)f * HW); return NT ? __builtin_nontemporal_load(p) : *p; };
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) ring[u] = ld(u);
    auto st = [&](int b) {
        f32x4 *o = reinterpret_cast<f32x4 *>(out + ((size_t)clip * TB + b) * HW + p0);
        if (NTS) __builtin_nontemporal_store(acc * (float)(b + 1), o); else *o = acc * (float)(b + 1);
    };
    int nb = 0;
    for (int f0 = 0; f0 < N; f0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            acc += ring[u];
            const int fn = f0 + u + DEPTH;
            ring[u] = ld(fn < N ? fn : N - 1);
        }
        if (SPREAD && (f0 + DEPTH) % 8 == 0 && nb < TB - 1) st(nb++);   // planes leave as the scan passes their segment, like the fused kernel
    }
    for (int b = nb; b < TB; ++b) st(b);
}

__global__ void __launch_bounds__(256) linear_kernel(const float *__restrict__ in, float *__restrict__ out, size_t n4_in, size_t n4_out)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4_in; i += stride) acc += reinterpret_cast<const f32x4 *>(in)[i];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4_out; i += stride) reinterpret_cast<f32x4 *>(out)[i] = acc;
}

// 8 pixels per work-item: each lane reads 32 contiguous bytes per frame (wave = 2 KiB contiguous)
template <int DEPTH>
__global__ void __launch_bounds__(256) pattern8_kernel(const float *__restrict__ in, float *__restrict__ out)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int blocks_per_clip = HW / 2048;
    const int clip = blockIdx.x / blocks_per_clip, blk = blockIdx.x % blocks_per_clip;
    const int p0 = (blk * 256 + threadIdx.x) * 8;
    const float *base = in + (size_t)clip * N * HW + p0;
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    f32x4 r0[DEPTH], r1[DEPTH];
    auto ld = [&](int f, int h) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + (size_t)f * HW) + h); };
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) { r0[u] = ld(u, 0); r1[u] = ld(u, 1); }
    for (int f0 = 0; f0 < N; f0 += DEPTH) {
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            a0 += r0[u]; a1 += r1[u];
            const int fn = f0 + u + DEPTH < N ? f0 + u + DEPTH : N - 1;
            r0[u] = ld(fn, 0); r1[u] = ld(fn, 1);
        }
    }
    for (int b = 0; b < TB; ++b) {
        f32x4 *o = reinterpret_cast<f32x4 *>(out + ((size_t)clip * TB + b) * HW + p0);
        o[0] = a0 * (float)(b + 1); o[1] = a1 * (float)(b + 1);
    }
}

// linear read with 4 independent 16-byte loads in flight per lane
__global__ void __launch_bounds__(256) linear4_kernel(const float *__restrict__ in, float *__restrict__ out, size_t n4_in, size_t n4_out)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    const size_t stride = (size_t)gridDim.x * 256;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(in);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4_in; i += 4 * stride) {
        a0 += __builtin_nontemporal_load(p + i); a1 += __builtin_nontemporal_load(p + i + stride);
        a2 += __builtin_nontemporal_load(p + i + 2 * stride); a3 += __builtin_nontemporal_load(p + i + 3 * stride);
    }
    for (; i < n4_in; i += stride) a0 += p[i];
    a0 += a1 + a2 + a3;
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n4_out; j += stride) reinterpret_cast<f32x4 *>(out)[j] = a0;
}

// reference ceilings: float4 copy (1:1 read:write, the guide's 6.29 TB/s figure) and a read-only sweep
__global__ void __launch_bounds__(256) copy_kernel(const float *__restrict__ in, float *__restrict__ out, size_t n4)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
        __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(in) + i), reinterpret_cast<f32x4 *>(out) + i);
}
__global__ void __launch_bounds__(256) read_kernel(const float *__restrict__ in, float *__restrict__ out, size_t n4)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0;
    const size_t stride = (size_t)gridDim.x * 256;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(in);
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) { a0 += __builtin_nontemporal_load(p + i); a1 += __builtin_nontemporal_load(p + i + stride); }
    a0 += a1;
    if (a0.x == 12345.0f) reinterpret_cast<f32x4 *>(out)[threadIdx.x] = a0;
}

int main()
{
    const size_t n_in = (size_t)B * N * HW, n_out = (size_t)B * TB * HW;
    float *in, *out;
    hipMalloc(&in, n_in * 4); hipMalloc(&out, n_out * 4);
    hipMemset(in, 0, n_in * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)(n_in + n_out) * 4;
    auto time = [&](auto launch, const char *name) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-34s %.4f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    const int grid = B * (HW / 1024);
    time([&] { pattern_kernel<4, true><<<grid, 256>>>(in, out); }, "esim pattern, depth 4, nt loads");
    time([&] { pattern_kernel<4, false><<<grid, 256>>>(in, out); }, "esim pattern, depth 4, plain loads");
    time([&] { pattern_kernel<4, true, 0, true><<<grid, 256>>>(in, out); }, "depth 4, nt loads + nt stores");
    time([&] { pattern_kernel<4, true, 0, false, true><<<grid, 256>>>(in, out); }, "depth 4, nt loads, stores spread");
    time([&] { pattern_kernel<4, true, 0, true, true><<<grid, 256>>>(in, out); }, "depth 4, nt, nt stores spread");
    time([&] { pattern_kernel<4, true, 1, true, true><<<grid, 256>>>(in, out); }, "same + XCD-contiguous remap");
    time([&] { pattern_kernel<4, true, 2, true, true><<<grid, 256>>>(in, out); }, "same + per-clip XCD remap");
    time([&] { pattern_kernel<8, true, 0, true, true><<<grid, 256>>>(in, out); }, "depth 8, nt, nt stores spread");
    time([&] { pattern_kernel<4, true, 0, true, true><<<grid, 256>>>(in, out); }, "depth 4, nt, nt stores spread (again)");
    time([&] { pattern_kernel<2, true><<<grid, 256>>>(in, out); }, "esim pattern, depth 2, nt loads");
    time([&] { pattern_kernel<8, true><<<grid, 256>>>(in, out); }, "esim pattern, depth 8, nt loads");
    time([&] { pattern_kernel<4, true, 1><<<grid, 256>>>(in, out); }, "esim pattern, XCD-contiguous remap");
    time([&] { pattern_kernel<4, true, 2><<<grid, 256>>>(in, out); }, "esim pattern, per-clip XCD remap");
    time([&] { pattern_kernel<8, true, 1><<<grid, 256>>>(in, out); }, "depth 8, XCD-contiguous remap");
    time([&] { pattern8_kernel<2><<<B * (HW / 2048), 256>>>(in, out); }, "8 px per item (2 KiB per wave), depth 2");
    time([&] { pattern8_kernel<4><<<B * (HW / 2048), 256>>>(in, out); }, "8 px per item (2 KiB per wave), depth 4");
    for (int g : {1024, 2048, 4096}) {
        char nm[64]; snprintf(nm, sizeof nm, "linear x4 unrolled nt, %d blocks", g);
        time([&] { linear4_kernel<<<g, 256>>>(in, out, n_in / 4, n_out / 4); }, nm);
    }
    time([&] { linear_kernel<<<256 * 8, 256>>>(in, out, n_in / 4, n_out / 4); }, "linear read + write, 2048 blocks");
    time([&] { linear_kernel<<<256 * 32, 256>>>(in, out, n_in / 4, n_out / 4); }, "linear read + write, 8192 blocks");
    {   // ceilings at other read:write mixes (bytes counted = read + written)
        const size_t n4 = n_in / 8;                 // half of the input buffer copied into the other half: 1.07 GB each way
        auto time2 = [&](auto launch, const char *name, double by) {
            for (int i = 0; i < 3; ++i) launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            printf("%-34s %.4f ms  %.0f GB/s\n", name, ms, by / ms / 1e6);
        };
        for (int g : {2048, 8192})
            time2([&] { copy_kernel<<<g, 256>>>(in, in + n4 * 4, n4); }, g == 2048 ? "float4 copy 1:1, 2048 blocks" : "float4 copy 1:1, 8192 blocks", 2.0 * n4 * 16);
        time2([&] { hipMemcpyAsync(in + n4 * 4, in, n4 * 16, hipMemcpyDeviceToDevice, 0); }, "hipMemcpy D2D 1:1", 2.0 * n4 * 16);
        for (int g : {2048, 8192})
            time2([&] { read_kernel<<<g, 256>>>(in, out, n_in / 4); }, g == 2048 ? "read-only sweep, 2048 blocks" : "read-only sweep, 8192 blocks", (double)n_in * 4);
    }
    return 0;
}
