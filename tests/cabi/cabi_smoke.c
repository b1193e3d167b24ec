/* Plain-C client of the C ABI (include/v2v_hip.h): no Python, no torch -- what a cgo/JNI/ctypes binding would do.
 * Builds a deterministic uint8 clip, runs the fused ESIM kernel (rng NONE = no random fields) in SUM and BILINEAR
 * mode and prints FNV-1a checksums of the voxel bytes; tests/test_cabi_native.py recomputes the same with the oracle.
 * Compile: gcc cabi_smoke.c -I../../include -L../../v2v_amd -lv2v_hip -L/opt/rocm/lib -lamdhip64 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "v2v_hip.h"

/* minimal HIP runtime prototypes (avoid needing a C++ compiler for hip_runtime.h) */
typedef int hipError_t;
hipError_t hipMalloc(void **ptr, size_t size);
hipError_t hipFree(void *ptr);
hipError_t hipMemcpy(void *dst, const void *src, size_t size, int kind);
hipError_t hipDeviceSynchronize(void);
hipError_t hipMemset(void *dst, int value, size_t size);
enum { H2D = 1, D2H = 2 };

static uint64_t fnv1a(const void *p, size_t n)
{
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(void)
{
    enum { B = 2, N = 11, H = 24, W = 32, K = N - 1 };
    if (v2v_version() != V2V_ABI_VERSION || v2v_device_count() < 1) { fprintf(stderr, "no device / ABI mismatch\n"); return 2; }
    const size_t n_in = (size_t)B * N * H * W;
    uint8_t *clip = malloc(n_in);
    uint32_t x = 2463534242u;
    for (size_t i = 0; i < n_in; ++i) {                  /* smooth-ish walk so event counts stay small */
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        const size_t prev = i >= (size_t)H * W ? i - (size_t)H * W : i;
        int v = (i >= (size_t)H * W ? clip[prev] : (int)(x & 255)) + (int)(x >> 28) - 8;
        clip[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
    const double params[2][5] = {{0.2, 0.2, 0, 0, 0}, {0.15, 0.35, 0, 0, 0}};
    void *d_in, *d_par, *d_out, *d_cnt;
    const size_t sum_bytes = sizeof(float) * B * (K / 5) * 5 * H * W, bil_bytes = sizeof(double) * B * 5 * H * W;
    if (hipMalloc(&d_in, n_in) || hipMalloc(&d_par, sizeof(params)) || hipMalloc(&d_out, bil_bytes > sum_bytes ? bil_bytes : sum_bytes) ||
        hipMalloc(&d_cnt, sizeof(int64_t) * B * 2)) return 3;
    hipMemcpy(d_in, clip, n_in, H2D);
    hipMemcpy(d_par, params, sizeof(params), H2D);
    hipMemset(d_cnt, 0, sizeof(int64_t) * B * 2);
    int rc = v2v_esim_voxel_hip(d_in, V2V_U8, B, N, H, W, (int64_t)N * H * W, (int64_t)H * W, (const double *)d_par, 5,
                                V2V_FLAG_NO_NOISE, V2V_RNG_NONE, 0, 0, NULL, V2V_BIN_SUM, 5, 1, d_out, V2V_F32,
                                (int64_t *)d_cnt, NULL);
    if (rc) { fprintf(stderr, "sum: %d %s\n", rc, v2v_last_error()); return 4; }
    hipDeviceSynchronize();
    float *h_sum = malloc(sum_bytes);
    int64_t cnt[B * 2];
    hipMemcpy(h_sum, d_out, sum_bytes, D2H);
    hipMemcpy(cnt, d_cnt, sizeof(cnt), D2H);
    printf("sum %016llx counts %lld %lld %lld %lld\n", (unsigned long long)fnv1a(h_sum, sum_bytes), (long long)cnt[0],
           (long long)cnt[1], (long long)cnt[2], (long long)cnt[3]);
    rc = v2v_esim_voxel_hip(d_in, V2V_U8, B, N, H, W, (int64_t)N * H * W, (int64_t)H * W, (const double *)d_par, 5, 0,
                            V2V_RNG_NONE, 0, 0, NULL, V2V_BIN_BILINEAR, 5, 1, d_out, V2V_F64, NULL, NULL);
    if (rc) { fprintf(stderr, "bilinear: %d %s\n", rc, v2v_last_error()); return 5; }
    hipDeviceSynchronize();
    double *h_bil = malloc(bil_bytes);
    hipMemcpy(h_bil, d_out, bil_bytes, D2H);
    printf("bilinear %016llx\n", (unsigned long long)fnv1a(h_bil, bil_bytes));
    /* error path: the reference's assert (N-1) % (num_bins*frames_per_bin) */
    rc = v2v_esim_voxel_hip(d_in, V2V_U8, B, N, H, W, (int64_t)N * H * W, (int64_t)H * W, (const double *)d_par, 5, 0,
                            V2V_RNG_NONE, 0, 0, NULL, V2V_BIN_SUM, 3, 1, d_out, V2V_F32, NULL, NULL);
    printf("bins_error %d\n", rc);
    FILE *f = fopen("cabi_clip.bin", "wb");
    if (f) { fwrite(clip, 1, n_in, f); fclose(f); }
    hipFree(d_in); hipFree(d_par); hipFree(d_out); hipFree(d_cnt);
    free(clip); free(h_sum); free(h_bil);
    return 0;
}
