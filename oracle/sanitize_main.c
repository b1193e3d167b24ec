/* Sanitizer driver for the scalar C oracle (CPU build only; GPU sanitizers are unavailable on this pool).
 * Built with -fsanitize=address,undefined by `make -C oracle sanitize` and run by tests/test_oracle_sanitize.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "v2v_oracle.c"

int main(void)
{
    enum { N = 9, H = 12, W = 10, HW = H * W, K = N - 1 };
    uint8_t *v8 = malloc(N * HW);
    float *v32 = malloc(sizeof(float) * N * HW);
    double lut64[256]; float lut32[256], lutv[256];
    for (int i = 0; i < 256; ++i) { lut64[i] = log(0.001 + pow(i / 255.0, 2.2)); lut32[i] = (float)lut64[i]; lutv[i] = (float)log(i / 255.0 + 0.01); }
    uint32_t x = 12345u;
    for (int i = 0; i < N * HW; ++i) { x = x * 1664525u + 1013904223u; v8[i] = (uint8_t)(x >> 24); v32[i] = (float)v8[i]; }
    const double params[5] = {0.2, 0.3, 0.05, 0.05, 0.7};
    double *out = malloc(sizeof(double) * K * HW);
    int64_t totals[2] = {0, 0};
    int rc = 0;
    rc |= oracle_esim_voxel_clip(v8, ORACLE_IN_U8, N, HW, lut64, lut32, params, 0, ORACLE_RNG_PHILOX, 77, 3, NULL, ORACLE_BIN_SUM, 4, 2, out, totals);
    rc |= oracle_esim_voxel_clip(v32, ORACLE_IN_F32, N, HW, lut64, lut32, params, 1, ORACLE_RNG_PHILOX, 77, 3, NULL, ORACLE_BIN_BILINEAR, 5, 1, out, totals);
    rc |= oracle_esim_voxel_batch(v8, ORACLE_IN_U8, 1, N, HW, lut64, lut32, params, 0, 0, ORACLE_RNG_NONE, 1, 0, ORACLE_BIN_SUM, 8, 1, out, totals);
    oracle_v2e_params P = {24.0, V2E_SPATIAL_TEMPORAL_INDEPENDENT, 0.5, 0.1, 0.0, 0.1, 30.0, 0.1, 1.0 / 240, 5.0, 0.1, 0.1, 1};
    rc |= oracle_v2e_voxel_clip(v8, ORACLE_IN_U8, N, HW, lutv, &P, ORACLE_RNG_PHILOX, 5, 1, NULL, ORACLE_BIN_SUM, 8, 1, out, totals);
    P.threshold_model = V2E_PN_RELATED;
    rc |= oracle_v2e_voxel_clip(v32, ORACLE_IN_F32, N, HW, lutv, &P, ORACLE_RNG_PHILOX, 5, 1, NULL, ORACLE_BIN_BILINEAR, 5, 1, out, totals);
    int64_t ts[5] = {0, 10, 20, 30, 40}, xs[5] = {0, 1, 2, 3, 9}, ys[5] = {0, 1, 2, 3, 11};
    int8_t ps[5] = {0, 1, 1, 0, 1};
    double vox[5 * HW];
    rc |= oracle_make_voxel(ts, xs, ys, ps, 5, 5, H, W, 0, vox);
    rc |= oracle_make_voxel(ts, xs, ys, ps, 5, 5, H, W, 1, vox);
    rc |= oracle_make_voxel(ts, xs, ys, ps, 0, 5, H, W, 1, vox);
    double q = oracle_floor_divide(24.550921417593624, 1.2275460708796813);
    printf("sanitize ok rc=%d q=%g totals=%lld/%lld\n", rc, q, (long long)totals[0], (long long)totals[1]);
    free(v8); free(v32); free(out);
    return rc != 0 || q != 19.0;
}
