/*
 * oracle/v2v_oracle.c -- scalar C restatement of the V2V video->voxel hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Built into oracle/libv2v_oracle.so by oracle/Makefile and loaded
 * (ctypes, oracle/clib.py) only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  The product (v2v_amd/, libv2v_hip.so) never links, loads or calls it.
 *
 * Parity status: PINNED against the golden vectors in tests/golden/ (captured from the imported
 * reference; see tests/golden/make_goldens.py) and against oracle/v2v_oracle.py.
 *
 * One pixel at a time, one frame pair at a time -- the obviously-correct form of
 *   data/v2v_core_esim.py:26-69   (ESIM frame-pair simulator)
 *   data/v2v_datasets.py:363-410  (sum binning of per-pair counts)
 *   utils/event_utils.py:692-728  (temporal-bilinear weights)
 *   data/testh5.py:60-90          (make_voxel event-list voxeliser)
 * np.floor_divide is third-party arithmetic (numpy 2.2.6 npy_divmod); restated below.
 *
 * Compile with -ffp-contract=off: the reference never fuses a multiply with an add.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* threads the batch drivers below run on (libgomp: OMP_NUM_THREADS, else the cores of the affinity mask): bench.py reports this as `cores` */
int oracle_omp_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_omp_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ np.floor_divide (float64) */
double oracle_floor_divide(double a, double b)
{
    if (b == 0.0) return a / b;
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0) {
        if ((b < 0) != (mod < 0)) { mod += b; div -= 1.0; }
    }
    double fl;
    if (div != 0.0) {
        fl = floor(div);
        if (div - fl > 0.5) fl += 1.0;
    } else {
        fl = copysign(0.0, a / b);
    }
    return fl;
}

void oracle_floor_divide_vec(const double *a, const double *b, double *q, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) q[i] = oracle_floor_divide(a[i], b[i]);
}

/* ------------------------------------------------------------------ Philox4x32-10 */
#define ORACLE_NOISE_ROUNDS 7   /* per-time-step noise fields (base noise, leak jitter, shot uniforms); all other fields: 10 */
static void philox4x32_r(uint32_t c[4], uint32_t k0, uint32_t k1, int rounds)
{
    for (int r = 0; r < rounds; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) { philox4x32_r(c, k0, k1, 10); }

void oracle_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
    philox4x32_10(c, key[0], key[1]);
    memcpy(out, c, sizeof(c));
}

/* 53-bit uniform, numpy legacy recipe (random_sample): (a>>5, b>>6) */
static double uniform53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

static float u32_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t float_as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/*
 * Device-native normals: TWO float32 deviates per 32-bit Philox word, one per 16-bit half, by direct table inversion.
 * Not taken from the reference (which uses numpy's MT19937 polar method, impossible to replay at bandwidth on a GPU); this is
 * the definition of the device-native noise fields, restated here independently of v2v_amd/csrc/v2v_rng.hpp.
 *   half-word n: sign = n >> 15, magnitude index i = (n >> 2) & 0x1FFF  <->  probability 1/2 + (i + 1/2) / 2^14 (midpoint grid;
 *   the two low bits are unused);  deviate = Phi^-1 of it = table[i], sign applied
 * gauss_icdf.inc is DATA generated once (tools/gen_gauss_icdf.py, scipy.special.ndtri) and committed; the device holds the
 * same text.  g0 comes from the high, g1 from the low half-word.
 */
static const float gauss_icdf_tab[8192] = {
#include "gauss_icdf.inc"
};

static float icdf14(uint32_t n)
{
    return u32_as_float(float_as_u32(gauss_icdf_tab[(n >> 2) & 0x1FFFu]) ^ ((n & 0x8000u) << 16));
}

static void gauss16_pair(uint32_t w, float *g0, float *g1)
{
    *g0 = icdf14(w >> 16);
    *g1 = icdf14(w & 0xFFFFu);
}

void oracle_gauss16(uint32_t w, float out[2]) { gauss16_pair(w, &out[0], &out[1]); }

/* many words at once: exhaustive accuracy checks in the tests */
void oracle_gauss16_many(const uint32_t *w, int64_t n, float *g0, float *g1)
{
    for (int64_t i = 0; i < n; ++i) gauss16_pair(w[i], &g0[i], &g1[i]);
}

static double px_uniform53(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream, uint32_t p)
{
    uint32_t c[4] = {p >> 1, field, clip, stream};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (p & 1u) ? uniform53(c[2], c[3]) : uniform53(c[0], c[1]);
}

/* normal `comp` (0/1) of pixel p's table-inversion deviate pair in block `field`: word p&3 of Philox block (p>>2, field, clip, stream) */
static float px_gauss_r(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream, uint32_t p, int comp, int rounds)
{
    uint32_t c[4] = {p >> 2, field, clip, stream};
    philox4x32_r(c, (uint32_t)seed, (uint32_t)(seed >> 32), rounds);
    float g0, g1;
    gauss16_pair(c[p & 3u], &g0, &g1);
    return comp ? g1 : g0;
}
static float px_gauss(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream, uint32_t p, int comp)
{
    return px_gauss_r(seed, clip, field, stream, p, comp, 10);
}

void oracle_philox_uniform_field(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream,
                                 int64_t n_pix, double *out)
{
    for (int64_t p = 0; p < n_pix; ++p) out[p] = px_uniform53(seed, clip, field, stream, (uint32_t)p);
}

void oracle_philox_gauss_field(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream,
                               int64_t n_pix, int comp, int rounds, float *out)
{
    for (int64_t p = 0; p < n_pix; ++p) out[p] = px_gauss_r(seed, clip, field, stream, (uint32_t)p, comp, rounds);
}
int oracle_noise_rounds(void) { return ORACLE_NOISE_ROUNDS; }

/* ------------------------------------------------------------------ ESIM + binning, one clip */
enum { ORACLE_IN_U8 = 0, ORACLE_IN_F32 = 1 };
enum { ORACLE_RNG_NONE = 0, ORACLE_RNG_PHILOX = 1, ORACLE_RNG_REPLAY = 2 };
enum { ORACLE_BIN_SUM = 0, ORACLE_BIN_BILINEAR = 1 };

typedef struct {
    const double *u_init;  /* [H*W]   rand  #1  (v2v_core_esim.py:29) */
    const double *u_hot;   /* [H*W]   rand  #2  (:37) */
    const double *g_hot;   /* [H*W]   randn #1  (:38) */
    const double *g_base;  /* [K,H*W] randn per pair (:44) */
} oracle_replay;

/* temporal-bilinear weight of pair k for bin b; same float64 expression as event_utils.py:715-719 */
static double bil_w(int64_t k, int64_t K, int b, int Tb)
{
    double t_norm = ((double)k - 0.0) / ((double)(K - 1) - 0.0) * (double)(Tb - 1);
    double w = 1.0 - fabs(t_norm - (double)b);
    return w > 0.0 ? w : 0.0;
}

/*
 * frames: one clip [N,H*W] (u8, or float32 holding integers 0..255).
 * lut64/lut32: golden G1 tables.  params: {pos,neg,base_std,hot_frac,hot_std}.
 * rng NONE: deterministic debugging mode: u_init = 0.5, never hot, all Gaussians 0.
 * out: SUM -> [L,Tb,H*W] float64 ; BILINEAR -> [Tb,H*W] float64.  totals[2] += ON, OFF event totals.
 */
int oracle_esim_voxel_clip(const void *frames, int in_dtype, int64_t N, int64_t HW,
                           const double *lut64, const float *lut32, const double params[5],
                           int noise_external, int rng_mode, uint64_t seed, uint32_t clip_id,
                           const oracle_replay *rp, int bin_mode, int Tb, int fpb,
                           double *out, int64_t *totals)
{
    const int64_t K = N - 1;
    if (K < 1 || Tb < 1 || fpb < 1) return -1;
    if (bin_mode == ORACLE_BIN_SUM && (K % ((int64_t)Tb * fpb)) != 0) return -2;
    if (bin_mode == ORACLE_BIN_BILINEAR && K < 2) return -3;
    const double pos = params[0], neg = params[1], base_std = params[2], hot_frac = params[3], hot_std = params[4];
    const uint8_t *f8 = (const uint8_t *)frames;
    const float *f32 = (const float *)frames;
    const int64_t n_out = (bin_mode == ORACLE_BIN_SUM) ? K / fpb : Tb;
    int64_t on_total = 0, off_total = 0;

    for (int64_t p = 0; p < HW; ++p) {
        double u0 = 0.5, u1 = 1.0, gh = 0.0;
        if (rng_mode == ORACLE_RNG_PHILOX) {
            u0 = px_uniform53(seed, clip_id, 0, 0, (uint32_t)p);
            if (hot_frac > 0.0) {          /* u >= 0: no hot pixel can exist otherwise */
                u1 = px_uniform53(seed, clip_id, 1, 0, (uint32_t)p);
                gh = (double)px_gauss(seed, clip_id, 2, 0, (uint32_t)p, 0);
            }
        } else if (rng_mode == ORACLE_RNG_REPLAY) {
            u0 = rp->u_init[p]; u1 = rp->u_hot[p]; gh = rp->g_hot[p];
        }
        double pot = u0 * (pos + neg) - neg;                       /* :29 */
        double hot = (u1 < hot_frac) ? hot_std * gh : 0.0;         /* :37-39 */
        for (int64_t o = 0; o < n_out; ++o) out[o * HW + p] = 0.0;

        for (int64_t k = 0; k < K; ++k) {
            if (in_dtype == ORACLE_IN_U8) {
                double d = lut64[f8[(k + 1) * HW + p]] - lut64[f8[k * HW + p]];
                pot += d;
            } else {
                float a = f32[k * HW + p], b = f32[(k + 1) * HW + p];
                float d = lut32[(int)b] - lut32[(int)a];           /* float32 subtraction */
                pot += (double)d;
            }
            double g = 0.0;
            if (rng_mode == ORACLE_RNG_PHILOX) { if (base_std != 0.0) g = (double)px_gauss_r(seed, clip_id, 3u + (uint32_t)(k >> 1), 0, (uint32_t)p, (int)(k & 1), ORACLE_NOISE_ROUNDS); }   /* 0*g adds nothing; block 3+m serves pairs 2m, 2m+1 */
            else if (rng_mode == ORACLE_RNG_REPLAY) g = rp->g_base[k * HW + p];
            double base = base_std * g;                            /* :44 */
            if (!noise_external) { pot += base; pot += hot; }      /* :48-49 */
            double on = 0.0, off = 0.0;
            if (pot >= pos) on = oracle_floor_divide(pot, pos);    /* :51-52 */
            if (pot <= -neg) off = oracle_floor_divide(-pot, neg); /* :54-55 */
            pot -= on * pos;                                       /* :57 */
            pot += off * neg;                                      /* :58 */
            double vox = on - off;
            if (noise_external) { vox = vox + base; vox = vox + hot; }
            on_total += (int64_t)on; off_total += (int64_t)off;
            if (bin_mode == ORACLE_BIN_SUM) {
                out[(k / fpb) * HW + p] += vox;                    /* [L,Tb] flattened == k / fpb */
            } else {
                for (int b = 0; b < Tb; ++b) {
                    double w = bil_w(k, K, b, Tb);
                    double contrib = vox * w;
                    out[(int64_t)b * HW + p] += contrib;
                }
            }
        }
    }
    if (totals) { totals[0] += on_total; totals[1] += off_total; }
    return 0;
}

/* Batch driver (OpenMP over clips when built with -fopenmp). frames [B,N,HW]; params [B,5] or stride 0. */
int oracle_esim_voxel_batch(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t HW,
                            const double *lut64, const float *lut32, const double *params,
                            int64_t params_stride, int noise_external, int rng_mode, uint64_t seed,
                            uint64_t clip_id0, int bin_mode, int Tb, int fpb, double *out,
                            int64_t *totals /* [B,2] or NULL */)
{
    const int64_t K = N - 1;
    const int64_t out_per_clip = ((bin_mode == ORACLE_BIN_SUM) ? K / fpb : Tb) * HW;
    const int64_t in_per_clip = N * HW * (in_dtype == ORACLE_IN_U8 ? 1 : 4);
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t c = 0; c < B; ++c) {
        int64_t t[2] = {0, 0};
        int r = oracle_esim_voxel_clip((const char *)frames + c * in_per_clip, in_dtype, N, HW, lut64, lut32,
                                       params + c * params_stride, noise_external, rng_mode, seed,
                                       (uint32_t)(clip_id0 + (uint64_t)c), NULL, bin_mode, Tb, fpb,
                                       out + c * out_per_clip, t);
        if (totals) { totals[2 * c] = t[0]; totals[2 * c + 1] = t[1]; }
        if (r != 0) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

/* ------------------------------------------------------------------ v2e model (data/v2v_core_v2e.py) */
/* exp(-lam), lam >= 0, from IEEE-exact primitives only (same definition on the device): used by the native
 * Poisson sampler, which is NOT NumPy's (NumPy's consumes a data-dependent number of MT19937 draws). */
static double exp_neg(double lam)
{
    const double x = -lam;
    if (x < -745.0) return 0.0;
    const double k = rint(x * 1.4426950408889634);
    double r = fma(-k, 0.693147180369123816490e+00, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;                 /* 1/13! */
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const int ki = (int)k;                          /* -1075 .. 0 */
    if (ki < -1022) return 0.0;
    uint64_t bits = (uint64_t)(1023 + ki) << 52;
    double scale; memcpy(&scale, &bits, 8);
    return p * scale;
}
double oracle_exp_neg(double lam) { return exp_neg(lam); }

/* Poisson by inversion from one 53-bit uniform (native mode). */
static double poisson_inv(double lam, double u)
{
    if (!(lam > 0.0)) return 0.0;
    double p = exp_neg(lam), s = p, x = 0.0;
    while (u > s && x < 1000.0) { x += 1.0; p = p * lam / x; s += p; }
    return x;
}
double oracle_poisson_inv(double lam, double u) { return poisson_inv(lam, u); }

/* expf for the log-normal noise-rate array (native mode; NumPy's float32 exp is a SIMD kernel we cannot match
 * bit for bit, so native mode defines its own from IEEE-exact primitives). */
static float expf_det(float x)
{
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    const float k = rintf(x * 1.44269502f);
    float r = fmaf(-k, 0.693359375f, x);
    r = fmaf(-k, -2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    p = fmaf(p, r2, r) + 1.0f;
    return p * u32_as_float((uint32_t)(127 + (int)k) << 23);
}
float oracle_expf_det(float x) { return expf_det(x); }

/* Native shot-noise uniforms: word p&3 of Philox block (p>>2, field) -> ON uniform from its high, OFF uniform from its
 * low 16 bits, midpoint grid (n + 1/2)/2^16. */
static float px_uniform16(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream, uint32_t p, int low)
{
    uint32_t c[4] = {p >> 2, field, clip, stream};
    philox4x32_r(c, (uint32_t)seed, (uint32_t)(seed >> 32), ORACLE_NOISE_ROUNDS);
    const uint32_t w = c[p & 3u];
    return ((float)(low ? (w & 0xFFFFu) : (w >> 16)) + 0.5f) * 1.52587890625e-05f;
}

/* exp(-lam) for the shot-noise sampler: degree-6 minimax polynomial on [0,1] (1.5e-8; exactly 1 at 0), expf_det above */
static float exp_neg_f32(float lam)
{
    if (lam > 1.0f) return expf_det(-lam);
    float e = 0x1.be1ddep-11f;
    e = fmaf(e, lam, -0x1.f60198p-8f);
    e = fmaf(e, lam, 0x1.51c0fcp-5f);
    e = fmaf(e, lam, -0x1.5507a6p-3f);
    e = fmaf(e, lam, 0x1.fff9acp-2f);
    e = fmaf(e, lam, -0x1.ffffcep-1f);
    e = fmaf(e, lam, 1.0f);
    return e;
}

/* Native Poisson sampler: float32 inversion from one uniform.  Counts 0..3 from the thresholds p0, p0(1+l), p0(1+l+l^2/2)
 * (p0 = exp(-l)); beyond that the running-sum loop.  Same sequence as v2v_amd/csrc/v2v_v2e.hpp, restated. */
static float poisson_inv_f32(float lam, float u)
{
    if (!(lam > 0.0f)) return 0.0f;               /* also NaN; on the device p0 = 1 or NaN gives the same 0 */
    const float p0 = exp_neg_f32(lam);
    const float hl = lam * 0.5f;
    const float q1 = lam + 1.0f;
    const float q2 = fmaf(hl, lam, q1);
    const float s1 = p0 * q1, s2 = p0 * q2;
    float x = (float)((u > p0) + (u > s1) + (u > s2));
    if (u > s2) {
        float p = p0 * (hl * lam), s = s2;
        x = 2.0f;
        do { x += 1.0f; p = p * (lam / x); s = s + p; } while (u > s && x < 64.0f);
    }
    return x;
}
float oracle_poisson_inv_f32(float lam, float u) { return poisson_inv_f32(lam, u); }

/* vector forms for the golden generator (tests/golden/make_goldens.py G14) and the statistical tests */
void oracle_expf_det_vec(const float *x, int64_t n, float *out) { for (int64_t i = 0; i < n; ++i) out[i] = expf_det(x[i]); }
void oracle_poisson_inv_f32_vec(const float *lam, const float *u, int64_t n, int64_t *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = (int64_t)poisson_inv_f32(lam[i], u[i]);
}
void oracle_philox_uniform16_field(uint64_t seed, uint32_t clip, uint32_t field, uint32_t stream, int64_t n_pix, int low, float *out)
{
    for (int64_t p = 0; p < n_pix; ++p) out[p] = px_uniform16(seed, clip, field, stream, (uint32_t)p, low);
}

enum { V2E_PN_RELATED = 0, V2E_SPATIAL_INDEPENDENT = 1, V2E_SPATIAL_TEMPORAL_INDEPENDENT = 2 };
enum { V2E_F_THRES_A = 0, V2E_F_NOISE_RATE = 2, V2E_F_FRAME0 = 16, V2E_F_STRIDE = 8 };
#define V2E_STREAM 1u

typedef struct {
    double fps;
    int threshold_model;
    double thres_mean_mean, thres_mean_std, thres_diff_mean, thres_diff_std;
    double cutoff_hz, leak_rate_hz, refractory_period_s, shot_noise_rate_hz, leak_jitter_fraction, noise_rate_cov_decades;
    int uint8_wrap;
} oracle_v2e_params;

typedef struct {
    const double *pos_thres, *neg_thres;   /* [HW] or [K,HW] (frame stride below) after clipping */
    int64_t thres_frame_stride;
    const float *noise_rate;               /* [HW] */
    const double *leak_randn;              /* [K,HW] or NULL */
    const int64_t *shot_pos, *shot_neg;    /* [K,HW] or NULL */
} oracle_v2e_replay;

static double v2e_floor_divide(double a, double b) { return oracle_floor_divide(a, b); }

/* thresholds of pixel p for frame field base `fb` (static: V2E_F_THRES_A) -- native mode */
static void v2e_native_thres(const oracle_v2e_params *P, uint64_t seed, uint32_t clip, uint32_t fa, uint32_t p,
                             double *pt, double *nt)
{
    const double ga = (double)px_gauss(seed, clip, fa, V2E_STREAM, p, 0);   /* the two deviates of the pixel's word */
    const double gb = (double)px_gauss(seed, clip, fa, V2E_STREAM, p, 1);
    double a, b;
    if (P->threshold_model == V2E_PN_RELATED) {
        const double mean = P->thres_mean_mean + P->thres_mean_std * ga;   /* normal(loc,scale) = loc + scale*g */
        const double diff = P->thres_diff_mean + P->thres_diff_std * gb;
        a = mean + (diff / 2); b = mean - (diff / 2);
    } else {
        a = P->thres_mean_mean + P->thres_mean_std * ga;
        b = P->thres_mean_mean + P->thres_mean_std * gb;
    }
    *pt = a < 0.01 ? 0.01 : a;             /* np.clip(a_min=0.01) */
    *nt = b < 0.01 ? 0.01 : b;
}

static double v2e_inten01_u8(uint8_t x, int wrap)
{
    return wrap ? (double)(uint8_t)(x + 20) / 275. : ((double)x + 20.0) / 275.;
}

/* Q(v) = rint(v * 2^20) clamped to +-(2^31 - 128), NaN -> 0: the fixed-point factors of the native shot-noise frame sum */
static int32_t shot_quant(double v)
{
    double s = v * 1048576.0;
    s = s > 2147483520.0 ? 2147483520.0 : s;
    s = s < -2147483520.0 ? -2147483520.0 : s;
    return s == s ? (int32_t)llrint(s) : 0;
}
/* the device keeps {low 32 bits, arithmetic high part} of the sum in two 64-bit accumulators and recombines them in float64 */
static double shot_sum_value(__int128 s)
{
    const int64_t hi = (int64_t)(s >> 32);
    const uint64_t lo = (uint64_t)(s & (__int128)0xFFFFFFFFu);
    return (double)hi * 4294967296.0 + (double)lo;
}

/*
 * One clip.  frames [N,HW] u8 or f32 (integer-valued).  lut = golden G1 v2e32.  rng: PHILOX or REPLAY.
 * out: SUM -> [K/fpb, HW] ; BILINEAR -> [Tb, HW] (float64).  Follows generate_events (v2v_core_v2e.py:401-553)
 * pixel by pixel with NumPy's dtype promotions made explicit (lp32 / base32 flags, see oracle/v2v_oracle.py).
 */
int oracle_v2e_voxel_clip(const void *frames, int in_dtype, int64_t N, int64_t HW, const float *lut,
                          const oracle_v2e_params *P, int rng_mode, uint64_t seed, uint32_t clip_id,
                          const oracle_v2e_replay *rp, int bin_mode, int Tb, int fpb, double *out, int64_t *totals)
{
    const int64_t K = N - 1;
    if (K < 1 || Tb < 1 || fpb < 1) return -1;
    if (bin_mode == ORACLE_BIN_SUM && (K % ((int64_t)Tb * fpb)) != 0) return -2;
    if (bin_mode == ORACLE_BIN_BILINEAR && K < 2) return -3;
    if (rng_mode != ORACLE_RNG_PHILOX && rng_mode != ORACLE_RNG_REPLAY) return -4;
    const uint8_t *f8 = (const uint8_t *)frames;
    const float *f32 = (const float *)frames;
    const int in_f32 = in_dtype == ORACLE_IN_F32;
    const int lp32 = (P->cutoff_hz <= 0) || in_f32;
    const int base32 = lp32 && !(P->leak_rate_hz > 0);
    const int temporal = P->threshold_model == V2E_SPATIAL_TEMPORAL_INDEPENDENT;
    const double pos_nominal = P->thres_mean_mean + P->thres_diff_mean / 2;
    const double neg_nominal = P->thres_mean_mean - P->thres_diff_mean / 2;
    const double tau = P->cutoff_hz > 0 ? 1 / (M_PI * 2 * P->cutoff_hz) : 0.0;
    const int64_t n_out = (bin_mode == ORACLE_BIN_SUM) ? K / fpb : Tb;
    const int shot = P->shot_noise_rate_hz > 0;
    int64_t on_total = 0, off_total = 0;

    /* native shot noise: per-frame INTEGER sums of Q(intensity factor) * Q(nominal / threshold), Q(v) = rint(v * 2^20) clamped
     * (v2v_amd/csrc/v2v_v2e.hpp: shot_quant) -- exact, hence free of any summation order; 128-bit accumulators here, split at
     * bit 32 and recombined in float64 exactly as the device does (one rounding) */
    __int128 *sum_pos = NULL, *sum_neg = NULL;
    if (shot && rng_mode == ORACLE_RNG_PHILOX) {
        sum_pos = (__int128 *)calloc((size_t)K, sizeof(__int128));
        sum_neg = (__int128 *)calloc((size_t)K, sizeof(__int128));
        for (int64_t p = 0; p < HW; ++p) {
            double pt, nt;
            v2e_native_thres(P, seed, clip_id, V2E_F_THRES_A, (uint32_t)p, &pt, &nt);
            for (int64_t k = 0; k < K; ++k) {
                const int64_t i = k + 1;
                if (temporal) v2e_native_thres(P, seed, clip_id, (uint32_t)(V2E_F_FRAME0 + V2E_F_STRIDE * i), (uint32_t)p, &pt, &nt);
                double fac;
                if (in_f32) { const float i01 = (f32[i * HW + p] + 20.0f) / 275.0f; fac = (double)(1.0f - 0.75f * i01); }
                else fac = 1 - 0.75 * v2e_inten01_u8(f8[i * HW + p], P->uint8_wrap);
                const int64_t q = shot_quant(fac);
                sum_pos[k] += (__int128)(q * (int64_t)shot_quant(pos_nominal / pt));
                sum_neg[k] += (__int128)(q * (int64_t)shot_quant(neg_nominal / nt));
            }
        }
    }

    for (int64_t p = 0; p < HW; ++p) {
        for (int64_t o = 0; o < n_out; ++o) out[o * HW + p] = 0.0;
        /* frame 0: lp = base = lin_log(frame0) (the low-pass update with delta_time = 0 is the identity) */
        const int x0 = in_f32 ? (int)f32[p] : (int)f8[p];
        double lp64 = (double)lut[x0], base64 = lp64;
        float lp_f = lut[x0], base_f = lp_f;
        double pt, nt;
        float nrate;
        if (rng_mode == ORACLE_RNG_PHILOX) {
            v2e_native_thres(P, seed, clip_id, V2E_F_THRES_A, (uint32_t)p, &pt, &nt);
            const float g = px_gauss(seed, clip_id, V2E_F_NOISE_RATE, V2E_STREAM, (uint32_t)p, 0);
            nrate = expf_det((float)(2.302585092994046 * P->noise_rate_cov_decades) * g);
        } else {
            pt = rp->pos_thres[p]; nt = rp->neg_thres[p]; nrate = rp->noise_rate[p];
        }
        for (int64_t k = 0; k < K; ++k) {
            const int64_t i = k + 1;
            const double dt = (double)i / P->fps - (double)(i - 1) / P->fps;      /* t_frame - t_previous */
            if (temporal) {
                if (rng_mode == ORACLE_RNG_PHILOX) v2e_native_thres(P, seed, clip_id, (uint32_t)(V2E_F_FRAME0 + V2E_F_STRIDE * i), (uint32_t)p, &pt, &nt);
                else { pt = rp->pos_thres[k * rp->thres_frame_stride + p]; nt = rp->neg_thres[k * rp->thres_frame_stride + p]; }
            }
            const float log_new = in_f32 ? lut[(int)f32[i * HW + p]] : lut[f8[i * HW + p]];
            double i01_64 = 0.0; float i01_32 = 0.0f;
            if (in_f32) i01_32 = (f32[i * HW + p] + 20.0f) / 275.0f;
            else i01_64 = v2e_inten01_u8(f8[i * HW + p], P->uint8_wrap);
            if (P->cutoff_hz > 0) {                                               /* low_pass_filter */
                if (in_f32) {
                    float eps = i01_32 * (float)(dt / tau);
                    if (eps > 1.0f) eps = 1.0f;
                    const float a = (1.0f - eps) * lp_f, b = eps * log_new;
                    lp_f = a + b;
                } else {
                    double eps = i01_64 * (dt / tau);
                    if (eps > 1.0) eps = 1.0;
                    const double a = (1 - eps) * lp64, b = eps * (double)log_new;
                    lp64 = a + b;
                }
            } else {
                lp_f = log_new;
            }
            if (P->leak_rate_hz > 0) {                                            /* subtract_leak_current */
                double g;
                if (rng_mode == ORACLE_RNG_PHILOX) g = (double)px_gauss_r(seed, clip_id, (uint32_t)(V2E_F_FRAME0 + V2E_F_STRIDE * (k >> 1) + 2), V2E_STREAM, (uint32_t)p, (int)(k & 1), ORACLE_NOISE_ROUNDS);
                else g = rp->leak_randn[k * HW + p];
                const float a32 = (float)P->leak_rate_hz * nrate;
                const double curr = (double)a32 * (1 - P->leak_jitter_fraction * g);
                const double dl = dt * curr * pt;
                base64 = base64 - dl;                                             /* base is float64 whenever leak > 0 */
            }
            double diff;
            if (lp32 && base32) { const float d = lp_f - base_f; diff = (double)d; }
            else diff = (lp32 ? (double)lp_f : lp64) - base64;
            const double pos_frame = diff > 0 ? diff : (diff == diff ? 0.0 : diff);
            const double nd = -diff;
            const double neg_frame = nd > 0 ? nd : (nd == nd ? 0.0 : nd);
            double fpos = v2e_floor_divide(pos_frame, pt);
            double fneg = v2e_floor_divide(neg_frame, nt);
            if (shot) {
                double sp, sn;
                if (rng_mode == ORACLE_RNG_PHILOX) {
                    double fac;
                    if (in_f32) fac = (double)(1.0f - 0.75f * i01_32); else fac = 1 - 0.75 * i01_64;
                    /* lambda = (intensity factor * threshold factor) * (rate/2 * dt / frame mean), all float32 (the frame
                     * mean itself comes from the exact integer sums above); float32 inversion */
                    const double mean_p = (shot_sum_value(sum_pos[k]) / 1099511627776.0) / (double)HW;
                    const double mean_n = (shot_sum_value(sum_neg[k]) / 1099511627776.0) / (double)HW;
                    const double f = (P->shot_noise_rate_hz / 2) * dt;
                    /* per pixel: intensity factor x float32 reciprocal threshold (biased low by 2^-22: the same value
                     * that estimates the floor-divide quotient on the device); per frame: nominal threshold x rate scale */
                    const float scale_p = (float)(f / mean_p) * (float)pos_nominal, scale_n = (float)(f / mean_n) * (float)neg_nominal;
                    const float fac32 = (float)fac;
                    const float inv_p = (float)((1.0 / pt) * 0x1.fffff8p-1), inv_n = (float)((1.0 / nt) * 0x1.fffff8p-1);
                    const float lam_p = (fac32 * inv_p) * scale_p;
                    const float lam_n = (fac32 * inv_n) * scale_n;
                    sp = (double)poisson_inv_f32(lam_p, px_uniform16(seed, clip_id, (uint32_t)(V2E_F_FRAME0 + V2E_F_STRIDE * i + 3), V2E_STREAM, (uint32_t)p, 0));
                    sn = (double)poisson_inv_f32(lam_n, px_uniform16(seed, clip_id, (uint32_t)(V2E_F_FRAME0 + V2E_F_STRIDE * i + 3), V2E_STREAM, (uint32_t)p, 1));
                } else {
                    sp = (double)rp->shot_pos[k * HW + p]; sn = (double)rp->shot_neg[k * HW + p];
                }
                fpos = fpos + sp; fneg = fneg + sn;
            }
            if (P->refractory_period_s > 0) {
                const double cap = (double)(int)(dt / P->refractory_period_s);
                if (fpos > cap) fpos = cap;
                if (fneg > cap) fneg = cap;
            }
            if (base32) {                                                          /* in-place += on a float32 array */
                base_f = (float)((double)base_f + fpos * pt);
                base_f = (float)((double)base_f - fneg * nt);
            } else {
                base64 = base64 + fpos * pt;
                base64 = base64 - fneg * nt;
            }
            const double vox = fpos - fneg;
            on_total += (int64_t)fpos; off_total += (int64_t)fneg;
            if (bin_mode == ORACLE_BIN_SUM) out[(k / fpb) * HW + p] += vox;
            else for (int b = 0; b < Tb; ++b) { const double c = vox * bil_w(k, K, b, Tb); out[(int64_t)b * HW + p] += c; }
        }
    }
    free(sum_pos); free(sum_neg);
    if (totals) { totals[0] += on_total; totals[1] += off_total; }
    return 0;
}

int oracle_v2e_voxel_batch(const void *frames, int in_dtype, int64_t B, int64_t N, int64_t HW, const float *lut,
                           const oracle_v2e_params *P, uint64_t seed, uint64_t clip_id0, int bin_mode, int Tb, int fpb,
                           double *out, int64_t *totals)
{
    const int64_t K = N - 1;
    const int64_t out_per_clip = ((bin_mode == ORACLE_BIN_SUM) ? K / fpb : Tb) * HW;
    const int64_t in_per_clip = N * HW * (in_dtype == ORACLE_IN_U8 ? 1 : 4);
    int rc = 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t c = 0; c < B; ++c) {
        int64_t t[2] = {0, 0};
        int r = oracle_v2e_voxel_clip((const char *)frames + c * in_per_clip, in_dtype, N, HW, lut, P, ORACLE_RNG_PHILOX,
                                      seed, (uint32_t)(clip_id0 + (uint64_t)c), NULL, bin_mode, Tb, fpb,
                                      out + c * out_per_clip, t);
        if (totals) { totals[2 * c] = t[0]; totals[2 * c + 1] = t[1]; }
        if (r != 0) {
#pragma omp critical
            rc = r;
        }
    }
    return rc;
}

/* ------------------------------------------------------------------ bgr_to_gray (data/v2v_datasets.py:19-22) */
/* gray = np.dot(img[..., :3], [0.5870, 0.1140, 0.2989]).astype(uint8).  For the 3-D / 4-D stacks the reference passes, NumPy
 * 2.2.6 (OpenBLAS 0.3.29 ddot: sequential FMA accumulation over the 3 channels) evaluates
 *     fma(r, w2, fma(g, w1, b * w0))            -- b, g, r = channels 0, 1, 2
 * which golden G15 (the reference's own function on all 2^24 colours) pins bit for bit.  (A 2-D [M,3] array goes through
 * dgemv and a different order, fma(r, w2, fma(b, w0, g * w1)); the reference never passes one.) */
void oracle_bgr_to_gray(const uint8_t *bgr, int64_t n_pix, uint8_t *gray)
{
    for (int64_t i = 0; i < n_pix; ++i) {
        const double b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        const double v = fma(r, 0.2989, fma(g, 0.1140, b * 0.5870));
        gray[i] = (uint8_t)v;                     /* truncating cast of a value in [0, 255) */
    }
}

/* ------------------------------------------------------------------ make_voxel (testh5.py:60-90) */
/* ts_us: int64 microseconds already shifted to ts[0]=0 (the Python side does the float math that
 * the reference does in numpy: ((ts-ts[0])*1e6).astype(int64)); ps01 in {0,1}. out [Tb,H,W] float64. */
int oracle_make_voxel(const int64_t *ts_us, const int64_t *xs, const int64_t *ys, const int8_t *ps01,
                      int64_t n, int Tb, int64_t H, int64_t W, int interpolate, double *out)
{
    memset(out, 0, sizeof(double) * (size_t)Tb * H * W);
    if (n == 0) return 0;
    if (!interpolate) {
        double t_per_bin = ((double)ts_us[n - 1] + 0.001) / (double)Tb;
        for (int64_t i = 0; i < n; ++i) {
            uint8_t b = (uint8_t)floor((double)ts_us[i] / t_per_bin);
            if (b >= Tb) return -4;
            out[((int64_t)b * H + ys[i]) * W + xs[i]] += (double)(ps01[i] * 2 - 1);
        }
    } else {
        double dt = (double)(ts_us[n - 1] - ts_us[0]);
        for (int b = 0; b < Tb; ++b)
            for (int64_t i = 0; i < n; ++i) {
                double t_norm = (double)(ts_us[i] - ts_us[0]) / (dt + 0.0001) * (double)(Tb - 1);
                double w = 1.0 - fabs(t_norm - (double)b);
                if (w < 0.0) w = 0.0;
                out[((int64_t)b * H + ys[i]) * W + xs[i]] += w * (double)(ps01[i] * 2 - 1);
            }
    }
    return 0;
}
