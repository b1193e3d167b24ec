// valu_rates.hip -- instruction-throughput microbenchmark for the ops the ESIM kernel is made of (gfx950).
// Each kernel issues a long unrolled run of ONE instruction on 8 independent register chains; every CU runs
// 4 waves per SIMD.  Reports cycles per wave-instruction per SIMD relative to v_fma_f32 (= 2 by the guide).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
constexpr int kIters = 2000;

#define DEF_KERNEL_F64(NAME, ASM)                                                              \
__global__ void __launch_bounds__(256) NAME(double *out, double seed) {                         \
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    double b = seed * 1.0000001;                                                                \
    for (int i = 0; i < kIters; ++i) {                                                          \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                    \
                     ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                    \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc"); \
    }                                                                                           \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;               \
}
#define A_ADD64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define A_MUL64(n) "v_mul_f64 %" #n ", %" #n ", %8\n"
#define A_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %8, %8\n"
#define A_FLOOR64(n) "v_floor_f64 %" #n ", %" #n "\n"
#define A_CMP64(n) "v_cmp_ge_f64 vcc, %" #n ", %8\n"
#define A_MAX64(n) "v_max_f64 %" #n ", %" #n ", %8\n"
#define A_RCP64(n) "v_rcp_f64 %" #n ", %" #n "\n"
#define A_TRUNC64(n) "v_trunc_f64 %" #n ", %" #n "\n"
DEF_KERNEL_F64(k_add_f64, A_ADD64)
DEF_KERNEL_F64(k_mul_f64, A_MUL64)
DEF_KERNEL_F64(k_fma_f64, A_FMA64)
DEF_KERNEL_F64(k_floor_f64, A_FLOOR64)
DEF_KERNEL_F64(k_cmp_ge_f64, A_CMP64)
DEF_KERNEL_F64(k_max_f64, A_MAX64)
DEF_KERNEL_F64(k_rcp_f64, A_RCP64)
DEF_KERNEL_F64(k_trunc_f64, A_TRUNC64)

#define DEF_KERNEL_F32(NAME, ASM)                                                              \
__global__ void __launch_bounds__(256) NAME(double *out, double seed) {                         \
    float a0 = (float)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = (float)seed * 1.0000001f;                                                         \
    for (int i = 0; i < kIters; ++i) {                                                          \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                    \
                     ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                    \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc", "s20", "s21"); \
    }                                                                                           \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;               \
}
#define A_FMA32(n) "v_fma_f32 %" #n ", %" #n ", %8, %8\n"
#define A_ADD32(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_CVTU32(n) "v_cvt_u32_f32 %" #n ", %" #n "\n"
#define A_CVTUB(n) "v_cvt_f32_ubyte0 %" #n ", %" #n "\n"
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define A_CMP32(n) "v_cmp_neq_f32 vcc, %" #n ", %8\n"
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define A_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define A_SQRT32(n) "v_sqrt_f32 %" #n ", %" #n "\n"
#define A_MAX32(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define A_CNDMASK64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %8\n"
#define A_ASHR(n) "v_ashrrev_i32 %" #n ", 31, %" #n "\n"
#define A_NOT(n) "v_not_b32 %" #n ", %" #n "\n"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %8\n"
#define A_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define A_SUB32(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define A_MUL32(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define A_ADDU32(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_CNDMASK_FRESH(n) "v_cmp_neq_f32 vcc, %" #n ", %8\nv_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
DEF_KERNEL_F32(k_cndmask_e64, A_CNDMASK64)
DEF_KERNEL_F32(k_bfi, A_BFI)
DEF_KERNEL_F32(k_ashr, A_ASHR)
DEF_KERNEL_F32(k_not, A_NOT)
DEF_KERNEL_F32(k_add3, A_ADD3)
DEF_KERNEL_F32(k_mov, A_MOV)
DEF_KERNEL_F32(k_sub32, A_SUB32)
DEF_KERNEL_F32(k_mul32, A_MUL32)
DEF_KERNEL_F32(k_addu32, A_ADDU32)
DEF_KERNEL_F32(k_cmp_cndmask, A_CNDMASK_FRESH)
DEF_KERNEL_F32(k_fma_f32, A_FMA32)
DEF_KERNEL_F32(k_add_f32, A_ADD32)
DEF_KERNEL_F32(k_cndmask, A_CNDMASK)
DEF_KERNEL_F32(k_cvt_u32_f32, A_CVTU32)
DEF_KERNEL_F32(k_cvt_f32_ubyte0, A_CVTUB)
DEF_KERNEL_F32(k_and_b32, A_AND)
DEF_KERNEL_F32(k_xor_b32, A_XOR)
DEF_KERNEL_F32(k_lshl_add_u32, A_LSHLADD)
DEF_KERNEL_F32(k_cmp_neq_f32, A_CMP32)
DEF_KERNEL_F32(k_mul_lo_u32, A_MULLO)
DEF_KERNEL_F32(k_mul_hi_u32, A_MULHI)
DEF_KERNEL_F32(k_sqrt_f32, A_SQRT32)
DEF_KERNEL_F32(k_max_f32, A_MAX32)


#define A_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %8\n"
#define A_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 7\n"
#define A_FFBH(n) "v_ffbh_u32 %" #n ", %" #n "\n"
#define A_FREXPM(n) "v_frexp_mant_f32 %" #n ", %" #n "\n"
#define A_FREXPE(n) "v_frexp_exp_i32_f32 %" #n ", %" #n "\n"
#define A_CVTF32U32(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define A_CVTSDWA(n) "v_cvt_f32_u32_sdwa %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define A_EXP32(n) "v_exp_f32 %" #n ", %" #n "\n"
#define A_LOG32(n) "v_log_f32 %" #n ", %" #n "\n"
#define A_RNDNE32(n) "v_rndne_f32 %" #n ", %" #n "\n"
#define A_MULU24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define A_MULHIU24(n) "v_mul_hi_u32_u24 %" #n ", %" #n ", %8\n"
#define A_LSHRREV(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define A_OR(n) "v_or_b32 %" #n ", %" #n ", %8\n"
#define A_FMAC(n) "v_fmac_f32 %" #n ", %8, %8\n"
#define A_FMAAK(n) "v_fmaak_f32 %" #n ", %" #n ", %8, 0x3f800001\n"
DEF_KERNEL_F32(k_and_or, A_ANDOR)
DEF_KERNEL_F32(k_alignbit, A_ALIGNBIT)
DEF_KERNEL_F32(k_ffbh, A_FFBH)
DEF_KERNEL_F32(k_frexp_mant, A_FREXPM)
DEF_KERNEL_F32(k_frexp_exp, A_FREXPE)
DEF_KERNEL_F32(k_cvt_f32_u32, A_CVTF32U32)
DEF_KERNEL_F32(k_cvt_sdwa, A_CVTSDWA)
DEF_KERNEL_F32(k_exp32, A_EXP32)
DEF_KERNEL_F32(k_log32, A_LOG32)
DEF_KERNEL_F32(k_rndne32, A_RNDNE32)
DEF_KERNEL_F32(k_mul_u24, A_MULU24)
DEF_KERNEL_F32(k_mulhi_u24, A_MULHIU24)
DEF_KERNEL_F32(k_lshrrev, A_LSHRREV)
DEF_KERNEL_F32(k_or_b32, A_OR)
DEF_KERNEL_F32(k_fmac, A_FMAC)
DEF_KERNEL_F32(k_fmaak, A_FMAAK)
// 32x32 -> 64 multiply-add (Philox): destination is a register pair
__global__ void __launch_bounds__(256) k_mad_u64_u32(double *out, double seed) {
    unsigned long long d0 = threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    unsigned b = (unsigned)seed * 2654435761u + 12345u;
    for (int i = 0; i < kIters; ++i) {
#define MAD(n) "v_mad_u64_u32 %" #n ", vcc, %8, %8, %" #n "\n"
        asm volatile(MAD(0) MAD(1) MAD(2) MAD(3) MAD(4) MAD(5) MAD(6) MAD(7) MAD(0) MAD(1) MAD(2) MAD(3) MAD(4) MAD(5) MAD(6) MAD(7)
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(b) : "vcc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = (double)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

// mixed-width conversions: f64 <-> f32 (source and destination differ in size, so separate chains)
__global__ void __launch_bounds__(256) k_cvt_f64_f32(double *out, double seed) {
    float s0 = (float)seed + threadIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int i = 0; i < kIters; ++i) {
        asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n"
                     "v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n"
                     "v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n"
                     "v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = d0 + d1 + d2 + d3;
}
__global__ void __launch_bounds__(256) k_cvt_f32_f64(double *out, double seed) {
    double s0 = seed + threadIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    float d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int i = 0; i < kIters; ++i) {
        asm volatile("v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n"
                     "v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n"
                     "v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n"
                     "v_cvt_f32_f64 %0, %4\nv_cvt_f32_f64 %1, %5\nv_cvt_f32_f64 %2, %6\nv_cvt_f32_f64 %3, %7\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (double)(d0 + d1 + d2 + d3);
}
__global__ void __launch_bounds__(256) k_cvt_i32_f64(double *out, double seed) {
    double s0 = seed + threadIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    int d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int i = 0; i < kIters; ++i) {
        asm volatile("v_cvt_i32_f64 %0, %4\nv_cvt_i32_f64 %1, %5\nv_cvt_i32_f64 %2, %6\nv_cvt_i32_f64 %3, %7\n"
                     "v_cvt_i32_f64 %0, %4\nv_cvt_i32_f64 %1, %5\nv_cvt_i32_f64 %2, %6\nv_cvt_i32_f64 %3, %7\n"
                     "v_cvt_i32_f64 %0, %4\nv_cvt_i32_f64 %1, %5\nv_cvt_i32_f64 %2, %6\nv_cvt_i32_f64 %3, %7\n"
                     "v_cvt_i32_f64 %0, %4\nv_cvt_i32_f64 %1, %5\nv_cvt_i32_f64 %2, %6\nv_cvt_i32_f64 %3, %7\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = (double)(d0 + d1 + d2 + d3);
}
__global__ void __launch_bounds__(256) k_cvt_f64_i32(double *out, double seed) {
    int s0 = (int)seed + threadIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
    for (int i = 0; i < kIters; ++i) {
        asm volatile("v_cvt_f64_i32 %0, %4\nv_cvt_f64_i32 %1, %5\nv_cvt_f64_i32 %2, %6\nv_cvt_f64_i32 %3, %7\n"
                     "v_cvt_f64_i32 %0, %4\nv_cvt_f64_i32 %1, %5\nv_cvt_f64_i32 %2, %6\nv_cvt_f64_i32 %3, %7\n"
                     "v_cvt_f64_i32 %0, %4\nv_cvt_f64_i32 %1, %5\nv_cvt_f64_i32 %2, %6\nv_cvt_f64_i32 %3, %7\n"
                     "v_cvt_f64_i32 %0, %4\nv_cvt_f64_i32 %1, %5\nv_cvt_f64_i32 %2, %6\nv_cvt_f64_i32 %3, %7\n"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(s0), "v"(s1), "v"(s2), "v"(s3));
    }
    out[blockIdx.x * 256 + threadIdx.x] = d0 + d1 + d2 + d3;
}
// LDS table lookups with data-dependent (pseudo-random / smooth) indices
__global__ void __launch_bounds__(256) k_lds_lut(double *out, double seed, int smooth) {
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)threadIdx.x * 0.5f;
    __syncthreads();
    unsigned idx = smooth ? (threadIdx.x / 2 + (unsigned)seed) : (threadIdx.x * 2654435761u + (unsigned)seed);
    float acc = 0;
    for (int i = 0; i < kIters * 2; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = lut[(idx >> (smooth ? 0 : 11)) & 255];
            acc += v;
            idx = smooth ? idx + 1 + (__float_as_uint(v) & 3) : idx * 1664525u + 1013904223u + __float_as_uint(v);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_pk_fma(double *out, double seed) {
    f32x2 a0 = {(float)seed + threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f32x2 b = {(float)seed * 1.0000001f, 0.5f};
    for (int i = 0; i < kIters; ++i) {
#define PK(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %8\n"
        asm volatile(PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7) PK(0) PK(1) PK(2) PK(3) PK(4) PK(5) PK(6) PK(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    f32x2 t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;
}
__global__ void __launch_bounds__(256) k_pk_mul(double *out, double seed) {
    f32x2 a0 = {(float)seed + threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f32x2 b = {(float)seed * 1.0000001f, 0.5f};
    for (int i = 0; i < kIters; ++i) {
#define PKM(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
        asm volatile(PKM(0) PKM(1) PKM(2) PKM(3) PKM(4) PKM(5) PKM(6) PKM(7) PKM(0) PKM(1) PKM(2) PKM(3) PKM(4) PKM(5) PKM(6) PKM(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    f32x2 t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;
}
__global__ void __launch_bounds__(256) k_pk_add(double *out, double seed) {
    f32x2 a0 = {(float)seed + threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f32x2 b = {(float)seed * 1.0000001f, 0.5f};
    for (int i = 0; i < kIters; ++i) {
#define PKA(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
        asm volatile(PKA(0) PKA(1) PKA(2) PKA(3) PKA(4) PKA(5) PKA(6) PKA(7) PKA(0) PKA(1) PKA(2) PKA(3) PKA(4) PKA(5) PKA(6) PKA(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    }
    f32x2 t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * 256 + threadIdx.x] = t.x + t.y;
}
// independent LDS lookups (throughput, not a dependent chain): 8 reads in flight per lane
__global__ void __launch_bounds__(256) k_lds_tp(double *out, double seed, int mode) {
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)threadIdx.x * 0.5f;
    __syncthreads();
    unsigned base = mode == 0 ? (threadIdx.x * 2654435761u) >> 11 : mode == 1 ? threadIdx.x / 2 : threadIdx.x;
    float acc = 0;
    for (int i = 0; i < kIters * 2; ++i) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lut[(base + 37u * j + (unsigned)i * (mode == 0 ? 101u : 1u)) & 255];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
typedef void (*kern_t)(double *, double);
int main() {
    double *out;
    const int blocks = 256 * 4;   // 4 blocks of 4 waves per CU -> 4 waves per SIMD
    hipMalloc(&out, sizeof(double) * blocks * 256);
    struct Item { const char *name; kern_t k; int instr_per_iter; };
    std::vector<Item> items = {
        {"v_fma_f32", k_fma_f32, 16}, {"v_add_f32", k_add_f32, 16}, {"v_max_f32", k_max_f32, 16}, {"v_cndmask_b32", k_cndmask, 16},
        {"v_cndmask_e64(sgpr)", k_cndmask_e64, 16}, {"v_bfi_b32", k_bfi, 16}, {"v_ashrrev_i32", k_ashr, 16}, {"v_not_b32", k_not, 16},
        {"v_add3_u32", k_add3, 16}, {"v_mov_b32", k_mov, 16}, {"v_sub_f32", k_sub32, 16}, {"v_mul_f32", k_mul32, 16}, {"v_add_u32", k_addu32, 16},
        {"cmp+cndmask pair", k_cmp_cndmask, 32},
        {"v_cvt_u32_f32", k_cvt_u32_f32, 16}, {"v_cvt_f32_ubyte0", k_cvt_f32_ubyte0, 16}, {"v_and_b32", k_and_b32, 16},
        {"v_xor_b32", k_xor_b32, 16}, {"v_lshl_add_u32", k_lshl_add_u32, 16}, {"v_cmp_neq_f32", k_cmp_neq_f32, 16},
        {"v_mul_lo_u32", k_mul_lo_u32, 16}, {"v_mul_hi_u32", k_mul_hi_u32, 16}, {"v_sqrt_f32", k_sqrt_f32, 16},
        {"v_add_f64", k_add_f64, 16}, {"v_mul_f64", k_mul_f64, 16}, {"v_fma_f64", k_fma_f64, 16}, {"v_max_f64", k_max_f64, 16},
        {"v_floor_f64", k_floor_f64, 16}, {"v_trunc_f64", k_trunc_f64, 16}, {"v_cmp_ge_f64", k_cmp_ge_f64, 16}, {"v_rcp_f64", k_rcp_f64, 16},
        {"v_cvt_f64_f32", k_cvt_f64_f32, 16}, {"v_cvt_f32_f64", k_cvt_f32_f64, 16}, {"v_cvt_i32_f64", k_cvt_i32_f64, 16},
        {"v_cvt_f64_i32", k_cvt_f64_i32, 16}, {"v_pk_fma_f32", k_pk_fma, 16},
        {"v_pk_mul_f32", k_pk_mul, 16}, {"v_pk_add_f32", k_pk_add, 16}, {"v_mad_u64_u32", k_mad_u64_u32, 16}, {"v_and_or_b32", k_and_or, 16},
        {"v_alignbit_b32", k_alignbit, 16}, {"v_ffbh_u32", k_ffbh, 16}, {"v_frexp_mant_f32", k_frexp_mant, 16}, {"v_frexp_exp_i32_f32", k_frexp_exp, 16},
        {"v_cvt_f32_u32", k_cvt_f32_u32, 16}, {"v_cvt_f32_u32_sdwa", k_cvt_sdwa, 16}, {"v_exp_f32", k_exp32, 16}, {"v_log_f32", k_log32, 16},
        {"v_rndne_f32", k_rndne32, 16}, {"v_mul_u32_u24", k_mul_u24, 16}, {"v_mul_hi_u32_u24", k_mulhi_u24, 16}, {"v_lshrrev_b32", k_lshrrev, 16},
        {"v_or_b32", k_or_b32, 16}, {"v_fmac_f32", k_fmac, 16}, {"v_fmaak_f32", k_fmaak, 16},
    };
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double base_ms = 0;
    for (auto &it : items) {
        it.k<<<blocks, 256>>>(out, 1.5);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) it.k<<<blocks, 256>>>(out, 1.5);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        if (base_ms == 0) base_ms = ms;
        // wave-instructions per SIMD = 4 waves * kIters * instr_per_iter
        const double winstr = 4.0 * kIters * it.instr_per_iter;
        printf("%-18s %8.3f ms   %6.2f ns/wave-instr/SIMD   rel cycles (v_fma_f32 = 2): %5.2f\n", it.name, ms, ms * 1e6 / winstr,
               2.0 * ms / base_ms);
    }
    for (int smooth = 0; smooth < 2; ++smooth) {
        k_lds_lut<<<blocks, 256>>>(out, 1.5, smooth);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) k_lds_lut<<<blocks, 256>>>(out, 1.5, smooth);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double winstr = 4.0 * kIters * 2 * 8;
        printf("lds_lut(%s)    %8.3f ms   %6.2f ns per ds_read_b32 wave-instr per SIMD (4 SIMDs share the LDS)\n", smooth ? "smooth" : "random", ms, ms * 1e6 / winstr);
    }
    for (int mode = 0; mode < 3; ++mode) {
        k_lds_tp<<<blocks, 256>>>(out, 1.5, mode);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) k_lds_tp<<<blocks, 256>>>(out, 1.5, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double winstr = 4.0 * kIters * 2 * 8;
        printf("lds_throughput(%s) %8.3f ms  %6.2f ns per ds_read_b32 wave-instr per SIMD (incl. ~3 VALU for index+add)\n",
               mode == 0 ? "random" : mode == 1 ? "pairs" : "linear", ms, ms * 1e6 / winstr);
    }
    return 0;
}
