// v2v_common.hpp -- build-time tunables and the raw-vector load/store helpers shared by the ESIM and v2e kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "v2v_args.hpp"
#include "v2v_rng.hpp"

namespace v2v {

// Tunables (overridable at build time for sweeps: make EXTRA="-DV2V_DEPTH=3 -DV2V_MIN_WAVES=4")
#ifndef V2V_DEPTH
#define V2V_DEPTH 4
#endif
#ifndef V2V_V2E_LP_TABLE
#define V2V_V2E_LP_TABLE 1   // v2e, float32 input: tabulated low-pass factors in the feature-specialised instances
#endif
#ifndef V2V_V2E_DEPTH
#define V2V_V2E_DEPTH 2     // frames in flight per work-item of the v2e kernel (even)
#endif
#ifndef V2V_V2E_MIN_WAVES
#define V2V_V2E_MIN_WAVES 3  // __launch_bounds__ occupancy target of the v2e kernel (waves per SIMD): 2 -> 3 waves is worth 12 % (2.27 -> 2.06 ms, config 3) despite ~25 spilled dwords; 4 spills the loop (3.3 ms)
#endif
#ifndef V2V_ESIM_ASYM4
#define V2V_ESIM_ASYM4 1   // general device-noise ESIM instances: by-polarity loop only, 2-frame ring, 4 waves per SIMD
#endif
#ifndef V2V_MIN_WAVES
#define V2V_MIN_WAVES 1
#endif
#ifndef V2V_NT_LOADS
#define V2V_NT_LOADS 1
#endif
#ifndef V2V_NT_STORES
#define V2V_NT_STORES 1     // voxel planes are written once and never re-read by the kernel: keep them out of the caches
#endif
constexpr int kDepth = V2V_DEPTH;   // frames in flight per work-item (register ring, reloaded right after use)

// compile-time unrolled loop: f(std::integral_constant<int, 0>{}), f(<1>), ...
template <int... U, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, U...>, F &&f) { (f(std::integral_constant<int, U>{}), ...); }

// ------------------------------------------------------------------------------------------------
// raw input vectors
template <int IN, int VEC> struct Raw;
template <> struct Raw<kInF32, 4> { float4 v; };
template <> struct Raw<kInF32, 2> { float2 v; };
template <> struct Raw<kInF32, 1> { float v; };
template <> struct Raw<kInU8, 4> { uint32_t v; };
template <> struct Raw<kInU8, 2> { uint16_t v; };
template <> struct Raw<kInU8, 1> { uint8_t v; };

template <int IN, int VEC, bool NT = (V2V_NT_LOADS != 0)>
__device__ __forceinline__ Raw<IN, VEC> load_raw(const void *base, int64_t elem_off)
{
    Raw<IN, VEC> r;
    if constexpr (IN == kInF32 && VEC == 4) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 *ptr = reinterpret_cast<const f32x4 *>(static_cast<const float *>(base) + elem_off);
        f32x4 t;
        if constexpr (NT) t = __builtin_nontemporal_load(ptr); else t = *ptr;
        r.v = make_float4(t.x, t.y, t.z, t.w);
    } else if constexpr (IN == kInF32 && VEC == 2) {
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        const f32x2v *ptr = reinterpret_cast<const f32x2v *>(static_cast<const float *>(base) + elem_off);
        f32x2v t;
        if constexpr (NT) t = __builtin_nontemporal_load(ptr); else t = *ptr;
        r.v = make_float2(t.x, t.y);
    } else if constexpr (IN == kInF32) r.v = static_cast<const float *>(base)[elem_off];
    else if constexpr (VEC == 2) {
        const uint16_t *ptr = reinterpret_cast<const uint16_t *>(static_cast<const uint8_t *>(base) + elem_off);
        if constexpr (NT) r.v = __builtin_nontemporal_load(ptr); else r.v = *ptr;
    } else if constexpr (VEC == 4) {
        const uint32_t *ptr = reinterpret_cast<const uint32_t *>(static_cast<const uint8_t *>(base) + elem_off);
        if constexpr (NT) r.v = __builtin_nontemporal_load(ptr); else r.v = *ptr;
    } else r.v = static_cast<const uint8_t *>(base)[elem_off];
    return r;
}

template <int IN> struct LutT { using type = double; };
template <> struct LutT<kInF32> { using type = float; };

template <int VEC>
__device__ __forceinline__ float raw_f32(const Raw<kInF32, VEC> &r, int j)
{
    if constexpr (VEC == 4) return (j == 0) ? r.v.x : (j == 1) ? r.v.y : (j == 2) ? r.v.z : r.v.w;
    else if constexpr (VEC == 2) return (j == 0) ? r.v.x : r.v.y;
    else return r.v;
}

template <int VEC, typename T>
__device__ __forceinline__ void store_vec(void *out, int64_t off, const T (&v)[VEC])
{
    T *o = static_cast<T *>(out) + off;
#if V2V_NT_STORES
    if constexpr (VEC == 4 && sizeof(T) == 8) {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(f64x2{v[0], v[1]}, reinterpret_cast<f64x2 *>(o));
        __builtin_nontemporal_store(f64x2{v[2], v[3]}, reinterpret_cast<f64x2 *>(o) + 1);
    } else if constexpr (VEC == 4) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4 *>(o));
    } else if constexpr (VEC == 2 && sizeof(T) == 8) {
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(f64x2{v[0], v[1]}, reinterpret_cast<f64x2 *>(o));
    } else if constexpr (VEC == 2) {
        typedef float f32x2v __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(f32x2v{v[0], v[1]}, reinterpret_cast<f32x2v *>(o));
    } else {
        __builtin_nontemporal_store(v[0], o);
    }
#else
    if constexpr (VEC == 4 && sizeof(T) == 8) {
        reinterpret_cast<double2 *>(o)[0] = make_double2(v[0], v[1]);
        reinterpret_cast<double2 *>(o)[1] = make_double2(v[2], v[3]);
    } else if constexpr (VEC == 4) {
        *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
    } else if constexpr (VEC == 2 && sizeof(T) == 8) {
        *reinterpret_cast<double2 *>(o) = make_double2(v[0], v[1]);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2 *>(o) = make_float2(v[0], v[1]);
    } else {
        o[0] = v[0];
    }
#endif
}

}  // namespace v2v
