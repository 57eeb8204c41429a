// Activation element types of the bf16 data path (BASELINE configs[2-4]): tensors between our own layers are bf16 in HBM
// (NCHW, as PyTorch lays them out); arithmetic, statistics and accumulation stay fp32 in registers.  bf16 -> fp32 is exact
// (a 16-bit shift), fp32 -> bf16 rounds to nearest even (v_cvt_pk_bf16_f32 on gfx950).  Every element-parallel kernel is
// templated on T in {float, bf16_t} through these overloads; the C ABI passes `dt` (0 = fp32, 1 = bf16) beside the pointers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
enum { C2M_F32 = 0, C2M_BF16 = 1 };

__device__ __forceinline__ float c2m_ld(const float* __restrict__ p, long i) { return p[i]; }
__device__ __forceinline__ float c2m_ld(const bf16_t* __restrict__ p, long i) { return (float)p[i]; }
__device__ __forceinline__ void c2m_st(float* __restrict__ p, long i, float v) { p[i] = v; }
__device__ __forceinline__ void c2m_st(bf16_t* __restrict__ p, long i, float v) { p[i] = (bf16_t)v; }

// four consecutive elements (16 B of fp32 / 8 B of bf16; the pointer must be aligned to that)
__device__ __forceinline__ float4 c2m_ld4(const float* __restrict__ p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 c2m_ld4(const bf16_t* __restrict__ p) {
    const uint2 r = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                       __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ void c2m_st4(float* __restrict__ p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void c2m_st4(bf16_t* __restrict__ p, float4 v) {
    typedef bf16_t bf16x4_t __attribute__((ext_vector_type(4)));
    const bf16x4_t q = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
    *reinterpret_cast<bf16x4_t*>(p) = q;
}
// two consecutive elements (8 B of fp32 / 4 B of bf16, aligned to that)
__device__ __forceinline__ float2 c2m_ld2(const float* __restrict__ p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 c2m_ld2(const bf16_t* __restrict__ p) {
    const unsigned r = *reinterpret_cast<const unsigned*>(p);
    return make_float2(__uint_as_float(r << 16), __uint_as_float(r & 0xffff0000u));
}
__device__ __forceinline__ void c2m_st2(float* __restrict__ p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void c2m_st2(bf16_t* __restrict__ p, float2 v) {
    typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t q = {(bf16_t)v.x, (bf16_t)v.y};
    *reinterpret_cast<bf16x2_t*>(p) = q;
}
// alignment (bytes) a pointer needs for c2m_ld4 / c2m_st4
template <class T> struct C2mVec4 { static constexpr uintptr_t mask = 4 * sizeof(T) - 1; };

// run BODY with `T` = the activation type selected by dt
#define C2M_DISPATCH_DT(dt, ...)                       \
    do {                                               \
        if ((dt) == C2M_BF16) { typedef bf16_t T; __VA_ARGS__ } \
        else { typedef float T; __VA_ARGS__ }          \
    } while (0)
