// common.h -- shared device/host helpers for libsvdx (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/svdx.h"

typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

void svdx_set_error(const char* fmt, ...);

#define SVDX_CHECK_ARG(cond, ...)                 \
    do {                                          \
        if (!(cond)) {                            \
            svdx_set_error(__VA_ARGS__);          \
            return -2;                            \
        }                                         \
    } while (0)

#define SVDX_LAUNCH_CHECK(name)                                                        \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            svdx_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return -1;                                                                 \
        }                                                                              \
    } while (0)

template <typename T> struct TT;
template <> struct TT<f16> {
    typedef f16x8 v8;
    typedef f16x4 v4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct TT<bf16> {
    typedef bf16x8 v8;
    typedef bf16x4 v4;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <typename T> __device__ __forceinline__ float to_f(T x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x) { return (T)x; }

// 16-byte vector of 8 activations <-> 8 floats
template <typename T> struct alignas(16) Vec8 { T v[8]; };
template <typename T> struct alignas(8) Vec4 { T v[4]; };

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&o)[8]) {
    Vec8<T> t = *reinterpret_cast<const Vec8<T>*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = to_f<T>(t.v[i]);
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&o)[8]) {
    Vec8<T> t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t.v[i] = from_f<T>(o[i]);
    *reinterpret_cast<Vec8<T>*>(p) = t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float silu_gradf_(float x) {
    float s = sigmoidf_(x);
    return s * (1.f + x * (1.f - s));
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * expf(-0.5f * x * x) * 0.3989422804014327f;
}

#define DISPATCH_DTYPE(dtype, ...)                                   \
    if ((dtype) == SVDX_F16) { typedef f16 T; __VA_ARGS__; }         \
    else if ((dtype) == SVDX_BF16) { typedef bf16 T; __VA_ARGS__; }  \
    else { svdx_set_error("bad dtype %d", (int)(dtype)); return -2; }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
