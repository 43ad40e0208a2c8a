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

// ---- launch plans (include/svdx.h: svdx_plan_*; common.cpp) --------------------------------------------------------------------------------
// Every kernel of the library is launched through hipLaunchKernelGGL; here that macro is re-pointed at svdx_launch, which issues the same
// launch through hipLaunchKernel and -- while the calling thread records a plan -- keeps the kernel's address, geometry and a copy of its
// by-value arguments.  A recorded plan re-issues the launches from C with no Python, torch or hipGraph involved (svdx_plan_replay).
bool svdx_plan_recording();
void svdx_plan_record(const void* fn, dim3 grid, dim3 block, unsigned lds, void* const* args, const size_t* sizes, int nargs);
#ifndef SVDX_SIM
#include <tuple>
#include <utility>
template <typename... KArgs, typename... Args>
inline void svdx_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, unsigned lds, hipStream_t st, Args&&... args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel argument count");
    const void* fn = reinterpret_cast<const void*>(kernel);
    if constexpr (sizeof...(KArgs) == 0) {
        (void)hipLaunchKernel(fn, grid, block, nullptr, lds, st);
        if (svdx_plan_recording()) svdx_plan_record(fn, grid, block, lds, nullptr, nullptr, 0);
    } else {
        std::tuple<std::remove_cv_t<KArgs>...> held(std::forward<Args>(args)...);    // converted to the kernel's parameter types, as <<< >>> would
        std::apply([&](auto&... a) {
            void* ptrs[] = {(void*)&a...};
            (void)hipLaunchKernel(fn, grid, block, ptrs, lds, st);
            if (svdx_plan_recording()) {
                const size_t sizes[] = {sizeof(a)...};
                svdx_plan_record(fn, grid, block, lds, ptrs, sizes, (int)sizeof...(a));
            }
        }, held);
    }
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    svdx_launch(kern, dim3(grid), dim3(block), (unsigned)(shmem), (hipStream_t)(stream), ##__VA_ARGS__)
#endif

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

// sum over the 64 lanes, result in every lane.  The 16 lanes of a DPP row are folded with four VALU instructions (row_mirror,
// row_half_mirror, two quad permutes -- no LDS crossbar round trips), the four rows with two ds_bpermute steps.  Fixed order:
// the result is bit-identical from run to run.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, false));   // lane i + lane 15 - i
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));   // + mirror inside each half
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float siluf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float silu_gradf_(float x) {
    float s = sigmoidf_(x);
    return s * (1.f + x * (1.f - s));
}

// Exact-erf GELU (what F.gelu computes) through Abramowitz-Stegun 7.1.26: |erf error| <= 1.5e-7, far inside the 16-bit output
// rounding.  One v_exp + one v_rcp + ~12 FMAs give the normal CDF and, from the same exponential, the density for the
// gradient -- the fused GEGLU GEMM epilogues are VALU-bound on this, so libm's branchy erff is not affordable there.
struct GeluParts { float cdf, pdf_x; };      // Phi(x), x * phi(x)
__device__ __forceinline__ GeluParts gelu_parts(float x) {
    const float ax = fabsf(x);
    const float E = __builtin_amdgcn_exp2f(-0.7213475204444817f * x * x);          // exp(-x^2/2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752f, ax, 1.f));
    float q = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    q = fmaf(t, q, 0.5f * 1.421413741f);
    q = fmaf(t, q, 0.5f * -0.284496736f);
    q = fmaf(t, q, 0.5f * 0.254829592f);
    const float h = q * t * E;                                                     // 0.5 * erfc(|x|/sqrt2)
    GeluParts r;
    r.cdf = x < 0.f ? h : 1.f - h;
    r.pdf_x = x * E * 0.3989422804014327f;
    return r;
}
__device__ __forceinline__ float gelu_erf(float x) { return x * gelu_parts(x).cdf; }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const GeluParts g = gelu_parts(x);
    return g.cdf + g.pdf_x;
}

// Fixed-point scales of the GroupNorm statistics (see the banner in norm.hip; shared with the GEMM epilogue that produces them):
// mode 0 = (sum, sum of squares) of the activations, mode 1 = the two backward sums; cnt = rows x channels of one group.
__host__ __device__ __forceinline__ void gn_fixed_scales(long cnt, int mode, int& k0, int& k1) {
    int lg = 0;
    while ((1L << lg) < cnt) ++lg;
    const int b0 = mode == 0 ? 16 : 18, b1 = mode == 0 ? 24 : 26;
    k0 = 62 - b0 - lg; k1 = 62 - b1 - lg;
    k0 = k0 < 0 ? 0 : (k0 > 40 ? 40 : k0);
    k1 = k1 < 0 ? 0 : (k1 > 40 ? 40 : k1);
}

#define DISPATCH_DTYPE(dtype, ...)                                   \
    if ((dtype) == SVDX_F16) { typedef f16 T; __VA_ARGS__; }         \
    else if ((dtype) == SVDX_BF16) { typedef bf16 T; __VA_ARGS__; }  \
    else { svdx_set_error("bad dtype %d", (int)(dtype)); return -2; }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
