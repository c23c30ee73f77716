// Common device/host helpers for the gfx950 (MI355X, CDNA4) ViT hot-path kernels.
// Wave = 64 lanes everywhere; no portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ucfvit_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// ---------------------------------------------------------------- errors
void ucfvit_set_error(const char* fmt, ...);

#define UCF_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ucfvit_set_error(__VA_ARGS__);       \
            return UCFVIT_ERR_INVALID_ARGUMENT;  \
        }                                        \
    } while (0)

#define UCF_LAUNCH_CHECK(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            ucfvit_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return UCFVIT_ERR_HIP;                                               \
        }                                                                        \
    } while (0)

static inline int ucf_is_aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---------------------------------------------------------------- scalar conversions
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }  // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

// 16-byte vector of T: 4 floats or 8 bf16
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec16<bf16> {
    static constexpr int N = 8;
    bf16x8 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

// 4 consecutive elements of T (16 B for float, 8 B for bf16)
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = x; }
};
template <> struct Vec4<bf16> {
    bf16x4 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    __device__ __forceinline__ void set(int i, float x) { v[i] = (bf16)x; }
};

// ---------------------------------------------------------------- wave reductions (64 lanes)
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

// exact (erf) GELU, as nn.GELU() default (reference: simple/arch.py:172)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// ---- bf16-path GELU: the erff / expf library calls cost ~80 VALU instructions per element, which made the fused epilogues of the
// fc1 forward and fc2 data-gradient GEMMs VALU-bound (measured B=166: +114 us and +265 us per launch over the plain epilogue).
// Phi(x) = 0.5 erfc(-x / sqrt 2) by Abramowitz-Stegun 7.1.28, erfc(u) = (1 + a1 u + ... + a6 u^6)^-16 for u >= 0, with the 1/sqrt 2
// folded into the coefficients: 6 FMAs, 4 squarings, one v_rcp_f32.  |error| < 1e-6 absolute in fp32 arithmetic (checked against
// scipy over [-12, 12], tests/test_hip_ops.py) = 1/2000 of a bf16 ulp at 1; only kernels whose OUTPUT is bf16 use it, the fp32
// GEMM path keeps erff.  Two elements at a time so the polynomial maps to v_pk_fma_f32 / v_pk_mul_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 norm_cdf_fast2(f32x2 x) {
    const f32x2 a = __builtin_elementwise_abs(x);
    f32x2 p = a * 5.38297490493278e-06f + 4.889063711743802e-05f;
    p = p * a + 3.8003574445610866e-05f;
    p = p * a + 0.0032776263542473316f;
    p = p * a + 0.02114100567996502f;
    p = p * a + 0.04986734688282013f;
    p = p * a + 1.0f;
    p = p * p;
    p = p * p;
    p = p * p;
    p = p * p;                                   // overflows to +inf for |x| > ~25: rcp(inf) = 0, the right limit
    f32x2 r;
    r[0] = 0.5f * __builtin_amdgcn_rcpf(p[0]);
    r[1] = 0.5f * __builtin_amdgcn_rcpf(p[1]);
    f32x2 o;
    o[0] = x[0] >= 0.f ? 1.0f - r[0] : r[0];
    o[1] = x[1] >= 0.f ? 1.0f - r[1] : r[1];
    return o;
}
// v[0..7] = gelu(v[0..7])
__device__ __forceinline__ void gelu_fast8(float* v) {
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        const f32x2 x = {v[r], v[r + 1]};
        const f32x2 y = x * norm_cdf_fast2(x);
        v[r] = y[0];
        v[r + 1] = y[1];
    }
}
// v[0..7] *= gelu'(h[0..7]),  gelu'(x) = Phi(x) + x phi(x)
__device__ __forceinline__ void gelu_grad_fast8(float* v, const float* h) {
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        const f32x2 x = {h[r], h[r + 1]};
        const f32x2 cdf = norm_cdf_fast2(x);
        const f32x2 e = x * x * -0.72134752044448170368f;          // -x^2/2 * log2(e)
        f32x2 pdf;
        pdf[0] = 0.39894228040143267794f * __builtin_amdgcn_exp2f(e[0]);
        pdf[1] = 0.39894228040143267794f * __builtin_amdgcn_exp2f(e[1]);
        const f32x2 gd = cdf + x * pdf;
        v[r] *= gd[0];
        v[r + 1] *= gd[1];
    }
}

// v[0..7] = gelu(v[0..7]) and d[0..7] = gelu'(v[0..7]) from one erfc evaluation
__device__ __forceinline__ void gelu_and_grad_fast8(float* v, float* d) {
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        const f32x2 x = {v[r], v[r + 1]};
        const f32x2 cdf = norm_cdf_fast2(x);
        const f32x2 e = x * x * -0.72134752044448170368f;
        f32x2 pdf;
        pdf[0] = 0.39894228040143267794f * __builtin_amdgcn_exp2f(e[0]);
        pdf[1] = 0.39894228040143267794f * __builtin_amdgcn_exp2f(e[1]);
        const f32x2 y = x * cdf, gd = cdf + x * pdf;
        v[r] = y[0];
        v[r + 1] = y[1];
        d[r] = gd[0];
        d[r + 1] = gd[1];
    }
}

// XCD-aware bijective block remap: blocks b and b+8 share an XCD (observed round-robin), so give each
// XCD label a contiguous chunk of the logical tile order (neighbouring tiles share operand panels -> L2 hits).
// Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}
