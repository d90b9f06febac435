// Shared device/host helpers for the vision-ft hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define VFT_WAVE 64

// ---------------------------------------------------------------------------------------------
// error plumbing: every C-ABI entry returns 0 on success, non-zero otherwise; the message is kept
// per host thread and read back with vft_last_error().
// ---------------------------------------------------------------------------------------------
extern "C" const char* vft_last_error(void);
void vft_set_error(const char* fmt, ...);

#define VFT_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            vft_set_error(__VA_ARGS__);          \
            return 1;                            \
        }                                        \
    } while (0)

#define VFT_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            vft_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 2;                                                            \
        }                                                                        \
    } while (0)

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }

__device__ __forceinline__ float bfbits2f(unsigned short u) {
    return __uint_as_float(((unsigned int)u) << 16);
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

// block-wide sum for blocks of NW waves; `red` must hold NW floats of LDS. All threads get the sum.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
// d/dx silu(x) = s + x*s*(1-s), s = sigmoid(x)
__device__ __forceinline__ float dsilu_f(float x) {
    float s = 1.f / (1.f + __expf(-x));
    return s * (1.f + x * (1.f - s));
}
// erf to 1.5e-7 absolute (Abramowitz-Stegun 7.1.26), branch-free: a third of libm erff's instructions — the GEGLU
// kernels were VALU-bound on it.  Far below the 2^-9 resolution of the bf16 tensors these functions feed.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float y = 1.f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}
// exact (erf) GELU as torch.nn.functional.gelu default
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_f(float x) {
    float cdf = 0.5f * (1.f + erf_fast(x * 0.70710678118654752f));
    float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// gelu(x) and gelu'(x) sharing one erf and one exp (exp(-x²/2) is also erf_fast's exp((x/√2)²))
__device__ __forceinline__ void gelu_both(float x, float& g, float& dg) {
    const float z = x * 0.70710678118654752f, az = fabsf(z);
    const float e = __expf(-az * az);  // = exp(-x²/2)
    const float t = __frcp_rn(fmaf(0.3275911f, az, 1.f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float cdf = 0.5f * (1.f + copysignf(1.f - p * t * e, z));
    g = x * cdf;
    dg = fmaf(x * 0.39894228040143268f, e, cdf);
}
