// Shared device helpers for the asrk kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/asrk.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f32_to_bf16(float v) { return (bf16_t)v; }

// exact (erf) GELU, as torch.nn.functional.gelu / nn.GELU() default
// gelu(x) = 0.5 x + 0.5 |x| erf(|x| / sqrt 2), erf by Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7;
// measured max |gelu error| 4.7e-7 over [-12, 12]): ~15 VALU issues, two of them transcendental,
// against ~30 for erff() -- this runs once per output element in the fc1 / conv epilogues.
__device__ __forceinline__ float gelu_erf(float x) {
    const float h = 0.5f * x;
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2 / 2)
    return fmaf(fabsf(h), fmaf(-p, e, 1.0f), h);
}

// d/dx of the exact GELU: Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
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

// 16-byte async global -> LDS copy: LDS destination is (wave-uniform base) + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

#define ASRK_CHECK_HIP(expr)                                  \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return ASRK_ERR_HIP;            \
    } while (0)
