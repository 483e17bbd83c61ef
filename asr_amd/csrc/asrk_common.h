// Shared device helpers for the asrk kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>
#include <utility>

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

// The same for two values at once on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of work per
// issue; an epilogue is pure VALU, there is no MFMA beside it to disturb).  Returns 2 * gelu(x) so that the caller can
// fold the 0.5 into its own scale.  11 packed + 2 plain + 4 transcendental issues per pair: 42 cycles per value against
// 68 for gelu_erf.  Coefficients negated so that erf = 1 + q e needs no sign flip.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf_x2(f32x2_t x) {
    f32x2_t ax;
    ax.x = __builtin_fabsf(x.x);
    ax.y = __builtin_fabsf(x.y);
    const f32x2_t one = {1.0f, 1.0f};
    const f32x2_t d = __builtin_elementwise_fma(ax, f32x2_t{0.23164189f, 0.23164189f}, one);  // 1 + 0.3275911 |x| / sqrt 2
    f32x2_t t;
    t.x = __builtin_amdgcn_rcpf(d.x);
    t.y = __builtin_amdgcn_rcpf(d.y);
    f32x2_t q = __builtin_elementwise_fma(t, f32x2_t{-1.061405429f, -1.061405429f}, f32x2_t{1.453152027f, 1.453152027f});
    q = __builtin_elementwise_fma(t, q, f32x2_t{-1.421413741f, -1.421413741f});
    q = __builtin_elementwise_fma(t, q, f32x2_t{0.284496736f, 0.284496736f});
    q = __builtin_elementwise_fma(t, q, f32x2_t{-0.254829592f, -0.254829592f});
    q = q * t;
    const f32x2_t s = x * f32x2_t{0.84932180028801904f, 0.84932180028801904f};  // sqrt(log2(e) / 2)
    const f32x2_t ss = s * -s;
    f32x2_t e;
    e.x = __builtin_amdgcn_exp2f(ss.x);  // exp(-x^2 / 2)
    e.y = __builtin_amdgcn_exp2f(ss.y);
    const f32x2_t w = __builtin_elementwise_fma(q, e, one);  // erf(|x| / sqrt 2)
    return __builtin_elementwise_fma(ax, w, x);              // x + |x| erf(|x| / sqrt 2) = 2 gelu(x)
}

// 2 * gelu(x) for the GEMM epilogues, two values at a time, in the logistic form gelu(x) = x * sigma(x q(x^2)) with a
// quadratic q fitted (minimax over |x| <= 7.25, where sigma has saturated to 1 - 7e-12) to the exact erf GELU:
// |err| <= 2.6e-5 absolute, evaluated in fp32 -- on N(0, 1.5^2) pre-activations rounded to bf16 the relative L2
// distance from the exact function is the rounding's own 1.6555e-3 to five digits.  6 packed + 2 plain + 4 transcendental
// issues per pair = 32 cycles per value against 42 for gelu_erf_x2 (11 + 2 + 4): the GELU epilogue of fc1 is bound by
// exactly this count.  The coefficients carry -log2(e), so the exponential is a bare v_exp_f32.
__device__ __forceinline__ f32x2_t gelu2_x2(f32x2_t x) {
    f32x2_t xc;
    xc.x = __builtin_amdgcn_fmed3f(x.x, -7.25f, 7.25f);
    xc.y = __builtin_amdgcn_fmed3f(x.y, -7.25f, 7.25f);
    const f32x2_t s = xc * xc;
    f32x2_t q = __builtin_elementwise_fma(s, f32x2_t{0.001014263f, 0.001014263f}, f32x2_t{-0.10677572f, -0.10677572f});
    q = __builtin_elementwise_fma(s, q, f32x2_t{-2.3011212f, -2.3011212f});
    const f32x2_t t = q * xc;  // -log2(e) x q(x^2)
    f32x2_t e;
    e.x = __builtin_amdgcn_exp2f(t.x);
    e.y = __builtin_amdgcn_exp2f(t.y);
    const f32x2_t d = __builtin_elementwise_fma(e, f32x2_t{0.5f, 0.5f}, f32x2_t{0.5f, 0.5f});  // (1 + e) / 2
    f32x2_t r;
    r.x = __builtin_amdgcn_rcpf(d.x);
    r.y = __builtin_amdgcn_rcpf(d.y);
    return x * r;  // 2 x sigma(.)
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

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{})
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// LDS byte address of a pointer into __shared__ memory (operand of hand-written ds_* instructions)
__device__ __forceinline__ uint32_t lds_addr32(const void* p) { return (uint32_t)(uintptr_t)LDS_PTR(p); }

// A wave-uniform pointer made an opaque scalar (v_readfirstlane of both halves), so that `p + 32-bit lane offset` is
// addressed as SGPR base + VGPR offset instead of a 64-bit VGPR pair.
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

// Opt a kernel in to `bytes` of dynamic LDS once per kernel and device (`done`: one bit per device, a function-local
// static of the launcher), instead of a hipFuncSetAttribute call in front of every launch.
inline int asrk_max_lds_once(const void* kern, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ASRK_ERR_HIP;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return ASRK_OK;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return ASRK_ERR_HIP;
    done.fetch_or(bit, std::memory_order_release);
    return ASRK_OK;
}

#define ASRK_CHECK_HIP(expr)                                  \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return ASRK_ERR_HIP;            \
    } while (0)
