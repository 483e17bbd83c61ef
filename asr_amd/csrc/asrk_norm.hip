// Memory-bound row kernels: LayerNorm, RMSNorm, casts, layout changes.  One wave per row, 16-byte
// accesses, fp32 statistics (two-pass in registers: mean, then centred variance).
#include "asrk_common.h"
#include "asrk_gemm.h"
#include "asrk_prof.h"

// Which multiply-adds fuse is written out (fmaf) and the compiler's own contraction is off for this file: the same row
// must come out bit-identical from every instantiation (generic / width-specialised LayerNorm, the LayerNorm + RMSNorm
// kernel against the two separate ones) -- tests compare them with torch.equal.
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float sumsq4(float a0, float a1, float a2, float a3) {
    return fmaf(a0, a0, a1 * a1) + fmaf(a2, a2, a3 * a3);
}

constexpr int MAX_V4 = 8;  // per-lane float4 chunks: D <= 64 * 4 * 8 = 2048

// LayerNorm(x) * w + b, eps inside the sqrt (torch.nn.LayerNorm; HF modeling_whisper.py:392,402,642)
// ADD: the residual step is fused in front, x += delta (delta = the bf16 output of the out-proj / fc2 GEMM, bias
// included; HF modeling_whisper.py:399,407), and the updated fp32 row is written back unless this is the
// encoder's last norm (KEEP_X false).  The fp32 read-modify-write of the residual stream thereby runs in this
// streaming kernel at >5 TB/s instead of in a GEMM epilogue, where every CU waits on it at once.
// NV: float4 chunks per lane when D == 256 NV exactly (768 -> 3, 1280 -> 5: the encoder widths; no per-chunk bounds
// test, no dead iterations -- at D = 1280 the generic form ran at 0.71 of the HBM peak against 0.81 at D = 768), or 0
// for any D % 4 == 0 up to 2048.
template <bool OUT_BF16, bool ADD, bool KEEP_X, int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(float* x, const bf16_t* __restrict__ delta,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        void* __restrict__ out, int M, int D, float eps) {
    constexpr int NI = NV ? NV : MAX_V4;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = NV ? 64 * NV : D >> 2;  // float4 per row
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    float4 v[NI];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        if (NV || c < nv) {
            if (ADD) {
                // the fp32 residual stream and the delta are touched once per layer step: non-temporal, so that what
                // stays in the memory-side cache is the bf16 output the next GEMM reads
                const f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr) + c);
                const bf16x4 dv = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(delta + (size_t)row * D) + c);
                f32x4 sum = xv + f32x4{(float)dv[0], (float)dv[1], (float)dv[2], (float)dv[3]};
                v[i] = float4{sum[0], sum[1], sum[2], sum[3]};
                if (KEEP_X) __builtin_nontemporal_store(sum, reinterpret_cast<f32x4*>(x + (size_t)row * D) + c);
            } else {
                v[i] = xr[c];
            }
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        if (NV || c < nv) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += sumsq4(a0, a1, a2, a3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int c = lane + 64 * i;
        if (NV || c < nv) {
            const float4 ww = reinterpret_cast<const float4*>(w)[c];
            const float4 bb = reinterpret_cast<const float4*>(b)[c];
            float4 o;
            o.x = fmaf((v[i].x - mean) * rstd, ww.x, bb.x);
            o.y = fmaf((v[i].y - mean) * rstd, ww.y, bb.y);
            o.z = fmaf((v[i].z - mean) * rstd, ww.z, bb.z);
            o.w = fmaf((v[i].w - mean) * rstd, ww.w, bb.w);
            if (OUT_BF16) {
                bf16x4 p;
                p[0] = (bf16_t)o.x; p[1] = (bf16_t)o.y; p[2] = (bf16_t)o.z; p[3] = (bf16_t)o.w;
                reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(out) + (size_t)row * D)[c] = p;
            } else {
                reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + (size_t)row * D)[c] = o;
            }
        }
    }
}

// LlamaRMSNorm (HF models/llama/modeling_llama.py:62-67): x * rsqrt(mean(x^2) + eps) in fp32, then
// gamma * (.) ; emitted as the bf16 operand of linear_1.  rstd is kept for the backward.
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                      bf16_t* __restrict__ xn, float* __restrict__ rstd_out, int M,
                                                      int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    float4 v[MAX_V4];
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            v[i] = xr[c];
            q += sumsq4(v[i].x, v[i].y, v[i].z, v[i].w);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c];
            bf16x4 p;
            p[0] = (bf16_t)(g.x * (v[i].x * rstd));
            p[1] = (bf16_t)(g.y * (v[i].y * rstd));
            p[2] = (bf16_t)(g.z * (v[i].z * rstd));
            p[3] = (bf16_t)(g.w * (v[i].w * rstd));
            reinterpret_cast<bf16x4*>(xn + (size_t)row * D)[c] = p;
        }
    }
}

// The encoder -> projector hand-off in one pass over the row (SURVEY 8 a8 + a9's prologue): x += delta, y = the encoder's
// final LayerNorm of it (HF modeling_whisper.py:642), then the projector's LlamaRMSNorm of y (src/modeling.py:78) emitted
// as the bf16 operand of linear_1, with its rstd for the backward.  y itself (last_hidden_state, fp32) is written only
// when the caller wants it (`y_out`: the training path keeps it for d gamma); without it the 4 bytes/element write and
// the RMSNorm's 4 bytes/element re-read are gone.  Same arithmetic, in the same order, as the two separate kernels.
template <bool ADD>
__global__ __launch_bounds__(256) void layernorm_rms_kernel(const float* __restrict__ x, const bf16_t* __restrict__ delta,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float* __restrict__ y_out, const float* __restrict__ gamma,
                                                            bf16_t* __restrict__ xn, float* __restrict__ rstd_out, int M, int D,
                                                            float eps, float rms_eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D >> 2;
    float4 v[MAX_V4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const f32x4 xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + (size_t)row * D) + c);
            f32x4 sum = xv;
            if (ADD) {
                const bf16x4 dv = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(delta + (size_t)row * D) + c);
                sum = xv + f32x4{(float)dv[0], (float)dv[1], (float)dv[2], (float)dv[3]};
            }
            v[i] = float4{sum[0], sum[1], sum[2], sum[3]};
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += sumsq4(a0, a1, a2, a3);
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 ww = reinterpret_cast<const float4*>(w)[c];
            const float4 bb = reinterpret_cast<const float4*>(b)[c];
            float4 o;
            o.x = fmaf((v[i].x - mean) * rstd, ww.x, bb.x);
            o.y = fmaf((v[i].y - mean) * rstd, ww.y, bb.y);
            o.z = fmaf((v[i].z - mean) * rstd, ww.z, bb.z);
            o.w = fmaf((v[i].w - mean) * rstd, ww.w, bb.w);
            if (y_out) reinterpret_cast<float4*>(y_out + (size_t)row * D)[c] = o;
            v[i] = o;
            q2 += sumsq4(o.x, o.y, o.z, o.w);
        }
    }
    const float rstd2 = rsqrtf(wave_sum(q2) / (float)D + rms_eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rstd2;
#pragma unroll
    for (int i = 0; i < MAX_V4; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 g = reinterpret_cast<const float4*>(gamma)[c];
            bf16x4 p;
            p[0] = (bf16_t)(g.x * (v[i].x * rstd2));
            p[1] = (bf16_t)(g.y * (v[i].y * rstd2));
            p[2] = (bf16_t)(g.z * (v[i].z * rstd2));
            p[3] = (bf16_t)(g.w * (v[i].w * rstd2));
            reinterpret_cast<bf16x4*>(xn + (size_t)row * D)[c] = p;
        }
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                            size_t n4, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        bf16x4 p;
        p[0] = (bf16_t)(v.x * scale); p[1] = (bf16_t)(v.y * scale);
        p[2] = (bf16_t)(v.z * scale); p[3] = (bf16_t)(v.w * scale);
        reinterpret_cast<bf16x4*>(dst)[i] = p;
    }
}

template <bool OUT_BF16, bool ADD, bool KEEP_X>
void launch_ln(dim3 grid, hipStream_t st, float* x, const bf16_t* delta, const float* w, const float* b, void* out, int M, int D,
               float eps) {
    if (D == 768)
        hipLaunchKernelGGL((layernorm_kernel<OUT_BF16, ADD, KEEP_X, 3>), grid, dim3(256), 0, st, x, delta, w, b, out, M, D, eps);
    else if (D == 1280)
        hipLaunchKernelGGL((layernorm_kernel<OUT_BF16, ADD, KEEP_X, 5>), grid, dim3(256), 0, st, x, delta, w, b, out, M, D, eps);
    else
        hipLaunchKernelGGL((layernorm_kernel<OUT_BF16, ADD, KEEP_X, 0>), grid, dim3(256), 0, st, x, delta, w, b, out, M, D, eps);
}

}  // namespace

int asrk_launch_layernorm(const float* x, const float* w, const float* b, void* out, bool out_bf16, int M, int D,
                          float eps, hipStream_t st) {
    if (!x || !w || !b || !out) return ASRK_ERR_NULL;
    if (M <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * MAX_V4) return ASRK_ERR_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)b & 15) || ((uintptr_t)out & 15)) return ASRK_ERR_ALIGN;
    const dim3 grid((M + 3) / 4);
    AsrkProfScope prof(ASRK_PROF_NORM, (double)M * D * (out_bf16 ? 6.0 : 8.0), st);
    float* xm = const_cast<float*>(x);  // the plain variant never writes x
    if (out_bf16)
        launch_ln<true, false, false>(grid, st, xm, nullptr, w, b, out, M, D, eps);
    else
        launch_ln<false, false, false>(grid, st, xm, nullptr, w, b, out, M, D, eps);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

// x += delta (fp32 += bf16), then LayerNorm of the updated row.  keep_x false skips the write-back of x.
int asrk_launch_add_layernorm(float* x, const bf16_t* delta, const float* w, const float* b, void* out, bool out_bf16,
                              bool keep_x, int M, int D, float eps, hipStream_t st) {
    if (!x || !delta || !w || !b || !out) return ASRK_ERR_NULL;
    if (M <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * MAX_V4) return ASRK_ERR_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)delta & 7) || ((uintptr_t)w & 15) || ((uintptr_t)b & 15) || ((uintptr_t)out & 15))
        return ASRK_ERR_ALIGN;
    const dim3 grid((M + 3) / 4);
    // algorithmic bytes per element: x read 4 + delta read 2 + x write 4 (if kept) + output 2 or 4
    AsrkProfScope prof(ASRK_PROF_NORM, (double)M * D * (6.0 + (keep_x ? 4.0 : 0.0) + (out_bf16 ? 2.0 : 4.0)), st);
    if (out_bf16 && keep_x)
        launch_ln<true, true, true>(grid, st, x, delta, w, b, out, M, D, eps);
    else if (out_bf16)
        launch_ln<true, true, false>(grid, st, x, delta, w, b, out, M, D, eps);
    else if (keep_x)
        launch_ln<false, true, true>(grid, st, x, delta, w, b, out, M, D, eps);
    else
        launch_ln<false, true, false>(grid, st, x, delta, w, b, out, M, D, eps);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

int asrk_launch_layernorm_rms(const float* x, const bf16_t* delta, const float* w, const float* b, float* y_out,
                              const AsrkRmsTail& t, int M, int D, float eps, hipStream_t st) {
    if (!x || !w || !b || !t.gamma || !t.xn) return ASRK_ERR_NULL;
    if (M <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * MAX_V4) return ASRK_ERR_SHAPE;
    if (((uintptr_t)x & 15) || ((uintptr_t)delta & 7) || ((uintptr_t)w & 15) || ((uintptr_t)b & 15) || ((uintptr_t)y_out & 15) ||
        ((uintptr_t)t.gamma & 15) || ((uintptr_t)t.xn & 7))
        return ASRK_ERR_ALIGN;
    const dim3 grid((M + 3) / 4);
    // algorithmic bytes per element: x read 4 (+ delta read 2) (+ y write 4) + xn write 2
    AsrkProfScope prof(ASRK_PROF_NORM, (double)M * D * (6.0 + (delta ? 2.0 : 0.0) + (y_out ? 4.0 : 0.0)), st);
    if (delta)
        hipLaunchKernelGGL((layernorm_rms_kernel<true>), grid, dim3(256), 0, st, x, delta, w, b, y_out, t.gamma, t.xn, t.rstd, M, D,
                           eps, t.eps);
    else
        hipLaunchKernelGGL((layernorm_rms_kernel<false>), grid, dim3(256), 0, st, x, delta, w, b, y_out, t.gamma, t.xn, t.rstd, M, D,
                           eps, t.eps);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

int asrk_launch_rmsnorm_bf16(const float* x, const float* gamma, bf16_t* xn, float* rstd, int M, int D, float eps,
                             hipStream_t st) {
    if (!x || !gamma || !xn) return ASRK_ERR_NULL;
    if (M <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * MAX_V4) return ASRK_ERR_SHAPE;
    hipLaunchKernelGGL(rmsnorm_kernel, dim3((M + 3) / 4), dim3(256), 0, st, x, gamma, xn, rstd, M, D, eps);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

int asrk_launch_cast_f32_bf16(const float* src, bf16_t* dst, size_t n, float scale, hipStream_t st) {
    if (!src || !dst) return ASRK_ERR_NULL;
    if (n & 3) return ASRK_ERR_SHAPE;
    const size_t n4 = n >> 2;
    if (n4 == 0) return ASRK_OK;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n4, scale);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

extern "C" int asrk_layernorm(const float* x, const float* w, const float* b, void* out, int out_dtype, int M, int D,
                              float eps, void* stream) {
    if (out_dtype != ASRK_F32 && out_dtype != ASRK_BF16) return ASRK_ERR_UNSUPPORTED;
    return asrk_launch_layernorm(x, w, b, out, out_dtype == ASRK_BF16, M, D, eps, (hipStream_t)stream);
}

extern "C" int asrk_add_layernorm(float* x, const void* delta, const float* w, const float* b, void* out, int out_dtype,
                                  int keep_x, int M, int D, float eps, void* stream) {
    if (out_dtype != ASRK_F32 && out_dtype != ASRK_BF16) return ASRK_ERR_UNSUPPORTED;
    return asrk_launch_add_layernorm(x, (const bf16_t*)delta, w, b, out, out_dtype == ASRK_BF16, keep_x != 0, M, D, eps,
                                     (hipStream_t)stream);
}
