// Internal launch interface of the bf16 MFMA GEMM (asrk_gemm.hip), shared by the encoder and
// projector compositions.
#pragma once
#include "asrk_common.h"

struct GemmArgs {
    const bf16_t* X;    // activations, row m at (m / x_rpb) * x_bs + (m % x_rpb) * x_rs, K contiguous
    const bf16_t* W;    // [N][K] row-major
    const float* bias;  // [N] or null
    const float* res;   // f32 residual rows (m % res_rows) * ldr, or null
    void* out;          // row m at (m / o_rpb) * o_bs + (m % o_rpb) * o_rs + o_off
    int M, N, K;
    int x_rpb;
    int64_t x_bs, x_rs;
    int o_rpb;
    int64_t o_bs, o_rs, o_off;
    int64_t ldr;
    int res_rows;
    float alpha;
    // optional: multiply by gelu'(pre[m][n]) (bf16, row stride ld_dg) after alpha - projector backward
    const bf16_t* dg_pre;
    int64_t ld_dg;
    // optional split-K: blockIdx.y = split s covers K tiles [s*kt_per_split, ...), writes out + s*o_split
    int splits;
    int64_t o_split;
    int nsplit;  // set by the 256-tile launcher: n-groups of the XCD grid (1, 2, 4, 8)
    int skew;    // set by the 256-tile launcher: start-up phase skew, in units of s_sleep(127) per phase step
};

// out = (((acc + bias) -> gelu?) * alpha [* gelu'(pre)]) + res
int asrk_gemm_launch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st);
int asrk_gemm256_launch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st);  // 256x256 tiles, splits == 1
int asrk_gemm256p_launch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st);  // same, persistent workgroups
bool asrk_gemm256p_fits(const GemmArgs& a);  // operands within the persistent kernel's 32-bit offsets

// Element part of the epilogue (everything except residual and store) for 4 consecutive n of row m.
// `bv` is the bias of those 4 columns, loaded once by the caller (it does not depend on the row: a load
// per accumulator made the epilogue a chain of 32 dependent L2 round trips, 9.8 k cycles per tile).
// 2 gelu of two values: the logistic form where the result is rounded to bf16 anyway (every bf16-output path uses the
// same one, so a clip's features do not depend on which tile kernel its batch size selects), the erf form (|err| <= 5e-7)
// for fp32 outputs (conv2 -> residual stream)
template <bool OUT_BF16>
__device__ __forceinline__ f32x2_t gemm_gelu2(f32x2_t x) { return OUT_BF16 ? gelu2_x2(x) : gelu_erf_x2(x); }

template <bool GELU, bool OUT_BF16>
__device__ __forceinline__ f32x4 gemm_epilogue_value(float alpha, const bf16_t* dg_pre, int64_t ld_dg, bool in_range, f32x4 v,
                                                     f32x4 bv, int m, int n) {
    v += bv;
    if (GELU) {
        const f32x2_t lo = gemm_gelu2<OUT_BF16>(f32x2_t{v[0], v[1]}), hi = gemm_gelu2<OUT_BF16>(f32x2_t{v[2], v[3]});
        v = f32x4{lo.x, lo.y, hi.x, hi.y} * (0.5f * alpha);
    } else {
        v *= alpha;
    }
    if (dg_pre && in_range) {
        const bf16x4 pz = *reinterpret_cast<const bf16x4*>(dg_pre + (int64_t)m * ld_dg + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_grad((float)pz[j]);
    }
    return v;
}
// The same from the argument block.  Kernels whose epilogue is unrolled should read alpha / dg_pre / ld_dg into
// locals once and call the form above: through the reference the compiler re-reads them from a scratch copy of
// the kernel arguments at every call (32 scratch loads, each behind an s_waitcnt vmcnt(0), per 256 x 256 tile).
template <bool GELU, bool OUT_BF16>
__device__ __forceinline__ f32x4 gemm_epilogue_value(const GemmArgs& a, f32x4 v, f32x4 bv, int m, int n) {
    return gemm_epilogue_value<GELU, OUT_BF16>(a.alpha, a.dg_pre, a.ld_dg, m < a.M && n < a.N, v, bv, m, n);
}

// Shared epilogue: v = 4 consecutive n (n .. n+3) of output row m.  `orow` is the element offset
// of the output row, `rrow` the residual row (or null).
template <bool GELU, bool OUT_BF16>
__device__ __forceinline__ void gemm_epilogue_store(const GemmArgs& a, f32x4 v, f32x4 bv, int m, int n, int64_t orow,
                                                    const float* rrow) {
    v += bv;
    if (GELU) {
        const f32x2_t lo = gemm_gelu2<OUT_BF16>(f32x2_t{v[0], v[1]}), hi = gemm_gelu2<OUT_BF16>(f32x2_t{v[2], v[3]});
        v = f32x4{lo.x, lo.y, hi.x, hi.y} * (0.5f * a.alpha);
    } else {
        v *= a.alpha;
    }
    if (a.dg_pre && m < a.M && n < a.N) {
        const bf16x4 pz = *reinterpret_cast<const bf16x4*>(a.dg_pre + (int64_t)m * a.ld_dg + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= gelu_erf_grad((float)pz[j]);
    }
    if (rrow) {
        const float4 rr = *reinterpret_cast<const float4*>(rrow + n);
        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
    }
    if (OUT_BF16) {
        bf16x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)v[j];
        *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(a.out) + orow + n) = o;
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.out) + orow + n) = v;
    }
}

// small helpers implemented in asrk_norm.hip
int asrk_launch_layernorm(const float* x, const float* w, const float* b, void* out, bool out_bf16, int M, int D,
                          float eps, hipStream_t st);
int asrk_launch_add_layernorm(float* x, const bf16_t* delta, const float* w, const float* b, void* out, bool out_bf16,
                              bool keep_x, int M, int D, float eps, hipStream_t st);
// The projector's RMSNorm fused behind the encoder's last LayerNorm (asrk_norm.hip: layernorm_rms_kernel)
struct AsrkRmsTail {
    const float* gamma;  // norm.weight of the projector
    bf16_t* xn;          // [M][D] bf16: gamma * y * rstd, the operand of linear_1
    float* rstd;         // [M] or null
    float eps;
    int dim;             // row width the caller sized xn / gamma for (the projector's audio_dim): must equal d_model
};
// x (+ delta, may be null) -> LayerNorm -> y (written to y_out unless null) -> RMSNorm -> t.xn, t.rstd
int asrk_launch_layernorm_rms(const float* x, const bf16_t* delta, const float* w, const float* b, float* y_out,
                              const AsrkRmsTail& t, int M, int D, float eps, hipStream_t st);
int asrk_launch_rmsnorm_bf16(const float* x, const float* gamma, bf16_t* xn, float* rstd, int M, int D, float eps,
                             hipStream_t st);
int asrk_launch_cast_f32_bf16(const float* src, bf16_t* dst, size_t n, float scale, hipStream_t st);
// q_log2e: q carries an extra factor log2(e) (folded into W_q / b_q by the caller)
// asrk_encoder.hip: the encoder forward; with `tail` its last norm also produces the projector's RMSNorm output and
// `out` (last_hidden_state, fp32) may be null
struct asrk_encoder;
int asrk_encoder_forward_impl(const asrk_encoder* enc, const float* feats, int B, int n_frames, float* out,
                              const AsrkRmsTail* tail, void* ws, size_t ws_bytes, hipStream_t st);
int asrk_launch_attention(const bf16_t* qkv, bf16_t* out, int B, int T, int H, bool q_log2e, hipStream_t st);
