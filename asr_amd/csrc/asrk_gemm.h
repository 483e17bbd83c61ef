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
};

// out = (((acc + bias) -> gelu?) * alpha [* gelu'(pre)]) + res
int asrk_gemm_launch(const GemmArgs& a, bool gelu, bool out_bf16, hipStream_t st);

// small helpers implemented in asrk_norm.hip
int asrk_launch_layernorm(const float* x, const float* w, const float* b, void* out, bool out_bf16, int M, int D,
                          float eps, hipStream_t st);
int asrk_launch_rmsnorm_bf16(const float* x, const float* gamma, bf16_t* xn, float* rstd, int M, int D, float eps,
                             hipStream_t st);
int asrk_launch_cast_f32_bf16(const float* src, bf16_t* dst, size_t n, float scale, hipStream_t st);
int asrk_launch_attention(const bf16_t* qkv, bf16_t* out, int B, int T, int H, hipStream_t st);
