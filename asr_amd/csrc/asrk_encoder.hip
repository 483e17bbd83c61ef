// Whisper-style audio encoder forward as a composition of the gfx950 kernels.
//
// Replaces WhisperEncoder.forward (reference src/modeling.py:33-40) -> HF WhisperEncoder.forward
// (transformers/models/whisper/modeling_whisper.py:592-646):
//   gelu(conv1) -> gelu(conv2, stride 2) -> permute -> + embed_positions -> L x encoder layer
//   (:379-413: LN -> attention -> +res -> LN -> fc1 -> gelu -> fc2 -> +res) -> final LN.
// The encoder is frozen and runs under no_grad in the reference (src/modeling.py:27-28,36), so
// only the forward exists.  Matmuls run in bf16 on MFMA with fp32 accumulation; LayerNorm,
// softmax statistics and the residual stream stay fp32 (the same split the reference gets from
// accelerate's bf16 autocast, configs/accelerate/a40.yaml:8).
//
// HBM layout of the workspace (B clips, M = 1500 B rows):
//   featT  bf16 [B][3002][n_mels]   time-major mel features, zero rows around each clip
//   h1     bf16 [B][3002][d]        gelu(conv1), time-major, zero rows around each clip
//   resid  f32  [M][d]              residual stream
//   xn     bf16 [M][d]              LayerNorm output (GEMM operand)
//   qkv    bf16 [M][3d]             q | k | v
//   ao     bf16 [M][d]              attention output
//   ffn    bf16 [M][ffn]            gelu(fc1)
#include "asrk_common.h"
#include "asrk_gemm.h"

#include <vector>

namespace {

constexpr int T_IN = 3000, T_ENC = 1500, T_PAD = T_IN + 2;

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// conv weight [co][ci][3] f32 -> [co][kk * C + ci] bf16, K padded with zeros to Kp
__global__ void pack_conv_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int C, int Kp) {
    const int co = blockIdx.x;
    for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
        float v = 0.f;
        if (k < 3 * C) {
            const int kk = k / C, ci = k % C;
            v = w[((size_t)co * C + ci) * 3 + kk];
        }
        out[(size_t)co * Kp + k] = (bf16_t)v;
    }
}

__global__ void scale_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src ? src[i] * scale : 0.f;
}

// feats f32 [B][C][3000] -> featT bf16 [B][3002][C] rows 1..3000 (tile: 64 frames x all C)
__global__ __launch_bounds__(256) void transpose_feats_kernel(const float* __restrict__ feats, bf16_t* __restrict__ featT,
                                                              int C) {
    extern __shared__ float tile[];  // [64][C + 1]
    const int t0 = blockIdx.x * 64, b = blockIdx.y;
    const float* src = feats + (size_t)b * C * T_IN;
    const int nt = min(64, T_IN - t0);
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        const int t = i & 63, c = i >> 6;
        if (t < nt) tile[t * (C + 1) + c] = src[(size_t)c * T_IN + t0 + t];
    }
    __syncthreads();
    bf16_t* dst = featT + ((size_t)b * T_PAD + 1 + t0) * C;
    const int half = C >> 1;
    for (int i = threadIdx.x; i < half * 64; i += 256) {
        const int c2 = i % half, t = i / half;
        if (t < nt) {
            bf16x2 v;
            v[0] = (bf16_t)tile[t * (C + 1) + 2 * c2];
            v[1] = (bf16_t)tile[t * (C + 1) + 2 * c2 + 1];
            *reinterpret_cast<bf16x2*>(dst + (size_t)t * C + 2 * c2) = v;
        }
    }
}

// zero the rows around each clip of a [B][3002][row] bf16 buffer (+ `tail` elements after the end)
__global__ void zero_pad_rows_kernel(bf16_t* __restrict__ buf, int row, int B, int tail) {
    const int b = blockIdx.x;  // 0..B
    size_t start, count;
    if (b == 0) { start = 0; count = row; }
    else if (b < B) { start = ((size_t)b * T_PAD - 1) * row; count = 2 * (size_t)row; }
    else { start = ((size_t)B * T_PAD - 1) * row; count = (size_t)row + tail; }
    for (size_t i = threadIdx.x; i < count; i += blockDim.x) buf[start + i] = (bf16_t)0.f;
}

struct LayerPacked {
    const bf16_t *wqkv, *wo, *w1, *w2;
    const float *bqkv, *bo, *b1, *b2, *ln1w, *ln1b, *ln2w, *ln2b;
};

}  // namespace

struct asrk_encoder {
    asrk_encoder_dims dims;
    int k1p;  // padded K of conv1
    char* blob;
    const bf16_t *conv1w, *conv2w;
    const float *conv1b, *conv2b, *pos, *lnw, *lnb;
    std::vector<LayerPacked> layers;
};

namespace {
struct WsLayout {
    size_t featT, h1, resid, xn, qkv, ao, delta, ffn, total;
};
WsLayout ws_layout(const asrk_encoder_dims& d, int B) {
    WsLayout w;
    const size_t M = (size_t)B * T_ENC;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    w.featT = take(((size_t)B * T_PAD + 1) * d.n_mels * 2 + 128);
    w.h1 = take(((size_t)B * T_PAD + 1) * d.d_model * 2 + 128);
    w.resid = take(M * d.d_model * 4);
    w.xn = take(M * d.d_model * 2);
    w.qkv = take(M * 3 * d.d_model * 2);
    w.ao = take(M * d.d_model * 2);
    w.delta = take(M * d.d_model * 2);
    w.ffn = take(M * d.ffn_dim * 2);
    w.total = off;
    return w;
}
}  // namespace

extern "C" size_t asrk_encoder_workspace_bytes(const asrk_encoder* enc, int B) {
    if (!enc || B <= 0) return 0;
    return ws_layout(enc->dims, B).total;
}

extern "C" int asrk_encoder_create(const asrk_encoder_dims* dims, const asrk_encoder_weights* w, void* stream,
                                   asrk_encoder** out) {
    if (!dims || !w || !out || !w->layers) return ASRK_ERR_NULL;
    const int d = dims->d_model, L = dims->n_layers, H = dims->n_heads, F = dims->ffn_dim, C = dims->n_mels;
    if (d <= 0 || L < 0 || H <= 0 || F <= 0 || C <= 0) return ASRK_ERR_SHAPE;
    if (d != H * 64 || (d % 64) || (F % 64) || (C % 8) || d > 2048) return ASRK_ERR_SHAPE;
    if (!w->conv1_w || !w->conv1_b || !w->conv2_w || !w->conv2_b || !w->pos || !w->ln_w || !w->ln_b) return ASRK_ERR_NULL;
    hipStream_t st = (hipStream_t)stream;
    const int k1p = (int)align_up((size_t)3 * C, 64);

    // ---- size the packed blob
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_c1w = take((size_t)d * k1p * 2), o_c2w = take((size_t)d * 3 * d * 2);
    const size_t o_c1b = take((size_t)d * 4), o_c2b = take((size_t)d * 4), o_pos = take((size_t)T_ENC * d * 4);
    const size_t o_lnw = take((size_t)d * 4), o_lnb = take((size_t)d * 4);
    struct LOff { size_t wqkv, wo, w1, w2, bqkv, bo, b1, b2, ln1w, ln1b, ln2w, ln2b; };
    std::vector<LOff> lo(L);
    for (int i = 0; i < L; ++i) {
        lo[i].wqkv = take((size_t)3 * d * d * 2); lo[i].wo = take((size_t)d * d * 2);
        lo[i].w1 = take((size_t)F * d * 2); lo[i].w2 = take((size_t)d * F * 2);
        lo[i].bqkv = take((size_t)3 * d * 4); lo[i].bo = take((size_t)d * 4);
        lo[i].b1 = take((size_t)F * 4); lo[i].b2 = take((size_t)d * 4);
        lo[i].ln1w = take((size_t)d * 4); lo[i].ln1b = take((size_t)d * 4);
        lo[i].ln2w = take((size_t)d * 4); lo[i].ln2b = take((size_t)d * 4);
    }
    char* blob = nullptr;
    if (hipMalloc(&blob, off) != hipSuccess) return ASRK_ERR_HIP;

    int rc = ASRK_OK;
    auto castw = [&](const float* src, size_t o, size_t n, float scale) {
        if (rc != ASRK_OK) return;
        if (!src) { rc = ASRK_ERR_NULL; return; }
        rc = asrk_launch_cast_f32_bf16(src, (bf16_t*)(blob + o), n, scale, st);
    };
    auto copyf = [&](const float* src, size_t o, int n, float scale, bool allow_null = false) {
        if (rc != ASRK_OK) return;
        if (!src && !allow_null) { rc = ASRK_ERR_NULL; return; }
        hipLaunchKernelGGL(scale_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, (float*)(blob + o), n, scale);
        if (hipGetLastError() != hipSuccess) rc = ASRK_ERR_HIP;
    };
    hipLaunchKernelGGL(pack_conv_kernel, dim3(d), dim3(256), 0, st, w->conv1_w, (bf16_t*)(blob + o_c1w), C, k1p);
    hipLaunchKernelGGL(pack_conv_kernel, dim3(d), dim3(256), 0, st, w->conv2_w, (bf16_t*)(blob + o_c2w), d, 3 * d);
    if (hipGetLastError() != hipSuccess) rc = ASRK_ERR_HIP;
    copyf(w->conv1_b, o_c1b, d, 1.f);
    copyf(w->conv2_b, o_c2b, d, 1.f);
    copyf(w->pos, o_pos, T_ENC * d, 1.f);
    copyf(w->ln_w, o_lnw, d, 1.f);
    copyf(w->ln_b, o_lnb, d, 1.f);
    // head_dim^-0.5 for head_dim 64, applied to q_proj output (:309), times log2(e): the attention kernel works in
    // log2 units (asrk_launch_attention, q_log2e)
    const float qscale = 0.125f * 1.4426950408889634f;
    for (int i = 0; i < L && rc == ASRK_OK; ++i) {
        const asrk_encoder_layer_weights& lw = w->layers[i];
        castw(lw.q_w, lo[i].wqkv, (size_t)d * d, qscale);
        castw(lw.k_w, lo[i].wqkv + (size_t)d * d * 2, (size_t)d * d, 1.f);
        castw(lw.v_w, lo[i].wqkv + (size_t)2 * d * d * 2, (size_t)d * d, 1.f);
        castw(lw.o_w, lo[i].wo, (size_t)d * d, 1.f);
        castw(lw.fc1_w, lo[i].w1, (size_t)F * d, 1.f);
        castw(lw.fc2_w, lo[i].w2, (size_t)d * F, 1.f);
        copyf(lw.q_b, lo[i].bqkv, d, qscale);
        copyf(nullptr, lo[i].bqkv + (size_t)d * 4, d, 1.f, true);  // k_proj has no bias (:279)
        copyf(lw.v_b, lo[i].bqkv + (size_t)2 * d * 4, d, 1.f);
        copyf(lw.o_b, lo[i].bo, d, 1.f);
        copyf(lw.fc1_b, lo[i].b1, F, 1.f);
        copyf(lw.fc2_b, lo[i].b2, d, 1.f);
        copyf(lw.ln1_w, lo[i].ln1w, d, 1.f);
        copyf(lw.ln1_b, lo[i].ln1b, d, 1.f);
        copyf(lw.ln2_w, lo[i].ln2w, d, 1.f);
        copyf(lw.ln2_b, lo[i].ln2b, d, 1.f);
    }
    if (rc != ASRK_OK) { (void)hipFree(blob); return rc; }

    asrk_encoder* e = new asrk_encoder();
    e->dims = *dims;
    e->k1p = k1p;
    e->blob = blob;
    e->conv1w = (const bf16_t*)(blob + o_c1w);
    e->conv2w = (const bf16_t*)(blob + o_c2w);
    e->conv1b = (const float*)(blob + o_c1b);
    e->conv2b = (const float*)(blob + o_c2b);
    e->pos = (const float*)(blob + o_pos);
    e->lnw = (const float*)(blob + o_lnw);
    e->lnb = (const float*)(blob + o_lnb);
    e->layers.resize(L);
    for (int i = 0; i < L; ++i) {
        LayerPacked& p = e->layers[i];
        p.wqkv = (const bf16_t*)(blob + lo[i].wqkv); p.wo = (const bf16_t*)(blob + lo[i].wo);
        p.w1 = (const bf16_t*)(blob + lo[i].w1); p.w2 = (const bf16_t*)(blob + lo[i].w2);
        p.bqkv = (const float*)(blob + lo[i].bqkv); p.bo = (const float*)(blob + lo[i].bo);
        p.b1 = (const float*)(blob + lo[i].b1); p.b2 = (const float*)(blob + lo[i].b2);
        p.ln1w = (const float*)(blob + lo[i].ln1w); p.ln1b = (const float*)(blob + lo[i].ln1b);
        p.ln2w = (const float*)(blob + lo[i].ln2w); p.ln2b = (const float*)(blob + lo[i].ln2b);
    }
    *out = e;
    return ASRK_OK;
}

extern "C" void asrk_encoder_destroy(asrk_encoder* enc) {
    if (!enc) return;
    (void)hipFree(enc->blob);
    delete enc;
}

extern "C" int asrk_encoder_forward(const asrk_encoder* enc, const float* feats, int B, int n_frames, float* out,
                                    void* ws, size_t ws_bytes, void* stream) {
    if (!out) return ASRK_ERR_NULL;
    return asrk_encoder_forward_impl(enc, feats, B, n_frames, out, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

int asrk_encoder_forward_impl(const asrk_encoder* enc, const float* feats, int B, int n_frames, float* out,
                              const AsrkRmsTail* tail, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!enc || !feats || !ws || (!out && !tail)) return ASRK_ERR_NULL;
    if (n_frames != T_IN) return ASRK_ERR_LENGTH;
    if (B <= 0 || B > 65535) return ASRK_ERR_SHAPE;
    if (((uintptr_t)feats & 15) || ((uintptr_t)out & 15) || ((uintptr_t)ws & 255)) return ASRK_ERR_ALIGN;
    const asrk_encoder_dims& dm = enc->dims;
    // the fused tail writes [M][d_model] bf16 into a slot the caller sized [M][tail->dim] and reads gamma[d_model]
    if (tail && (tail->dim != dm.d_model || !tail->gamma || !tail->xn)) return tail->dim != dm.d_model ? ASRK_ERR_SHAPE : ASRK_ERR_NULL;
    const int d = dm.d_model, F = dm.ffn_dim, C = dm.n_mels, H = dm.n_heads;
    const WsLayout wl = ws_layout(dm, B);
    if (ws_bytes < wl.total) return ASRK_ERR_WORKSPACE;
    char* w = (char*)ws;
    bf16_t* featT = (bf16_t*)(w + wl.featT);
    bf16_t* h1 = (bf16_t*)(w + wl.h1);
    float* resid = (float*)(w + wl.resid);
    bf16_t* xn = (bf16_t*)(w + wl.xn);
    bf16_t* qkv = (bf16_t*)(w + wl.qkv);
    bf16_t* ao = (bf16_t*)(w + wl.ao);
    bf16_t* delta = (bf16_t*)(w + wl.delta);
    bf16_t* ffn = (bf16_t*)(w + wl.ffn);
    const int M = B * T_ENC;
    int rc;

    // ---- stem input: time-major bf16 features with zero rows around each clip
    hipLaunchKernelGGL(zero_pad_rows_kernel, dim3(B + 1), dim3(256), 0, st, featT, C, B, 64);
    hipLaunchKernelGGL(zero_pad_rows_kernel, dim3(B + 1), dim3(256), 0, st, h1, d, B, 64);
    hipLaunchKernelGGL(transpose_feats_kernel, dim3((T_IN + 63) / 64, B), dim3(256), (size_t)64 * (C + 1) * 4, st, feats,
                       featT, C);
    if (hipGetLastError() != hipSuccess) return ASRK_ERR_HIP;

    // ---- conv1 (k3, pad 1) + GELU as a GEMM over overlapping rows: K = 3 C (padded to k1p)
    {
        GemmArgs a{};
        a.X = featT; a.W = enc->conv1w; a.bias = enc->conv1b; a.res = nullptr; a.out = h1;
        a.M = B * T_IN; a.N = d; a.K = enc->k1p;
        a.x_rpb = T_IN; a.x_bs = (int64_t)T_PAD * C; a.x_rs = C;
        a.o_rpb = T_IN; a.o_bs = (int64_t)T_PAD * d; a.o_rs = d; a.o_off = d;
        a.ldr = 0; a.res_rows = 1; a.alpha = 1.f; a.splits = 1;
        if ((rc = asrk_gemm_launch(a, true, true, st)) != ASRK_OK) return rc;
    }
    // ---- conv2 (k3, stride 2, pad 1) + GELU + positional embedding -> fp32 residual stream
    {
        GemmArgs a{};
        a.X = h1; a.W = enc->conv2w; a.bias = enc->conv2b; a.res = enc->pos; a.out = resid;
        a.M = M; a.N = d; a.K = 3 * d;
        a.x_rpb = T_ENC; a.x_bs = (int64_t)T_PAD * d; a.x_rs = 2 * d;
        a.o_rpb = M; a.o_bs = 0; a.o_rs = d; a.o_off = 0;
        a.ldr = d; a.res_rows = T_ENC; a.alpha = 1.f; a.splits = 1;
        if ((rc = asrk_gemm_launch(a, true, false, st)) != ASRK_OK) return rc;
    }
    auto linear = [&](const bf16_t* x, int K, const bf16_t* wt, const float* bias, int N, void* o, bool gelu, bool obf16,
                      const float* res) {
        GemmArgs a{};
        a.X = x; a.W = wt; a.bias = bias; a.res = res; a.out = o;
        a.M = M; a.N = N; a.K = K;
        a.x_rpb = M; a.x_bs = 0; a.x_rs = K;
        a.o_rpb = M; a.o_bs = 0; a.o_rs = N; a.o_off = 0;
        a.ldr = N; a.res_rows = M; a.alpha = 1.f; a.splits = 1;
        return asrk_gemm_launch(a, gelu, obf16, st);
    };
    // The residual stream `resid` stays fp32.  The out-proj and fc2 GEMMs emit their result (bias included) as a
    // bf16 delta, and the residual add x += delta is fused into the LayerNorm that follows it (ln2, the next
    // layer's ln1, or the encoder's final norm): the add is exact in fp32, only the delta is rounded to bf16,
    // like every other GEMM operand on this path.
    const size_t nl = enc->layers.size();
    if (nl == 0) {
        if (tail) return asrk_launch_layernorm_rms(resid, nullptr, enc->lnw, enc->lnb, out, *tail, M, d, 1e-5f, st);
        return asrk_launch_layernorm(resid, enc->lnw, enc->lnb, out, false, M, d, 1e-5f, st);
    }
    if ((rc = asrk_launch_layernorm(resid, enc->layers[0].ln1w, enc->layers[0].ln1b, xn, true, M, d, 1e-5f, st)) != ASRK_OK)
        return rc;
    for (size_t i = 0; i < nl; ++i) {
        const LayerPacked& L = enc->layers[i];
        if ((rc = linear(xn, d, L.wqkv, L.bqkv, 3 * d, qkv, false, true, nullptr)) != ASRK_OK) return rc;
        if ((rc = asrk_launch_attention(qkv, ao, B, T_ENC, H, true, st)) != ASRK_OK) return rc;
        if ((rc = linear(ao, d, L.wo, L.bo, d, delta, false, true, nullptr)) != ASRK_OK) return rc;
        if ((rc = asrk_launch_add_layernorm(resid, delta, L.ln2w, L.ln2b, xn, true, true, M, d, 1e-5f, st)) != ASRK_OK) return rc;
        if ((rc = linear(xn, d, L.w1, L.b1, F, ffn, true, true, nullptr)) != ASRK_OK) return rc;
        if ((rc = linear(ffn, F, L.w2, L.b2, d, delta, false, true, nullptr)) != ASRK_OK) return rc;
        if (i + 1 < nl) {
            const LayerPacked& Ln = enc->layers[i + 1];
            rc = asrk_launch_add_layernorm(resid, delta, Ln.ln1w, Ln.ln1b, xn, true, true, M, d, 1e-5f, st);
        } else if (tail) {
            rc = asrk_launch_layernorm_rms(resid, delta, enc->lnw, enc->lnb, out, *tail, M, d, 1e-5f, st);
        } else {
            rc = asrk_launch_add_layernorm(resid, delta, enc->lnw, enc->lnb, out, false, false, M, d, 1e-5f, st);
        }
        if (rc != ASRK_OK) return rc;
    }
    return ASRK_OK;
}
