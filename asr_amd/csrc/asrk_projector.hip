// AudioProjector forward / backward on gfx950.
//
// Replaces reference src/modeling.py:77-84 (forward) and the autograd backward of its five
// parameters (norm.weight, linear_1.{weight,bias}, linear_2.{weight,bias}):
//   y = 0.01 * ( gelu( (gamma * x * rsqrt(mean(x^2) + 1e-6)) W1^T + b1 ) W2^T + b2 )
// RMSNorm follows HF LlamaRMSNorm (transformers/models/llama/modeling_llama.py:62-67).
// GEMMs run in bf16 on MFMA with fp32 accumulation (what the reference gets under accelerate's
// bf16 autocast); statistics, bias/gamma gradients and every reduction are fp32.  fp32 master
// weights are cast to bf16 (and transposed for the backward) by asrk_projector_pack into a buffer the caller keeps until
// the parameters change; without one (params->packed NULL) every call makes its own copies in the workspace.
//
// saved (for backward), M rows:  rstd f32 [M] | xn bf16 [M][A] | pre bf16 [M][T] | h bf16 [M][T]
#include "asrk_common.h"
#include "asrk_gemm.h"

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr float RMS_EPS = 1e-6f;   // src/modeling.py:66
constexpr float OUT_SCALE = 0.01f; // src/modeling.py:83

// h = gelu(pre)  (bf16 -> bf16, 8 elements per lane)
__global__ __launch_bounds__(256) void gelu_bf16_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ h, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
        const bf16x8 v = reinterpret_cast<const bf16x8*>(pre)[i];
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)gelu_erf((float)v[j]);
        reinterpret_cast<bf16x8*>(h)[i] = o;
    }
}

// grad_out (f32 or bf16) * scale -> bf16
template <bool IN_BF16>
__global__ __launch_bounds__(256) void scale_to_bf16_kernel(const void* __restrict__ src, bf16_t* __restrict__ dst, size_t n4,
                                                            float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float v[4];
        if (IN_BF16) {
            const bf16x4 t = reinterpret_cast<const bf16x4*>(src)[i];
            for (int j = 0; j < 4; ++j) v[j] = (float)t[j];
        } else {
            const float4 t = reinterpret_cast<const float4*>(src)[i];
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        }
        bf16x4 o;
        for (int j = 0; j < 4; ++j) o[j] = (bf16_t)(v[j] * scale);
        reinterpret_cast<bf16x4*>(dst)[i] = o;
    }
}

// bf16 [R][C] -> bf16 [C][Rp] (Rp >= R, columns R..Rp-1 zero filled); 64 x 64 tiles through LDS
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R,
                                                             int C, int Rp) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i & 63, r = i >> 6;
        bf16_t v = (bf16_t)0.f;
        if (r0 + r < R && c0 + c < C) v = src[(size_t)(r0 + r) * C + c0 + c];
        tile[r][c] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i & 63, c = i >> 6;
        if (c0 + c < C && r0 + r < Rp) dst[(size_t)(c0 + c) * Rp + r0 + r] = tile[r][c];
    }
}

// f32 [R][C] -> bf16 [C][R]  (small weight matrices)
__global__ __launch_bounds__(256) void transpose_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                                 int R, int C) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i & 63, r = i >> 6;
        if (r0 + r < R && c0 + c < C) tile[r][c] = src[(size_t)(r0 + r) * C + c0 + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i & 63, c = i >> 6;
        if (c0 + c < C && r0 + r < R) dst[(size_t)(c0 + c) * R + r0 + r] = (bf16_t)tile[r][c];
    }
}

// column sums of a bf16 [M][N] matrix over a chunk of rows -> part[chunk][N]  (fp32)
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16_t* __restrict__ a, int M, int N, int rows_per_block,
                                                             float* __restrict__ part) {
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (c >= N) return;
    float s0 = 0.f, s1 = 0.f;
    for (int r = r0; r < r1; ++r) {
        const bf16x2 v = *reinterpret_cast<const bf16x2*>(a + (size_t)r * N + c);
        s0 += (float)v[0];
        s1 += (float)v[1];
    }
    part[(size_t)blockIdx.y * N + c] = s0;
    part[(size_t)blockIdx.y * N + c + 1] = s1;
}

// out[i] = scale * sum_s part[s][i]   (deterministic split-K / partial-sum combine)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ part, int S, size_t n, float* __restrict__ out,
                                                           float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += part[(size_t)k * n + i];
        out[i] = s * scale;
    }
}

// RMSNorm backward.  dxn = dL/d(gamma * xhat) (bf16 [M][D]); per block of `rows_per_block` rows:
//   part[block][k] = sum_rows dxn[r][k] * xhat[r][k]           (-> d gamma)
//   dx[r][k] = rstd * (g_k dxn_k - xhat_k * mean_k(g dxn xhat))  (optional)
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dxn, const float* __restrict__ x,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          int M, int D, int rows_per_block, float* __restrict__ part,
                                                          float* __restrict__ dx) {
    extern __shared__ float red[];  // [4][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    const int nv = D >> 2;
    float4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = r0 + wave; r < r1; r += 4) {
        const float rs = rstd[r];
        float4 xh[8], gd[8];
        float dot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 xv = reinterpret_cast<const float4*>(x + (size_t)r * D)[c];
                const bf16x4 dv = reinterpret_cast<const bf16x4*>(dxn + (size_t)r * D)[c];
                const float4 g = reinterpret_cast<const float4*>(gamma)[c];
                xh[i] = make_float4(xv.x * rs, xv.y * rs, xv.z * rs, xv.w * rs);
                const float d0 = (float)dv[0], d1 = (float)dv[1], d2 = (float)dv[2], d3 = (float)dv[3];
                acc[i].x += d0 * xh[i].x; acc[i].y += d1 * xh[i].y; acc[i].z += d2 * xh[i].z; acc[i].w += d3 * xh[i].w;
                gd[i] = make_float4(g.x * d0, g.y * d1, g.z * d2, g.w * d3);
                dot += (gd[i].x * xh[i].x + gd[i].y * xh[i].y) + (gd[i].z * xh[i].z + gd[i].w * xh[i].w);
            }
        }
        if (dx) {
            const float mean = wave_sum(dot) / (float)D;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = lane + 64 * i;
                if (c < nv) {
                    float4 o;
                    o.x = rs * (gd[i].x - xh[i].x * mean); o.y = rs * (gd[i].y - xh[i].y * mean);
                    o.z = rs * (gd[i].z - xh[i].z * mean); o.w = rs * (gd[i].w - xh[i].w * mean);
                    reinterpret_cast<float4*>(dx + (size_t)r * D)[c] = o;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) reinterpret_cast<float4*>(red + wave * D)[c] = acc[i];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < D; k += 256)
        part[(size_t)blockIdx.x * D + k] = (red[k] + red[D + k]) + (red[2 * D + k] + red[3 * D + k]);
}

struct SavedLayout { size_t rstd, xn, pre, h, total; };
SavedLayout saved_layout(int A, int T, int M) {
    SavedLayout s;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = align_up(off + b, 256); return o; };
    s.rstd = take((size_t)M * 4);
    s.xn = take((size_t)M * A * 2);
    s.pre = take((size_t)M * T * 2);
    s.h = take((size_t)M * T * 2);
    s.total = off;
    return s;
}

constexpr int COLSUM_ROWS = 512;
constexpr int RMSBWD_ROWS = 256;

struct WsLayout {
    // forward
    size_t w1, w2, xn, h;
    // backward
    size_t dy, dyT, hT, w2T, dpre, dpreT, xnT, w1T, dxn, slabs, part, total;
    int Mp, splits;
};
WsLayout ws_layout(int A, int T, int M) {
    WsLayout w{};
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = align_up(off + b, 256); return o; };
    w.Mp = (int)align_up((size_t)M, 64);
    int splits = w.Mp / 64 / 8;  // >= 8 K tiles per split
    splits = splits < 1 ? 1 : (splits > 32 ? 32 : splits);
    w.splits = splits;
    w.w1 = take((size_t)T * A * 2);
    w.w2 = take((size_t)T * T * 2);
    w.xn = take((size_t)M * A * 2);   // forward without `saved`; backward reuses as dxn
    w.h = take((size_t)M * T * 2);    // forward without `saved`; backward reuses as dpre
    w.dxn = w.xn;
    w.dpre = w.h;
    w.dy = take((size_t)M * T * 2);
    w.dyT = take((size_t)T * w.Mp * 2);    // reused for dpreT
    w.dpreT = w.dyT;
    w.hT = take((size_t)T * w.Mp * 2);
    w.xnT = take((size_t)A * w.Mp * 2);
    w.w2T = take((size_t)T * T * 2);
    w.w1T = take((size_t)A * T * 2);
    const size_t maxw = (size_t)T * (T > A ? T : A);
    w.slabs = take((size_t)splits * maxw * 4);
    const size_t nchunk = (size_t)(M + COLSUM_ROWS - 1) / COLSUM_ROWS;
    const size_t nblk = (size_t)(M + RMSBWD_ROWS - 1) / RMSBWD_ROWS;
    const size_t pw = (nchunk * T > nblk * A ? nchunk * T : nblk * A);
    w.part = take(pw * 4);
    w.total = off;
    return w;
}

int check_dims(const asrk_projector_dims* d, int M) {
    if (!d) return ASRK_ERR_NULL;
    if (d->audio_dim <= 0 || d->text_dim <= 0 || M <= 0) return ASRK_ERR_SHAPE;
    if ((d->audio_dim % 64) || (d->text_dim % 64) || d->audio_dim > 2048) return ASRK_ERR_SHAPE;
    return ASRK_OK;
}

unsigned grid_for(size_t n, unsigned cap = 8192) {
    size_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// bf16 copies of the weights: W1 [T][A] | W2 [T][T] | W1^T [A][T] | W2^T [T][T]
struct PackedLayout { size_t w1, w2, w1T, w2T, total; };
PackedLayout packed_layout(int A, int T) {
    PackedLayout p;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off = align_up(off + b, 256); return o; };
    p.w1 = take((size_t)T * A * 2);
    p.w2 = take((size_t)T * T * 2);
    p.w1T = take((size_t)A * T * 2);
    p.w2T = take((size_t)T * T * 2);
    p.total = off;
    return p;
}

GemmArgs plain_gemm(const bf16_t* X, int64_t ldx, const bf16_t* W, void* out, int64_t ldo, int M, int N, int K) {
    GemmArgs a{};
    a.X = X; a.W = W; a.out = out;
    a.M = M; a.N = N; a.K = K;
    a.x_rpb = M; a.x_bs = 0; a.x_rs = ldx;
    a.o_rpb = M; a.o_bs = 0; a.o_rs = ldo; a.o_off = 0;
    a.ldr = 0; a.res_rows = 1; a.alpha = 1.f; a.splits = 1;
    return a;
}

}  // namespace

extern "C" size_t asrk_projector_workspace_bytes(const asrk_projector_dims* d, int M) {
    if (check_dims(d, M) != ASRK_OK) return 0;
    return ws_layout(d->audio_dim, d->text_dim, M).total;
}

extern "C" size_t asrk_projector_saved_bytes(const asrk_projector_dims* d, int M) {
    if (check_dims(d, M) != ASRK_OK) return 0;
    return saved_layout(d->audio_dim, d->text_dim, M).total;
}

extern "C" size_t asrk_projector_packed_bytes(const asrk_projector_dims* d) {
    if (check_dims(d, 1) != ASRK_OK) return 0;
    return packed_layout(d->audio_dim, d->text_dim).total;
}

extern "C" int asrk_projector_pack(const asrk_projector_dims* d, const asrk_projector_params* p, void* packed, void* stream) {
    int rc = check_dims(d, 1);
    if (rc != ASRK_OK) return rc;
    if (!p || !packed || !p->w1 || !p->w2) return ASRK_ERR_NULL;
    if ((uintptr_t)packed & 255) return ASRK_ERR_ALIGN;
    const int A = d->audio_dim, T = d->text_dim;
    const PackedLayout pl = packed_layout(A, T);
    hipStream_t st = (hipStream_t)stream;
    char* b = (char*)packed;
    if ((rc = asrk_launch_cast_f32_bf16(p->w1, (bf16_t*)(b + pl.w1), (size_t)T * A, 1.f, st)) != ASRK_OK) return rc;
    if ((rc = asrk_launch_cast_f32_bf16(p->w2, (bf16_t*)(b + pl.w2), (size_t)T * T, 1.f, st)) != ASRK_OK) return rc;
    hipLaunchKernelGGL(transpose_f32_bf16_kernel, dim3((T + 63) / 64, (A + 63) / 64), dim3(256), 0, st, p->w1, (bf16_t*)(b + pl.w1T), T, A);
    hipLaunchKernelGGL(transpose_f32_bf16_kernel, dim3((T + 63) / 64, (T + 63) / 64), dim3(256), 0, st, p->w2, (bf16_t*)(b + pl.w2T), T, T);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

namespace {

int forward_check(const asrk_projector_dims* d, const asrk_projector_params* p, int M, const void* out, int out_dtype,
                  const void* saved, const void* ws, size_t ws_bytes) {
    int rc = check_dims(d, M);
    if (rc != ASRK_OK) return rc;
    if (!p || !out || !ws || !p->gamma || !p->w1 || !p->b1 || !p->w2 || !p->b2) return ASRK_ERR_NULL;
    if (out_dtype != ASRK_F32 && out_dtype != ASRK_BF16) return ASRK_ERR_UNSUPPORTED;
    if (((uintptr_t)out & 15) || ((uintptr_t)ws & 255) || ((uintptr_t)saved & 255) || ((uintptr_t)p->packed & 255)) return ASRK_ERR_ALIGN;
    if (ws_bytes < ws_layout(d->audio_dim, d->text_dim, M).total) return ASRK_ERR_WORKSPACE;
    return ASRK_OK;
}

// where this call's RMSNorm output lives: in `saved` when the backward will want it, else in the workspace
void xn_slot(int A, int T, int M, void* saved, void* ws, bf16_t** xn, float** rstd) {
    if (saved) {
        const SavedLayout sl = saved_layout(A, T, M);
        *rstd = (float*)((char*)saved + sl.rstd);
        *xn = (bf16_t*)((char*)saved + sl.xn);
    } else {
        *rstd = nullptr;
        *xn = (bf16_t*)((char*)ws + ws_layout(A, T, M).xn);
    }
}

// linear_1 -> GELU -> linear_2 -> * 0.01 from the RMSNorm output already sitting in its slot (arguments checked)
int forward_from_xn(const asrk_projector_dims* d, const asrk_projector_params* p, int M, void* out, int out_dtype, void* saved,
                    void* ws, hipStream_t st) {
    const int A = d->audio_dim, T = d->text_dim;
    const WsLayout wl = ws_layout(A, T, M);
    char* w = (char*)ws;
    int rc;
    const bf16_t *w1, *w2;
    if (p->packed) {
        const PackedLayout pl = packed_layout(A, T);
        w1 = (const bf16_t*)((const char*)p->packed + pl.w1);
        w2 = (const bf16_t*)((const char*)p->packed + pl.w2);
    } else {
        if ((rc = asrk_launch_cast_f32_bf16(p->w1, (bf16_t*)(w + wl.w1), (size_t)T * A, 1.f, st)) != ASRK_OK) return rc;
        if ((rc = asrk_launch_cast_f32_bf16(p->w2, (bf16_t*)(w + wl.w2), (size_t)T * T, 1.f, st)) != ASRK_OK) return rc;
        w1 = (const bf16_t*)(w + wl.w1);
        w2 = (const bf16_t*)(w + wl.w2);
    }
    bf16_t *xn, *pre = nullptr, *h;
    float* rstd;
    xn_slot(A, T, M, saved, ws, &xn, &rstd);
    if (saved) {
        const SavedLayout sl = saved_layout(A, T, M);
        pre = (bf16_t*)((char*)saved + sl.pre); h = (bf16_t*)((char*)saved + sl.h);
    } else {
        h = (bf16_t*)(w + wl.h);
    }
    GemmArgs g1 = plain_gemm(xn, A, w1, saved ? (void*)pre : (void*)h, T, M, T, A);
    g1.bias = p->b1;
    if (saved) {
        // training: keep the pre-activation for gelu'(.) in the backward, then h = gelu(pre)
        if ((rc = asrk_gemm_launch(g1, false, true, st)) != ASRK_OK) return rc;
        const size_t n8 = (size_t)M * T / 8;
        hipLaunchKernelGGL(gelu_bf16_kernel, dim3(grid_for(n8)), dim3(256), 0, st, pre, h, n8);
        if (hipGetLastError() != hipSuccess) return ASRK_ERR_HIP;
    } else {
        if ((rc = asrk_gemm_launch(g1, true, true, st)) != ASRK_OK) return rc;
    }
    GemmArgs g2 = plain_gemm(h, T, w2, out, T, M, T, T);
    g2.bias = p->b2;
    g2.alpha = OUT_SCALE;  // (h W2^T + b2) * 0.01
    return asrk_gemm_launch(g2, false, out_dtype == ASRK_BF16, st);
}

}  // namespace

extern "C" int asrk_projector_forward(const asrk_projector_dims* d, const asrk_projector_params* p, const float* x, int M,
                                      void* out, int out_dtype, void* saved, void* ws, size_t ws_bytes, void* stream) {
    int rc = forward_check(d, p, M, out, out_dtype, saved, ws, ws_bytes);
    if (rc != ASRK_OK) return rc;
    if (!x) return ASRK_ERR_NULL;
    if ((uintptr_t)x & 15) return ASRK_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    bf16_t* xn;
    float* rstd;
    xn_slot(d->audio_dim, d->text_dim, M, saved, ws, &xn, &rstd);
    if ((rc = asrk_launch_rmsnorm_bf16(x, p->gamma, xn, rstd, M, d->audio_dim, RMS_EPS, st)) != ASRK_OK) return rc;
    return forward_from_xn(d, p, M, out, out_dtype, saved, ws, st);
}

// Encoder and projector in one call: the encoder's last LayerNorm hands the projector its RMSNorm output directly
// (asrk_norm.hip: layernorm_rms_kernel).  `hidden` (last_hidden_state, fp32 [B*1500][d_model]) is written only when
// non-NULL; a caller that will run asrk_projector_backward needs it (that call's `x`).
extern "C" int asrk_encoder_projector_forward(const asrk_encoder* enc, const float* feats, int B, int n_frames, float* hidden,
                                              void* enc_ws, size_t enc_ws_bytes, const asrk_projector_dims* d,
                                              const asrk_projector_params* p, void* out, int out_dtype, void* saved, void* ws,
                                              size_t ws_bytes, void* stream) {
    if (B <= 0 || B > 65535) return ASRK_ERR_SHAPE;
    const int M = B * 1500;
    int rc = forward_check(d, p, M, out, out_dtype, saved, ws, ws_bytes);
    if (rc != ASRK_OK) return rc;
    if (saved && !hidden) return ASRK_ERR_NULL;  // the backward reads the projector's fp32 input
    hipStream_t st = (hipStream_t)stream;
    AsrkRmsTail tail;
    tail.gamma = p->gamma;
    tail.eps = RMS_EPS;
    tail.dim = d->audio_dim;  // checked against the encoder's d_model before anything is launched
    xn_slot(d->audio_dim, d->text_dim, M, saved, ws, &tail.xn, &tail.rstd);
    if ((rc = asrk_encoder_forward_impl(enc, feats, B, n_frames, hidden, &tail, enc_ws, enc_ws_bytes, st)) != ASRK_OK) return rc;
    return forward_from_xn(d, p, M, out, out_dtype, saved, ws, st);
}

extern "C" int asrk_projector_backward(const asrk_projector_dims* d, const asrk_projector_params* p, const void* grad_out,
                                       int grad_dtype, int M, const void* saved, const asrk_projector_grads* grads,
                                       float* grad_x, const float* x, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_dims(d, M);
    if (rc != ASRK_OK) return rc;
    if (!p || !grad_out || !saved || !grads || !ws || !x) return ASRK_ERR_NULL;
    if (!p->gamma || !p->w1 || !p->w2 || !grads->gamma || !grads->w1 || !grads->b1 || !grads->w2 || !grads->b2)
        return ASRK_ERR_NULL;
    if (grad_dtype != ASRK_F32 && grad_dtype != ASRK_BF16) return ASRK_ERR_UNSUPPORTED;
    if (((uintptr_t)grad_out & 15) || ((uintptr_t)ws & 255) || ((uintptr_t)saved & 255) || ((uintptr_t)x & 15) ||
        ((uintptr_t)grad_x & 15) || ((uintptr_t)p->packed & 255))
        return ASRK_ERR_ALIGN;
    const int A = d->audio_dim, T = d->text_dim;
    const WsLayout wl = ws_layout(A, T, M);
    if (ws_bytes < wl.total) return ASRK_ERR_WORKSPACE;
    const SavedLayout sl = saved_layout(A, T, M);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    const char* s = (const char*)saved;
    const float* rstd = (const float*)(s + sl.rstd);
    const bf16_t* xn = (const bf16_t*)(s + sl.xn);
    const bf16_t* pre = (const bf16_t*)(s + sl.pre);
    const bf16_t* h = (const bf16_t*)(s + sl.h);
    bf16_t* dy = (bf16_t*)(w + wl.dy);
    bf16_t* dyT = (bf16_t*)(w + wl.dyT);
    bf16_t* hT = (bf16_t*)(w + wl.hT);
    bf16_t* xnT = (bf16_t*)(w + wl.xnT);
    const bf16_t *w2T, *w1T;
    if (p->packed) {
        const PackedLayout pl = packed_layout(A, T);
        w1T = (const bf16_t*)((const char*)p->packed + pl.w1T);
        w2T = (const bf16_t*)((const char*)p->packed + pl.w2T);
    } else {
        hipLaunchKernelGGL(transpose_f32_bf16_kernel, dim3((T + 63) / 64, (T + 63) / 64), dim3(256), 0, st, p->w2, (bf16_t*)(w + wl.w2T), T, T);
        hipLaunchKernelGGL(transpose_f32_bf16_kernel, dim3((T + 63) / 64, (A + 63) / 64), dim3(256), 0, st, p->w1, (bf16_t*)(w + wl.w1T), T, A);
        w2T = (const bf16_t*)(w + wl.w2T);
        w1T = (const bf16_t*)(w + wl.w1T);
    }
    bf16_t* dpre = (bf16_t*)(w + wl.dpre);
    bf16_t* dxn = (bf16_t*)(w + wl.dxn);
    float* slabs = (float*)(w + wl.slabs);
    float* part = (float*)(w + wl.part);
    const int Mp = wl.Mp, S = wl.splits;
    const int nchunk = (M + COLSUM_ROWS - 1) / COLSUM_ROWS;

    auto colsum = [&](const bf16_t* a, int N, float* outv) {
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((N / 2 + 255) / 256, nchunk), dim3(256), 0, st, a, M, N, COLSUM_ROWS, part);
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for(N)), dim3(256), 0, st, part, nchunk, (size_t)N, outv, 1.f);
    };
    auto wgrad = [&](const bf16_t* gT, const bf16_t* aT, int N_out, int N_in, float* outw) -> int {
        // outw[i][j] = sum_m g[m][i] * a[m][j]  = NT GEMM over the transposed copies, split over K = Mp
        GemmArgs g = plain_gemm(gT, Mp, aT, slabs, N_in, N_out, N_in, Mp);
        g.splits = S;
        g.o_split = (int64_t)N_out * N_in;
        int r = asrk_gemm_launch(g, false, false, st);
        if (r != ASRK_OK) return r;
        hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for((size_t)N_out * N_in)), dim3(256), 0, st, slabs, S,
                           (size_t)N_out * N_in, outw, 1.f);
        return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
    };

    // 1. dy = bf16(grad_out * 0.01)
    const size_t n4 = (size_t)M * T / 4;
    if (grad_dtype == ASRK_BF16)
        hipLaunchKernelGGL(scale_to_bf16_kernel<true>, dim3(grid_for(n4)), dim3(256), 0, st, grad_out, dy, n4, OUT_SCALE);
    else
        hipLaunchKernelGGL(scale_to_bf16_kernel<false>, dim3(grid_for(n4)), dim3(256), 0, st, grad_out, dy, n4, OUT_SCALE);
    // 2. d b2 = colsum(dy);  d W2 = dy^T h
    colsum(dy, T, grads->b2);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(Mp / 64, T / 64), dim3(256), 0, st, dy, dyT, M, T, Mp);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(Mp / 64, T / 64), dim3(256), 0, st, h, hT, M, T, Mp);
    if (hipGetLastError() != hipSuccess) return ASRK_ERR_HIP;
    if ((rc = wgrad(dyT, hT, T, T, grads->w2)) != ASRK_OK) return rc;
    // 3. d pre = (dy W2) * gelu'(pre)
    {
        GemmArgs g = plain_gemm(dy, T, w2T, dpre, T, M, T, T);
        g.dg_pre = pre;
        g.ld_dg = T;
        if ((rc = asrk_gemm_launch(g, false, true, st)) != ASRK_OK) return rc;
    }
    // 4. d b1 = colsum(dpre);  d W1 = dpre^T xn
    colsum(dpre, T, grads->b1);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(Mp / 64, T / 64), dim3(256), 0, st, dpre, dyT, M, T, Mp);
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3(Mp / 64, A / 64), dim3(256), 0, st, xn, xnT, M, A, Mp);
    if (hipGetLastError() != hipSuccess) return ASRK_ERR_HIP;
    if ((rc = wgrad(dyT, xnT, T, A, grads->w1)) != ASRK_OK) return rc;
    // 5. d xn = dpre W1;  d gamma = sum_rows dxn * xhat;  optional d x
    {
        GemmArgs g = plain_gemm(dpre, T, w1T, dxn, A, M, A, T);
        if ((rc = asrk_gemm_launch(g, false, true, st)) != ASRK_OK) return rc;
    }
    const int nblk = (M + RMSBWD_ROWS - 1) / RMSBWD_ROWS;
    hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(nblk), dim3(256), (size_t)4 * A * 4, st, dxn, x, rstd, p->gamma, M, A,
                       RMSBWD_ROWS, part, grad_x);
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(grid_for(A)), dim3(256), 0, st, part, nblk, (size_t)A, grads->gamma, 1.f);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}
