// Audio-embedding splice: replaces the per-sample Python loop of
// ASRModel.forward (src/modeling.py:222-276) -- for every sample, find the first <|audio_chunk|> token c
// (:223-227, one .item() sync per sample in the reference), and build
//     inputs_embeds[b] = cat(text_embeds[b, :c], audio_embeds[b], text_embeds[b, c+1:])      (:229-237)
//     labels[b]        = cat(labels[b, :c], -100 x A, labels[b, c+1:])                        (:239-245)
// Every sample has the same input length L, so every row comes out L - 1 + A long, the reference's padding branch
// (:247-270) never pads and its attention mask is all ones (:266) -- reproduced as is.
// Pure data movement: HBM-bound, 16-byte accesses, one workgroup per output row group; bit-exact.
#include "asrk_common.h"
#include "asrk_prof.h"

namespace {

// first index of `token` in each row of ids [B][L], -1 if absent (one workgroup per row)
__global__ __launch_bounds__(256) void find_token_kernel(const long long* __restrict__ ids, int L, long long token,
                                                         int* __restrict__ first) {
    __shared__ int best;
    if (threadIdx.x == 0) best = 0x7fffffff;
    __syncthreads();
    const long long* row = ids + (size_t)blockIdx.x * L;
    int mine = 0x7fffffff;
    for (int i = threadIdx.x; i < L; i += blockDim.x)
        if (row[i] == token) { mine = i; break; }
    if (mine != 0x7fffffff) atomicMin(&best, mine);
    __syncthreads();
    if (threadIdx.x == 0) first[blockIdx.x] = best == 0x7fffffff ? -1 : best;
}

// source row of output position t of sample b: text row t (t < c), audio row t - c (c <= t < c + A), text row t - A + 1

// Forward: out[b][t][:] for t in [0, L - 1 + A); ROWB = bytes per embedding row (multiple of 16)
__global__ __launch_bounds__(256) void splice_fwd_kernel(const char* __restrict__ text, const char* __restrict__ audio,
                                                         const int* __restrict__ chunk, const long long* __restrict__ labels,
                                                         int L, int A, int rowb, char* __restrict__ out,
                                                         long long* __restrict__ out_labels, long long* __restrict__ out_mask) {
    const int b = blockIdx.y;
    const int T = L - 1 + A;
    int c = chunk[b];
    c = c < 0 ? 0 : c;  // absent token: the host raises (ValueError in the reference); keep the copy in bounds
    const int rows_per_block = 8;  // 8 waves... 256 threads = 4 waves, 2 rows each
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = 0; r < rows_per_block / 4; ++r) {
        const int t = blockIdx.x * rows_per_block + wave * (rows_per_block / 4) + r;
        if (t >= T) return;
        const bool is_audio = t >= c && t < c + A;
        const int srow = t < c ? t : (is_audio ? t - c : t - A + 1);
        const char* src = is_audio ? audio + ((size_t)b * A + srow) * rowb : text + ((size_t)b * L + srow) * rowb;
        char* dst = out + ((size_t)b * T + t) * rowb;
        for (int i = lane * 16; i < rowb; i += 64 * 16) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
        if (lane == 0) {
            if (out_labels) out_labels[(size_t)b * T + t] = is_audio ? -100ll : labels[(size_t)b * L + srow];
            if (out_mask) out_mask[(size_t)b * T + t] = 1ll;
        }
    }
}

// Backward: the splice is a permutation of rows, so its adjoint copies rows back.  grad_text row c (the placeholder
// token, dropped by the forward) gets zeros.
__global__ __launch_bounds__(256) void splice_bwd_kernel(const char* __restrict__ grad_out, const int* __restrict__ chunk, int L,
                                                         int A, int rowb, char* __restrict__ grad_text,
                                                         char* __restrict__ grad_audio) {
    const int b = blockIdx.y;
    const int T = L - 1 + A;
    int c = chunk[b];
    c = c < 0 ? 0 : c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // rows [0, T): scatter back; row T (one extra) zeroes grad_text[b][c]
    for (int r = 0; r < 2; ++r) {
        const int t = blockIdx.x * 8 + wave * 2 + r;
        if (t > T) return;
        if (t == T) {
            if (grad_text) {
                char* dst = grad_text + ((size_t)b * L + c) * rowb;
                for (int i = lane * 16; i < rowb; i += 64 * 16) *reinterpret_cast<uint4*>(dst + i) = uint4{0u, 0u, 0u, 0u};
            }
            return;
        }
        const bool is_audio = t >= c && t < c + A;
        const int srow = t < c ? t : (is_audio ? t - c : t - A + 1);
        char* dst = is_audio ? (grad_audio ? grad_audio + ((size_t)b * A + srow) * rowb : nullptr)
                             : (grad_text ? grad_text + ((size_t)b * L + srow) * rowb : nullptr);
        if (!dst) continue;
        const char* src = grad_out + ((size_t)b * T + t) * rowb;
        for (int i = lane * 16; i < rowb; i += 64 * 16) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
    }
}

int elem_bytes(int dtype) { return dtype == ASRK_F32 ? 4 : (dtype == ASRK_BF16 ? 2 : 0); }

}  // namespace

extern "C" int asrk_find_token_i64(const int64_t* ids, int B, int L, int64_t token, int* first_idx, void* stream) {
    if (!ids || !first_idx) return ASRK_ERR_NULL;
    if (B <= 0 || L <= 0) return ASRK_ERR_SHAPE;
    hipLaunchKernelGGL(find_token_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const long long*)ids, L, (long long)token,
                       first_idx);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

extern "C" int asrk_embed_splice(const void* text, const void* audio, const int* chunk_idx, const int64_t* labels, int B, int L,
                                 int A, int D, int dtype, void* out_embeds, int64_t* out_labels, int64_t* out_mask,
                                 void* stream) {
    if (!text || !audio || !chunk_idx || !out_embeds || (out_labels && !labels)) return ASRK_ERR_NULL;
    const int eb = elem_bytes(dtype);
    if (!eb) return ASRK_ERR_UNSUPPORTED;
    if (B <= 0 || L <= 0 || A <= 0 || D <= 0 || B > 65535 || ((size_t)D * eb) % 16) return ASRK_ERR_SHAPE;
    if (((uintptr_t)text & 15) || ((uintptr_t)audio & 15) || ((uintptr_t)out_embeds & 15)) return ASRK_ERR_ALIGN;
    const int T = L - 1 + A;
    // algorithmic bytes: every output row is read once and written once (+ labels / mask, 8 bytes each per position)
    AsrkProfScope prof(ASRK_PROF_SPLICE, 2.0 * B * T * D * eb + (out_labels ? 16.0 : 0.0) * B * T + (out_mask ? 8.0 : 0.0) * B * T,
                       (hipStream_t)stream);
    hipLaunchKernelGGL(splice_fwd_kernel, dim3((T + 7) / 8, B), dim3(256), 0, (hipStream_t)stream, (const char*)text,
                       (const char*)audio, chunk_idx, (const long long*)labels, L, A, D * eb, (char*)out_embeds,
                       (long long*)out_labels, (long long*)out_mask);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}

extern "C" int asrk_embed_splice_backward(const void* grad_out, const int* chunk_idx, int B, int L, int A, int D, int dtype,
                                          void* grad_text, void* grad_audio, void* stream) {
    if (!grad_out || !chunk_idx) return ASRK_ERR_NULL;
    const int eb = elem_bytes(dtype);
    if (!eb) return ASRK_ERR_UNSUPPORTED;
    if (B <= 0 || L <= 0 || A <= 0 || D <= 0 || B > 65535 || ((size_t)D * eb) % 16) return ASRK_ERR_SHAPE;
    if (((uintptr_t)grad_out & 15) || ((uintptr_t)grad_text & 15) || ((uintptr_t)grad_audio & 15)) return ASRK_ERR_ALIGN;
    if (!grad_text && !grad_audio) return ASRK_OK;
    const int T = L - 1 + A;
    hipLaunchKernelGGL(splice_bwd_kernel, dim3((T + 1 + 7) / 8, B), dim3(256), 0, (hipStream_t)stream, (const char*)grad_out,
                       chunk_idx, L, A, D * eb, (char*)grad_text, (char*)grad_audio);
    return hipGetLastError() == hipSuccess ? ASRK_OK : ASRK_ERR_HIP;
}
