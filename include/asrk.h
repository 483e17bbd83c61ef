/*
 * asrk.h - C ABI of libasrk.so: the MI355X (gfx950) audio hot path of alexkroman/asr.
 *
 *   log-mel front-end  ->  Whisper-style audio encoder (frozen, forward only)  ->  AudioProjector
 *
 * The reference is pure Python on torch / HF transformers and has no FFI of its own; each entry
 * point below names the reference call it replaces (file:line, "HF:" = the third-party
 * transformers package the reference calls).  asr_amd/ binds these with ctypes (see
 * INTEGRATION.md for the stub a maintainer would add on the reference side).
 *
 * Conventions
 *   - every pointer marked "dev" is device memory owned by the caller (a torch tensor's
 *     data_ptr()); rows are contiguous; base addresses must be 16-byte aligned;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     all work is enqueued asynchronously on it, the library never synchronises;
 *   - return value 0 = ASRK_OK, negative = error (asrk_strerror); nothing throws or aborts;
 *   - forward / backward entry points allocate nothing: scratch comes in through `ws`.
 *     Only the *_create calls allocate (immutable tables / packed bf16 weights), once.
 */
#ifndef ASRK_H
#define ASRK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASRK_VERSION 100

enum {
    ASRK_OK = 0,
    ASRK_ERR_NULL = -1,        /* required pointer is NULL */
    ASRK_ERR_SHAPE = -2,       /* unsupported / inconsistent dimension */
    ASRK_ERR_ALIGN = -3,       /* pointer not 16-byte aligned */
    ASRK_ERR_HIP = -4,         /* HIP runtime call or kernel launch failed */
    ASRK_ERR_WORKSPACE = -5,   /* workspace too small */
    ASRK_ERR_UNSUPPORTED = -6, /* dtype / mode not implemented */
    ASRK_ERR_LENGTH = -7       /* mel features are not 3000 frames long (HF modeling_whisper.py:612-616) */
};

enum { ASRK_F32 = 0, ASRK_BF16 = 1, ASRK_I16 = 2 /* PCM input of asrk_collate_pcm only */ };

int asrk_version(void);
const char* asrk_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Log-mel front-end.
 * Replaces HF WhisperFeatureExtractor._torch_extract_fbank_features
 * (HF:models/whisper/feature_extraction_whisper.py:135-168), which the reference reaches from
 * src/train.py:54-59 (eval) and src/train.py:172-180 (collator).
 * ---------------------------------------------------------------------------------------- */
typedef struct asrk_logmel_plan asrk_logmel_plan;

/* mel_filters: HOST pointer, float32 [201][n_mels] row-major = HF's `self.mel_filters`
 * (feature_extraction_whisper.py:95-103) cast to float32 (:156).  Builds the per-device constant
 * tables (Hann window, twiddles, sparse mel rows) on the current device. */
int asrk_logmel_plan_create(int n_mels, const float* mel_filters, asrk_logmel_plan** plan);
void asrk_logmel_plan_destroy(asrk_logmel_plan* plan);

/* wave  dev f32 [B][n_samples]   (already padded/truncated, :300-307; n_samples > 200)
 * out   dev f32 [B][n_mels][n_samples/160]
 * valid_samples  dev int32 [B] or NULL: the number of samples of each clip in front of its zero padding (the lengths
 *       HF's padder works from, :300-307).  Only an accelerator: tiles that lie entirely in the padding are written as
 *       silence (log10(1e-10), exactly what the transform gives for zeros) without being loaded or transformed; the
 *       result is the same bit for bit, and samples at or after valid_samples[b] MUST be zero.
 * ws    dev scratch of asrk_logmel_workspace_bytes(B, n_samples), 256-byte aligned: the per-clip maxima (:160-162),
 *       one 128-byte line per clip, and one minimum per wave and tile, by which the floor pass skips what needs none.
 * STFT is Hann-400 / hop-160 / center + reflect pad; the last STFT frame is dropped (:154).
 * The plan's tables live on the device that was current in asrk_logmel_plan_create: a call made with another device
 * current returns ASRK_ERR_UNSUPPORTED (one plan per GPU). */
size_t asrk_logmel_workspace_bytes(int B, int n_samples);
int asrk_logmel_f32(const asrk_logmel_plan* plan, const float* wave, const int* valid_samples, int B, int n_samples,
                    float* out, void* ws, size_t ws_bytes, void* stream);

/* The same computation for callers that cannot allocate -- code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  It has NO counterpart in the reference (which never manages device memory itself).
 * The per-clip maxima, which asrk_logmel_f32 keeps in the caller's `ws`, live in a static device array of
 * ASRK_LOGMEL_NOALLOC_MAX_CLIPS 128-byte lines (512 KiB, held by every GPU the library runs on until the process ends)
 * whose per-device address asrk_logmel_plan_create looks up once; the per-tile minima are not kept, so the floor pass
 * visits every tile (same result, bit for bit).  The call itself performs no hipMalloc / hipFree / event
 * creation and is not timed by the measurement hook.  B > ASRK_LOGMEL_NOALLOC_MAX_CLIPS returns ASRK_ERR_WORKSPACE;
 * a plan created on another device returns ASRK_ERR_UNSUPPORTED.  All calls on one GPU share that array: calls
 * that overlap on one GPU (two streams, two threads) race -- serialise them, or use asrk_logmel_f32 with separate
 * workspaces.  The array is cleared at the start of every call. */
#define ASRK_LOGMEL_NOALLOC_MAX_CLIPS 4096
int asrk_logmel_f32_noalloc(const asrk_logmel_plan* plan, const float* wave, int B, int n_samples, float* out,
                            void* stream);

/* ------------------------------------------------------------------------------------------
 * Audio encoder forward.
 * Replaces WhisperEncoder.forward (src/modeling.py:33-40) -> HF WhisperEncoder.forward
 * (HF:models/whisper/modeling_whisper.py:592-646) and everything it calls (:215-413).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int d_model, n_layers, n_heads, ffn_dim, n_mels;
} asrk_encoder_dims;

/* fp32 device pointers in the HF state-dict layouts (Linear weight = [out][in]). */
typedef struct {
    const float *q_w, *q_b, *k_w, *v_w, *v_b, *o_w, *o_b; /* self_attn.{q,k,v,out}_proj; k has no bias (:279) */
    const float *ln1_w, *ln1_b;                            /* self_attn_layer_norm */
    const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
    const float *ln2_w, *ln2_b;                            /* final_layer_norm */
} asrk_encoder_layer_weights;

typedef struct {
    const float *conv1_w, *conv1_b; /* [d][n_mels][3], [d]   (:566) */
    const float *conv2_w, *conv2_b; /* [d][d][3], [d]        (:567) */
    const float* pos;               /* embed_positions.weight [1500][d] (:569) */
    const float *ln_w, *ln_b;       /* layer_norm (:573) */
    const asrk_encoder_layer_weights* layers; /* HOST array [n_layers] */
} asrk_encoder_weights;

typedef struct asrk_encoder asrk_encoder;

/* Packs the frozen weights into the bf16 layouts the kernels read (QKV fused, q pre-scaled by
 * head_dim^-0.5 as :309 does, conv taps reordered).  Requires head_dim == 64, d_model % 128 == 0,
 * ffn_dim % 128 == 0, n_mels % 8 == 0.  The fp32 sources may be freed once `stream` has drained. */
int asrk_encoder_create(const asrk_encoder_dims* dims, const asrk_encoder_weights* w, void* stream,
                        asrk_encoder** enc);
void asrk_encoder_destroy(asrk_encoder* enc);
size_t asrk_encoder_workspace_bytes(const asrk_encoder* enc, int B);

/* feats dev f32 [B][n_mels][n_frames]; n_frames must be 3000 (ASRK_ERR_LENGTH otherwise, :612-616)
 * out   dev f32 [B][1500][d_model]  = last_hidden_state (:642-646).
 * The attention mask the reference passes is ignored by HF (:607-609), so it is not an argument. */
int asrk_encoder_forward(const asrk_encoder* enc, const float* feats, int B, int n_frames, float* out,
                         void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * AudioProjector forward / backward.
 * Replaces AudioProjector.forward (src/modeling.py:77-84: LlamaRMSNorm(eps 1e-6) -> linear_1 ->
 * exact GELU -> linear_2 -> *0.01) and the autograd backward of its five parameters; the cast at
 * src/modeling.py:220,301 is folded in through `out_dtype`.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int audio_dim, text_dim;
} asrk_projector_dims;

typedef struct {
    const float *gamma, *w1, *b1, *w2, *b2; /* dev fp32 master params: norm.weight, linear_{1,2}.{weight,bias} */
    const void* packed; /* dev buffer written by asrk_projector_pack for THESE w1 / w2, or NULL: the call then makes
                           its own bf16 copies (two casts in forward, two transposes in backward, every call) */
} asrk_projector_params;

typedef struct {
    float *gamma, *w1, *b1, *w2, *b2; /* dev fp32, OVERWRITTEN with this call's gradients */
} asrk_projector_grads;

size_t asrk_projector_workspace_bytes(const asrk_projector_dims* dims, int M);
size_t asrk_projector_saved_bytes(const asrk_projector_dims* dims, int M);

/* The bf16 operands of the four GEMMs that read a weight (W1, W2 for the forward; W1^T, W2^T for the backward) in one
 * caller-owned device buffer of asrk_projector_packed_bytes(), 256-byte aligned.  The fp32 parameters stay the master
 * copy (the optimizer of src/train.py updates them): pack again after every update, never for a frozen projector.
 * The reference gets the same copies from autocast's weight-cast cache (torch.autocast(cache_enabled=True)). */
size_t asrk_projector_packed_bytes(const asrk_projector_dims* dims);
int asrk_projector_pack(const asrk_projector_dims* dims, const asrk_projector_params* p, void* packed, void* stream);

/* x dev f32 [M][audio_dim]; out dev [M][text_dim] in out_dtype (ASRK_F32 | ASRK_BF16).
 * saved: dev buffer of asrk_projector_saved_bytes() kept for backward, or NULL for inference. */
int asrk_projector_forward(const asrk_projector_dims* dims, const asrk_projector_params* p, const float* x,
                           int M, void* out, int out_dtype, void* saved, void* ws, size_t ws_bytes,
                           void* stream);

/* grad_out dev [M][text_dim] in grad_dtype; `saved` from the matching forward; x = that forward's
 * input.  grad_x (dev f32 [M][audio_dim]) may be NULL: the reference's projector input is produced
 * under no_grad (src/modeling.py:36), so it never needs it. */
int asrk_projector_backward(const asrk_projector_dims* dims, const asrk_projector_params* p,
                            const void* grad_out, int grad_dtype, int M, const void* saved,
                            const asrk_projector_grads* grads, float* grad_x, const float* x, void* ws,
                            size_t ws_bytes, void* stream);

/* Encoder and projector in one call (src/modeling.py:213-220 / :296-301: self.projector(self.encoder(...)) followed by
 * the cast to the decoder's dtype): the encoder's final LayerNorm (HF modeling_whisper.py:642) also applies the
 * projector's RMSNorm and emits the bf16 operand of linear_1, so last_hidden_state makes no round trip through HBM.
 * `hidden` dev f32 [B][1500][d_model] receives last_hidden_state when non-NULL; it must be given when `saved` is (it is
 * asrk_projector_backward's `x`).  enc_ws / ws: the two workspaces of the separate calls; out / saved as in
 * asrk_projector_forward with M = 1500 B.  Results are bit-identical to asrk_encoder_forward + asrk_projector_forward. */
int asrk_encoder_projector_forward(const asrk_encoder* enc, const float* feats, int B, int n_frames, float* hidden,
                                   void* enc_ws, size_t enc_ws_bytes, const asrk_projector_dims* dims,
                                   const asrk_projector_params* p, void* out, int out_dtype, void* saved, void* ws,
                                   size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Building blocks (exported so the parity tests can check each kernel on its own through the
 * same ABI; the entry points above are compositions of these).
 * ---------------------------------------------------------------------------------------- */

/* out[m][n] = epilogue( sum_k x[m][k] * w[n][k] ),  x bf16 [M][K] (ldx), w bf16 [N][K] (K%64==0),
 * epilogue: (+bias[n]) -> (gelu) -> (*alpha) -> (+res[(m % res_rows)][n], f32, ldr) -> out_dtype.
 * HF: every nn.Linear on the path (modeling_whisper.py:279-282, 375-376; src/modeling.py:68-70). */
int asrk_gemm_bf16(const void* x, int64_t ldx, const void* w, const float* bias, const float* res, int64_t ldr,
                   int res_rows, void* out, int64_t ldo, int out_dtype, int M, int N, int K, int gelu,
                   float alpha, void* stream);

/* LayerNorm over the last dim (eps 1e-5, torch default; modeling_whisper.py:371,377,573):
 * x f32 [M][D] -> out (ASRK_BF16 | ASRK_F32) [M][D]. */
int asrk_layernorm(const float* x, const float* w, const float* b, void* out, int out_dtype, int M, int D,
                   float eps, void* stream);

/* The residual step fused in front of a LayerNorm: x[m][:] += delta[m][:] (fp32 += bf16; modeling_whisper.py:399,407),
 * then out = LayerNorm(x) as above (:402, next layer's :392, or :642).  x is updated in place unless keep_x == 0
 * (the encoder's last norm, whose input is not needed again).  delta bf16 [M][D] is the out-proj / fc2 result. */
int asrk_add_layernorm(float* x, const void* delta, const float* w, const float* b, void* out, int out_dtype,
                       int keep_x, int M, int D, float eps, void* stream);

/* ---- on-device collate --------------------------------------------------------------
 * The pad / truncate / mask step of the feature extractor (HF feature_extraction_whisper.py:300-307, 332-341;
 * feature_extraction_sequence_utils.py:224-337) from one ragged device buffer: clip b = pcm[offsets[b] .. offsets[b+1])
 * (ASRK_F32, or ASRK_I16 scaled by 1/32768), offsets int64 [B+1] on the device.
 *   wave [B][target] f32: samples, right-padded with 0.0 / truncated to `target`;
 *   frame_mask [B][target / hop] int32: 1 where frame * hop < min(len, target)   (= HF's attention_mask[:, ::hop]).
 * Either output may be NULL. */
int asrk_collate_pcm(const void* pcm, int pcm_dtype, const int64_t* offsets, int B, int target, int hop, float* wave,
                     int* frame_mask, void* stream);

/* ---- audio-embedding splice (src/modeling.py:222-276) ------------------------------
 * first_idx[b] = index of the first `token` in ids[b][0..L), -1 if absent (the reference's
 * (input_ids[i] == audio_chunk_id).nonzero()[0], :223-227, without its per-sample host sync). */
int asrk_find_token_i64(const int64_t* ids, int B, int L, int64_t token, int* first_idx, void* stream);

/* out_embeds[b] = cat(text[b][:c], audio[b], text[b][c+1:]) with c = chunk_idx[b] (:229-237), [B][L-1+A][D];
 * out_labels[b] = cat(labels[b][:c], -100 x A, labels[b][c+1:]) (:239-245) and out_mask = 1 (:266; all rows have the
 * same length, the reference's padding branch never pads) -- both optional.  dtype ASRK_BF16 | ASRK_F32, D * size % 16 == 0. */
int asrk_embed_splice(const void* text, const void* audio, const int* chunk_idx, const int64_t* labels, int B, int L, int A,
                      int D, int dtype, void* out_embeds, int64_t* out_labels, int64_t* out_mask, void* stream);

/* Adjoint of asrk_embed_splice (autograd of torch.cat, :229-237): rows of grad_out go back to grad_audio [B][A][D] and
 * grad_text [B][L][D] (row c, the dropped placeholder, is zeroed).  Either output may be NULL. */
int asrk_embed_splice_backward(const void* grad_out, const int* chunk_idx, int B, int L, int A, int D, int dtype,
                               void* grad_text, void* grad_audio, void* stream);

/* Bidirectional multi-head attention, head_dim 64, no mask, scaling 1.0 (q already scaled)
 * (modeling_whisper.py:215-238, 341-351).  qkv bf16 [B*T][3*H*64] (q | k | v per row),
 * out bf16 [B*T][H*64]. */
int asrk_attention_bf16(const void* qkv, void* out, int B, int T, int H, void* stream);
/* The same with q pre-multiplied by log2(e) as well (scores in log2 units): the form asrk_encoder_forward uses, the
 * factor being folded into W_q and b_q when the weights are packed. */
int asrk_attention_log2_bf16(const void* qkv, void* out, int B, int T, int H, void* stream);

/* ------------------------------------------------------------------------------------------
 * Measurement hook (no counterpart in the reference, which has no profiler: SURVEY section 5).
 * When enabled, every GEMM / attention / log-mel / LayerNorm launch is bracketed by HIP events on
 * its own stream; asrk_profile_read sums one category -- one per rocprof symbol: 0 gemm256p_kernel (persistent
 * 256 x 256 tiles), 5 gemm256_kernel (one 256 x 256 tile per workgroup), 6 gemm_bf16_kernel (128 x 128 tiles), all
 * with work = 2MNK flops; 1 attention: 4*B*H*T*T*64 flops; 2 log-mel: algorithmic bytes; 3 LayerNorm: bytes;
 * 4 embed splice: bytes -- after the caller has synchronised.  bench.py uses it for the `roofline` object.  Off by
 * default.
 * ---------------------------------------------------------------------------------------- */
int asrk_profile_enable(int on);
int asrk_profile_read(int category, double* total_ms, double* total_work, int* launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* ASRK_H */
