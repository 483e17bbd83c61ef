"""Audio-embedding splice: the per-sample loop of the reference's ``ASRModel.forward``
(src/modeling.py:222-276) as one call on the GPU.

    inputs_embeds, labels, attention_mask = splice_audio_embeddings(
        text_embeds, audio_embeds, input_ids, labels, audio_chunk_id)

replaces lines 222-276: for every sample the first ``<|audio_chunk|>`` token is replaced by that
sample's audio embeddings, labels get ``-100`` over the audio span, and the attention mask is all ones
(every row comes out ``L - 1 + A`` long, so the reference's padding branch never pads).  The reference
does one ``.item()`` host sync per sample; here the token search runs on the GPU and a single sync
checks the whole batch for the reference's ``ValueError``.  Differentiable w.r.t. both embedding
inputs (the adjoint of ``torch.cat``).  Runs in libasrk.so; there is no CPU path.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

_DT = {torch.float32: _lib.ASRK_F32, torch.bfloat16: _lib.ASRK_BF16}


def find_first_token(input_ids: torch.Tensor, token_id: int) -> torch.Tensor:
    """int32 [B]: index of the first ``token_id`` per row, -1 if absent (src/modeling.py:223-227)."""
    if input_ids.dim() != 2 or input_ids.dtype != torch.int64:
        raise ValueError("input_ids must be an int64 [B, L] tensor")
    lib = _lib.load()
    ids = input_ids.contiguous()
    out = torch.empty((ids.shape[0],), device=ids.device, dtype=torch.int32)
    _lib.check(lib.asrk_find_token_i64(_lib.ptr(ids), ids.shape[0], ids.shape[1], int(token_id), _lib.ptr(out),
                                       _lib.current_stream()))
    return out


class _SpliceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, text, audio, chunk_idx, labels):
        lib = _lib.load()
        B, L, D = text.shape
        A = audio.shape[1]
        T = L - 1 + A
        out = torch.empty((B, T, D), device=text.device, dtype=text.dtype)
        out_labels = torch.empty((B, T), device=text.device, dtype=torch.int64) if labels is not None else None
        out_mask = torch.empty((B, T), device=text.device, dtype=torch.int64)
        _lib.check(lib.asrk_embed_splice(_lib.ptr(text), _lib.ptr(audio), _lib.ptr(chunk_idx), _lib.ptr(labels), B, L, A, D,
                                         _DT[text.dtype], _lib.ptr(out), _lib.ptr(out_labels), _lib.ptr(out_mask),
                                         _lib.current_stream()))
        ctx.save_for_backward(chunk_idx)
        ctx.dims = (B, L, A, D)
        ctx.mark_non_differentiable(out_mask)
        if out_labels is not None:
            ctx.mark_non_differentiable(out_labels)
        return out, out_labels, out_mask

    @staticmethod
    def backward(ctx, g_out, _g_labels, _g_mask):
        lib = _lib.load()
        (chunk_idx,) = ctx.saved_tensors
        B, L, A, D = ctx.dims
        g_out = g_out.contiguous()
        need_text, need_audio = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        g_text = torch.empty((B, L, D), device=g_out.device, dtype=g_out.dtype) if need_text else None
        g_audio = torch.empty((B, A, D), device=g_out.device, dtype=g_out.dtype) if need_audio else None
        _lib.check(lib.asrk_embed_splice_backward(_lib.ptr(g_out), _lib.ptr(chunk_idx), B, L, A, D, _DT[g_out.dtype],
                                                  _lib.ptr(g_text), _lib.ptr(g_audio), _lib.current_stream()))
        return g_text, g_audio, None, None


def splice_audio_embeddings(text_embeds: torch.Tensor, audio_embeds: torch.Tensor, input_ids: torch.Tensor,
                            labels: Optional[torch.Tensor], audio_chunk_id: int
                            ) -> Tuple[torch.Tensor, Optional[torch.Tensor], torch.Tensor]:
    """(inputs_embeds [B, L-1+A, D], labels [B, L-1+A] or None, attention_mask [B, L-1+A] int64)."""
    if text_embeds.dim() != 3 or audio_embeds.dim() != 3 or text_embeds.shape[0] != audio_embeds.shape[0] \
            or text_embeds.shape[2] != audio_embeds.shape[2]:
        raise ValueError("text_embeds [B, L, D] and audio_embeds [B, A, D] must agree in B and D")
    if text_embeds.dtype not in _DT or audio_embeds.dtype != text_embeds.dtype:
        raise ValueError("embeddings must share one dtype, float32 or bfloat16 (the reference casts the audio "
                         "embeddings to the decoder's dtype first, src/modeling.py:220)")
    if tuple(input_ids.shape) != tuple(text_embeds.shape[:2]) or (labels is not None and labels.shape != input_ids.shape):
        raise ValueError("input_ids / labels must be [B, L]")
    if labels is not None and labels.dtype != torch.int64:
        raise ValueError("labels must be int64")
    chunk_idx = find_first_token(input_ids, audio_chunk_id)
    if int(chunk_idx.min().item()) < 0:  # the only host sync: the reference's error, for the whole batch at once
        raise ValueError("'<|audio_chunk|>' token not found in input_ids.")
    return _SpliceFn.apply(text_embeds.contiguous(), audio_embeds.contiguous(), chunk_idx,
                           labels.contiguous() if labels is not None else None)
