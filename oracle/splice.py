"""CPU restatement (NumPy) of the reference's audio-embedding splice, src/modeling.py:222-276.
TEST INFRASTRUCTURE ONLY: imported by tests/ as the checker, never by the product path.

Pinned by construction rather than by a fixture: the function is pure data movement, and
tests/test_oracle_golden.py checks it against a literal torch.cat / torch.full / torch.stack evaluation of the same
lines on random inputs (the reference's ASRModel cannot be instantiated offline: it needs hub weights)."""
import numpy as np


def first_token(input_ids: np.ndarray, token: int) -> np.ndarray:
    """(input_ids[i] == token).nonzero()[0] per row, -1 when absent (src/modeling.py:223-227)."""
    hit = input_ids == token
    idx = hit.argmax(axis=1).astype(np.int32)
    idx[~hit.any(axis=1)] = -1
    return idx


def splice(text_embeds, audio_embeds, input_ids, labels, token):
    """Returns inputs_embeds [B, L-1+A, D], labels [B, L-1+A], attention_mask [B, L-1+A] (src/modeling.py:229-276)."""
    B, L, D = text_embeds.shape
    A = audio_embeds.shape[1]
    c = first_token(input_ids, token)
    if (c < 0).any():
        raise ValueError("'<|audio_chunk|>' token not found in input_ids.")  # :225
    T = L - 1 + A
    out = np.empty((B, T, D), dtype=text_embeds.dtype)
    lab = np.empty((B, T), dtype=np.int64)
    for b in range(B):
        ci = int(c[b])
        out[b, :ci] = text_embeds[b, :ci]                    # :231
        out[b, ci:ci + A] = audio_embeds[b]                   # :232
        out[b, ci + A:] = text_embeds[b, ci + 1:]             # :233
        lab[b, :ci] = labels[b, :ci]                          # :240
        lab[b, ci:ci + A] = -100                              # :242
        lab[b, ci + A:] = labels[b, ci + 1:]                  # :241
    mask = np.ones((B, T), dtype=np.int64)                    # :266 (pad_len is always 0: all rows are T long)
    return out, lab, mask


def splice_backward(grad_out, chunk_idx, L, A):
    """Adjoint of the torch.cat of :229-237: (grad_text [B, L, D], grad_audio [B, A, D])."""
    B, T, D = grad_out.shape
    gt = np.zeros((B, L, D), dtype=grad_out.dtype)
    ga = np.empty((B, A, D), dtype=grad_out.dtype)
    for b in range(B):
        ci = int(chunk_idx[b])
        gt[b, :ci] = grad_out[b, :ci]
        ga[b] = grad_out[b, ci:ci + A]
        gt[b, ci + 1:] = grad_out[b, ci + A:]
    return gt, ga
