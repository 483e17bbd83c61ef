"""Greedy-decode evaluation harness: the reference's ``evaluate_samples``
(src/train.py:36-82) and the prompt assembly of ``ASRModel.generate`` (src/modeling.py:296-334) on top of this
package's extractor / encoder / projector / splice, plus a word error rate that needs no ``evaluate`` / ``jiwer``.

The decoder itself (an HF causal LM) is outside this path: ``evaluate_samples`` takes any ``model`` with the
reference's ``generate(input_features=..., attention_mask=..., max_new_tokens=..., do_sample=False)`` signature.
WER follows the definition the reference's metric implements (``evaluate.load("wer")`` -> jiwer: words split on
whitespace, (substitutions + deletions + insertions) / reference words, pooled over the corpus).  jiwer is not
installed in this environment, so agreement with it is by definition and known-answer tests only: parity unpinned.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence, Tuple

import torch


def _edit_distance(ref: Sequence[str], hyp: Sequence[str]) -> int:
    """Word-level Levenshtein distance (unit costs)."""
    prev = list(range(len(hyp) + 1))
    for i, r in enumerate(ref, 1):
        cur = [i] + [0] * len(hyp)
        for j, h in enumerate(hyp, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (r != h))
        prev = cur
    return prev[-1]


def word_error_rate(predictions: Iterable[str], references: Iterable[str]) -> float:
    """Corpus WER = sum of word edit distances / sum of reference lengths (src/train.py:79-80)."""
    predictions, references = list(predictions), list(references)
    if len(predictions) != len(references):
        raise ValueError("predictions and references differ in length")
    errors = total = 0
    for p, r in zip(predictions, references):
        rw, pw = r.split(), p.split()
        errors += _edit_distance(rw, pw)
        total += len(rw)
    if total == 0:
        raise ValueError("references contain no words")
    return errors / total


def build_generate_embeds(prompt_embeds: torch.Tensor, prompt_ids: torch.Tensor, audio_embeds: torch.Tensor,
                          audio_chunk_id: int) -> torch.Tensor:
    """initial_embeds of ``ASRModel.generate`` (src/modeling.py:311-327) through the splice kernel.

    prompt_embeds [1, P, D], prompt_ids [1, P], audio_embeds [B, A, D] -> [B, P-1+A, D].  As in the reference, the
    prompt is spliced with the FIRST sample's audio embeddings and then expanded over the batch (:317-327) -- a
    batch > 1 therefore decodes sample 0 B times there; kept identical here, batch 1 is the only meaningful use."""
    from .splice import splice_audio_embeddings

    if prompt_ids.dim() != 2 or prompt_ids.shape[0] != 1:
        raise ValueError("prompt_ids must be [1, P]")
    try:
        first, _, _ = splice_audio_embeddings(prompt_embeds[:1], audio_embeds[:1].to(prompt_embeds.dtype), prompt_ids, None,
                                              audio_chunk_id)
    except ValueError as e:
        raise ValueError("'<|audio_chunk|>' token not found in instruction.") from e  # :314-315
    return first.expand(audio_embeds.shape[0], -1, -1) if audio_embeds.shape[0] > 1 else first


@torch.no_grad()
def evaluate_samples(model, tokenizer, feature_extractor, eval_samples, device=None) -> Tuple[List[str], List[str], float]:
    """Same contract as the reference's ``evaluate_samples`` (src/train.py:36-82): greedy decode of every sample,
    one at a time, returns (predictions, references, wer)."""
    if device is None:
        device = model.device
    predictions, references = [], []
    for sample in eval_samples:
        inputs = feature_extractor(sample["audio"]["array"], sampling_rate=16000, return_tensors="pt",
                                   return_attention_mask=True)
        input_features = inputs.input_features.to(device)
        attention_mask = inputs.attention_mask.to(device) if hasattr(inputs, "attention_mask") else None
        generated_ids = model.generate(input_features=input_features, attention_mask=attention_mask, max_new_tokens=100,
                                       do_sample=False)
        predictions.append(tokenizer.decode(generated_ids[0], skip_special_tokens=True))
        references.append(sample.get("text", sample.get("sentence", "")))
    return predictions, references, word_error_rate(predictions, references)
