"""CPU restatement of the Whisper log-mel front-end (TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows, step by step, the third-party code the reference calls at
src/train.py:54-59 (eval, single clip) and src/train.py:172-180 (collator, batch):

* HF transformers/models/whisper/feature_extraction_whisper.py:193-346  (__call__: validation,
  batching, pad / truncate, mask rescale)
* HF transformers/feature_extraction_sequence_utils.py:224-337         (_pad / _truncate, int32 mask)
* HF transformers/models/whisper/feature_extraction_whisper.py:135-168  (_torch_extract_fbank_features)
* HF transformers/audio_utils.py:448-517, 541-560, 638-729              (slaney mel filter bank)

Everything here is numpy/scipy; there is no torch.stft call, so the restatement is an
independent statement of the arithmetic (explicit reflect pad, framing, Hann window, 400-point
real DFT, power, mel projection, log10 clamp, per-clip max-8 floor, (x+4)/4).
"""
from __future__ import annotations

import numpy as np
import scipy.fft

SAMPLING_RATE = 16000
N_FFT = 400
HOP = 160
N_SAMPLES = 480000  # chunk_length 30 s * 16 kHz   (feature_extraction_whisper.py:91)
N_BINS = 1 + N_FFT // 2


# ----------------------------------------------------------------------------------------------
# mel filter bank  (audio_utils.py:638-729 with norm="slaney", mel_scale="slaney")
# ----------------------------------------------------------------------------------------------
def _hz_to_mel_slaney(freq):
    # audio_utils.py:470-481
    freq = np.asarray(freq, dtype=np.float64)
    mels = 3.0 * freq / 200.0
    logstep = 27.0 / np.log(6.4)
    log_region = freq >= 1000.0
    with np.errstate(divide="ignore", invalid="ignore"):
        mels = np.where(log_region, 15.0 + np.log(np.maximum(freq, 1e-300) / 1000.0) * logstep, mels)
    return mels


def _mel_to_hz_slaney(mels):
    # audio_utils.py:506-517
    mels = np.asarray(mels, dtype=np.float64)
    freq = 200.0 * mels / 3.0
    logstep = np.log(6.4) / 27.0
    log_region = mels >= 15.0
    return np.where(log_region, 1000.0 * np.exp(logstep * (mels - 15.0)), freq)


def mel_filter_bank(n_mels: int) -> np.ndarray:
    """[201, n_mels] float64, as WhisperFeatureExtractor.__init__ builds it
    (feature_extraction_whisper.py:95-103)."""
    mel_min = float(_hz_to_mel_slaney(0.0))
    mel_max = float(_hz_to_mel_slaney(8000.0))
    mel_freqs = np.linspace(mel_min, mel_max, n_mels + 2)
    filter_freqs = _mel_to_hz_slaney(mel_freqs)
    fft_freqs = np.linspace(0, SAMPLING_RATE // 2, N_BINS)  # audio_utils.py:713
    # audio_utils.py:556-560
    filter_diff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2 : n_mels + 2] - filter_freqs[:n_mels])  # audio_utils.py:719
    fb = fb * enorm[None, :]
    return fb


# ----------------------------------------------------------------------------------------------
# host-side batching / padding  (feature_extraction_whisper.py:265-319, sequence_utils.py:224-337)
# ----------------------------------------------------------------------------------------------
def pad_batch(raw_speech, sampling_rate=None, padding="max_length", max_length=None, truncation=True):
    """Returns (wave [B, L] float32, sample_mask [B, L] int32)."""
    if sampling_rate is not None and sampling_rate != SAMPLING_RATE:
        raise ValueError(f"sampling rate {sampling_rate} != {SAMPLING_RATE}")
    is_batched_numpy = isinstance(raw_speech, np.ndarray) and raw_speech.ndim > 1
    if is_batched_numpy and raw_speech.ndim > 2:
        raise ValueError("Only mono-channel audio is supported")
    is_batched = is_batched_numpy or (
        isinstance(raw_speech, (list, tuple)) and isinstance(raw_speech[0], (np.ndarray, tuple, list))
    )
    clips = [np.asarray(c, dtype=np.float32) for c in raw_speech] if is_batched else [np.asarray(raw_speech, dtype=np.float32)]
    max_length = max_length if max_length else N_SAMPLES
    if truncation:
        clips = [c[:max_length] for c in clips]
    if padding in (True, "longest"):
        tgt = max(len(c) for c in clips)
    elif padding == "max_length":
        tgt = max_length
    else:
        raise ValueError("oracle supports padding='max_length' / 'longest' only")
    wave = np.zeros((len(clips), tgt), dtype=np.float32)
    mask = np.zeros((len(clips), tgt), dtype=np.int32)
    for i, c in enumerate(clips):
        n = min(len(c), tgt)
        wave[i, :n] = c[:n]
        mask[i, : len(c)] = 1
    return wave, mask


def rescale_mask(sample_mask: np.ndarray) -> np.ndarray:
    # feature_extraction_whisper.py:332-341
    m = sample_mask[:, ::HOP]
    if sample_mask.shape[1] % HOP != 0:
        m = m[:, :-1]
    return m


# ----------------------------------------------------------------------------------------------
# log-mel arithmetic  (feature_extraction_whisper.py:135-168)
# ----------------------------------------------------------------------------------------------
def hann_window_periodic(dtype=np.float32) -> np.ndarray:
    # torch.hann_window(400) is periodic: 0.5 - 0.5 cos(2 pi n / N)   (:141)
    n = np.arange(N_FFT, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * n / N_FFT)).astype(dtype)


def log_mel(wave: np.ndarray, n_mels: int = 80, dtype=np.float32, mel_filters: np.ndarray | None = None) -> np.ndarray:
    """wave [B, L] -> [B, n_mels, L // 160] in `dtype` arithmetic (float32 mirrors the
    reference's torch path; float64 is the high-precision cross-check)."""
    wave = np.ascontiguousarray(wave, dtype=dtype)
    if wave.ndim == 1:
        wave = wave[None]
    B, L = wave.shape
    if L <= N_FFT // 2:
        raise ValueError("reflect padding needs more than n_fft//2 samples")
    n_frames = L // HOP  # 1 + L//160 STFT frames, last one dropped (:154)
    fb = (mel_filter_bank(n_mels) if mel_filters is None else mel_filters).astype(dtype)  # (:156)
    win = hann_window_periodic(dtype)
    out = np.empty((B, n_mels, n_frames), dtype=dtype)
    for b in range(B):
        # torch.stft(center=True, pad_mode="reflect")   (:149)
        padded = np.pad(wave[b], (N_FFT // 2, N_FFT // 2), mode="reflect")
        frames = np.lib.stride_tricks.sliding_window_view(padded, N_FFT)[::HOP][:n_frames]
        spec = scipy.fft.rfft(frames * win[None, :], axis=-1)  # complex64 for float32 input
        power = (spec.real.astype(dtype) ** 2 + spec.imag.astype(dtype) ** 2).astype(dtype)  # abs()**2 (:154)
        mel = (power @ fb).T  # mel_filters.T @ magnitudes (:157)
        log_spec = np.log10(np.maximum(mel, dtype(1e-10)))  # (:159)
        log_spec = np.maximum(log_spec, log_spec.max() - dtype(8.0))  # per-clip max (:160-162)
        out[b] = (log_spec + dtype(4.0)) / dtype(4.0)  # (:165)
    return out


def extract(raw_speech, n_mels=80, sampling_rate=None, padding="max_length", max_length=None,
            return_attention_mask=True, dtype=np.float32):
    """Full __call__ restatement: returns dict(input_features f32 [B,n_mels,F], attention_mask i32 [B,F])."""
    wave, mask = pad_batch(raw_speech, sampling_rate, padding, max_length)
    feats = log_mel(wave, n_mels, dtype=dtype).astype(np.float32)
    res = {"input_features": feats}
    if return_attention_mask:
        res["attention_mask"] = rescale_mask(mask)
    return res
