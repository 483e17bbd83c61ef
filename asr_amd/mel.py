"""Host-side constant: the slaney mel filter bank the Whisper front-end uses.

Mirrors HF ``mel_filter_bank(201, n_mels, 0, 8000, 16000, norm="slaney", mel_scale="slaney")``
(transformers/audio_utils.py:638-729 with :448-517 and :541-560), which
``WhisperFeatureExtractor.__init__`` calls (feature_extraction_whisper.py:95-103).  Built once in
float64 and handed to ``asrk_logmel_plan_create`` as float32 [201][n_mels].
"""
from __future__ import annotations

import numpy as np

SAMPLING_RATE = 16000
N_FFT = 400
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLING_RATE
NB_MAX_FRAMES = N_SAMPLES // HOP_LENGTH

_MIN_LOG_HZ = 1000.0
_MIN_LOG_MEL = 15.0
_LOGSTEP = 27.0 / np.log(6.4)


def _hz_to_mel(f: np.ndarray) -> np.ndarray:
    f = np.atleast_1d(np.asarray(f, dtype=np.float64))
    mel = 3.0 * f / 200.0
    hi = f >= _MIN_LOG_HZ
    mel[hi] = _MIN_LOG_MEL + np.log(f[hi] / _MIN_LOG_HZ) * _LOGSTEP
    return mel


def _mel_to_hz(mel: np.ndarray) -> np.ndarray:
    mel = np.atleast_1d(np.asarray(mel, dtype=np.float64))
    f = 200.0 * mel / 3.0
    hi = mel >= _MIN_LOG_MEL
    f[hi] = _MIN_LOG_HZ * np.exp((np.log(6.4) / 27.0) * (mel[hi] - _MIN_LOG_MEL))
    return f


def slaney_mel_filter_bank(n_mels: int, n_bins: int = 1 + N_FFT // 2, f_min: float = 0.0, f_max: float = 8000.0,
                           sampling_rate: int = SAMPLING_RATE) -> np.ndarray:
    """float64 [n_bins, n_mels] triangular, area-normalised (slaney) filters."""
    edges_mel = np.linspace(_hz_to_mel(f_min)[0], _hz_to_mel(f_max)[0], n_mels + 2)
    edges_hz = _mel_to_hz(edges_mel)
    bin_hz = np.linspace(0, sampling_rate // 2, n_bins)
    width = np.diff(edges_hz)
    dist = edges_hz[None, :] - bin_hz[:, None]
    falling = -dist[:, :-2] / width[:-1]
    rising = dist[:, 2:] / width[1:]
    bank = np.maximum(0.0, np.minimum(falling, rising))
    bank *= (2.0 / (edges_hz[2:] - edges_hz[:-2]))[None, :]
    return bank
