"""asr_amd: MI355X-native audio hot path (log-mel -> Whisper-style encoder -> projector) of
alexkroman/asr, behind the reference's own call signatures.  All arithmetic runs in libasrk.so
(hand-written gfx950 HIP kernels, C ABI in include/asrk.h); there is no CPU fallback."""
from . import _lib  # noqa: F401
from .feature_extraction import BatchFeature, LogMelFeatureExtractor, WhisperFeatureExtractor  # noqa: F401

__all__ = ["BatchFeature", "LogMelFeatureExtractor", "WhisperFeatureExtractor"]
