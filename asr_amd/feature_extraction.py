"""Drop-in for ``transformers.WhisperFeatureExtractor`` on the reference's two call sites.

    src/train.py:54-59    fe(audio_array, sampling_rate=16000, return_tensors="pt", return_attention_mask=True)
    src/train.py:172-180  fe(audio_arrays, sampling_rate=..., return_tensors="pt", return_attention_mask=True,
                             padding="max_length", max_length=480000)

Same keyword arguments, same return keys (attribute AND item access, as the callers use
``inputs.input_features`` / ``hasattr(inputs, "attention_mask")``), same ValueErrors
(HF feature_extraction_whisper.py:265-271, 279-280).  The host side only validates, pads into a
pinned staging buffer and builds the frame mask (HF :285-307, 332-341; sequence_utils.py:224-337);
the STFT / mel / log arithmetic (HF :135-168) runs in the HIP kernel behind ``asrk_logmel_f32``.

Differences that are deliberate: tensors come back on the GPU (``.to(device)`` in the callers is
then a no-op) unless ``return_tensors="np"``; raw speech may also be passed as a CUDA tensor
``[B, n_samples]`` (already padded) to skip the host staging entirely.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .mel import HOP_LENGTH, N_FFT, N_SAMPLES, SAMPLING_RATE, slaney_mel_filter_bank


class BatchFeature(dict):
    """Minimal mapping with attribute access, like transformers.BatchFeature."""

    def __getattr__(self, item):
        try:
            return self[item]
        except KeyError as e:
            raise AttributeError(item) from e

    def to(self, *args, **kwargs):
        for k, v in list(self.items()):
            if isinstance(v, torch.Tensor):
                if v.is_floating_point():
                    self[k] = v.to(*args, **kwargs)
                else:  # HF only casts floating tensors; integer masks just move device
                    dev = kwargs.get("device", next((a for a in args if isinstance(a, (str, torch.device))), None))
                    self[k] = v.to(device=dev) if dev is not None else v
        return self


class LogMelFeatureExtractor:
    """Whisper log-mel front-end running on MI355X.  ``WhisperFeatureExtractor`` is an alias."""

    model_input_names = ["input_features"]

    def __init__(self, feature_size: int = 80, sampling_rate: int = SAMPLING_RATE, hop_length: int = HOP_LENGTH,
                 chunk_length: int = 30, n_fft: int = N_FFT, padding_value: float = 0.0, dither: float = 0.0,
                 return_attention_mask: bool = False, device: Union[str, torch.device, None] = None, **kwargs):
        if (sampling_rate, hop_length, n_fft) != (SAMPLING_RATE, HOP_LENGTH, N_FFT):
            raise ValueError("the MI355X front-end implements Whisper's 16 kHz / n_fft 400 / hop 160 analysis only")
        if dither != 0.0:
            raise ValueError("dither is not implemented (the reference never sets it)")
        self.feature_size = feature_size
        self.sampling_rate = sampling_rate
        self.hop_length = hop_length
        self.chunk_length = chunk_length
        self.n_fft = n_fft
        self.padding_value = padding_value
        self.dither = dither
        self.return_attention_mask = return_attention_mask
        self.padding_side = "right"
        self.n_samples = chunk_length * sampling_rate
        self.nb_max_frames = self.n_samples // hop_length
        self.mel_filters = slaney_mel_filter_bank(feature_size)  # float64 [201, n_mels], as HF keeps it
        self.device = torch.device(device) if device is not None else None
        self._plans = {}  # device index -> plan handle

    # ------------------------------------------------------------------ serialisation (HF on-disk format)
    def to_dict(self):
        return {
            "feature_extractor_type": "WhisperFeatureExtractor", "feature_size": self.feature_size,
            "sampling_rate": self.sampling_rate, "hop_length": self.hop_length, "chunk_length": self.chunk_length,
            "n_fft": self.n_fft, "padding_value": self.padding_value, "dither": self.dither,
            "return_attention_mask": self.return_attention_mask, "padding_side": self.padding_side,
            "n_samples": self.n_samples, "nb_max_frames": self.nb_max_frames, "processor_class": "WhisperProcessor",
        }

    def save_pretrained(self, save_directory: str, **kwargs):
        """Writes preprocessor_config.json in the format stock transformers reads (src/modeling.py:157-161)."""
        import json
        import os

        os.makedirs(save_directory, exist_ok=True)
        path = os.path.join(save_directory, "preprocessor_config.json")
        with open(path, "w") as f:
            json.dump(self.to_dict(), f, indent=2, sort_keys=True)
        return [path]

    @classmethod
    def from_pretrained(cls, path: str, **kwargs):
        import json
        import os

        with open(os.path.join(path, "preprocessor_config.json")) as f:
            cfg = json.load(f)
        keys = ("feature_size", "sampling_rate", "hop_length", "chunk_length", "n_fft", "padding_value", "dither",
                "return_attention_mask")
        return cls(**{k: cfg[k] for k in keys if k in cfg}, **kwargs)

    # ------------------------------------------------------------------ device plumbing
    def _device(self, device) -> torch.device:
        dev = torch.device(device) if device is not None else self.device
        if dev is not None and dev.type != "cuda":
            # HF's `device="cpu"` asks for CPU arithmetic and CPU tensors; this drop-in has neither, and quietly running
            # on a GPU instead would surprise the caller.  The reference's call sites never pass `device`.
            raise _lib.AsrkError(
                f"LogMelFeatureExtractor computes on a ROCm GPU only (device={str(dev)!r} requested): pass device=None or "
                "'cuda[:i]', and use return_tensors='np' for host arrays.  There is no CPU fallback.")
        if not torch.cuda.is_available():
            raise _lib.AsrkError("LogMelFeatureExtractor needs a ROCm GPU: there is no CPU fallback")
        if dev is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        return dev

    def _plan(self, dev: torch.device):
        h = self._plans.get(dev.index)
        if h is None:
            lib = _lib.load()
            filt = np.ascontiguousarray(self.mel_filters.astype(np.float32))  # HF :156 casts to float32
            handle = ctypes.c_void_p()
            with torch.cuda.device(dev):
                _lib.check(lib.asrk_logmel_plan_create(self.feature_size, filt.ctypes.data_as(ctypes.c_void_p),
                                                        ctypes.byref(handle)), "asrk_logmel_plan_create")
            self._plans[dev.index] = h = handle
        return h

    def extract_from_device(self, wave: torch.Tensor, lengths=None) -> torch.Tensor:
        """wave: CUDA float32 [B, n_samples] (padded) -> CUDA float32 [B, n_mels, n_samples // 160].

        ``lengths`` (optional; array-like or tensor of B sample counts): the valid length of each clip in front of its
        ZERO padding.  It changes no value -- the kernel only skips the transform of tiles that lie entirely in the
        padding, which come out as exact silence either way."""
        _lib.assert_not_forked()
        if wave.dim() != 2 or wave.dtype != torch.float32 or not wave.is_cuda:
            raise ValueError("expected a CUDA float32 tensor of shape [batch, n_samples]")
        wave = wave.contiguous()
        B, L = wave.shape
        if L <= self.n_fft // 2:
            raise ValueError(f"reflect padding needs more than {self.n_fft // 2} samples, got {L}")
        lib = _lib.load()
        with torch.cuda.device(wave.device):
            out = torch.empty((B, self.feature_size, L // self.hop_length), dtype=torch.float32, device=wave.device)
            ws = torch.empty((max(int(lib.asrk_logmel_workspace_bytes(B, L)), 256),), dtype=torch.uint8, device=wave.device)
            lens = None
            if lengths is not None:
                lens = torch.as_tensor(lengths).to(device=wave.device, dtype=torch.int32, non_blocking=True).contiguous()
                if lens.numel() != B:
                    raise ValueError(f"lengths has {lens.numel()} entries for a batch of {B}")
            _lib.check(lib.asrk_logmel_f32(self._plan(wave.device), _lib.ptr(wave), _lib.ptr(lens), B, L, _lib.ptr(out),
                                           _lib.ptr(ws), ws.numel(), _lib.current_stream()), "asrk_logmel_f32")
        return out

    # ------------------------------------------------------------------ __call__ (HF :193-346)
    def __call__(self, raw_speech, truncation: bool = True, pad_to_multiple_of: Optional[int] = None,
                 return_tensors: Optional[str] = None, return_attention_mask: Optional[bool] = None,
                 padding: Optional[str] = "max_length", max_length: Optional[int] = None,
                 sampling_rate: Optional[int] = None, do_normalize: Optional[bool] = None,
                 device: Union[str, torch.device, None] = None, **kwargs) -> BatchFeature:
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(
                f"The model corresponding to this feature extractor: {self.__class__.__name__} was trained using a"
                f" sampling rate of {self.sampling_rate}. Please make sure that the provided `raw_speech` input"
                f" was sampled with {self.sampling_rate} and not {sampling_rate}."
            )
        if do_normalize:
            raise ValueError("do_normalize is not implemented (the reference never sets it)")
        if return_attention_mask is None:
            return_attention_mask = self.return_attention_mask
        if str(return_tensors) not in ("pt", "np", "None"):
            raise ValueError("return_tensors must be 'pt', 'np' or None")
        _lib.assert_not_forked()  # DataLoader workers: fail with the num_workers=0 / FeaturePrefetcher guidance, not inside HIP

        if isinstance(raw_speech, torch.Tensor) and raw_speech.is_cuda:  # on-device collate path
            if raw_speech.dim() == 1:
                raw_speech = raw_speech[None]
            if raw_speech.dim() > 2:
                raise ValueError(f"Only mono-channel audio is supported for input to {self}")
            feats = self.extract_from_device(raw_speech.to(torch.float32))
            res = BatchFeature(input_features=feats)
            if return_attention_mask:
                res["attention_mask"] = torch.ones((feats.shape[0], feats.shape[2]), dtype=torch.int32,
                                                   device=feats.device)
            return self._convert(res, return_tensors)
        if isinstance(raw_speech, torch.Tensor):
            raw_speech = raw_speech.numpy()

        dev = self._device(device)
        host, lengths, target = self.prepare_host(raw_speech, truncation=truncation, pad_to_multiple_of=pad_to_multiple_of,
                                                   padding=padding, max_length=max_length)
        with torch.cuda.device(dev):
            wave = host.to(dev, non_blocking=True)
            feats = self.extract_from_device(wave, lengths if self.padding_value == 0.0 else None)
        res = BatchFeature(input_features=feats)
        if return_attention_mask:
            res["attention_mask"] = torch.from_numpy(self.frame_mask(lengths, target)).to(dev, non_blocking=True)
        return self._convert(res, return_tensors)

    # ------------------------------------------------------------------ host-side steps (no GPU needed)
    def prepare_host(self, raw_speech, truncation: bool = True, pad_to_multiple_of: Optional[int] = None,
                     padding: Optional[str] = "max_length", max_length: Optional[int] = None):
        """Batching, truncation and right-padding of HF __call__ (:278-307; sequence_utils.py:187-220,
        296-337).  Returns (float32 tensor [B, target]; lengths; target).  The tensor is pinned when this process has
        already initialised the GPU; a DataLoader worker (no GPU use allowed there) gets pageable memory, so this
        step stays usable inside forked workers."""
        is_batched_numpy = isinstance(raw_speech, np.ndarray) and raw_speech.ndim > 1
        if is_batched_numpy and raw_speech.ndim > 2:
            raise ValueError(f"Only mono-channel audio is supported for input to {self}")
        is_batched = is_batched_numpy or (
            isinstance(raw_speech, (list, tuple)) and len(raw_speech) > 0
            and isinstance(raw_speech[0], (np.ndarray, tuple, list, torch.Tensor))
        )
        clips: Sequence = raw_speech if is_batched else [raw_speech]
        clips = [np.asarray(c.cpu() if isinstance(c, torch.Tensor) else c, dtype=np.float32).reshape(-1) for c in clips]
        max_length = max_length if max_length else self.n_samples
        if pad_to_multiple_of is not None and max_length % pad_to_multiple_of != 0:
            max_length = (max_length // pad_to_multiple_of + 1) * pad_to_multiple_of
        if truncation:
            clips = [c[:max_length] for c in clips]
        if padding in (True, "longest"):
            target = max(len(c) for c in clips)
            if pad_to_multiple_of is not None and target % pad_to_multiple_of != 0:
                target = (target // pad_to_multiple_of + 1) * pad_to_multiple_of
        elif padding == "max_length":
            target = max(max_length, max(len(c) for c in clips)) if not truncation else max_length
        else:
            raise ValueError("padding must be 'max_length' or 'longest': the device kernel works on one dense batch")
        B = len(clips)
        host = torch.empty((B, target), dtype=torch.float32, pin_memory=torch.cuda.is_initialized())
        host_np = host.numpy()
        lengths = np.empty((B,), dtype=np.int64)
        for i, c in enumerate(clips):
            n = min(len(c), target)
            host_np[i, :n] = c[:n]
            host_np[i, n:] = self.padding_value
            lengths[i] = n
        return host, lengths, target

    def frame_mask(self, lengths: np.ndarray, target: int) -> np.ndarray:
        """int32 [B, target // 160]: the sample mask taken every hop, trimmed when `target` is not a
        multiple of the hop (HF :332-341)."""
        n_mask = target // self.hop_length
        frame_idx = np.arange(n_mask, dtype=np.int64)[None, :] * self.hop_length
        return (frame_idx < np.asarray(lengths)[:, None]).astype(np.int32)

    @staticmethod
    def _convert(res: BatchFeature, return_tensors) -> BatchFeature:
        if return_tensors == "pt":
            return res
        return BatchFeature({k: v.cpu().numpy() for k, v in res.items()})


WhisperFeatureExtractor = LogMelFeatureExtractor
