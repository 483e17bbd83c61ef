"""On-device collate: raw clips -> log-mel features without padding on the host.

The reference's ``DataCollator`` (src/train.py:172-180) hands a list of 1-30 s float arrays to the HF feature
extractor, which pads every clip to 30 s on the CPU, and the training loop then copies the features to the GPU.
Here the host only concatenates the VALID samples of the batch into one pinned buffer (float32, or int16 PCM for
half the bytes), one H2D copy moves it, and ``asrk_collate_pcm`` pads / truncates / builds the frame mask on the
GPU (HF feature_extraction_whisper.py:300-307, 332-341) in front of the log-mel kernel.

    fe = LogMelFeatureExtractor(feature_size=80)
    batch = device_collate(fe, clips)                    # BatchFeature: input_features, attention_mask (GPU)
    for batch in FeaturePrefetcher(fe, iterable_of_clip_lists): ...   # copy of batch n+1 under compute of batch n

No CPU fallback: raises AsrkError when libasrk.so or a GPU is missing.
"""
from __future__ import annotations

from typing import Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .feature_extraction import BatchFeature, LogMelFeatureExtractor


def pack_ragged(clips: Sequence, dtype=np.float32, pin: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """(pcm [sum(len)] float32|int16, offsets int64 [B+1]).  CPU only, so it may run inside forked DataLoader workers;
    both tensors are pinned when this process has already initialised the GPU (never the case in such a worker)."""
    if dtype not in (np.float32, np.int16):
        raise ValueError("dtype must be numpy float32 or int16")
    arrs = []
    for c in clips:
        a = np.asarray(c.cpu() if isinstance(c, torch.Tensor) else c)
        if a.ndim != 1:
            raise ValueError("Only mono-channel audio is supported")  # HF :279-280
        if dtype == np.int16 and a.dtype != np.int16:
            raise ValueError("int16 packing needs int16 PCM clips (no silent requantisation)")
        arrs.append(a.astype(dtype, copy=False))
    pin = torch.cuda.is_initialized() if pin is None else pin
    lens = np.array([len(a) for a in arrs], dtype=np.int64)
    offsets = torch.empty((len(arrs) + 1,), dtype=torch.int64, pin_memory=pin)
    offsets[0] = 0
    offsets[1:] = torch.from_numpy(np.cumsum(lens))
    pcm = torch.empty((max(int(lens.sum()), 1),), dtype=torch.float32 if dtype == np.float32 else torch.int16, pin_memory=pin)
    view = pcm.numpy()
    for a, o in zip(arrs, offsets[:-1].tolist()):
        view[o:o + len(a)] = a
    return pcm, offsets


def collate_on_device(pcm: torch.Tensor, offsets: torch.Tensor, target: int, hop: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """pcm / offsets on the GPU -> (wave float32 [B, target], frame mask int32 [B, target // hop])."""
    if not pcm.is_cuda or not offsets.is_cuda or offsets.dtype != torch.int64 or pcm.dtype not in (torch.float32, torch.int16):
        raise ValueError("pcm (float32|int16) and offsets (int64) must be CUDA tensors")
    lib = _lib.load()
    B = offsets.numel() - 1
    with torch.cuda.device(pcm.device):
        wave = torch.empty((B, target), dtype=torch.float32, device=pcm.device)
        mask = torch.empty((B, target // hop), dtype=torch.int32, device=pcm.device)
        _lib.check(lib.asrk_collate_pcm(_lib.ptr(pcm), _lib.ASRK_F32 if pcm.dtype == torch.float32 else _lib.ASRK_I16,
                                        _lib.ptr(offsets), B, target, hop, _lib.ptr(wave), _lib.ptr(mask),
                                        _lib.current_stream()), "asrk_collate_pcm")
    return wave, mask


def _valid_lengths(offsets: torch.Tensor, target: int) -> torch.Tensor:
    """Samples of each clip in front of the zero padding asrk_collate_pcm writes (clips are truncated to `target`)."""
    return (offsets[1:] - offsets[:-1]).clamp(max=target).to(torch.int32)


def device_collate(fe: LogMelFeatureExtractor, clips: Sequence, dtype=np.float32, max_length: Optional[int] = None,
                   device=None) -> BatchFeature:
    """Same result as ``fe(clips, padding="max_length", max_length=..., return_attention_mask=True)``."""
    dev = fe._device(device)
    pcm, offsets = pack_ragged(clips, dtype)
    target = max_length if max_length else fe.n_samples
    with torch.cuda.device(dev):
        d_off = offsets.to(dev, non_blocking=True)
        wave, mask = collate_on_device(pcm.to(dev, non_blocking=True), d_off, target, fe.hop_length)
        feats = fe.extract_from_device(wave, _valid_lengths(d_off, target))
    return BatchFeature(input_features=feats, attention_mask=mask)


class FeaturePrefetcher:
    """Iterates ``BatchFeature``s for an iterable of clip lists, keeping ``depth`` batches ahead: packing runs on the
    host, the H2D copy on a side stream, and the compute stream only waits for the copy's event -- so the PCIe
    transfer of batch n+1 runs under the kernels of batch n."""

    def __init__(self, fe: LogMelFeatureExtractor, batches: Iterable[Sequence], dtype=np.float32,
                 max_length: Optional[int] = None, device=None, depth: int = 2):
        self.fe, self.batches, self.dtype, self.depth = fe, batches, dtype, max(1, depth)
        self.target = max_length if max_length else fe.n_samples
        self.dev = fe._device(device)
        self.copy_stream = torch.cuda.Stream(device=self.dev)

    def _stage(self, clips):
        # an item may already be packed -- (pcm, offsets) pinned tensors, e.g. from DataLoader workers
        if isinstance(clips, tuple) and len(clips) == 2 and all(isinstance(t, torch.Tensor) for t in clips):
            pcm, offsets = clips
        else:
            pcm, offsets = pack_ragged(clips, self.dtype)
        with torch.cuda.stream(self.copy_stream):
            d_pcm = pcm.to(self.dev, non_blocking=True)
            d_off = offsets.to(self.dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return pcm, offsets, d_pcm, d_off, ev  # the pinned sources stay referenced until the batch is consumed

    def __iter__(self) -> Iterator[BatchFeature]:
        it = iter(self.batches)
        queue = []
        for clips in it:
            queue.append(self._stage(clips))
            if len(queue) > self.depth:
                yield self._finish(queue.pop(0))
        while queue:
            yield self._finish(queue.pop(0))

    def _finish(self, staged) -> BatchFeature:
        _pcm, _off, d_pcm, d_off, ev = staged
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(ev)
        d_pcm.record_stream(cur)
        d_off.record_stream(cur)
        with torch.cuda.device(self.dev):
            wave, mask = collate_on_device(d_pcm, d_off, self.target, self.fe.hop_length)
            feats = self.fe.extract_from_device(wave, _valid_lengths(d_off, self.target))
        return BatchFeature(input_features=feats, attention_mask=mask)
