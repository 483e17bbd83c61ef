"""Drop-in for the reference's ``WhisperEncoder`` wrapper (src/modeling.py:20-40).

Same call signature -- ``encoder(input_features, attention_mask=None) -> FloatTensor [B, 1500, d]``
-- same ``.d_model`` attribute, same frozen / eval / no-grad behaviour (src/modeling.py:27-31,36),
same ``ValueError`` when the mel length is not 3000 (HF modeling_whisper.py:612-616), and the same
parameter names (``whisper.encoder.conv1.weight`` ... ``whisper.encoder.layer_norm.bias``) so a
reference checkpoint loads with ``load_state_dict``.  The torch submodules below only HOLD the
parameters; the arithmetic runs in libasrk.so through ``asrk_encoder_forward``.
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _lib


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """Whisper's fixed positional table (HF modeling_whisper.py:55-64)."""
    if channels % 2 != 0:
        raise ValueError("channels must be even")
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length).view(-1, 1) * inv.view(1, -1)
    return torch.cat([t.sin(), t.cos()], dim=1)


class _Attention(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.k_proj = nn.Linear(d, d, bias=False)  # HF :279
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)


class _Layer(nn.Module):
    def __init__(self, d, ffn):
        super().__init__()
        self.self_attn = _Attention(d)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, ffn)
        self.fc2 = nn.Linear(ffn, d)
        self.final_layer_norm = nn.LayerNorm(d)


class _EncoderParams(nn.Module):
    """Parameter container laid out like HF WhisperEncoder (modeling_whisper.py:566-573)."""

    def __init__(self, d, layers, ffn, n_mels, max_source_positions=1500):
        super().__init__()
        self.conv1 = nn.Conv1d(n_mels, d, kernel_size=3, padding=1)
        self.conv2 = nn.Conv1d(d, d, kernel_size=3, stride=2, padding=1)
        self.embed_positions = nn.Embedding(max_source_positions, d)
        with torch.no_grad():
            self.embed_positions.weight.copy_(sinusoids(max_source_positions, d))
        self.layers = nn.ModuleList([_Layer(d, ffn) for _ in range(layers)])
        self.layer_norm = nn.LayerNorm(d)


class _WhisperShell(nn.Module):
    def __init__(self, enc):
        super().__init__()
        self.encoder = enc


class WhisperEncoder(nn.Module):
    """Frozen Whisper-style audio encoder on MI355X.  Defaults are the whisper-small dimensions the
    reference hard-codes (src/modeling.py:23-25); other sizes (e.g. large-v3: 1280/32/20/5120/128)
    are the same code."""

    def __init__(self, d_model: int = 768, encoder_layers: int = 12, encoder_attention_heads: int = 12,
                 encoder_ffn_dim: int = 3072, num_mel_bins: int = 80):
        super().__init__()
        if d_model != 64 * encoder_attention_heads:
            raise ValueError("the MI355X attention kernel implements Whisper's head_dim = 64")
        self.whisper = _WhisperShell(_EncoderParams(d_model, encoder_layers, encoder_ffn_dim, num_mel_bins))
        for p in self.whisper.parameters():
            p.requires_grad = False
        self.d_model = d_model
        self.n_heads = encoder_attention_heads
        self.n_layers = encoder_layers
        self.ffn_dim = encoder_ffn_dim
        self.n_mels = num_mel_bins
        self.whisper.eval()
        self._handle = None
        self._packed_key = None
        self._packed_sentinel = None
        self._stale = True
        self._ws = None

    # ------------------------------------------------------------------ weights
    @classmethod
    def from_state_dict(cls, sd: dict, n_heads: Optional[int] = None, prefix: str = "", device="cuda"):
        """Build from an HF-encoder-keyed state dict (``conv1.weight`` ...); dims are read off the tensors."""
        g = lambda k: sd[prefix + k]  # noqa: E731
        d = g("conv1.weight").shape[0]
        n_mels = g("conv1.weight").shape[1]
        layers = 1 + max(int(k[len(prefix):].split(".")[1]) for k in sd if k.startswith(prefix + "layers."))
        ffn = g("layers.0.fc1.weight").shape[0]
        m = cls(d, layers, n_heads or d // 64, ffn, n_mels)
        tgt = {k: torch.as_tensor(v) for k, v in sd.items() if k.startswith(prefix)}
        m.whisper.encoder.load_state_dict({k[len(prefix):]: v for k, v in tgt.items()}, strict=True)
        return m.to(device)

    # The packed bf16 copy follows the fp32 parameters through everything that replaces or rewrites them wholesale --
    # .to() / .cuda() / .float() (``_apply``) and ``load_state_dict`` mark it stale -- and through in-place edits of
    # the two sentinel parameters checked on every call (the first and the last one).  The full (address, version) key
    # over all ~200-500 parameters costs 0.2-1.2 ms of host time per forward, so it is taken only when one of those
    # says something changed, on every train() / eval() switch, and on every 256th call; after an in-place edit of some
    # OTHER frozen weight call ``refresh_weights()`` to have it seen at once.  Updates through ``p.data`` bump no
    # version counter and are seen by nothing: call ``refresh_weights()`` after those.
    def _apply(self, fn, *args, **kwargs):
        self._stale = True
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._stale = True
        return super().load_state_dict(*args, **kwargs)

    def train(self, mode: bool = True):
        # a mode switch is where fine-tuning code tends to touch weights: take the full key again on the next call
        self._stale = True
        return super().train(mode)

    def refresh_weights(self):
        """Re-pack the frozen weights on the next forward (needed only after editing them in place)."""
        self._stale = True

    def _sentinel(self):
        enc = self.whisper.encoder
        a, b = enc.conv1.weight, enc.layer_norm.bias
        return (a.data_ptr(), a._version, b.data_ptr(), b._version)

    _FULL_CHECK_EVERY = 256  # calls between two full (address, version) keys over every parameter

    def _pack(self):
        self._calls = getattr(self, "_calls", 0) + 1
        if (self._handle is not None and not self._stale and self._sentinel() == self._packed_sentinel
                and self._calls % self._FULL_CHECK_EVERY):
            return self._handle
        enc = self.whisper.encoder
        params = list(enc.parameters())
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._handle is not None and key == self._packed_key:
            self._stale, self._packed_sentinel = False, self._sentinel()
            return self._handle
        dev = params[0].device
        if dev.type != "cuda":
            raise _lib.AsrkError("WhisperEncoder must live on a ROCm GPU (there is no CPU fallback); call .to('cuda')")
        lib = _lib.load()
        self._release()
        f32 = lambda p: p.detach().to(torch.float32).contiguous()  # noqa: E731
        keep = []

        def P(t):
            t = f32(t)
            keep.append(t)
            return ctypes.c_void_p(t.data_ptr())

        layer_arr = (_lib.EncoderLayerWeights * max(1, self.n_layers))()
        for i, L in enumerate(enc.layers):
            lw = layer_arr[i]
            a = L.self_attn
            lw.q_w, lw.q_b = P(a.q_proj.weight), P(a.q_proj.bias)
            lw.k_w = P(a.k_proj.weight)
            lw.v_w, lw.v_b = P(a.v_proj.weight), P(a.v_proj.bias)
            lw.o_w, lw.o_b = P(a.out_proj.weight), P(a.out_proj.bias)
            lw.ln1_w, lw.ln1_b = P(L.self_attn_layer_norm.weight), P(L.self_attn_layer_norm.bias)
            lw.fc1_w, lw.fc1_b = P(L.fc1.weight), P(L.fc1.bias)
            lw.fc2_w, lw.fc2_b = P(L.fc2.weight), P(L.fc2.bias)
            lw.ln2_w, lw.ln2_b = P(L.final_layer_norm.weight), P(L.final_layer_norm.bias)
        w = _lib.EncoderWeights()
        w.conv1_w, w.conv1_b = P(enc.conv1.weight), P(enc.conv1.bias)
        w.conv2_w, w.conv2_b = P(enc.conv2.weight), P(enc.conv2.bias)
        w.pos = P(enc.embed_positions.weight)
        w.ln_w, w.ln_b = P(enc.layer_norm.weight), P(enc.layer_norm.bias)
        w.layers = ctypes.cast(layer_arr, ctypes.POINTER(_lib.EncoderLayerWeights))
        dims = _lib.EncoderDims(self.d_model, self.n_layers, self.n_heads, self.ffn_dim, self.n_mels)
        handle = ctypes.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.asrk_encoder_create(ctypes.byref(dims), ctypes.byref(w), _lib.current_stream(),
                                               ctypes.byref(handle)), "asrk_encoder_create")
            torch.cuda.current_stream().synchronize()  # packing kernels have read the fp32 sources
        self._handle, self._packed_key = handle, key
        self._stale, self._packed_sentinel = False, self._sentinel()
        return handle

    def _release(self):
        if self._handle is not None:
            _lib.load().asrk_encoder_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # ------------------------------------------------------------------ forward (src/modeling.py:33-40)
    def forward(self, input_features: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        # attention_mask is accepted and ignored, exactly as HF does (modeling_whisper.py:607-609, 636-640)
        with torch.no_grad():
            x, handle, ws = self._prepare(input_features)
            B = x.shape[0]
            out = torch.empty((B, 1500, self.d_model), dtype=torch.float32, device=x.device)
            if B == 0:
                return out
            with torch.cuda.device(x.device):
                _lib.check(_lib.load().asrk_encoder_forward(handle, _lib.ptr(x), B, 3000, _lib.ptr(out), _lib.ptr(ws),
                                                            ws.numel(), _lib.current_stream()), "asrk_encoder_forward")
            return out

    def _check_features(self, input_features: torch.Tensor):
        if input_features.shape[-1] != 3000:
            raise ValueError(
                f"Whisper expects the mel input features to be of length 3000, but found "
                f"{input_features.shape[-1]}. Make sure to pad the input mel features to 3000."
            )
        if input_features.dim() != 3 or input_features.shape[1] != self.n_mels:
            raise ValueError(f"expected input_features of shape [batch, {self.n_mels}, 3000]")

    def _prepare(self, input_features: torch.Tensor):
        """Checked fp32 features on the encoder's device, the packed-weights handle and a workspace for this batch
        (shared by ``forward`` and ``asr_amd.projector.encode_and_project``)."""
        self._check_features(input_features)
        with torch.no_grad():
            handle = self._pack()
            dev = self.whisper.encoder.conv1.weight.device
            x = input_features.to(device=dev, dtype=torch.float32).contiguous()
            B = x.shape[0]
            with torch.cuda.device(dev):
                need = _lib.load().asrk_encoder_workspace_bytes(handle, max(B, 1))
                if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                    self._ws = None
                    self._ws = torch.empty((need,), dtype=torch.uint8, device=dev)
            return x, handle, self._ws
