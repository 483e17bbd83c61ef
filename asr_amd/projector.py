"""Drop-in for the reference's ``AudioProjector`` (src/modeling.py:63-84).

Same constructor ``(audio_dim, text_dim)``, same parameter names (``norm.weight``,
``linear_1.{weight,bias}``, ``linear_2.{weight,bias}``), same init (src/modeling.py:72-75), same
forward signature, trainable through autograd.  Forward and backward run in libasrk.so
(``asrk_projector_forward`` / ``asrk_projector_backward``).
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.nn as nn

from . import _lib

_DT = {torch.float32: _lib.ASRK_F32, torch.bfloat16: _lib.ASRK_BF16}


class _RMSNormParams(nn.Module):
    """Holds ``weight`` like HF LlamaRMSNorm (models/llama/modeling_llama.py:53-60)."""

    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps


def _params_struct(gamma, w1, b1, w2, b2, packed=None):
    return _lib.ProjectorParams(*[ctypes.c_void_p(t.data_ptr()) for t in (gamma, w1, b1, w2, b2)],
                                ctypes.c_void_p(packed.data_ptr() if packed is not None else None))


_WS = {}


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    ws = _WS.get(dev)
    if ws is None or ws.numel() < nbytes:
        _WS[dev] = None
        ws = _WS[dev] = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    return ws


class _PackedWeights:
    """The bf16 copies of linear_1 / linear_2 (and their transposes for the backward) made by
    ``asrk_projector_pack``, kept until a parameter changes: the key is (storage address, ``_version``)
    of both weights, which every in-place update (an optimizer step, ``load_state_dict``, ``.to()``)
    changes.  The reference gets the same reuse from autocast's weight-cast cache."""

    def __init__(self):
        self.key = None
        self.buf = None

    def get(self, lib, dims, w1, w2, ps) -> torch.Tensor:
        key = (w1.data_ptr(), w1._version, w2.data_ptr(), w2._version, w1.device, w1.dtype)
        if key != self.key:
            if self.buf is None or self.buf.device != ps[1].device:
                self.buf = torch.empty((lib.asrk_projector_packed_bytes(ctypes.byref(dims)),), dtype=torch.uint8,
                                       device=ps[1].device)
            pstruct = _params_struct(*ps)
            _lib.check(lib.asrk_projector_pack(ctypes.byref(dims), ctypes.byref(pstruct), _lib.ptr(self.buf),
                                               _lib.current_stream()), "asrk_projector_pack")
            self.key = key
        return self.buf


class _ProjectorFn(torch.autograd.Function):
    """``x`` is the projector's input, or -- with ``encoder`` given -- the encoder's ``input_features``:
    then both run in one ``asrk_encoder_projector_forward`` call and ``last_hidden_state`` exists only
    if the backward will need it."""

    @staticmethod
    def forward(ctx, x, gamma, w1, b1, w2, b2, out_dtype, grad_enabled, cache, encoder):
        lib = _lib.load()
        A, T = w1.shape[1], w1.shape[0]
        dims = _lib.ProjectorDims(A, T)
        ps = [t.detach().to(torch.float32).contiguous() for t in (gamma, w1, b1, w2, b2)]
        need_grad = grad_enabled and any(ctx.needs_input_grad[1:6])  # no saved activations under no_grad
        if encoder is None:
            x2 = x.detach().reshape(-1, A).to(torch.float32).contiguous()
            out_shape = (*x.shape[:-1], T)
            need_grad = need_grad or (grad_enabled and ctx.needs_input_grad[0])
        else:
            if encoder.d_model != A:
                raise ValueError(f"projector audio_dim {A} does not match the encoder's d_model {encoder.d_model}")
            feats, handle, enc_ws = encoder._prepare(x)
            out_shape = (feats.shape[0], 1500, T)
            x2 = torch.empty((feats.shape[0] * 1500, A), dtype=torch.float32, device=feats.device) if need_grad else None
        dev = ps[1].device
        M = out_shape[0] * 1500 if encoder is not None else x2.shape[0]
        out = torch.empty((M, T), dtype=out_dtype, device=dev)
        if M == 0:
            return out.reshape(out_shape)
        with torch.cuda.device(dev):
            packed = cache.get(lib, dims, w1, w2, ps) if cache is not None else None
            ws = _workspace(dev, lib.asrk_projector_workspace_bytes(ctypes.byref(dims), M))
            saved = None
            if need_grad:
                saved = torch.empty((lib.asrk_projector_saved_bytes(ctypes.byref(dims), M),), dtype=torch.uint8, device=dev)
            pstruct = _params_struct(*ps, packed=packed)
            if encoder is None:
                _lib.check(lib.asrk_projector_forward(ctypes.byref(dims), ctypes.byref(pstruct), _lib.ptr(x2), M,
                                                      _lib.ptr(out), _DT[out_dtype], _lib.ptr(saved), _lib.ptr(ws),
                                                      ws.numel(), _lib.current_stream()), "asrk_projector_forward")
            else:
                _lib.check(lib.asrk_encoder_projector_forward(handle, _lib.ptr(feats), feats.shape[0], 3000, _lib.ptr(x2),
                                                              _lib.ptr(enc_ws), enc_ws.numel(), ctypes.byref(dims),
                                                              ctypes.byref(pstruct), _lib.ptr(out), _DT[out_dtype],
                                                              _lib.ptr(saved), _lib.ptr(ws), ws.numel(),
                                                              _lib.current_stream()), "asrk_encoder_projector_forward")
        if need_grad:
            ctx.save_for_backward(x2, saved, *ps)
            ctx.packed = packed  # valid in the backward: autograd forbids in-place changes of saved parameters
            ctx.dims = (A, T, M)
            ctx.x_shape = x.shape
            ctx.fused = encoder is not None
        return out.reshape(out_shape)

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        x2, saved, gamma, w1, b1, w2, b2 = ctx.saved_tensors
        A, T, M = ctx.dims
        dims = _lib.ProjectorDims(A, T)
        go = grad_out.reshape(M, T)
        if go.dtype not in _DT:
            go = go.to(torch.float32)
        go = go.contiguous()
        dev = x2.device
        g = [torch.empty_like(t) for t in (gamma, w1, b1, w2, b2)]
        gx = torch.empty_like(x2) if (ctx.needs_input_grad[0] and not ctx.fused) else None
        with torch.cuda.device(dev):
            ws = _workspace(dev, lib.asrk_projector_workspace_bytes(ctypes.byref(dims), M))
            pstruct = _params_struct(gamma, w1, b1, w2, b2, packed=ctx.packed)
            gstruct = _lib.ProjectorGrads(*[ctypes.c_void_p(t.data_ptr()) for t in g])
            _lib.check(lib.asrk_projector_backward(ctypes.byref(dims), ctypes.byref(pstruct), _lib.ptr(go), _DT[go.dtype],
                                                   M, _lib.ptr(saved), ctypes.byref(gstruct), _lib.ptr(gx), _lib.ptr(x2),
                                                   _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                       "asrk_projector_backward")
        return (gx.reshape(ctx.x_shape) if gx is not None else None, g[0], g[1], g[2], g[3], g[4], None, None, None, None)


class AudioProjector(nn.Module):
    def __init__(self, audio_dim: int, text_dim: int, out_dtype: Optional[torch.dtype] = None):
        super().__init__()
        self.norm = _RMSNormParams(audio_dim, eps=1e-6)
        self.linear_1 = nn.Linear(audio_dim, text_dim, bias=True)
        self.act = nn.GELU()
        self.linear_2 = nn.Linear(text_dim, text_dim, bias=True)
        # reference init, src/modeling.py:72-75
        nn.init.normal_(self.linear_1.weight, mean=0.0, std=0.01)
        nn.init.zeros_(self.linear_1.bias)
        nn.init.normal_(self.linear_2.weight)
        nn.init.zeros_(self.linear_2.bias)
        self.out_dtype = out_dtype  # None: fp32, or bf16 under torch.autocast(bfloat16) like the reference
        self._packed = _PackedWeights()

    def refresh_weights(self):
        """Re-make the bf16 weight copies on the next call.  Needed only after an update that bumps no version counter
        (``p.data.copy_()`` / ``p.data.add_()``, as some sharding wrappers and third-party optimizers do)."""
        self._packed.key = None

    def _out_dtype(self):
        if self.out_dtype is not None:
            return self.out_dtype
        return torch.bfloat16 if (torch.is_autocast_enabled() and
                                  torch.get_autocast_gpu_dtype() == torch.bfloat16) else torch.float32

    def _apply_fn(self, x, encoder):
        if self.linear_1.weight.device.type != "cuda":
            raise _lib.AsrkError("AudioProjector runs on a ROCm GPU only (there is no CPU fallback); call .to('cuda')")
        return _ProjectorFn.apply(x, self.norm.weight, self.linear_1.weight, self.linear_1.bias, self.linear_2.weight,
                                  self.linear_2.bias, self._out_dtype(), torch.is_grad_enabled(), self._packed, encoder)

    def forward(self, audio_features: torch.Tensor) -> torch.Tensor:
        if not audio_features.is_cuda:
            raise _lib.AsrkError("AudioProjector runs on a ROCm GPU only (there is no CPU fallback)")
        return self._apply_fn(audio_features, None)


def encode_and_project(encoder, projector: AudioProjector, input_features: torch.Tensor) -> torch.Tensor:
    """``projector(encoder(input_features))`` (src/modeling.py:213-220, :296-301) as ONE library call: the encoder's
    final LayerNorm feeds the projector's RMSNorm directly and ``last_hidden_state`` is materialised only when the
    projector is being trained (its backward reads it).  Same values, bit for bit, as the two separate modules."""
    return projector._apply_fn(input_features, encoder)
