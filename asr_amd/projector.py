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


def _params_struct(gamma, w1, b1, w2, b2):
    return _lib.ProjectorParams(*[ctypes.c_void_p(t.data_ptr()) for t in (gamma, w1, b1, w2, b2)])


_WS = {}


def _workspace(dev: torch.device, nbytes: int) -> torch.Tensor:
    ws = _WS.get(dev)
    if ws is None or ws.numel() < nbytes:
        _WS[dev] = None
        ws = _WS[dev] = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    return ws


class _ProjectorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, w1, b1, w2, b2, out_dtype, grad_enabled):
        lib = _lib.load()
        A, T = w1.shape[1], w1.shape[0]
        x2 = x.detach().reshape(-1, A).to(torch.float32).contiguous()
        M = x2.shape[0]
        dims = _lib.ProjectorDims(A, T)
        ps = [t.detach().to(torch.float32).contiguous() for t in (gamma, w1, b1, w2, b2)]
        need_grad = grad_enabled and any(ctx.needs_input_grad[:6])  # no saved activations under no_grad
        out = torch.empty((M, T), dtype=out_dtype, device=x2.device)
        with torch.cuda.device(x2.device):
            ws = _workspace(x2.device, lib.asrk_projector_workspace_bytes(ctypes.byref(dims), M))
            saved = None
            if need_grad:
                saved = torch.empty((lib.asrk_projector_saved_bytes(ctypes.byref(dims), M),), dtype=torch.uint8,
                                    device=x2.device)
            pstruct = _params_struct(*ps)
            _lib.check(lib.asrk_projector_forward(ctypes.byref(dims), ctypes.byref(pstruct), _lib.ptr(x2), M,
                                                  _lib.ptr(out), _DT[out_dtype], _lib.ptr(saved), _lib.ptr(ws),
                                                  ws.numel(), _lib.current_stream()), "asrk_projector_forward")
        if need_grad:
            ctx.save_for_backward(x2, saved, *ps)
            ctx.dims = (A, T, M)
            ctx.x_shape = x.shape
        return out.reshape(*x.shape[:-1], T)

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
        gx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        with torch.cuda.device(dev):
            ws = _workspace(dev, lib.asrk_projector_workspace_bytes(ctypes.byref(dims), M))
            pstruct = _params_struct(gamma, w1, b1, w2, b2)
            gstruct = _lib.ProjectorGrads(*[ctypes.c_void_p(t.data_ptr()) for t in g])
            _lib.check(lib.asrk_projector_backward(ctypes.byref(dims), ctypes.byref(pstruct), _lib.ptr(go), _DT[go.dtype],
                                                   M, _lib.ptr(saved), ctypes.byref(gstruct), _lib.ptr(gx), _lib.ptr(x2),
                                                   _lib.ptr(ws), ws.numel(), _lib.current_stream()),
                       "asrk_projector_backward")
        return (gx.reshape(ctx.x_shape) if gx is not None else None, g[0], g[1], g[2], g[3], g[4], None, None)


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

    def forward(self, audio_features: torch.Tensor) -> torch.Tensor:
        if not audio_features.is_cuda:
            raise _lib.AsrkError("AudioProjector runs on a ROCm GPU only (there is no CPU fallback)")
        out_dtype = self.out_dtype
        if out_dtype is None:
            out_dtype = torch.bfloat16 if (torch.is_autocast_enabled() and
                                           torch.get_autocast_gpu_dtype() == torch.bfloat16) else torch.float32
        return _ProjectorFn.apply(audio_features, self.norm.weight, self.linear_1.weight, self.linear_1.bias,
                                  self.linear_2.weight, self.linear_2.bias, out_dtype, torch.is_grad_enabled())
