"""CPU fp32 restatement of the reference AudioProjector forward and backward
(TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows src/modeling.py:63-84 (RMSNorm(eps=1e-6) -> Linear -> exact-erf GELU -> Linear -> *0.01)
and HF transformers/models/llama/modeling_llama.py:62-67 (LlamaRMSNorm: fp32 upcast,
x * rsqrt(mean(x^2) + eps), then weight * x).

The backward is written out by hand (no autograd) so that it is an independent statement of the
gradients the reference's autograd would produce for gamma, W1, b1, W2, b2.  The projector input
comes from the frozen no-grad encoder (src/modeling.py:36), so no input gradient is needed, but
it is returned as well for completeness.
"""
from __future__ import annotations

import math

import torch

EPS = 1e-6
OUT_SCALE = 0.01


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(x)


def projector_forward(sd: dict, x, return_saved: bool = False):
    x = _t(x).to(torch.float32)
    g = _t(sd["norm.weight"])
    w1, b1 = _t(sd["linear_1.weight"]), _t(sd["linear_1.bias"])
    w2, b2 = _t(sd["linear_2.weight"]), _t(sd["linear_2.bias"])
    rstd = torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + EPS)
    xhat = x * rstd
    xn = g * xhat
    pre = xn @ w1.T + b1
    h = 0.5 * pre * (1.0 + torch.erf(pre / math.sqrt(2.0)))
    y = (h @ w2.T + b2) * OUT_SCALE
    if return_saved:
        return y, (x, rstd, xhat, xn, pre, h)
    return y


def projector_backward(sd: dict, x, grad_out):
    """Returns dict of gradients keyed like the state dict (+ 'input')."""
    y, (x, rstd, xhat, xn, pre, h) = projector_forward(sd, x, return_saved=True)
    g = _t(sd["norm.weight"])
    w1 = _t(sd["linear_1.weight"])
    w2 = _t(sd["linear_2.weight"])
    dy = _t(grad_out).to(torch.float32) * OUT_SCALE  # through the *0.01
    M = dy.numel() // dy.shape[-1]
    dy2 = dy.reshape(M, -1)
    h2 = h.reshape(M, -1)
    pre2 = pre.reshape(M, -1)
    xn2 = xn.reshape(M, -1)
    xhat2 = xhat.reshape(M, -1)
    grads = {}
    grads["linear_2.weight"] = dy2.T @ h2
    grads["linear_2.bias"] = dy2.sum(0)
    dh = dy2 @ w2
    cdf = 0.5 * (1.0 + torch.erf(pre2 / math.sqrt(2.0)))
    pdf = torch.exp(-0.5 * pre2 * pre2) / math.sqrt(2.0 * math.pi)
    dpre = dh * (cdf + pre2 * pdf)
    grads["linear_1.weight"] = dpre.T @ xn2
    grads["linear_1.bias"] = dpre.sum(0)
    dxn = dpre @ w1
    grads["norm.weight"] = (dxn * xhat2).sum(0)
    dxhat = dxn * g
    D = x.shape[-1]
    rstd2 = rstd.reshape(M, 1)
    dx = rstd2 * (dxhat - xhat2 * (dxhat * xhat2).sum(-1, keepdim=True) / D)
    grads["input"] = dx.reshape(x.shape)
    return y, grads
