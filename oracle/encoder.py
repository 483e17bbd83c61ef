"""CPU fp32 restatement of the Whisper-style audio encoder forward
(TEST INFRASTRUCTURE, see oracle/__init__.py).

Follows the third-party arithmetic the reference wrapper calls (src/modeling.py:33-40):

* HF transformers/models/whisper/modeling_whisper.py:612-646  encoder forward (length check,
  conv stem + GELU, permute, + embed_positions, layer loop, final LayerNorm)
* :379-413 encoder layer (pre-LN, attention, residual, pre-LN, fc1 + GELU, fc2, residual)
* :284-356 attention (q scaled by head_dim**-0.5 BEFORE the score matmul, k_proj has no bias,
  no mask, scaling=1.0) and :215-238 eager softmax(QK^T)V
* :55-64 sinusoidal positions

Parameters come in as a plain dict keyed like the HF encoder's state dict ("conv1.weight",
"layers.0.self_attn.q_proj.weight", ...).  torch is used only as a CPU tensor library
(matmul / conv1d / erf); every step is written out.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(x)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def attention(x, sd, prefix, n_heads):
    B, T, D = x.shape
    hd = D // n_heads
    q = (x @ _t(sd[prefix + "q_proj.weight"]).T + _t(sd[prefix + "q_proj.bias"])) * (hd ** -0.5)  # :309
    k = x @ _t(sd[prefix + "k_proj.weight"]).T  # no bias (:279)
    v = x @ _t(sd[prefix + "v_proj.weight"]).T + _t(sd[prefix + "v_proj.bias"])
    q = q.view(B, T, n_heads, hd).transpose(1, 2)
    k = k.view(B, T, n_heads, hd).transpose(1, 2)
    v = v.view(B, T, n_heads, hd).transpose(1, 2)
    s = q @ k.transpose(2, 3)  # scaling = 1.0 (:348)
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, T, D)
    return o @ _t(sd[prefix + "out_proj.weight"]).T + _t(sd[prefix + "out_proj.bias"])


def encoder_forward(sd: dict, input_features, n_heads: int, n_layers: int | None = None,
                    return_intermediates: bool = False):
    """input_features [B, n_mels, 3000] -> last_hidden_state [B, 1500, d] (float32)."""
    x = _t(input_features).to(torch.float32)
    if x.shape[-1] != 3000:  # modeling_whisper.py:612-616
        raise ValueError(
            f"Whisper expects the mel input features to be of length 3000, but found {x.shape[-1]}."
        )
    if n_layers is None:
        n_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
    inter = {}
    with torch.no_grad():
        h = gelu_erf(F.conv1d(x, _t(sd["conv1.weight"]), _t(sd["conv1.bias"]), padding=1))  # :618
        inter["conv1"] = h
        h = gelu_erf(F.conv1d(h, _t(sd["conv2.weight"]), _t(sd["conv2.bias"]), stride=2, padding=1))  # :619
        h = h.permute(0, 2, 1) + _t(sd["embed_positions.weight"])[:1500]  # :621-624
        inter["stem"] = h
        for i in range(n_layers):
            p = f"layers.{i}."
            r = h
            y = layer_norm(h, _t(sd[p + "self_attn_layer_norm.weight"]), _t(sd[p + "self_attn_layer_norm.bias"]))
            h = r + attention(y, sd, p + "self_attn.", n_heads)
            r = h
            y = layer_norm(h, _t(sd[p + "final_layer_norm.weight"]), _t(sd[p + "final_layer_norm.bias"]))
            y = gelu_erf(y @ _t(sd[p + "fc1.weight"]).T + _t(sd[p + "fc1.bias"]))
            h = r + (y @ _t(sd[p + "fc2.weight"]).T + _t(sd[p + "fc2.bias"]))
            inter[f"layer{i}"] = h
        h = layer_norm(h, _t(sd["layer_norm.weight"]), _t(sd["layer_norm.bias"]))  # :642
    if return_intermediates:
        return h, inter
    return h
