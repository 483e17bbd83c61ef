"""Deterministic synthetic inputs shared by tests/golden/make_golden.py and the tests.

Everything is integer hashing + exact float conversions (plus np.sin/np.cos on float64 for the
tonal signals), so the same arrays are regenerated bit-for-bit wherever the tests run and the
fixtures only need to hold the expected OUTPUTS.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def det_uniform(n: int, seed: int) -> np.ndarray:
    """n float64 values in [0, 1), exactly reproducible."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x100000001B3)
    h = _splitmix64(idx)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def det_normal(shape, seed: int, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """Approximately N(mean, std^2) float32 (Irwin-Hall sum of 12 uniforms: only adds, so exact)."""
    n = int(np.prod(shape))
    acc = np.zeros(n, dtype=np.float64)
    for k in range(12):
        acc += det_uniform(n, seed * 12 + k + 1)
    acc -= 6.0
    return (acc * std + mean).astype(np.float32).reshape(shape)


# ---------------------------------------------------------------------------- audio clips
def clip_noise(n: int, seed: int = 1, amp: float = 0.1) -> np.ndarray:
    return np.clip(det_normal((n,), seed, std=amp), -1.0, 1.0).astype(np.float32)


def clip_tone_chirp(n: int) -> np.ndarray:
    t = np.arange(n, dtype=np.float64) / 16000.0
    dur = n / 16000.0
    tone = 0.3 * np.sin(2 * np.pi * 440.0 * t)
    chirp = 0.2 * np.sin(2 * np.pi * (200.0 * t + 0.5 * (6000.0 - 200.0) / dur * t * t))
    return (tone + chirp).astype(np.float32)


def clip_square(n: int, period: int = 100) -> np.ndarray:
    return np.where((np.arange(n) // (period // 2)) % 2 == 0, 1.0, -1.0).astype(np.float32)


def clip_speechlike(n: int, seed: int = 7) -> np.ndarray:
    """Amplitude-modulated noise + harmonics: a wide dynamic range with silent gaps."""
    t = np.arange(n, dtype=np.float64) / 16000.0
    env = np.clip(np.sin(2 * np.pi * 1.7 * t), 0.0, None) ** 2
    harm = sum(np.sin(2 * np.pi * 120.0 * k * t) / k for k in range(1, 12))
    x = env * (0.15 * harm + det_normal((n,), seed, std=0.03).astype(np.float64))
    return np.clip(x, -1.0, 1.0).astype(np.float32)


LOGMEL_CASES = {
    # name: (builder, n_samples)
    "noise30": (lambda: clip_noise(480000, seed=1), 480000),
    "tonechirp30": (lambda: clip_tone_chirp(480000), 480000),
    "short3005": (lambda: clip_noise(48077, seed=2), 48077),  # 3.005 s, not a multiple of 160
    "silence": (lambda: np.zeros(480000, dtype=np.float32), 480000),
    "square30": (lambda: clip_square(480000), 480000),
    "speechlike12": (lambda: clip_speechlike(192000), 192000),
    "tiny250": (lambda: clip_noise(250, seed=3), 250),  # shorter than one window
}


# ---------------------------------------------------------------------------- encoder / projector params
def encoder_state_dict(d_model: int, n_layers: int, ffn: int, n_mels: int, seed: int = 0, w_std: float | None = None):
    """Deterministic WhisperEncoder-shaped parameters keyed exactly like HF's state dict
    (modeling_whisper.py:566-573; sinusoidal embed_positions as :55-64)."""
    import math

    sd = {}
    s = [seed * 1000]

    def nxt(shape, std):
        s[0] += 1
        return det_normal(shape, s[0], std=std)

    std = w_std if w_std is not None else 1.0 / math.sqrt(d_model)
    sd["conv1.weight"] = nxt((d_model, n_mels, 3), 1.0 / math.sqrt(3 * n_mels))
    sd["conv1.bias"] = nxt((d_model,), 0.05)
    sd["conv2.weight"] = nxt((d_model, d_model, 3), 1.0 / math.sqrt(3 * d_model))
    sd["conv2.bias"] = nxt((d_model,), 0.05)
    half = d_model // 2
    inc = math.log(10000.0) / (half - 1)
    inv = np.exp(-inc * np.arange(half, dtype=np.float32)).astype(np.float32)
    st = np.arange(1500, dtype=np.float32)[:, None] * inv[None, :]
    sd["embed_positions.weight"] = np.concatenate([np.sin(st), np.cos(st)], axis=1).astype(np.float32)
    for i in range(n_layers):
        p = f"layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nm}.weight"] = nxt((d_model, d_model), std)
            if nm != "k_proj":
                sd[p + f"self_attn.{nm}.bias"] = nxt((d_model,), 0.05)
        sd[p + "self_attn_layer_norm.weight"] = 1.0 + nxt((d_model,), 0.1)
        sd[p + "self_attn_layer_norm.bias"] = nxt((d_model,), 0.05)
        sd[p + "fc1.weight"] = nxt((ffn, d_model), std)
        sd[p + "fc1.bias"] = nxt((ffn,), 0.05)
        sd[p + "fc2.weight"] = nxt((d_model, ffn), 1.0 / math.sqrt(ffn))
        sd[p + "fc2.bias"] = nxt((d_model,), 0.05)
        sd[p + "final_layer_norm.weight"] = 1.0 + nxt((d_model,), 0.1)
        sd[p + "final_layer_norm.bias"] = nxt((d_model,), 0.05)
    sd["layer_norm.weight"] = 1.0 + nxt((d_model,), 0.1)
    sd["layer_norm.bias"] = nxt((d_model,), 0.05)
    return sd


def projector_state_dict(audio_dim: int, text_dim: int, seed: int = 0):
    """Deterministic AudioProjector parameters; scales follow the reference init
    (src/modeling.py:72-75: W1 ~ N(0, 0.01^2), W2 ~ N(0, 1)) with non-zero biases / gamma so the
    bias and gamma paths are exercised."""
    b = seed * 1000 + 500
    return {
        "norm.weight": 1.0 + det_normal((audio_dim,), b + 1, std=0.1),
        "linear_1.weight": det_normal((text_dim, audio_dim), b + 2, std=0.01),
        "linear_1.bias": det_normal((text_dim,), b + 3, std=0.02),
        "linear_2.weight": det_normal((text_dim, text_dim), b + 4, std=1.0),
        "linear_2.bias": det_normal((text_dim,), b + 5, std=0.02),
    }


# ---------------------------------------------------------------------------- golden cases (inputs only)
# The cases tests/golden/make_golden.py generated the fixtures for.  They live here, not in the generator, so that no
# test imports the script that imports transformers / the reference (it never travels to the GPU box as a dependency).
ENCODER_CASES = {
    # name: dict(d_model, layers, heads, ffn, n_mels, batch, feat_seed, rows)
    "tiny": dict(d_model=128, layers=2, heads=2, ffn=512, n_mels=80, batch=2, seed=11, rows=None),
    "mid": dict(d_model=256, layers=4, heads=4, ffn=1024, n_mels=128, batch=1, seed=12, rows=None),
    "small": dict(d_model=768, layers=12, heads=12, ffn=3072, n_mels=80, batch=1, seed=13,
                  rows=list(range(0, 1500, 25)) + [1, 2, 3, 749, 1496, 1497, 1498, 1499]),
}

PROJECTOR_CASES = {
    "p128": dict(audio_dim=128, text_dim=192, batch=2, seed=21),
    "p768": dict(audio_dim=768, text_dim=960, batch=1, seed=22),
}


def encoder_features(case):
    c = ENCODER_CASES[case]
    x = det_normal((c["batch"], c["n_mels"], 3000), seed=c["seed"] * 7 + 1, std=0.45)
    return np.clip(x, -1.0, 1.6).astype(np.float32)


def projector_inputs(case):
    c = PROJECTOR_CASES[case]
    x = det_normal((c["batch"], 1500, c["audio_dim"]), seed=c["seed"] * 5 + 1, std=1.3)
    gy = det_normal((c["batch"], 1500, c["text_dim"]), seed=c["seed"] * 5 + 2, std=1.0)
    return x, gy
