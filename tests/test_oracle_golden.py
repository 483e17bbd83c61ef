"""CPU: pins the oracle (oracle/) against the golden fixtures generated from the installed HF
transformers code and the reference's AudioProjector (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from oracle import encoder as oe
from oracle import logmel as ol
from oracle import projector as op
from tests import signals


@pytest.mark.parametrize("n_mels", [80, 128])
def test_mel_filter_bank_is_bit_identical(golden_dir, n_mels):
    g = np.load(f"{golden_dir}/melfilt_{n_mels}.npz")["mel_filters"]
    assert np.array_equal(ol.mel_filter_bank(n_mels), g)
    assert (g > 0).sum(axis=1).max() <= 2  # every FFT bin feeds at most two filters (sparse kernel relies on it)


@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("case", list(signals.LOGMEL_CASES))
def test_logmel_oracle_matches_hf(golden_dir, n_mels, case):
    g = np.load(f"{golden_dir}/logmel_{n_mels}.npz")
    r = ol.extract(signals.LOGMEL_CASES[case][0](), n_mels=n_mels, sampling_rate=16000)
    assert r["input_features"].shape == (1, n_mels, 3000) and r["input_features"].dtype == np.float32
    assert np.abs(r["input_features"][0] - g[case + "__features"]).max() <= 1e-4
    assert r["attention_mask"].dtype == np.int32 and np.array_equal(r["attention_mask"][0], g[case + "__mask"])


def test_logmel_oracle_batch_longest_and_errors(golden_dir):
    g = np.load(f"{golden_dir}/logmel_80.npz")
    clips = [signals.LOGMEL_CASES[n][0]() for n in ("short3005", "tiny250")]
    r = ol.extract(clips, n_mels=80, sampling_rate=16000, padding="longest")
    assert np.abs(r["input_features"] - g["longest__features"]).max() <= 1e-4
    assert np.array_equal(r["attention_mask"], g["longest__mask"])
    with pytest.raises(ValueError):
        ol.extract(clips[0], sampling_rate=8000)
    with pytest.raises(ValueError):
        ol.extract(np.zeros((2, 2, 10), dtype=np.float32), sampling_rate=16000)
    # float64 arithmetic agrees with the float32 path far inside the tolerance
    a = ol.log_mel(clips[0][None, :16000], 80, dtype=np.float32)
    b = ol.log_mel(clips[0][None, :16000], 80, dtype=np.float64)
    assert np.abs(a - b).max() < 2e-5


@pytest.mark.parametrize("case", ["tiny", "mid", "small"])
def test_encoder_oracle_matches_hf(golden_dir, case):
    c = signals.ENCODER_CASES[case]
    g = np.load(f"{golden_dir}/encoder_{case}.npz")
    sd = signals.encoder_state_dict(c["d_model"], c["layers"], c["ffn"], c["n_mels"], seed=c["seed"])
    y = oe.encoder_forward(sd, signals.encoder_features(case), c["heads"]).numpy()
    ref, mine = (g["out"], y) if "out" in g else (g["out_rows"], y[:, g["rows"], :])
    assert np.abs(mine - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    with pytest.raises(ValueError):
        oe.encoder_forward(sd, np.zeros((1, c["n_mels"], 1500), dtype=np.float32), c["heads"])


@pytest.mark.parametrize("case", ["p128", "p768"])
def test_projector_oracle_matches_reference(golden_dir, case):
    c = signals.PROJECTOR_CASES[case]
    g = np.load(f"{golden_dir}/projector_{case}.npz")
    sd = signals.projector_state_dict(c["audio_dim"], c["text_dim"], seed=c["seed"])
    x, gy = signals.projector_inputs(case)
    y, grads = op.projector_backward(sd, x, gy)
    y = y.numpy()
    if "out" in g:
        assert np.abs(y - g["out"]).max() <= 1e-6
    else:
        assert np.abs(y[:, ::20] - g["out_rows"]).max() <= 1e-6
    for k in ("norm.weight", "linear_1.weight", "linear_1.bias", "linear_2.weight", "linear_2.bias"):
        ref = g["grad__" + k.replace(".", "__")]
        m = grads[k].numpy()
        if m.shape != ref.shape:
            m = m[::8, ::8]
        assert np.linalg.norm(m - ref) <= 2e-6 * np.linalg.norm(ref), k
