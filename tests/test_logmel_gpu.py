"""GPU parity of the HIP log-mel front-end (through the C ABI) against the oracle and the HF
golden fixtures.  Tolerance: 1e-4 absolute (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import logmel as oracle_logmel
from tests import signals

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def extractors():
    from asr_amd import LogMelFeatureExtractor

    return {n: LogMelFeatureExtractor(feature_size=n) for n in (80, 128)}


@pytest.mark.parametrize("n_mels", [80, 128])
@pytest.mark.parametrize("case", list(signals.LOGMEL_CASES))
def test_single_clip_matches_golden(extractors, golden_dir, margins, n_mels, case):
    g = np.load(f"{golden_dir}/logmel_{n_mels}.npz")
    clip = signals.LOGMEL_CASES[case][0]()
    r = extractors[n_mels](clip, sampling_rate=16000, return_tensors="pt", return_attention_mask=True)
    assert r.input_features.is_cuda and r.input_features.dtype == torch.float32
    feats = r.input_features.cpu().numpy()
    assert feats.shape == (1, n_mels, 3000)
    ref = g[case + "__features"]
    err = np.abs(feats[0] - ref).max()
    margins.add(f"log-mel vs HF fixture, {case}/{n_mels} mel", err, TOL, "abs")
    assert err <= TOL, f"{case}/{n_mels}: max abs err {err:.3e}"
    assert r.attention_mask.dtype == torch.int32
    assert np.array_equal(r.attention_mask.cpu().numpy()[0], g[case + "__mask"])


@pytest.mark.parametrize("n_mels", [80, 128])
def test_collator_batch_matches_golden(extractors, golden_dir, margins, n_mels):
    g = np.load(f"{golden_dir}/logmel_{n_mels}.npz")
    names = ["short3005", "speechlike12", "tiny250", "noise30"]
    clips = [signals.LOGMEL_CASES[n][0]() for n in names]
    r = extractors[n_mels](clips, sampling_rate=16000, return_tensors="pt", return_attention_mask=True,
                           padding="max_length", max_length=480000)
    feats = r["input_features"].cpu().numpy()
    assert feats.shape == (4, n_mels, 3000)
    assert np.array_equal(r["attention_mask"].cpu().numpy(), g["batch__mask"])
    for i, n in enumerate(names):
        err = np.abs(feats[i] - g[n + "__features"]).max()
        margins.add(f"log-mel collator batch vs HF fixture, {n}/{n_mels} mel", err, TOL, "abs")
        assert err <= TOL, n
    # per-clip max: a clip's features do not depend on its batch neighbours
    alone = extractors[n_mels](clips[1], sampling_rate=16000, return_tensors="np").input_features
    assert np.array_equal(alone[0], feats[1])


@pytest.mark.parametrize("n_mels", [80, 128])
def test_padding_longest(extractors, golden_dir, margins, n_mels):
    g = np.load(f"{golden_dir}/logmel_{n_mels}.npz")
    clips = [signals.LOGMEL_CASES["short3005"][0](), signals.LOGMEL_CASES["tiny250"][0]()]
    r = extractors[n_mels](clips, sampling_rate=16000, return_tensors="np", return_attention_mask=True,
                           padding="longest")
    assert r.input_features.shape == g["longest__features"].shape
    err = np.abs(r.input_features - g["longest__features"]).max()
    margins.add(f"log-mel padding=longest vs HF fixture, {n_mels} mel", err, TOL, "abs")
    assert err <= TOL
    assert np.array_equal(r.attention_mask, g["longest__mask"])


def test_matches_oracle_on_random_batch(extractors, margins):
    rng = np.random.default_rng(5)
    lens = [480000, 16000, 77777, 401, 239999]
    clips = [np.clip(rng.standard_normal(n) * 0.2, -1, 1).astype(np.float32) for n in lens]
    r = extractors[80](clips, sampling_rate=16000, return_tensors="np", return_attention_mask=True)
    o = oracle_logmel.extract(clips, n_mels=80, sampling_rate=16000)
    err = np.abs(r.input_features - o["input_features"]).max()
    margins.add("log-mel ragged batch vs oracle, 80 mel", err, TOL, "abs")
    assert err <= TOL
    assert np.array_equal(r.attention_mask, o["attention_mask"])


def test_errors_and_edge_cases(extractors):
    fe = extractors[80]
    with pytest.raises(ValueError):
        fe(np.zeros(1000, dtype=np.float32), sampling_rate=8000)
    with pytest.raises(ValueError):
        fe(np.zeros((2, 2, 100), dtype=np.float32), sampling_rate=16000)
    # list of python floats, float64 array, truncation of > 30 s
    r = fe([0.0] * 500, sampling_rate=16000, return_tensors="pt")
    assert r.input_features.shape == (1, 80, 3000)
    long_clip = np.ones(500000, dtype=np.float64) * 0.01
    r = fe(long_clip, sampling_rate=16000, return_tensors="pt", return_attention_mask=True)
    assert r.input_features.shape == (1, 80, 3000) and int(r.attention_mask.sum()) == 3000
    # device-resident input
    w = torch.zeros((3, 480000), device="cuda")
    f = fe.extract_from_device(w)
    assert f.shape == (3, 80, 3000) and torch.all(f == -1.5)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_full_size_properties(extractors, n_mels):
    """BASELINE config 2 size (256 x 30 s, 80 and 128 mel bins): properties that need no oracle run."""
    fe = extractors[n_mels]
    g = torch.Generator(device="cuda").manual_seed(3)
    w = (torch.randn((256, 480000), device="cuda", generator=g) * 0.1).clamp_(-1, 1)
    w[7, 100000:] = 0.0  # silence tail
    f = fe.extract_from_device(w)
    torch.cuda.synchronize()
    assert torch.isfinite(f).all()
    mx = f.amax(dim=(1, 2))
    mn = f.amin(dim=(1, 2))
    assert torch.all(mn >= mx - 2.0 - 1e-6)  # max-8 floor in log10 units == 2.0 after /4
    assert tuple(f.shape) == (256, n_mels, 3000)
    assert torch.all(f[7, :, 700:] == f[7, 0, 2999])  # silence tail sits on the floor
    # clip-in-batch == clip-alone, bit for bit
    assert torch.equal(fe.extract_from_device(w[5:6].clone())[0], f[5])
    # scaling the waveform by 2 shifts log-mel by log10(4)/4 where the floor is not active
    f2 = fe.extract_from_device(w[:4] * 2.0)
    d = (f2 - f[:4])
    assert torch.allclose(d[:3], torch.full_like(d[:3], float(np.log10(4.0) / 4.0)), atol=2e-5)


@pytest.mark.parametrize("n_mels", [80, 128])
def test_valid_lengths_change_no_bit(extractors, n_mels):
    """asrk_logmel_f32's optional `valid_samples`: tiles that lie entirely in a clip's zero padding are written as silence
    without being transformed.  Same bits as the full computation, for lengths at tile and hop boundaries, a clip that
    fills the window, an empty clip, and a non-multiple-of-24-frames window (reflection at the end)."""
    fe = extractors[n_mels]
    for total, lens in ((480000, [480000, 0, 1, 3839, 3840, 3841, 200000, 479599, 479999, 77777]),
                        (100160, [100160, 50000, 96000, 99759, 99760, 123, 4000])):
        g = torch.Generator(device="cuda").manual_seed(total)
        w = (torch.randn((len(lens), total), device="cuda", generator=g) * 0.1).clamp_(-1, 1)
        for i, n in enumerate(lens):
            w[i, n:] = 0.0
        full = fe.extract_from_device(w)
        fast = fe.extract_from_device(w, lens)
        assert torch.equal(full, fast)
        assert torch.equal(fe.extract_from_device(w, torch.tensor(lens, device="cuda")), full)
    with pytest.raises(ValueError):
        fe.extract_from_device(w, [1, 2])


def test_plan_of_another_device_is_refused(extractors):
    """asrk_logmel_f32 / asrk_logmel_f32_noalloc compare the device a plan was created on with the current one (its
    tables are that GPU's memory).  The box has one GPU, so the test patches the plan's recorded device -- `struct
    asrk_logmel_plan { int n_mels; int device; ... }`, asrk_logmel.hip -- and restores it."""
    import ctypes

    from asr_amd import _lib

    lib = _lib.load()
    fe = extractors[80]
    w = torch.zeros((2, 480000), device="cuda")
    ref = fe.extract_from_device(w)
    plan = fe._plan(w.device)
    dev_field = ctypes.c_int.from_address(plan.value + 4)
    assert ctypes.c_int.from_address(plan.value).value == 80 and dev_field.value == w.device.index
    out = torch.empty_like(ref)
    ws = torch.empty((int(lib.asrk_logmel_workspace_bytes(2, 480000)),), dtype=torch.uint8, device="cuda")
    dev_field.value = w.device.index + 1
    try:
        rc = lib.asrk_logmel_f32(plan, _lib.ptr(w), _lib.ptr(None), 2, 480000, _lib.ptr(out), _lib.ptr(ws), ws.numel(),
                                 _lib.current_stream())
        rc2 = lib.asrk_logmel_f32_noalloc(plan, _lib.ptr(w), 2, 480000, _lib.ptr(out), _lib.current_stream())
    finally:
        dev_field.value = w.device.index
    assert rc == _lib.ASRK_ERR_UNSUPPORTED and rc2 == _lib.ASRK_ERR_UNSUPPORTED
    assert torch.equal(fe.extract_from_device(w), ref)
