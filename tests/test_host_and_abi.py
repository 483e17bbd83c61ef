"""CPU: host-side logic of the drop-in modules, the C-ABI surface of libasrk.so (load + exported
symbols only: no compute without a GPU), and the host emulation of the kernel's 400-point FFT."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from asr_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "asrk.h")).read()
    declared = set(re.findall(r"\b(asrk_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/asrk.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    loaded = _lib.load()
    assert loaded.asrk_version() >= 100
    assert loaded.asrk_strerror(0) == b"ok" and b"3000" in loaded.asrk_strerror(-7)


def test_mel_bank_matches_hf_bit_for_bit(golden_dir):
    from asr_amd.mel import slaney_mel_filter_bank

    for n in (80, 128):
        assert np.array_equal(slaney_mel_filter_bank(n), np.load(f"{golden_dir}/melfilt_{n}.npz")["mel_filters"])


def test_extractor_host_side_matches_hf_masks(golden_dir):
    from asr_amd import BatchFeature, LogMelFeatureExtractor, WhisperFeatureExtractor
    from tests import signals

    assert WhisperFeatureExtractor is LogMelFeatureExtractor
    fe = LogMelFeatureExtractor(feature_size=80)
    assert fe.n_samples == 480000 and fe.nb_max_frames == 3000 and fe.mel_filters.shape == (201, 80)
    g = np.load(f"{golden_dir}/logmel_80.npz")
    names = ["short3005", "speechlike12", "tiny250", "noise30"]
    clips = [signals.LOGMEL_CASES[n][0]() for n in names]
    host, lengths, target = fe.prepare_host(clips, padding="max_length", max_length=480000)
    assert tuple(host.shape) == (4, 480000) and host.dtype == torch.float32
    assert np.array_equal(fe.frame_mask(lengths, target), g["batch__mask"])
    assert np.array_equal(host[0, :48077].numpy(), clips[0]) and float(host[0, 48077:].abs().max()) == 0.0
    host, lengths, target = fe.prepare_host([clips[0], clips[2]], padding="longest")
    assert target == 48077 and np.array_equal(fe.frame_mask(lengths, target), g["longest__mask"])
    # single clip given as a python list / float64 array / > 30 s (truncated)
    host, lengths, target = fe.prepare_host([0.0] * 100)
    assert tuple(host.shape) == (1, 480000) and lengths[0] == 100
    host, lengths, target = fe.prepare_host(np.ones(500000, dtype=np.float64))
    assert lengths[0] == 480000
    with pytest.raises(ValueError):
        fe(np.zeros(100, dtype=np.float32), sampling_rate=8000)
    with pytest.raises(ValueError):
        fe.prepare_host(np.zeros((2, 2, 100), dtype=np.float32))
    with pytest.raises(ValueError):
        LogMelFeatureExtractor(sampling_rate=8000)
    bf = BatchFeature(input_features=torch.zeros(1, 2), attention_mask=torch.ones(1, 2, dtype=torch.int32))
    assert hasattr(bf, "attention_mask") and bf["input_features"] is bf.input_features and not hasattr(bf, "nope")
    assert bf.to(torch.float64).attention_mask.dtype == torch.int32


def test_extractor_save_load_roundtrip(tmp_path):
    import json

    from asr_amd import LogMelFeatureExtractor

    fe = LogMelFeatureExtractor(feature_size=128, return_attention_mask=True)
    fe.save_pretrained(str(tmp_path))
    cfg = json.load(open(tmp_path / "preprocessor_config.json"))
    assert cfg["feature_extractor_type"] == "WhisperFeatureExtractor" and cfg["feature_size"] == 128
    assert cfg["n_fft"] == 400 and cfg["hop_length"] == 160 and cfg["chunk_length"] == 30
    fe2 = LogMelFeatureExtractor.from_pretrained(str(tmp_path))
    assert fe2.feature_size == 128 and fe2.return_attention_mask is True


def test_extractor_reads_a_stock_written_preprocessor_config(golden_dir, tmp_path):
    """Row f4, ours <- stock: tests/golden/hf_preprocessor_config_*.json were written by stock transformers'
    WhisperFeatureExtractor.save_pretrained (tests/golden/make_golden.py::gen_preprocessor); the reference puts such a
    file next to the model (src/modeling.py:157-161)."""
    import json
    import shutil

    from asr_amd import LogMelFeatureExtractor

    for n_mels in (80, 128):
        d = tmp_path / f"stock{n_mels}"
        d.mkdir()
        shutil.copy(f"{golden_dir}/hf_preprocessor_config_{n_mels}.json", d / "preprocessor_config.json")
        stock = json.load(open(d / "preprocessor_config.json"))
        fe = LogMelFeatureExtractor.from_pretrained(str(d))
        for k in ("feature_size", "sampling_rate", "hop_length", "chunk_length", "n_fft", "padding_value", "dither",
                  "return_attention_mask", "n_samples", "nb_max_frames", "padding_side"):
            assert getattr(fe, k) == stock[k], k
        assert fe.feature_size == n_mels and fe.mel_filters.shape == (201, n_mels)
        # and what we write holds every key the stock file holds, with the same values
        mine = fe.to_dict()
        for k, v in stock.items():
            if not k.startswith("_"):
                assert mine[k] == v, k


def test_stock_transformers_reads_our_preprocessor_config(tmp_path):
    """Row f4, stock <- ours: stock WhisperFeatureExtractor.from_pretrained loads the file our save_pretrained writes,
    with equal fields and a bit-equal filter bank.  Runs where transformers is importable (the build container)."""
    transformers = pytest.importorskip("transformers")
    from asr_amd import LogMelFeatureExtractor

    for n_mels in (80, 128):
        ours = LogMelFeatureExtractor(feature_size=n_mels, return_attention_mask=True)
        d = tmp_path / f"ours{n_mels}"
        ours.save_pretrained(str(d))
        hf = transformers.WhisperFeatureExtractor.from_pretrained(str(d))
        assert type(hf).__name__ == "WhisperFeatureExtractor"
        for k in ("feature_size", "sampling_rate", "hop_length", "chunk_length", "n_fft", "padding_value", "dither",
                  "return_attention_mask", "n_samples", "nb_max_frames", "padding_side"):
            assert getattr(hf, k) == getattr(ours, k), k
        assert np.array_equal(np.asarray(hf.mel_filters), ours.mel_filters)
        # and back: what stock transformers re-saves is read by ours
        d2 = tmp_path / f"hf{n_mels}"
        hf.save_pretrained(str(d2))
        again = LogMelFeatureExtractor.from_pretrained(str(d2))
        assert again.to_dict() == ours.to_dict()


def test_forked_child_is_refused_with_guidance():
    """the reference runs the extractor inside forked DataLoader workers (src/train.py:172-180);
    a HIP context does not survive fork(), so the drop-in must notice and say what to do instead."""
    from asr_amd import LogMelFeatureExtractor, _lib

    _lib.load()
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:  # child: report through the pipe, never return into pytest
        msg = "no error"
        try:
            try:
                _lib.load()
            except _lib.AsrkError as e:
                msg = "load: " + str(e)
            try:
                LogMelFeatureExtractor()(np.zeros(1600, dtype=np.float32), sampling_rate=16000)
                msg += " | call: no error"
            except _lib.AsrkError as e:
                msg += " | call: " + str(e)
            # the CPU-only host steps stay usable in a worker
            host, lengths, _ = LogMelFeatureExtractor().prepare_host([np.ones(10, dtype=np.float32)])
            msg += f" | host ok {tuple(host.shape)} {int(lengths[0])}"
        finally:
            os.write(w, msg.encode())
            os._exit(0)
    os.close(w)
    out = b""
    while True:
        chunk = os.read(r, 65536)
        if not chunk:
            break
        out += chunk
    os.waitpid(pid, 0)
    out = out.decode()
    assert out.count("forked child process") == 2, out
    assert "num_workers=0" in out and "FeaturePrefetcher" in out
    assert "host ok (1, 480000) 10" in out
    _lib.assert_not_forked()  # the parent itself is fine


def test_device_cpu_is_an_explicit_error():
    """HF's device="cpu" means CPU arithmetic; the drop-in refuses instead of silently computing on a GPU."""
    from asr_amd import LogMelFeatureExtractor, _lib

    with pytest.raises(_lib.AsrkError, match="GPU only"):
        LogMelFeatureExtractor()(np.zeros(1600, dtype=np.float32), sampling_rate=16000, device="cpu")
    with pytest.raises(_lib.AsrkError, match="GPU only"):
        LogMelFeatureExtractor(device="cpu")(np.zeros(1600, dtype=np.float32), sampling_rate=16000)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_product_fails_loudly_without_a_gpu():
    """No CPU fallback anywhere: without a GPU the product raises instead of computing."""
    from asr_amd import LogMelFeatureExtractor, _lib
    from asr_amd.encoder import WhisperEncoder
    from asr_amd.projector import AudioProjector

    with pytest.raises(_lib.AsrkError):
        LogMelFeatureExtractor()(np.zeros(16000, dtype=np.float32), sampling_rate=16000)
    enc = WhisperEncoder(128, 1, 2, 256, 80)
    with pytest.raises(ValueError, match="length 3000"):
        enc(torch.zeros(1, 80, 100))
    with pytest.raises(_lib.AsrkError):
        enc(torch.zeros(1, 80, 3000))
    with pytest.raises(_lib.AsrkError):
        AudioProjector(128, 192)(torch.zeros(1, 1500, 128))


def test_encoder_and_projector_parameter_contract():
    from asr_amd.encoder import WhisperEncoder, sinusoids
    from asr_amd.projector import AudioProjector

    enc = WhisperEncoder()  # defaults = whisper-small dims the reference hard-codes
    assert enc.d_model == 768 and not enc.whisper.training
    assert all(not p.requires_grad for p in enc.parameters())
    assert sum(p.numel() for p in enc.parameters()) == 88154112  # == HF WhisperEncoder at whisper-small dims
    sd = enc.state_dict()
    for k in ("whisper.encoder.conv1.weight", "whisper.encoder.conv2.bias", "whisper.encoder.embed_positions.weight",
              "whisper.encoder.layers.11.self_attn.q_proj.bias", "whisper.encoder.layers.0.self_attn.k_proj.weight",
              "whisper.encoder.layers.3.self_attn_layer_norm.weight", "whisper.encoder.layers.3.final_layer_norm.bias",
              "whisper.encoder.layers.5.fc1.weight", "whisper.encoder.layers.5.fc2.bias", "whisper.encoder.layer_norm.weight"):
        assert k in sd, k
    assert "whisper.encoder.layers.0.self_attn.k_proj.bias" not in sd
    assert tuple(sd["whisper.encoder.conv1.weight"].shape) == (768, 80, 3)
    # fixed sinusoidal table as HF builds it (modeling_whisper.py:55-64)
    from tests import signals
    ref = signals.encoder_state_dict(768, 0, 3072, 80)["embed_positions.weight"]
    assert np.abs(sinusoids(1500, 768).numpy() - ref).max() < 5e-4  # f32 exp/sin of arguments up to 1499
    torch.manual_seed(0)
    p = AudioProjector(768, 960)
    assert abs(p.linear_1.weight.std().item() - 0.01) < 2e-4 and abs(p.linear_2.weight.std().item() - 1.0) < 5e-3
    assert sum(x.numel() for x in p.parameters()) == 1661568 and all(x.requires_grad for x in p.parameters())
    assert set(p.state_dict()) == {"norm.weight", "linear_1.weight", "linear_1.bias", "linear_2.weight", "linear_2.bias"}


def test_fft400_host_emulation_matches_numpy(tmp_path):
    """The kernel's paired-frame 400-point FFT index algebra (stage 1, the 20 x 20 exchange image, stage 2, the mirror
    lane / register of the spectrum separation), run lane by lane over one wave on the host."""
    exe = tmp_path / "fft400_host"
    subprocess.check_call(["g++", "-O2", "-o", str(exe), os.path.join(ROOT, "tests", "host", "fft400_host.cpp")])
    rng = np.random.default_rng(0)
    s = (rng.standard_normal(1200) * 0.1).astype(np.float32)
    s[320:880] = 0.0  # frames 2 and 3 (one pair, samples 320..879) are silent and must give exact zeros
    s.tofile(tmp_path / "s.bin")
    subprocess.check_call([str(exe), str(tmp_path / "s.bin"), str(tmp_path / "p.bin")])
    p = np.fromfile(tmp_path / "p.bin", dtype=np.float32).reshape(6, -1)[:, :201]  # rows are padded to 16-byte chunks
    n = np.arange(400)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * n / 400)
    ref = np.stack([np.abs(np.fft.rfft(s[t * 160: t * 160 + 400].astype(np.float64) * w)) ** 2 for t in range(6)])
    assert np.abs(p - ref).max() <= 2e-6 * ref.max()
    assert np.all(p[2] == 0.0) and np.all(p[3] == 0.0)
