"""Generate the golden fixtures under tests/golden/ (run ONCE in the build container).

    python tests/golden/make_golden.py

Sources of truth (none of them travels to the GPU box; only the .npz outputs do):
  * installed third-party ``transformers`` (5.15.0 here; the reference pins 4.56.1):
    WhisperFeatureExtractor(feature_size=80|128) built by local constructor, and
    WhisperEncoder(WhisperConfig(...)) with parameters overwritten from tests/signals.py;
  * the reference's own ``AudioProjector`` class, imported from /root/reference/src/modeling.py.
    That module's top-level ``from peft import LoraConfig, TaskType, get_peft_model`` needs the
    ``peft`` package, which is absent here; the three names are only used inside ``LLMDecoder``
    (out of scope), so an empty placeholder module is registered for the import to succeed.
    Nothing on the projector path touches it.

Inputs are regenerated deterministically by tests/signals.py, so fixtures hold expected OUTPUTS.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from tests import signals  # noqa: E402

torch.set_num_threads(8)
torch.backends.cuda.matmul.allow_tf32 = False


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB", {k: v.shape for k, v in arrs.items()})


# ------------------------------------------------------------------------------------ log-mel
def gen_logmel():
    import transformers
    from transformers import WhisperFeatureExtractor

    meta = f"transformers {transformers.__version__} torch {torch.__version__} numpy {np.__version__}"
    for n_mels in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=n_mels)
        save(f"melfilt_{n_mels}.npz", mel_filters=fe.mel_filters, meta=np.array(meta))
        out = {}
        for name, (build, _n) in signals.LOGMEL_CASES.items():
            clip = build()
            # eval-style call: src/train.py:54-59
            r = fe(clip, sampling_rate=16000, return_tensors="np", return_attention_mask=True)
            out[name + "__features"] = r["input_features"][0].astype(np.float32)
            out[name + "__mask"] = r["attention_mask"][0].astype(np.int32)
        # collator-style batched call: src/train.py:172-180
        names = ["short3005", "speechlike12", "tiny250", "noise30"]
        clips = [signals.LOGMEL_CASES[n][0]() for n in names]
        r = fe(clips, sampling_rate=16000, return_tensors="np", return_attention_mask=True,
               padding="max_length", max_length=480000)
        out["batch__mask"] = r["attention_mask"].astype(np.int32)
        # batch features equal the single-clip ones (per-clip max); store a checksum row only
        out["batch__features_rowsum"] = r["input_features"].astype(np.float64).sum(axis=(1, 2))
        for i, n in enumerate(names):
            assert np.array_equal(r["input_features"][i], out[n + "__features"]), n
        # padding="longest" (non-30 s length) - HF supports it, exercises generic n_samples
        r = fe([clips[0], clips[2]], sampling_rate=16000, return_tensors="np", return_attention_mask=True,
               padding="longest")
        out["longest__features"] = r["input_features"].astype(np.float32)
        out["longest__mask"] = r["attention_mask"].astype(np.int32)
        # the numpy/float64 path of the same class, as a precision cross-check (:105-133)
        padded = np.zeros((1, 480000), dtype=np.float64)
        padded[0, : len(clips[0])] = clips[0]
        r64 = fe._np_extract_fbank_features(padded, "cpu")
        out["short3005__features_np64"] = r64[0].astype(np.float32)
        save(f"logmel_{n_mels}.npz", meta=np.array(meta), **out)


# ------------------------------------------------------------------------------------ encoder
ENCODER_CASES = signals.ENCODER_CASES  # the cases and their inputs live in tests/signals.py
encoder_features = signals.encoder_features


def gen_encoder():
    import transformers
    from transformers import WhisperConfig
    from transformers.models.whisper.modeling_whisper import WhisperEncoder

    meta = f"transformers {transformers.__version__} torch {torch.__version__}"
    for name, c in ENCODER_CASES.items():
        cfg = WhisperConfig(d_model=c["d_model"], encoder_layers=c["layers"], encoder_attention_heads=c["heads"],
                            encoder_ffn_dim=c["ffn"], num_mel_bins=c["n_mels"], decoder_layers=1,
                            decoder_attention_heads=c["heads"], decoder_ffn_dim=64, vocab_size=128,
                            max_source_positions=1500)
        enc = WhisperEncoder(cfg).eval()
        sd = signals.encoder_state_dict(c["d_model"], c["layers"], c["ffn"], c["n_mels"], seed=c["seed"])
        missing = enc.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        print(name, "attn impl:", cfg._attn_implementation, missing)
        x = torch.from_numpy(encoder_features(name))
        with torch.no_grad():
            y = enc(x, attention_mask=None).last_hidden_state.float().numpy()
            # attention_mask is accepted and ignored (modeling_whisper.py:607-609, 636-640)
            y2 = enc(x, attention_mask=torch.zeros(x.shape[0], 3000, dtype=torch.int32)).last_hidden_state.float().numpy()
        assert np.array_equal(y, y2)
        rows = c["rows"]
        arrs = dict(meta=np.array(meta))
        if rows is None:
            arrs["out"] = y
        else:
            arrs["rows"] = np.array(rows, dtype=np.int32)
            arrs["out_rows"] = y[:, rows, :]
            arrs["out_abs_mean"] = np.array(np.abs(y).mean())
        save(f"encoder_{name}.npz", **arrs)


# ------------------------------------------------------------------------------------ projector
PROJECTOR_CASES = signals.PROJECTOR_CASES
projector_inputs = signals.projector_inputs


def gen_projector():
    placeholder = types.ModuleType("peft")
    placeholder.LoraConfig = placeholder.TaskType = placeholder.get_peft_model = None
    sys.modules.setdefault("peft", placeholder)
    sys.path.insert(0, "/root/reference/src")
    import modeling as ref_modeling  # the reference's own module

    for name, c in PROJECTOR_CASES.items():
        proj = ref_modeling.AudioProjector(c["audio_dim"], c["text_dim"])
        sd = signals.projector_state_dict(c["audio_dim"], c["text_dim"], seed=c["seed"])
        proj.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        x, gy = projector_inputs(name)
        xt = torch.from_numpy(x)
        y = proj(xt)
        y.backward(torch.from_numpy(gy))
        g = {k.replace(".", "__"): p.grad.numpy() for k, p in proj.named_parameters()}
        arrs = {}
        if name == "p128":
            arrs["out"] = y.detach().numpy()
            for k, v in g.items():
                arrs["grad__" + k] = v
        else:
            arrs["out_rows"] = y.detach().numpy()[:, ::20, :]
            for k, v in g.items():
                arrs["grad__" + k] = v if v.ndim == 1 else v[::8, ::8]
        save(f"projector_{name}.npz", **arrs)
        # reference init statistics (src/modeling.py:72-75), for the init parity test
    torch.manual_seed(0)
    proj = ref_modeling.AudioProjector(768, 960)
    print("init std W1 %.4f W2 %.4f b1 %.1f b2 %.1f gamma %.1f" % (
        proj.linear_1.weight.std(), proj.linear_2.weight.std(), proj.linear_1.bias.abs().max(),
        proj.linear_2.bias.abs().max(), proj.norm.weight.mean()))


# ------------------------------------------------------------------------------------ preprocessor_config.json
def gen_preprocessor():
    """the file stock transformers writes with save_pretrained (what
    src/modeling.py:157-161 puts next to the model), kept as a fixture so that the CPU suite can check
    that our extractor reads a stock-written file without importing transformers."""
    import json
    import tempfile

    import transformers
    from transformers import WhisperFeatureExtractor

    for n_mels in (80, 128):
        with tempfile.TemporaryDirectory() as d:
            WhisperFeatureExtractor(feature_size=n_mels, return_attention_mask=True).save_pretrained(d)
            cfg = json.load(open(os.path.join(d, "preprocessor_config.json")))
        cfg["_written_by"] = f"transformers {transformers.__version__} WhisperFeatureExtractor.save_pretrained"
        with open(os.path.join(HERE, f"hf_preprocessor_config_{n_mels}.json"), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        print(f"hf_preprocessor_config_{n_mels}.json:", sorted(cfg))


if __name__ == "__main__":
    which = sys.argv[1:] or ["logmel", "encoder", "projector", "preprocessor"]
    if "preprocessor" in which:
        gen_preprocessor()
    if "logmel" in which:
        gen_logmel()
    if "encoder" in which:
        gen_encoder()
    if "projector" in which:
        gen_projector()
