#!/usr/bin/env python
"""Greedy-decode WER on locally supplied checkpoints.

    python tools/eval_wer.py --whisper-dir DIR --decoder-dir DIR --manifest FILE.jsonl [--projector FILE.safetensors]

Wires `asr_amd.evaluation.evaluate_samples` (the reference's src/train.py:38-82 contract) to an `ASRModel`-shaped object
whose audio path is this package's (log-mel -> encoder -> projector -> prompt splice, src/modeling.py:299-327) and
whose decoder is a stock HF causal LM run by transformers' `generate` (src/modeling.py:329-334).  Nothing is fetched:
every directory must be local, weights are read with safetensors only (no pickle), and when any of them is missing the
tool prints "skipped: no weights" and exits 0 -- there are no Whisper / SmolLM2 weights in the build or test
environment, so the north-star's "WER within 0.1 absolute" claim is UNPINNED until someone runs this where they exist.

  --whisper-dir   HF Whisper checkpoint (config.json + model.safetensors with `model.encoder.*` keys, e.g. a local copy
                  of openai/whisper-small, the name the reference hard-codes at src/modeling.py:23-25)
  --decoder-dir   HF causal LM + tokenizer (the reference default is SmolLM2-360M-Instruct, configs/hydra/model/*.yaml);
                  a LoRA-trained decoder must be merged into plain weights first (peft is not a dependency here)
  --projector     safetensors holding `audio_projector.{norm.weight,linear_1.*,linear_2.*}` (a checkpoint written by the
                  reference's save_pretrained); without it the projector keeps its reference init and the WER only
                  exercises the plumbing
  --manifest      JSON lines: {"audio": "clip.npy" | "clip.wav", "text": "reference transcript"}; 16 kHz mono,
                  .npy float32 or 16-bit PCM .wav (paths relative to the manifest)
  --reference-json  optional: {"wer": x} produced by the reference on the same manifest; the tool then reports the
                  absolute difference against the 0.1 bound of BASELINE.json's north_star
"""
import argparse
import json
import os
import sys
import wave as wave_mod

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

INSTRUCTION_TEMPLATE = "User: Please transcribe the following audio recording.\n<|audio_chunk|>\nAssistant: "  # src/modeling.py:131-133
AUDIO_TOKENS = ["<|audio_start|>", "<|audio_end|>", "<|audio_pad|>", "<|audio_sep|>", "<|audio_chunk|>"]  # :165-171


def skipped(why):
    print(f"skipped: no weights ({why})")
    return 0


def read_clip(path):
    if path.endswith(".npy"):
        a = np.load(path, allow_pickle=False)
        return (a.astype(np.float32) / 32768.0) if a.dtype == np.int16 else a.astype(np.float32)
    with wave_mod.open(path, "rb") as w:
        if w.getframerate() != 16000 or w.getnchannels() != 1 or w.getsampwidth() != 2:
            raise ValueError(f"{path}: expected 16 kHz mono 16-bit PCM")
        return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / 32768.0


def load_manifest(path):
    base = os.path.dirname(os.path.abspath(path))
    samples = []
    with open(path) as fh:
        for line in fh:
            if not line.strip():
                continue
            rec = json.loads(line)
            p = rec["audio"] if os.path.isabs(rec["audio"]) else os.path.join(base, rec["audio"])
            samples.append({"audio": {"array": read_clip(p), "sampling_rate": 16000}, "text": rec["text"]})
    return samples


class LocalASRModel:
    """The three attributes and the `generate` signature `evaluate_samples` needs (src/modeling.py:290-334)."""

    def __init__(self, encoder, projector, decoder, tokenizer, audio_chunk_id):
        self.encoder, self.audio_projector, self.decoder, self.tokenizer = encoder, projector, decoder, tokenizer
        self.audio_chunk_id = audio_chunk_id
        self.device = next(decoder.parameters()).device

    def generate(self, input_features, attention_mask=None, max_new_tokens=100, **kw):
        import torch

        from asr_amd.evaluation import build_generate_embeds
        from asr_amd.projector import encode_and_project

        with torch.no_grad():
            audio_embeds = encode_and_project(self.encoder, self.audio_projector, input_features).to(self.decoder.dtype)
            prompt_ids = self.tokenizer(INSTRUCTION_TEMPLATE, return_tensors="pt").input_ids.to(input_features.device)
            prompt_embeds = self.decoder.get_input_embeddings()(prompt_ids)
            initial = build_generate_embeds(prompt_embeds, prompt_ids, audio_embeds, self.audio_chunk_id)
            return self.decoder.generate(inputs_embeds=initial, max_new_tokens=max_new_tokens, **kw)


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--whisper-dir")
    ap.add_argument("--decoder-dir")
    ap.add_argument("--projector")
    ap.add_argument("--manifest")
    ap.add_argument("--reference-json")
    ap.add_argument("--limit", type=int, default=0)
    args = ap.parse_args(argv)

    for what, p, inner in (("--whisper-dir", args.whisper_dir, "model.safetensors"),
                           ("--decoder-dir", args.decoder_dir, "config.json"), ("--manifest", args.manifest, None)):
        if not p:
            return skipped(f"{what} not given")
        full = os.path.join(p, inner) if inner else p
        if not os.path.exists(full):
            return skipped(f"{full} not found")

    import torch
    from safetensors.torch import load_file

    if not torch.cuda.is_available():
        print("error: eval_wer.py needs a ROCm GPU (asr_amd has no CPU fallback)", file=sys.stderr)
        return 2
    from asr_amd import LogMelFeatureExtractor
    from asr_amd.encoder import WhisperEncoder
    from asr_amd.evaluation import evaluate_samples
    from asr_amd.projector import AudioProjector
    from transformers import AutoModelForCausalLM, AutoTokenizer

    wcfg = json.load(open(os.path.join(args.whisper_dir, "config.json")))
    sd = load_file(os.path.join(args.whisper_dir, "model.safetensors"))
    pref = "model.encoder." if any(k.startswith("model.encoder.") for k in sd) else "encoder."
    enc_sd = {k[len(pref):]: v.float() for k, v in sd.items() if k.startswith(pref)}
    if "embed_positions.weight" not in enc_sd:
        return skipped("the Whisper checkpoint holds no encoder weights")
    encoder = WhisperEncoder.from_state_dict(enc_sd, n_heads=wcfg["encoder_attention_heads"])
    fe = LogMelFeatureExtractor(feature_size=wcfg.get("num_mel_bins", 80))

    tokenizer = AutoTokenizer.from_pretrained(args.decoder_dir, local_files_only=True)
    decoder = AutoModelForCausalLM.from_pretrained(args.decoder_dir, local_files_only=True, use_safetensors=True,
                                                   dtype=torch.bfloat16).cuda().eval()
    if tokenizer.pad_token is None:
        tokenizer.pad_token = tokenizer.eos_token  # src/modeling.py:105-107
    if tokenizer.add_special_tokens({"additional_special_tokens": AUDIO_TOKENS}) > 0:  # :173-190
        decoder.resize_token_embeddings(len(tokenizer))
    chunk_id = tokenizer.convert_tokens_to_ids("<|audio_chunk|>")
    projector = AudioProjector(encoder.d_model, decoder.config.hidden_size)
    if args.projector and os.path.exists(args.projector):
        psd = {k.split("audio_projector.", 1)[1]: v.float() for k, v in load_file(args.projector).items()
               if "audio_projector." in k}
        projector.load_state_dict(psd, strict=True)
    else:
        print("note: no --projector given: the projector keeps its init; this run checks plumbing, not accuracy", file=sys.stderr)
    projector = projector.cuda()

    samples = load_manifest(args.manifest)
    if args.limit:
        samples = samples[: args.limit]
    model = LocalASRModel(encoder, projector, decoder, tokenizer, chunk_id)
    preds, refs, wer = evaluate_samples(model, tokenizer, fe, samples, device=model.device)
    res = {"samples": len(samples), "wer": wer, "predictions": preds[:5], "references": refs[:5]}
    if args.reference_json and os.path.exists(args.reference_json):
        ref_wer = float(json.load(open(args.reference_json))["wer"])
        res.update(reference_wer=ref_wer, abs_diff=abs(wer - ref_wer), within_0p1=abs(wer - ref_wer) <= 0.1)
    print(json.dumps(res))
    return 0


if __name__ == "__main__":
    sys.exit(main())
