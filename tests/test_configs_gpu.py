"""Single-GPU side of BASELINE.json configs[3] and configs[4], at the sizes bench.py launches them.

configs[3] (`pipeline-large` / `train-large`): whisper-large-v3 dims (d 1280, 32 layers, 20 heads, ffn 5120, 128 mel),
16 clips per GPU -- the 32-layer, M = 24 000 launch itself, checked clip by clip against the CPU oracle, then the
training step as bench.py runs it: fused encoder + projector forward, the projector's custom backward on a synthetic
upstream gradient, gradients landing in `FlatGradAllReducer`'s flat buffer (world size 1 here; the N > 1 exchange is
covered on gloo by tests/test_distributed_gloo.py).  Reference: src/modeling.py:218-220 (encoder -> projector),
src/modeling.py:33-40 (no-grad encoder), :77-84 (projector).

configs[4]: a ragged batch of 64 clips of 1-30 s through the collator path (src/train.py:172-180: pad to 30 s + frame
mask + log-mel) -> encoder -> projector, whisper-small dims: masks, the per-clip independence of the front-end bit for
bit, and three clips against the whole oracle chain.

Weights are too large for fixtures (637 M / 88 M parameters): both sides build them from one torch CPU generator, and
the checker is the CPU oracle, itself pinned to HF / the reference by tests/golden (tests/test_oracle_golden.py).
Tolerance: rel 1e-2 on activations (BASELINE.json north_star), 2e-2 on gradients (as the other projector tests).
"""
import math

import numpy as np
import pytest
import torch

from oracle import encoder as oracle_encoder
from oracle import logmel as oracle_logmel
from oracle import projector as oracle_projector
from tests import signals

pytestmark = pytest.mark.gpu
REL = 1e-2


def rel_l2(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).flatten()
    b = torch.as_tensor(b, dtype=torch.float64).flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def torch_encoder_state_dict(d, layers, ffn, n_mels, seed):
    """HF-keyed encoder parameters from one CPU generator (seconds even at large-v3 size, where the integer-hash
    generator of tests/signals.py takes minutes).  Scales as signals.encoder_state_dict."""
    g = torch.Generator().manual_seed(seed)

    def rnd(*shape, std, mean=0.0):
        return torch.randn(shape, generator=g) * std + mean

    sd = {"conv1.weight": rnd(d, n_mels, 3, std=(3 * n_mels) ** -0.5), "conv1.bias": rnd(d, std=0.05),
          "conv2.weight": rnd(d, d, 3, std=(3 * d) ** -0.5), "conv2.bias": rnd(d, std=0.05)}
    half = d // 2
    inv = torch.exp(-(math.log(10000.0) / (half - 1)) * torch.arange(half, dtype=torch.float32))
    st = torch.arange(1500, dtype=torch.float32)[:, None] * inv[None, :]
    sd["embed_positions.weight"] = torch.cat([st.sin(), st.cos()], dim=1)
    for i in range(layers):
        p = f"layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nm}.weight"] = rnd(d, d, std=d ** -0.5)
            if nm != "k_proj":
                sd[p + f"self_attn.{nm}.bias"] = rnd(d, std=0.05)
        sd[p + "self_attn_layer_norm.weight"] = rnd(d, std=0.1, mean=1.0)
        sd[p + "self_attn_layer_norm.bias"] = rnd(d, std=0.05)
        sd[p + "fc1.weight"], sd[p + "fc1.bias"] = rnd(ffn, d, std=d ** -0.5), rnd(ffn, std=0.05)
        sd[p + "fc2.weight"], sd[p + "fc2.bias"] = rnd(d, ffn, std=ffn ** -0.5), rnd(d, std=0.05)
        sd[p + "final_layer_norm.weight"] = rnd(d, std=0.1, mean=1.0)
        sd[p + "final_layer_norm.bias"] = rnd(d, std=0.05)
    sd["layer_norm.weight"], sd["layer_norm.bias"] = rnd(d, std=0.1, mean=1.0), rnd(d, std=0.05)
    return sd


def build_projector(audio_dim, text_dim, seed):
    from asr_amd.projector import AudioProjector

    psd = {k: torch.from_numpy(v) for k, v in signals.projector_state_dict(audio_dim, text_dim, seed=seed).items()}
    proj = AudioProjector(audio_dim, text_dim)
    proj.load_state_dict(psd, strict=True)
    return proj.cuda(), psd


# ------------------------------------------------------------------------------------------------ configs[3]
@pytest.mark.timeout(600)
def test_large_v3_full_depth_batch16_forward_and_train_step(margins):
    from asr_amd.distributed import FlatGradAllReducer
    from asr_amd.encoder import WhisperEncoder
    from asr_amd.projector import encode_and_project

    d, layers, heads, ffn, n_mels, T, B = 1280, 32, 20, 5120, 128, 960, 16
    sd = torch_encoder_state_dict(d, layers, ffn, n_mels, seed=404)
    enc = WhisperEncoder.from_state_dict(sd, n_heads=heads)
    proj, psd = build_projector(d, T, seed=41)
    g = torch.Generator().manual_seed(405)
    feats = (torch.randn((B, n_mels, 3000), generator=g) * 0.45).clamp_(-1.0, 1.6)
    g_up = (torch.randn((B, 1500, T), generator=g) * 1e-2).to(torch.bfloat16)  # bench.py train-large: bf16 upstream gradient

    # the oracle on every clip alone (2.3 TFLOP each in fp32: a few seconds per clip on the box's host cores)
    hidden_ref = torch.cat([oracle_encoder.encoder_forward(sd, feats[b:b + 1], heads, n_layers=layers) for b in range(B)])

    # --- forward, as `pipeline-large` launches it: 32 layers, M = 24 000
    x = feats.cuda()
    y = enc(x)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (B, 1500, d) and bool(torch.isfinite(y).all())
    worst = 0.0
    for b in range(B):
        e = rel_l2(y[b].cpu(), hidden_ref[b])
        mx = float((y[b].cpu() - hidden_ref[b]).abs().max() / hidden_ref[b].abs().max())
        worst = max(worst, e)
        if b in (0, 11):
            print(f"large-v3 x 32 layers, B=16, clip {b}: rel-L2 {e:.3e}, max-abs/max {mx:.3e}")
            margins.add(f"encoder large-v3 dims, 32 layers, B=16 (pipeline-large launch), clip {b} vs oracle", e, REL, "rel-L2")
        assert e <= REL and mx <= 5e-2, f"clip {b}: rel-L2 {e:.3e}, max {mx:.3e}"
    margins.add("encoder large-v3 dims, 32 layers, B=16, worst of the 16 clips vs oracle", worst, REL, "rel-L2")
    # clip 11 alone takes other tile kernels (M = 1500): bf16 accuracy, and the launch is reproducible bit for bit
    y11 = enc(x[11:12])
    assert rel_l2(y[11].cpu(), y11[0].cpu()) <= 5e-3
    assert torch.equal(enc(x), y)

    # --- the train step of bench.py (`train-large`): grads land in the reducer's flat buffer
    reducer = FlatGradAllReducer(proj.parameters())
    views = [(p.grad.data_ptr(), p.grad.numel()) for p in reducer.params]
    for rep in range(2):  # twice: the second step must start from zeroed gradients, not accumulate
        reducer.zero_()
        emb = encode_and_project(enc, proj, x)
        assert emb.requires_grad and tuple(emb.shape) == (B, 1500, T)
        emb.backward(g_up.cuda())
        assert reducer.all_reduce() is None  # world size 1: nothing to exchange
    torch.cuda.synchronize()
    off = 0
    for p, (ptr, n) in zip(reducer.params, views):  # every .grad is still the same view of reducer.flat
        assert p.grad.data_ptr() == ptr == reducer.flat.data_ptr() + 4 * off and p.grad.numel() == n
        off += n
    y_ref, grads_ref = oracle_projector.projector_backward(psd, hidden_ref, g_up.float())
    fe = rel_l2(emb.detach().float().cpu(), y_ref)
    print(f"train-large step: embeddings rel-L2 {fe:.3e}")
    margins.add("train-large step (large-v3 x 32, B=16): projector output vs oracle chain", fe, REL, "rel-L2")
    assert fe <= REL
    flat = reducer.flat.cpu()
    off = 0
    for (name, p) in [(n_, p_) for n_, p_ in proj.named_parameters() if p_.requires_grad]:
        mine = flat[off: off + p.numel()].view_as(p)
        off += p.numel()
        e = rel_l2(mine, grads_ref[name])
        print(f"   grad {name} (from the flat buffer): rel-L2 {e:.3e}")
        margins.add(f"train-large step: grad {name} in the flat buffer vs oracle", e, 2e-2, "rel-L2")
        assert bool(torch.isfinite(mine).all()) and e <= 2e-2, f"grad {name}: {e:.3e}"


# ------------------------------------------------------------------------------------------------ configs[4]
@pytest.mark.timeout(300)
def test_ragged_batch64_collate_encoder_projector(margins):
    from asr_amd import LogMelFeatureExtractor
    from asr_amd.collate import device_collate
    from asr_amd.encoder import WhisperEncoder
    from asr_amd.projector import encode_and_project

    d, layers, heads, ffn, n_mels, T, B = 768, 12, 12, 3072, 80, 960, 64
    sd = torch_encoder_state_dict(d, layers, ffn, n_mels, seed=505)
    enc = WhisperEncoder.from_state_dict(sd, n_heads=heads)
    proj, psd = build_projector(d, T, seed=51)
    rng = np.random.default_rng(506)
    lens = rng.integers(16000, 480001, size=B)  # U[1 s, 30 s] in samples, not multiples of the hop
    lens[3], lens[17] = 480000, 16000           # both ends of the range are in the batch
    clips = [np.clip(0.1 * rng.standard_normal(int(n)), -1.0, 1.0).astype(np.float32) for n in lens]
    fe = LogMelFeatureExtractor(feature_size=n_mels)

    batch = device_collate(fe, clips)  # pad + mask + log-mel on the GPU (src/train.py:172-180)
    feats, mask = batch.input_features, batch.attention_mask
    assert tuple(feats.shape) == (B, n_mels, 3000) and feats.dtype == torch.float32 and feats.is_cuda
    assert mask.dtype == torch.int32 and tuple(mask.shape) == (B, 3000)
    assert np.array_equal(mask.sum(dim=1).cpu().numpy(), -(-lens // 160))  # ceil(len / 160) frames
    assert bool(((mask[:, 1:] - mask[:, :-1]) <= 0).all())  # ones, then zeros
    # the collator path equals the extractor's own host path, and a clip's features do not depend on its neighbours
    host = fe(clips, sampling_rate=16000, return_tensors="pt", return_attention_mask=True, padding="max_length",
              max_length=480000)
    assert torch.equal(host.input_features, feats) and torch.equal(host.attention_mask, mask)
    for b in (0, 17, 40):
        alone = fe(clips[b], sampling_rate=16000, return_tensors="pt", return_attention_mask=True)
        assert torch.equal(alone.input_features[0], feats[b]) and torch.equal(alone.attention_mask[0], mask[b])

    with torch.no_grad():
        emb = encode_and_project(enc, proj, feats)
        two = proj(enc(feats, attention_mask=mask))  # the reference's two calls, mask accepted and ignored
    torch.cuda.synchronize()
    assert tuple(emb.shape) == (B, 1500, T) and bool(torch.isfinite(emb).all()) and torch.equal(emb, two)
    for b in (3, 17, 40):  # 30 s, 1 s and a middling clip against the whole oracle chain
        o = oracle_logmel.extract([clips[b]], n_mels=n_mels, sampling_rate=16000)
        assert np.array_equal(o["attention_mask"][0], mask[b].cpu().numpy())
        ef = float(np.abs(o["input_features"][0] - feats[b].cpu().numpy()).max())
        h = oracle_encoder.encoder_forward(sd, o["input_features"], heads, n_layers=layers)
        ref = oracle_projector.projector_forward(psd, h)[0]
        e = rel_l2(emb[b].cpu(), ref)
        print(f"ragged batch, clip {b} ({lens[b] / 16000:.2f} s): log-mel max abs {ef:.2e}, embeddings rel-L2 {e:.3e}")
        margins.add(f"configs[4] ragged batch of 64, clip {b}: log-mel vs oracle", ef, 1e-4, "abs")
        margins.add(f"configs[4] ragged batch of 64, clip {b}: embeddings vs oracle chain", e, REL, "rel-L2")
        assert ef <= 1e-4 and e <= REL
    # clip-in-batch == clip-alone through the encoder + projector to bf16 accuracy (other tile kernels at M = 1500)
    with torch.no_grad():
        e1 = encode_and_project(enc, proj, feats[40:41])
    assert rel_l2(emb[40].cpu(), e1[0].cpu()) <= 5e-3


# ------------------------------------------------------------------------------------------------ other checkpoint sizes
# The reference hard-codes openai/whisper-small (src/modeling.py:24) and BASELINE.json adds large-v3; the encoder class
# takes its dimensions from the checkpoint, so the sizes in between must go through the same kernels' other tile
# choices (d = 384 is not a multiple of the 256-column tile, 6 / 8 / 16 heads).  Two layers each, three clips.
@pytest.mark.parametrize("name,d,heads,ffn,n_mels", [("whisper-tiny", 384, 6, 1536, 80), ("whisper-base", 512, 8, 2048, 80),
                                                      ("whisper-medium", 1024, 16, 4096, 80)])
def test_other_whisper_sizes_two_layers(name, d, heads, ffn, n_mels, margins):
    from asr_amd.encoder import WhisperEncoder
    from asr_amd.projector import encode_and_project

    layers, B, T = 2, 3, 576
    sd = torch_encoder_state_dict(d, layers, ffn, n_mels, seed=600 + heads)
    enc = WhisperEncoder.from_state_dict(sd, n_heads=heads)
    proj, psd = build_projector(d, T, seed=60 + heads)
    g = torch.Generator().manual_seed(700 + heads)
    feats = (torch.randn((B, n_mels, 3000), generator=g) * 0.45).clamp_(-1.0, 1.6)
    h_ref = oracle_encoder.encoder_forward(sd, feats, heads, n_layers=layers)
    y_ref = oracle_projector.projector_forward(psd, h_ref)
    with torch.no_grad():
        h = enc(feats.cuda())
        y = encode_and_project(enc, proj, feats.cuda())
    torch.cuda.synchronize()
    eh, ey = rel_l2(h.cpu(), h_ref), rel_l2(y.float().cpu(), y_ref)
    print(f"{name} dims x {layers} layers: hidden rel-L2 {eh:.3e}, embeddings rel-L2 {ey:.3e}")
    margins.add(f"encoder {name} dims, {layers} layers, B={B} vs oracle", eh, REL, "rel-L2")
    margins.add(f"encoder + projector {name} dims vs oracle chain", ey, REL, "rel-L2")
    assert eh <= REL and ey <= REL
    with torch.no_grad():  # a clip's result does not depend on its neighbours beyond bf16 tile-choice noise
        h1 = enc(feats[1:2].cuda())
    assert rel_l2(h[1].cpu(), h1[0].cpu()) <= 5e-3


def test_the_whole_path_is_capturable_in_a_hip_graph():
    """Callers that replay the step as a HIP graph (tight serving loops): every entry point only enqueues work on the
    caller's stream -- no synchronisation, no allocation of its own beyond what the caller's allocator hands it -- so
    waveform -> log-mel -> encoder -> projector can be captured once and replayed; the replay is bit-identical to
    stream launches, also after the input buffer has been overwritten in place."""
    from asr_amd import LogMelFeatureExtractor
    from asr_amd.encoder import WhisperEncoder
    from asr_amd.projector import encode_and_project

    d, layers, heads, ffn, n_mels, T, B = 384, 2, 6, 1536, 80, 576, 4
    enc = WhisperEncoder.from_state_dict(torch_encoder_state_dict(d, layers, ffn, n_mels, seed=808), n_heads=heads)
    proj, _ = build_projector(d, T, seed=81)
    fe = LogMelFeatureExtractor(feature_size=n_mels)
    g = torch.Generator().manual_seed(809)
    wave = (0.1 * torch.randn((B, 480000), generator=g)).clamp_(-1, 1).cuda()
    wave2 = (0.1 * torch.randn((B, 480000), generator=g)).clamp_(-1, 1).cuda()

    def step():
        with torch.no_grad():
            return encode_and_project(enc, proj, fe.extract_from_device(wave, None))

    eager = step().clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    wave.copy_(wave2)  # new audio in the captured input buffer
    graph.replay()
    torch.cuda.synchronize()
    eager2 = step()
    assert torch.equal(out, eager2) and not torch.equal(eager2, eager)
