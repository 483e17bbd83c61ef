"""On-device collate: asrk_collate_pcm / asr_amd.collate against the oracle's
pad_batch + rescale_mask (oracle/logmel.py, pinned to HF by the log-mel fixtures) and against the extractor's
own host path."""
import numpy as np
import pytest
import torch

from oracle import logmel as ol


def clips_for(seed, lens):
    rng = np.random.default_rng(seed)
    return [np.clip(0.2 * rng.standard_normal(n), -1, 1).astype(np.float32) for n in lens]


def test_pack_ragged_host():
    from asr_amd.collate import pack_ragged

    clips = clips_for(0, [5, 0, 17, 3])
    pcm, off = pack_ragged(clips, np.float32, pin=False)
    assert off.tolist() == [0, 5, 5, 22, 25] and pcm.dtype == torch.float32
    assert np.array_equal(pcm.numpy()[5:22], clips[2])
    pcm16, _ = pack_ragged([np.arange(4, dtype=np.int16)], np.int16, pin=False)
    assert pcm16.dtype == torch.int16 and pcm16.tolist() == [0, 1, 2, 3]
    with pytest.raises(ValueError):
        pack_ragged([np.zeros((2, 3), dtype=np.float32)], pin=False)
    with pytest.raises(ValueError):
        pack_ragged([np.zeros(3, dtype=np.float32)], np.int16, pin=False)


@pytest.mark.gpu
@pytest.mark.parametrize("target", [480000, 4000, 1605])
def test_collate_kernel_matches_oracle(target):
    from asr_amd.collate import collate_on_device, pack_ragged

    lens = [48077, 0, 1, 159, 160, 161, 3999, 4000, 4001, 600000 if target == 480000 else 5000]
    clips = clips_for(1, lens)
    pcm, off = pack_ragged(clips)
    wave, mask = collate_on_device(pcm.cuda(), off.cuda(), target, 160)
    ref_wave, ref_mask = ol.pad_batch(clips, max_length=target)
    assert np.array_equal(wave.cpu().numpy(), ref_wave)
    assert np.array_equal(mask.cpu().numpy(), ol.rescale_mask(ref_mask))
    # int16 PCM: x / 32768 exactly
    c16 = [(c * 32767).astype(np.int16) for c in clips]
    pcm16, off16 = pack_ragged(c16, np.int16)
    w16, m16 = collate_on_device(pcm16.cuda(), off16.cuda(), target, 160)
    ref16, _ = ol.pad_batch([c.astype(np.float32) / 32768.0 for c in c16], max_length=target)
    assert np.array_equal(w16.cpu().numpy(), ref16) and torch.equal(m16, mask)


@pytest.mark.gpu
def test_device_collate_equals_host_path_and_prefetcher():
    from asr_amd import LogMelFeatureExtractor
    from asr_amd.collate import FeaturePrefetcher, device_collate

    fe = LogMelFeatureExtractor(feature_size=80)
    batches = [clips_for(10 + i, [48077 + 1000 * i, 160000, 480000, 20000]) for i in range(4)]
    for clips in batches[:2]:
        a = fe(clips, sampling_rate=16000, return_tensors="pt", return_attention_mask=True, padding="max_length",
               max_length=480000)
        b = device_collate(fe, clips)
        assert torch.equal(a.input_features, b.input_features) and torch.equal(a.attention_mask, b.attention_mask)
    got = list(FeaturePrefetcher(fe, batches, depth=2))
    assert len(got) == len(batches)
    for clips, g in zip(batches, got):
        ref = device_collate(fe, clips)
        assert torch.equal(g.input_features, ref.input_features) and torch.equal(g.attention_mask, ref.attention_mask)
