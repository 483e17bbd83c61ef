"""Audio-embedding splice (src/modeling.py:222-276).

CPU: the NumPy oracle against a literal torch.cat / torch.full evaluation of the reference lines.
GPU: the HIP kernels (through asr_amd.splice, i.e. the C ABI) against the oracle -- bit-exact, it is a copy --
and their autograd against torch.cat's."""
import numpy as np
import pytest
import torch

from oracle import splice as osp

CHUNK = 49153  # stands in for tokenizer.convert_tokens_to_ids("<|audio_chunk|>")


def make_case(B, L, A, D, seed, dtype=np.float32):
    rng = np.random.default_rng(seed)
    text = rng.standard_normal((B, L, D)).astype(dtype)
    audio = rng.standard_normal((B, A, D)).astype(dtype)
    ids = rng.integers(0, 49000, size=(B, L), dtype=np.int64)
    labels = rng.integers(-100, 49000, size=(B, L), dtype=np.int64)
    pos = rng.integers(0, L, size=B)
    pos[0] = 0          # placeholder first ...
    if B > 1:
        pos[1] = L - 1  # ... and last
    for b in range(B):
        ids[b, pos[b]] = CHUNK
        if b % 2 == 0 and pos[b] + 2 < L:
            ids[b, pos[b] + 2] = CHUNK  # a second occurrence must be ignored (nonzero()[0], :227)
    return text, audio, ids, labels, pos


def torch_literal(text, audio, ids, labels):
    """The reference lines evaluated with the same torch calls (checker for the oracle)."""
    embs, labs = [], []
    for i in range(ids.shape[0]):
        c = (ids[i] == CHUNK).nonzero()[0].item()
        embs.append(torch.cat([text[i, :c], audio[i], text[i, c + 1:]], dim=0))
        labs.append(torch.cat([labels[i, :c], torch.full((audio.shape[1],), -100, dtype=labels.dtype), labels[i, c + 1:]], dim=0))
    return torch.stack(embs), torch.stack(labs)


@pytest.mark.parametrize("B,L,A,D", [(3, 17, 5, 8), (2, 40, 1500, 16), (1, 1, 3, 4)])
def test_oracle_matches_torch_literal(B, L, A, D):
    text, audio, ids, labels, pos = make_case(B, L, A, D, seed=B * 100 + L)
    e, lab, mask = osp.splice(text, audio, ids, labels, CHUNK)
    te, tl = torch_literal(torch.from_numpy(text), torch.from_numpy(audio), torch.from_numpy(ids), torch.from_numpy(labels))
    assert np.array_equal(e, te.numpy()) and np.array_equal(lab, tl.numpy())
    assert mask.shape == lab.shape and (mask == 1).all() and e.shape == (B, L - 1 + A, D)
    assert np.array_equal(osp.first_token(ids, CHUNK), pos.astype(np.int32))
    # adjoint: <splice(x), g> == <x, splice^T(g)>
    g = np.random.default_rng(5).standard_normal(e.shape).astype(np.float32)
    gt, ga = osp.splice_backward(g, pos, L, A)
    lhs = float((e.astype(np.float64) * g).sum())
    rhs = float((text.astype(np.float64) * gt).sum() + (audio.astype(np.float64) * ga).sum())
    assert abs(lhs - rhs) <= 1e-6 * max(1.0, abs(lhs))
    assert (gt[np.arange(B), pos] == 0).all()


def test_oracle_raises_without_placeholder():
    text, audio, ids, labels, _ = make_case(2, 9, 4, 8, seed=3)
    ids[1][ids[1] == CHUNK] = 7
    with pytest.raises(ValueError, match="audio_chunk"):
        osp.splice(text, audio, ids, labels, CHUNK)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,L,A,D", [(3, 17, 5, 8), (4, 96, 1500, 960), (1, 1, 3, 8), (2, 33, 7, 576)])
def test_splice_gpu_bit_exact(B, L, A, D, dtype):
    from asr_amd.splice import find_first_token, splice_audio_embeddings

    text, audio, ids, labels, pos = make_case(B, L, A, D, seed=B + L + A)
    tt = torch.from_numpy(text).to(dtype).cuda()
    ta = torch.from_numpy(audio).to(dtype).cuda()
    ti, tl = torch.from_numpy(ids).cuda(), torch.from_numpy(labels).cuda()
    assert np.array_equal(find_first_token(ti, CHUNK).cpu().numpy(), pos.astype(np.int32))
    e, lab, mask = splice_audio_embeddings(tt, ta, ti, tl, CHUNK)
    # the oracle on the same (already rounded) values: a copy must be bit-exact
    re, rl, rm = osp.splice(tt.float().cpu().numpy(), ta.float().cpu().numpy(), ids, labels, CHUNK)
    assert e.dtype == dtype and tuple(e.shape) == (B, L - 1 + A, D)
    assert np.array_equal(e.float().cpu().numpy(), re)
    assert lab.dtype == torch.int64 and np.array_equal(lab.cpu().numpy(), rl)
    assert mask.dtype == torch.int64 and np.array_equal(mask.cpu().numpy(), rm)
    e2, lab2, _ = splice_audio_embeddings(tt, ta, ti, None, CHUNK)
    assert lab2 is None and torch.equal(e2, e)


@pytest.mark.gpu
def test_splice_gpu_autograd_and_errors():
    from asr_amd.splice import splice_audio_embeddings

    B, L, A, D = 3, 21, 9, 64
    text, audio, ids, labels, pos = make_case(B, L, A, D, seed=11)
    tt = torch.from_numpy(text).cuda().requires_grad_(True)
    ta = torch.from_numpy(audio).cuda().requires_grad_(True)
    ti, tl = torch.from_numpy(ids).cuda(), torch.from_numpy(labels).cuda()
    e, _, _ = splice_audio_embeddings(tt, ta, ti, tl, CHUNK)
    g = torch.randn_like(e)
    e.backward(g)
    gt, ga = osp.splice_backward(g.cpu().numpy(), pos, L, A)
    assert np.array_equal(tt.grad.cpu().numpy(), gt) and np.array_equal(ta.grad.cpu().numpy(), ga)
    # only the audio side needs a gradient in training (frozen decoder embeddings)
    ta2 = torch.from_numpy(audio).cuda().requires_grad_(True)
    e3, _, _ = splice_audio_embeddings(torch.from_numpy(text).cuda(), ta2, ti, tl, CHUNK)
    e3.backward(g)
    assert np.array_equal(ta2.grad.cpu().numpy(), ga)
    bad = ti.clone()
    bad[1][bad[1] == CHUNK] = 7
    with pytest.raises(ValueError, match="audio_chunk"):
        splice_audio_embeddings(tt, ta, bad, tl, CHUNK)
    with pytest.raises(ValueError):
        splice_audio_embeddings(tt, ta.to(torch.bfloat16), ti, tl, CHUNK)
