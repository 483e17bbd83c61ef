"""asrk_logmel_f32_noalloc -- the one entry point that allocates nothing (include/asrk.h) -- against asrk_logmel_f32,
through the C ABI.  Two calls in a row on the same GPU, the second with a SMALLER input (fewer, shorter clips): the
static scratch keeps the first call's per-clip maxima, so the second call only passes if the entry point clears what
it uses.  Bit-exact: both entry points run the same kernels.  Collected last (zz) on purpose: the round's end runs
pytest -x, and a failure here must not hide the other suites."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _both(lib, fe, wave):
    B, L = wave.shape
    plan = fe._plan(wave.device)
    ref = torch.full((B, fe.feature_size, L // 160), float("nan"), device="cuda")
    ws = torch.empty((max(int(lib.asrk_logmel_workspace_bytes(B, L)), 256),), dtype=torch.uint8, device="cuda")
    assert lib.asrk_logmel_f32(plan, _p(wave), _p(None), B, L, _p(ref), _p(ws), ws.numel(), _stream()) == 0
    out = torch.full_like(ref, float("nan"))
    rc = lib.asrk_logmel_f32_noalloc(plan, _p(wave), B, L, _p(out), _stream())
    assert rc == 0, lib.asrk_strerror(rc)
    torch.cuda.synchronize()
    return out, ref


@pytest.mark.parametrize("n_mels", [80, 128])
def test_noalloc_entry_point_matches_the_workspace_one(n_mels):
    from asr_amd import LogMelFeatureExtractor, _lib

    lib = _lib.load()
    fe = LogMelFeatureExtractor(feature_size=n_mels)
    g = torch.Generator(device="cuda").manual_seed(17 + n_mels)
    # first call: 9 loud 30 s clips (large per-clip maxima are left behind in the scratch)
    w1 = (torch.randn((9, 480000), device="cuda", generator=g) * 0.5).clamp_(-1, 1)
    out1, ref1 = _both(lib, fe, w1)
    assert torch.isfinite(out1).all() and torch.equal(out1, ref1)
    # second call, smaller and much quieter: 4 clips of 3.005 s (not a multiple of the hop), one of them silent.  A stale
    # maximum from the first call would lift the max-8 floor and change these outputs.
    w2 = (torch.randn((4, 48077), device="cuda", generator=g) * 1e-3).clamp_(-1, 1).contiguous()
    w2[2] = 0.0
    out2, ref2 = _both(lib, fe, w2)
    assert torch.isfinite(out2).all() and torch.equal(out2, ref2)
    assert torch.all(out2[2] == -1.5)  # silence: log10(1e-10) = -10 everywhere -> (-10 + 4) / 4
    # and against the extractor's normal path (which allocates its workspace through torch)
    assert torch.equal(out2, fe.extract_from_device(w2))


def test_noalloc_entry_point_limits():
    from asr_amd import LogMelFeatureExtractor, _lib

    lib = _lib.load()
    fe = LogMelFeatureExtractor(feature_size=80)
    plan = fe._plan(torch.device("cuda", torch.cuda.current_device()))
    w = torch.zeros((1, 1600), device="cuda")
    out = torch.zeros((1, 80, 10), device="cuda")
    # more clips than the static scratch holds: refused before anything is launched (pointers are never dereferenced)
    assert lib.asrk_logmel_f32_noalloc(plan, _p(w), 4097, 1600, _p(out), _stream()) == -5  # ASRK_ERR_WORKSPACE
    assert lib.asrk_logmel_f32_noalloc(plan, _p(None), 1, 1600, _p(out), _stream()) == -1
    assert lib.asrk_logmel_f32_noalloc(plan, _p(w), 0, 1600, _p(out), _stream()) == 0
    assert lib.asrk_logmel_f32_noalloc(plan, _p(w), 1, 1600, _p(out), _stream()) == 0
    torch.cuda.synchronize()
    assert torch.all(out == -1.5)
