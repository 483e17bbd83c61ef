"""GPU parity AT THE SHAPES THE BENCHMARK RUNS (bench.py pipeline-small: 64 x 30 s, whisper-small dims; pipeline-large /
train-large: 16 x 30 s, large-v3 dims).  The other GPU suites check the kernels at sizes the CPU oracle can follow; here
the production launches themselves -- M = 96 000 / 24 000 rows, K up to 5120 = 80 K-slices per tile, several rounds of
tiles per CU on the persistent kernel -- are checked, through the C ABI:

  * every production GEMM shape on integer operands with non-zeros in EVERY K slice: all partial sums are small integers,
    exact in fp32 in any order and exact in bf16, so the comparison is bit-exact (HF modeling_whisper.py:279-282,
    375-376; src/modeling.py:68-70);
  * the fc1 shape with the fused GELU: exact integer pre-activations, so the output must equal the float64 erf GELU
    to one bf16 rounding;
  * the whole whisper-small encoder at B = 64 against the HF golden rows (clip 0 = the fixture's input, clips 1..63
    distinct), and clip 0 inside the batch (256-tile kernels) against clip 0 alone (128-tile kernels);
  * the projector forward / backward at large-v3 width 1280 -> 960 (BASELINE.json configs[3]).
"""
import ctypes

import numpy as np
import pytest
import torch

from tests import signals

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from asr_amd import _lib

    return _lib.load()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _int_operands(M, N, K, seed):
    """x in {-1, 0, 1} everywhere; w has exactly 2 non-zeros (+-1, +-2) in EVERY 64-wide K slice of every row, at
    positions that differ from row to row and slice to slice.  |sum| <= 3 * K / 64 + 3 <= 243 for K <= 5120."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.randint(-1, 2, (M, K), device="cuda", generator=g).to(torch.bfloat16)
    nsl = K // 64
    w = torch.zeros((N, nsl, 64), device="cuda")
    pos = torch.randint(0, 64, (N, nsl, 2), device="cuda", generator=g)
    val = torch.randint(1, 3, (N, nsl, 2), device="cuda", generator=g).float() * (torch.randint(0, 2, (N, nsl, 2), device="cuda", generator=g).float() * 2 - 1)
    val[..., 1] = torch.sign(val[..., 1])  # second entry +-1: |row slice| <= 3 even when both land on one position
    w.scatter_(2, pos[..., :1], val[..., :1])
    w.scatter_add_(2, pos[..., 1:], val[..., 1:])
    assert bool((w.abs().sum(dim=2) > 0).any(dim=0).all()), "every K slice must carry non-zeros"
    w = w.reshape(N, K).to(torch.bfloat16)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    return x, w, bias


def _ref_chunks(x, w, bias, rows=16000):
    """fp32 reference in row chunks (exact: integer sums far below 2^24)."""
    for r0 in range(0, x.shape[0], rows):
        yield r0, x[r0:r0 + rows].float() @ w.float().T + bias


# (M, N, K): whisper-small at 64 clips -- QKV, fc1 (plain here, GELU below), fc2, out-proj; large-v3 at 16 clips -- QKV, fc1,
# fc2, out-proj.  All of them dispatch to gemm256p_kernel (bf16 output).
PROD_SHAPES = [(96000, 2304, 768), (96000, 3072, 768), (96000, 768, 3072), (96000, 768, 768),
               (24000, 3840, 1280), (24000, 5120, 1280), (24000, 1280, 5120), (24000, 1280, 1280)]


@pytest.mark.parametrize("M,N,K", PROD_SHAPES)
def test_production_gemm_shapes_bit_exact(lib, M, N, K):
    x, w, bias = _int_operands(M, N, K, seed=M + 3 * N + 7 * K)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    rc = lib.asrk_gemm_bf16(_p(x), K, _p(w), _p(bias), _p(None), 0, 0, _p(out), N, 1, M, N, K, 0, 1.0, _stream())
    assert rc == 0, lib.asrk_strerror(rc)
    torch.cuda.synchronize()
    mx = 0.0
    for r0, ref in _ref_chunks(x, w, bias):
        mx = max(mx, ref.abs().max().item())
        got = out[r0:r0 + ref.shape[0]].float()
        if not torch.equal(got, ref):
            bad = (got != ref).nonzero()
            pytest.fail(f"M={M} N={N} K={K}: {bad.shape[0]} wrong elements in rows {r0}.., first at {bad[0].tolist()}")
    assert 8 <= mx <= 256, mx  # the sums are non-trivial and exactly representable in bf16


@pytest.mark.parametrize("M,N,K", [(96000, 3072, 768), (24000, 5120, 1280)])
def test_production_fc1_gelu_shape(lib, M, N, K):
    """fc1 with the fused GELU epilogue (gemm256p_kernel<8, 1, true>) at its production shapes: the pre-activations
    are exact integers, so out must be gelu(float64) rounded once to bf16, plus the 2.6e-5 absolute bound of the epilogue's
    logistic-form evaluation (asrk_common.h: gelu2_x2)."""
    x, w, bias = _int_operands(M, N, K, seed=11 + M + N)
    bias = bias * 0.25  # quarter-integer pre-activations: exercises the erf between the integers too (still exact in fp32)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    rc = lib.asrk_gemm_bf16(_p(x), K, _p(w), _p(bias), _p(None), 0, 0, _p(out), N, 1, M, N, K, 1, 1.0, _stream())
    assert rc == 0, lib.asrk_strerror(rc)
    torch.cuda.synchronize()
    worst = 0.0
    for r0, pre in _ref_chunks(x, w, bias, rows=8000):
        pre = pre.double()
        ref = 0.5 * pre * (1.0 + torch.erf(pre / np.sqrt(2.0)))
        err = (out[r0:r0 + pre.shape[0]].double() - ref).abs()
        bound = ref.abs() * 2.0 ** -8 + 3e-5
        assert bool((err <= bound).all()), f"rows {r0}..: max excess {(err - bound).max().item():.3e}"
        worst = max(worst, err.max().item())
    print(f"fc1+GELU M={M} N={N} K={K}: max abs err {worst:.3e}")


def test_whisper_small_encoder_at_bench_batch_matches_golden(golden_dir, margins):
    """bench.py's pipeline-small encoder launch: whisper-small dims, B = 64 (M = 96 000).  Clip 0 is the input of
    tests/golden/encoder_small.npz (HF WhisperEncoder fp32, modeling_whisper.py:592-646); clips 1..63 are distinct."""
    from asr_amd.encoder import WhisperEncoder

    c = signals.ENCODER_CASES["small"]
    sd = signals.encoder_state_dict(c["d_model"], c["layers"], c["ffn"], c["n_mels"], seed=c["seed"])
    enc = WhisperEncoder.from_state_dict(sd, n_heads=c["heads"])
    g = np.load(f"{golden_dir}/encoder_small.npz")
    x0 = torch.from_numpy(signals.encoder_features("small")).cuda()  # [1, 80, 3000]
    B = 64
    gen = torch.Generator(device="cuda").manual_seed(2024)
    x = (torch.randn((B, c["n_mels"], 3000), device="cuda", generator=gen) * 0.45).clamp_(-1.0, 1.6)
    x[0] = x0[0]
    x[37] = x0[0]  # the same clip again, elsewhere in the batch
    y = enc(x)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (B, 1500, c["d_model"]) and bool(torch.isfinite(y).all())
    rows = g["rows"]
    ref = g["out_rows"][0]
    for b in (0, 37):
        mine = y[b].cpu().numpy()[rows, :]
        err = rel_l2(mine, ref)
        mx = np.abs(mine - ref).max() / np.abs(ref).max()
        print(f"B=64 whisper-small, clip {b}: rel-L2 {err:.3e}, max-abs/max {mx:.3e}")
        margins.add(f"encoder whisper-small at B=64 (bench shape), clip {b} vs HF fp32 fixture", err, 1e-2, "rel-L2")
        assert err <= 1e-2 and mx <= 5e-2
    assert torch.equal(y[0], y[37])  # clips are independent, bit for bit
    assert not torch.equal(y[0], y[1])
    # the same clip alone takes the 128-tile kernels: agreement to bf16 accuracy
    y1 = enc(x0)
    e = rel_l2(y[0].cpu().numpy(), y1[0].cpu().numpy())
    print(f"clip 0 in the 64-batch vs alone: rel-L2 {e:.3e}")
    margins.add("encoder whisper-small, clip 0 in the 64-batch vs alone", e, 5e-3, "rel-L2")
    assert e <= 5e-3
    # every clip's output is a LayerNorm output: mean ~ beta, finite spread (a wrong tile anywhere shows as an outlier)
    stats = y.reshape(B, -1).std(dim=1)
    assert bool((stats > 0.1).all()) and bool((stats < 10).all()), stats


def test_projector_large_v3_width_forward_backward_vs_torch_fp32(margins):
    """AudioProjector(1280, 960) on 16 x 1500 rows (bench.py train-large; BASELINE.json configs[3]): forward and the
    five parameter gradients against the module's arithmetic (src/modeling.py:77-84) in fp32 torch autograd."""
    from asr_amd.projector import AudioProjector

    A, T, B = 1280, 960, 16
    proj = AudioProjector(A, T)
    sd = signals.projector_state_dict(A, T, seed=31)
    proj.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    proj = proj.cuda()
    g = torch.Generator(device="cuda").manual_seed(6)
    x = torch.randn((B, 1500, A), device="cuda", generator=g)
    gy = torch.randn((B, 1500, T), device="cuda", generator=g) * 1e-2
    y = proj(x)
    y.backward(gy)
    mine = {n: p.grad.clone() for n, p in proj.named_parameters()}
    ref_p = {n: p.detach().clone().requires_grad_(True) for n, p in proj.named_parameters()}
    xn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * ref_p["norm.weight"]
    h = torch.nn.functional.gelu(xn @ ref_p["linear_1.weight"].T + ref_p["linear_1.bias"])
    y_ref = (h @ ref_p["linear_2.weight"].T + ref_p["linear_2.bias"]) * 0.01
    y_ref.backward(gy)
    fe = rel_l2(y.detach().cpu().numpy(), y_ref.detach().cpu().numpy())
    print(f"projector 1280->960, M = {B * 1500}: forward rel-L2 {fe:.3e}")
    margins.add("projector 1280->960 forward vs fp32 torch", fe, 1e-2, "rel-L2")
    assert fe <= 1e-2
    for n in mine:
        e = rel_l2(mine[n].cpu().numpy(), ref_p[n].grad.cpu().numpy())
        print(f"   grad {n}: rel-L2 {e:.3e}")
        margins.add(f"projector 1280->960 grad {n} vs fp32 torch autograd", e, 2e-2, "rel-L2")
        assert e <= 2e-2, f"grad {n}: {e:.3e}"
    # bf16 hand-off (src/modeling.py:220) at this width
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        yb = proj(x)
    assert yb.dtype == torch.bfloat16 and rel_l2(yb.float().cpu().numpy(), y_ref.detach().cpu().numpy()) <= 1e-2


def test_gemm_bias_residual_bf16_above_the_256_tile_threshold(lib):
    """(+ res) -> bf16 at a shape whose tile count would dispatch to the 256-tile kernels, whose bf16 epilogues carry
    no residual add: the launcher keeps this combination on the 128-tile kernel (include/asrk.h: `(+res) -> out_dtype`)."""
    M, N, K = 40003, 960, 192
    g = torch.Generator(device="cuda").manual_seed(99)
    x = (torch.randn((M, K), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda", generator=g) / np.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn((N,), device="cuda", generator=g)
    res = torch.randn((M, N), device="cuda", generator=g) * 4.0
    ref = x.float() @ w.float().T + bias + res
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    rc = lib.asrk_gemm_bf16(_p(x), K, _p(w), _p(bias), _p(res), N, M, _p(out), N, 1, M, N, K, 0, 1.0, _stream())
    assert rc == 0, lib.asrk_strerror(rc)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    assert err <= 1e-2 * max(ref.abs().max().item(), 1.0), err
    # the residual really is in there
    assert (out.float() - (ref - res)).abs().max().item() > 1.0
