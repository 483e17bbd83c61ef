"""GPU parity of the building-block kernels, each called through the C ABI (ctypes) and compared
with a plain fp32 torch computation of the same op on the same (bf16-rounded) operands."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from asr_amd import _lib

    return _lib.load()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / np.sqrt(2.0)))


# the last two shapes have >= 512 tiles of 256 x 256, so they dispatch to the persistent kernel (bf16 outputs) and to the
# one-tile-per-workgroup 256 kernel (fp32 outputs); M is ragged in both, N = 960 also has a ragged last n-tile
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (1500, 768, 768), (300, 960, 768), (4500, 2304, 768),
                                   (77, 132, 3072), (3000, 768, 256), (66017, 512, 128), (40003, 960, 192)])
@pytest.mark.parametrize("mode", ["plain_bf16", "bias_gelu_bf16", "bias_res_f32", "alpha_f32"])
def test_gemm(lib, M, N, K, mode):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    x = (torch.randn((M, K), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda", generator=g) / np.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn((N,), device="cuda", generator=g)
    res = torch.randn((M, N), device="cuda", generator=g)
    ref = x.float() @ w.float().T
    use_bias = mode != "plain_bf16"
    gelu = mode == "bias_gelu_bf16"
    use_res = mode == "bias_res_f32"
    alpha = 0.01 if mode == "alpha_f32" else 1.0
    out_bf16 = mode.endswith("bf16")
    if use_bias:
        ref = ref + bias
    if gelu:
        ref = _gelu(ref)
    ref = ref * alpha
    if use_res:
        ref = ref + res
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
    rc = lib.asrk_gemm_bf16(_p(x), K, _p(w), _p(bias if use_bias else None), _p(res if use_res else None), N, M, _p(out), N,
                            1 if out_bf16 else 0, M, N, K, int(gelu), alpha, _stream())
    assert rc == 0, lib.asrk_strerror(rc)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    scale = ref.abs().max().item()
    tol = (1e-2 if out_bf16 else 2e-5 * np.sqrt(K)) * max(scale, 1.0)
    assert torch.isfinite(out.float()).all()
    assert err <= tol, f"max err {err:.3e} vs tol {tol:.3e} (scale {scale:.2f})"


def test_gemm_identity_asymmetric(lib):
    """A = I against an asymmetric B catches a transposed / permuted accumulator map exactly."""
    K = N = 128
    M = 128
    x = torch.eye(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.arange(N * K, device="cuda").reshape(N, K) % 251).float().sub(125.0).to(torch.bfloat16)  # exact in bf16
    out = torch.empty((M, N), device="cuda", dtype=torch.float32)
    rc = lib.asrk_gemm_bf16(_p(x), K, _p(w), _p(None), _p(None), 0, 0, _p(out), N, 0, M, N, K, 0, 1.0, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(out, w.float().T)


@pytest.mark.parametrize("out_bf16", [True, False])
@pytest.mark.parametrize("M,N,K", [(66017, 512, 64), (33001, 1024, 128)])
def test_gemm_large_tiles_exact(lib, M, N, K, out_bf16):
    """The 256 x 256-tile kernels (persistent for bf16, one tile per workgroup for fp32) on small-integer operands:
    every product sum is an integer <= 256 in magnitude, exact in fp32 and in bf16, so the result must be bit-exact
    -- any wrong tile, row map, swizzle or ragged-edge handling shows as a mismatch, not as a tolerance."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x = torch.randint(-1, 2, (M, K), device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randint(-2, 3, (N, K), device="cuda", generator=g).to(torch.bfloat16)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
    ref = x.float() @ w.float().T + bias
    assert ref.abs().max().item() <= 256
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
    rc = lib.asrk_gemm_bf16(_p(x), K, _p(w), _p(bias), _p(None), 0, 0, _p(out), N, int(out_bf16), M, N, K, 0, 1.0, _stream())
    assert rc == 0, lib.asrk_strerror(rc)
    torch.cuda.synchronize()
    assert torch.equal(out.float(), ref)


@pytest.mark.parametrize("M,out_bf16", [(4096, False), (70000, True)])
def test_gemm_gelu_epilogue_accuracy(lib, M, out_bf16):
    """W = I, so out = alpha * gelu(x + bias) element by element, against float64 erf.  fp32 output (128-tile and
    one-tile 256 kernels; the packed-fp32 erf form): |err| <= 1e-6 * max(1, |x|); bf16 output at a size that takes the
    persistent kernel's straight-line epilogue (the logistic form, |err| <= 2.6e-5 before the rounding): one bf16
    rounding plus that bound.  Inputs reach +-40, far beyond the fit's clamp at 7.25."""
    K = N = 256
    g = torch.Generator(device="cuda").manual_seed(M)
    x = (torch.rand((M, K), device="cuda", generator=g) * 24.0 - 12.0).to(torch.bfloat16)
    x[0, :8] = torch.tensor([0.0, -0.0, 1e-3, -1e-3, 12.0, -12.0, 40.0, -40.0], device="cuda").to(torch.bfloat16)
    w = torch.eye(N, K, device="cuda").to(torch.bfloat16)
    bias = torch.rand((N,), device="cuda", generator=g) - 0.5
    alpha = 1.25
    pre = x.double() + bias.double()
    ref = alpha * 0.5 * pre * (1.0 + torch.erf(pre / np.sqrt(2.0)))
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
    rc = lib.asrk_gemm_bf16(_p(x), K, _p(w), _p(bias), _p(None), 0, 0, _p(out), N, int(out_bf16), M, N, K, 1, alpha, _stream())
    assert rc == 0, lib.asrk_strerror(rc)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    if out_bf16:
        # fp32 value within alpha * 2.6e-5 of the truth, then one bf16 rounding (relative 2^-9 incl. ties)
        err = (out.double() - ref).abs()
        assert (err <= ref.abs() * 2.0 ** -8 + 3.5e-5).all(), err.max().item()
    else:
        err = (out.double() - ref).abs() / pre.abs().clamp(min=1.0)
        assert err.max().item() <= 1.5e-6, err.max().item()


def test_persistent_gemm_sweep():
    """The persistent kernel's tile stream over many small shapes (see tests/persistent_gemm_sweep.py), forced by the
    measurement switches.  Those exist only in the diagnostic build of the same sources (asr_amd/libasrk_diag.so,
    `make -C asr_amd/csrc diag`, built by __graft_entry__.build()): the product library reads no environment
    variable.  The switches are read once per process, hence one child process for the whole sweep."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    diag = os.path.join(root, "asr_amd", "libasrk_diag.so")
    assert os.path.exists(diag), "build it: make -C asr_amd/csrc diag"
    env = dict(os.environ, ASRK_LIB_PATH=diag, ASRK_GEMM_TILE="256", ASRK_GEMM_CFG="4", PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "persistent_gemm_sweep.py")], env=env, capture_output=True,
                       text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "bit-exact" in r.stdout


def test_gemm_rejects_bad_shapes(lib):
    x = torch.zeros((64, 96), device="cuda", dtype=torch.bfloat16)
    out = torch.zeros((64, 64), device="cuda")
    assert lib.asrk_gemm_bf16(_p(x), 96, _p(x), _p(None), _p(None), 0, 0, _p(out), 64, 0, 64, 64, 96, 0, 1.0, _stream()) != 0
    assert lib.asrk_gemm_bf16(_p(None), 64, _p(x), _p(None), _p(None), 0, 0, _p(out), 64, 0, 64, 64, 64, 0, 1.0, _stream()) != 0


@pytest.mark.parametrize("M,D", [(1, 128), (1500, 768), (333, 1280), (64, 256)])
@pytest.mark.parametrize("out_bf16", [False, True])
def test_layernorm(lib, M, D, out_bf16):
    g = torch.Generator(device="cuda").manual_seed(M + D)
    x = torch.randn((M, D), device="cuda", generator=g) * 3 + 1.5
    w = torch.randn((D,), device="cuda", generator=g)
    b = torch.randn((D,), device="cuda", generator=g)
    ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-5)
    out = torch.empty((M, D), device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
    rc = lib.asrk_layernorm(_p(x), _p(w), _p(b), _p(out), int(out_bf16), M, D, 1e-5, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    err = (out.float() - ref).abs().max().item()
    assert err <= (4e-2 if out_bf16 else 2e-5), err


@pytest.mark.parametrize("M,D", [(1, 128), (1500, 768), (333, 1280)])
@pytest.mark.parametrize("out_bf16,keep_x", [(True, True), (False, False), (False, True)])
def test_add_layernorm(lib, M, D, out_bf16, keep_x):
    """x += delta (fp32 += bf16) fused in front of the LayerNorm; the fp32 sum is written back bit-exactly."""
    g = torch.Generator(device="cuda").manual_seed(7 * M + D)
    x = torch.randn((M, D), device="cuda", generator=g) * 3 + 1.5
    delta = (torch.randn((M, D), device="cuda", generator=g) * 2).to(torch.bfloat16)
    w = torch.randn((D,), device="cuda", generator=g)
    b = torch.randn((D,), device="cuda", generator=g)
    x_new = x + delta.float()
    ref = torch.nn.functional.layer_norm(x_new, (D,), w, b, 1e-5)
    xin = x.clone()
    out = torch.empty((M, D), device="cuda", dtype=torch.bfloat16 if out_bf16 else torch.float32)
    rc = lib.asrk_add_layernorm(_p(xin), _p(delta), _p(w), _p(b), _p(out), int(out_bf16), int(keep_x), M, D, 1e-5, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(xin, x_new if keep_x else x)
    err = (out.float() - ref).abs().max().item()
    assert err <= (4e-2 if out_bf16 else 2e-5), err
    assert lib.asrk_add_layernorm(_p(xin), _p(None), _p(w), _p(b), _p(out), int(out_bf16), 1, M, D, 1e-5, _stream()) != 0


@pytest.mark.parametrize("B,T,H", [(1, 64, 1), (2, 128, 2), (1, 1500, 2), (2, 1500, 12), (1, 100, 3), (3, 257, 1)])
def test_attention(lib, B, T, H):
    d = 64 * H
    g = torch.Generator(device="cuda").manual_seed(B * 100 + T + H)
    qkv = torch.randn((B * T, 3 * d), device="cuda", generator=g)
    qkv[:, :d] *= 0.35  # q already carries the head_dim^-0.5 scaling
    qkv = qkv.to(torch.bfloat16)
    out = torch.full((B * T, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    rc = lib.asrk_attention_bf16(_p(qkv), _p(out), B, T, H, _stream())
    assert rc == 0
    torch.cuda.synchronize()
    f = qkv.float().reshape(B, T, 3, H, 64)
    q, k, v = f[:, :, 0].transpose(1, 2), f[:, :, 1].transpose(1, 2), f[:, :, 2].transpose(1, 2)
    p = torch.softmax(q @ k.transpose(2, 3), dim=-1)
    ref = (p @ v).transpose(1, 2).reshape(B * T, d)
    err = (out.float() - ref).abs().max().item()
    assert torch.isfinite(out.float()).all()
    assert err <= 2e-2, f"max err {err:.3e}"


@pytest.mark.parametrize("scale,shift", [(0.35, 0.0), (6.0, 0.0), (3.0, 9.0), (0.35, -9.0)])
def test_attention_log2_domain(lib, scale, shift):
    """The encoder's form: q pre-multiplied by log2(e).  Small scores, large ones (|s| up to a few hundred in log2
    units), a large common offset and late jumps of the running maximum exercise the lazy reference moves."""
    B, T, H = 2, 1500, 3
    d = 64 * H
    g = torch.Generator(device="cuda").manual_seed(int(scale * 100) + int(shift))
    q = torch.randn((B * T, d), device="cuda", generator=g) * scale
    k = torch.randn((B * T, d), device="cuda", generator=g) + shift * torch.sign(q.mean(0, keepdim=True) + 1e-3)
    v = torch.randn((B * T, d), device="cuda", generator=g)
    k[B * T // 2 + 100:B * T // 2 + 110] *= 4.0  # a late jump of the running maximum for many queries
    qkv_nat = torch.cat([q, k, v], dim=1).to(torch.bfloat16)                    # natural-log scores (reference)
    qkv_l2 = torch.cat([q * 1.4426950408889634, k, v], dim=1).to(torch.bfloat16)
    qkv_l2[:, d:] = qkv_nat[:, d:]
    out = torch.full((B * T, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    assert lib.asrk_attention_log2_bf16(_p(qkv_l2), _p(out), B, T, H, _stream()) == 0
    torch.cuda.synchronize()
    # reference on the SAME rounded q: undo the log2(e) factor in fp32
    qf = (qkv_l2[:, :d].float() / 1.4426950408889634).view(B, T, H, 64).transpose(1, 2)
    kf = qkv_l2[:, d:2 * d].float().view(B, T, H, 64).transpose(1, 2)
    vf = qkv_l2[:, 2 * d:].float().view(B, T, H, 64).transpose(1, 2)
    ref = torch.softmax(qf @ kf.transpose(-1, -2), dim=-1) @ vf
    ref = ref.transpose(1, 2).reshape(B * T, d)
    assert torch.isfinite(out.float()).all()
    err = (out.float() - ref).abs().max().item()
    assert err <= 3e-2 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("case", ["late_spike", "all_far_negative", "ordinary", "inf_free_tail"])
def test_attention_log2_reference_free_path_and_its_fallback(lib, case):
    """The encoder's attention (log2 domain) takes p = 2^s WITHOUT a reference and checks after the key loop that every
    row sum stayed inside [2^-100, 2^100) and the accumulators are finite; otherwise the workgroup repeats the loop in
    the eager form.  `late_spike`: one key in the 5th tile scores ~ +370 log2 units for one query (the shortcut
    overflows to inf -> fallback).  `all_far_negative`: every score of one query is ~ -290 (row sum underflows ->
    fallback).  `ordinary`: |s| of a few units, the shortcut's result stands.  `inf_free_tail`: T = 1500 whose last
    tile holds 28 keys (the half-tile instance).  Reference: float64 softmax."""
    T, H = (1500, 2) if case == "inf_free_tail" else (384, 1)
    d = 64 * H
    g = torch.Generator(device="cuda").manual_seed(17)
    q = torch.randn((T, d), device="cuda", generator=g) * 0.4
    k = torch.randn((T, d), device="cuda", generator=g)
    v = torch.randn((T, d), device="cuda", generator=g)
    if case == "late_spike":
        q[7, :64] = 2.0
        k[300, :64] = 2.0        # score 256 natural units = 369 log2 units, in tile 4
    if case == "all_far_negative":
        k[:, :64] = k[:, :64] * 0.05 + 1.0
        q[9, :64] = -3.125       # every score ~ -200 natural units
    LOG2E = 1.4426950408889634
    qkv = torch.cat([q * LOG2E, k, v], dim=1).to(torch.bfloat16)
    out = torch.full((T, d), float("nan"), device="cuda", dtype=torch.bfloat16)
    assert lib.asrk_attention_log2_bf16(_p(qkv), _p(out), 1, T, H, _stream()) == 0
    torch.cuda.synchronize()
    f = qkv.double()
    qd = (f[:, :d] / LOG2E).view(T, H, 64).transpose(0, 1)
    kd = f[:, d:2 * d].view(T, H, 64).transpose(0, 1)
    vd = f[:, 2 * d:].view(T, H, 64).transpose(0, 1)
    sc = qd @ kd.transpose(-1, -2)
    ref = (torch.softmax(sc, dim=-1) @ vd).transpose(0, 1).reshape(T, d)
    if case == "late_spike":
        assert sc[0, 7].max().item() > 200 and sc[0, 7].argmax().item() == 300
    if case == "all_far_negative":
        assert sc[0, 9].max().item() < -150
    assert torch.isfinite(out.float()).all()
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-2 * max(1.0, ref.abs().max().item()), err
    # the same call again gives the same bits (the fallback leaves no state behind)
    out2 = torch.empty_like(out)
    assert lib.asrk_attention_log2_bf16(_p(qkv), _p(out2), 1, T, H, _stream()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, out2)


def test_attention_online_softmax_rescale(lib):
    """Force the running-max rescale: one key, late in the sequence, dominates one query."""
    B, T, H = 1, 320, 1
    g = torch.Generator(device="cuda").manual_seed(9)
    qkv = torch.randn((T, 192), device="cuda", generator=g) * 0.3
    qkv[5, 0:64] = 2.0          # query 5
    qkv[300, 64:128] = 4.0      # key 300 (5th tile) -> score 512 for query 5
    qkv[10, 64:128] = 1.0       # an earlier, smaller spike in tile 0
    qkv = qkv.to(torch.bfloat16)
    out = torch.empty((T, 64), device="cuda", dtype=torch.bfloat16)
    assert lib.asrk_attention_bf16(_p(qkv), _p(out), B, T, H, _stream()) == 0
    torch.cuda.synchronize()
    f = qkv.double()
    p = torch.softmax(f[:, :64] @ f[:, 64:128].T, dim=-1)
    ref = p @ f[:, 128:]
    assert (out.double() - ref).abs().max().item() <= 2e-2
    assert (out[5].double() - f[300, 128:]).abs().max().item() <= 2e-2  # query 5 attends to key 300 only
