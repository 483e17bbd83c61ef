"""Helper of tests/test_kernels_gpu.py::test_persistent_gemm_sweep (not collected by pytest: no test_ prefix).

Run in a process of its own with ASRK_GEMM_TILE=256 ASRK_GEMM_CFG=4, which sends EVERY bf16-output GEMM with K >= 192 to
the persistent kernel whatever its size, so that its tile stream is exercised with 1, 2, 3 ... tiles per workgroup,
3 to 12 K slices per tile, ragged M and a ragged last n-tile.  Operands are small integers: every sum is an integer
below 2^8 in magnitude, exact in fp32 and in bf16, so the comparison is bit-exact.
"""
import ctypes
import itertools
import sys

import torch

from asr_amd import _lib


def main():
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())  # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(1234)
    n_cases = 0
    for M, N, K in itertools.product((256, 700, 2051, 9000, 70000), (256, 520, 768, 960), (192, 256, 320, 768)):
        if M == 70000 and (N, K) not in ((256, 192), (768, 320), (960, 256)):
            continue
        x = torch.randint(-1, 2, (M, K), device="cuda", generator=g).to(torch.bfloat16)
        w = torch.randint(-1, 2, (N, K), device="cuda", generator=g).to(torch.bfloat16)
        # keep |sum| small: zero all but 64 columns of K
        w[:, 64:] = 0
        bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).float()
        for use_bias in (True, False):
            ref = x.float() @ w.float().T + (bias if use_bias else 0.0)
            assert ref.abs().max().item() <= 128
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            rc = lib.asrk_gemm_bf16(p(x), K, p(w), p(bias if use_bias else None), p(None), 0, 0, p(out), N, 1, M, N, K, 0, 1.0, st)
            assert rc == 0, (M, N, K, lib.asrk_strerror(rc))
            torch.cuda.synchronize()
            if not torch.equal(out.float(), ref):
                bad = (out.float() != ref).nonzero()
                print(f"MISMATCH M={M} N={N} K={K} bias={use_bias}: {bad.shape[0]} elements, first {bad[0].tolist()}")
                return 1
            n_cases += 1
    print(f"persistent GEMM sweep: {n_cases} cases bit-exact")
    return 0


if __name__ == "__main__":
    sys.exit(main())
