"""Probe: one asrk_gemm_bf16 call over M rows against two calls over [0, M1) and [M1, M) (the last, nearly empty round
of 256-row tiles replaced by a second, smaller launch).  python tools/gemm_split_probe.py"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import _lib  # noqa: E402

lib = _lib.load()
p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
wa = torch.randn(8192, 8192, device="cuda").bfloat16()
t_end = time.time() + 0.5
while time.time() < t_end:
    (wa @ wa).sum().item()
M = 96000
for name, N, K, gelu in (("out", 768, 768, 0), ("fc2", 768, 3072, 0), ("qkv", 2304, 768, 0), ("fc1", 3072, 768, 1)):
    x = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn((N,), device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)

    def call(r0, r1):
        rc = lib.asrk_gemm_bf16(p(x[r0:]), K, p(w), p(bias), None, N, r1 - r0, p(out[r0:]), N, 1, r1 - r0, N, K, gelu, 1.0, st)
        assert rc == 0, rc

    def timed(f, iters=20):
        for _ in range(5):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / iters

    tiles_n = (N + 255) // 256
    base = timed(lambda: call(0, M))
    ref = out.clone()
    line = f"{name:4s} N={N} K={K}: one call {base:7.1f} us;"
    # M1 = 8 XCDs x m1 tiles of 256 rows
    res = []
    for m1 in [int(v) for v in os.environ.get("M1_TILES", "8,16,20,24,28,30,32,34,36,38,40,42,44,46").split(",")]:
        M1 = m1 * 8 * 256
        if M1 <= 0 or M1 >= M:
            continue
        out.zero_()
        t = timed(lambda: (call(0, M1), call(M1, M)))
        same = bool(torch.equal(out, ref))
        res.append((t, m1, same))
    line += "  " + "  ".join(f"{m1}:{t:.0f}{'' if same else '!'}" for t, m1, same in res)
    best = min(res)
    line += f"   best {best[1]} tiles/XCD: {best[0]:.1f} us ({100 * (best[0] / base - 1):+.1f} %)"
    print(line, flush=True)
