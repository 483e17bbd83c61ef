"""Diagnostic: phase times of the persistent 256x256 GEMM (asrk_gemm256p.hip) from in-kernel s_memtime stamps
(needs asr_amd/libasrk_stamps.so from tools/build_diag.sh; run with ASRK_GEMM_TILE=256 ASRK_GEMM_CFG=4)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ.get("ASRK_DIAG_LIB", os.path.join(ROOT, "asr_amd", "libasrk_stamps.so")))
from ctypes import c_float, c_int, c_int64, c_void_p  # noqa: E402

lib.asrk_gemm_bf16.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int,
                               c_int, c_int, c_int, c_int, c_float, c_void_p]
lib.asrk_debug_read_pstamps.argtypes = [c_void_p, c_int]
M, N, K, gelu = (int(v) for v in sys.argv[1:5])
f32res = int(sys.argv[5]) if len(sys.argv) > 5 else 0
x = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16)
w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn((N,), device="cuda")
out = torch.randn((M, N), device="cuda") if f32res else torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
p = lambda t: c_void_p(t.data_ptr())  # noqa: E731
st = c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    assert lib.asrk_gemm_bf16(p(x), K, p(w), p(bias), p(out) if f32res else None, N, M, p(out), N, 0 if f32res else 1,
                              M, N, K, gelu, 1.0, st) == 0
torch.cuda.synchronize()
nb = 256
buf = np.zeros((nb, 8), dtype=np.uint64)
assert lib.asrk_debug_read_pstamps(buf.ctypes.data_as(c_void_p), nb * 8) == 0
t = buf.astype(np.int64)
med = lambda v: float(np.median(v))  # noqa: E731
tiles = ((M + 255) // 256) * ((N + 255) // 256)


def _rounds(bm):  # schedule_cost of asrk_gemm256p.hip (256 CUs)
    tm, tn = (M + bm - 1) // bm, (N + 255) // 256
    return -(-(((tm + 7) // 8) * tn) // 32)


mt6 = os.environ.get("ASRK_GEMM_MT") == "6" or (os.environ.get("ASRK_GEMM_MT") != "8" and _rounds(192) * 0.84 < _rounds(256))
print(f"tile height chosen by the launcher: {192 if mt6 else 256} rows (a 192-row slice is 3/4 of the MFMA work)")
print(f"M={M} N={N} K={K} gelu={gelu} f32res={f32res}: {tiles} tiles, {tiles / 256:.2f} per workgroup; shader clocks")
print(f"  tile 0: load wait {med(t[:, 1] - t[:, 0]):8.0f}   K loop {med(t[:, 2] - t[:, 1]):8.0f}   epilogue {med(t[:, 3] - t[:, 2]):8.0f}")
print(f"  tile 1: wait at first barrier {med(t[:, 5] - t[:, 3]):8.0f}   K loop + epilogue {med(t[:, 6] - t[:, 5]):8.0f}")
print(f"  whole workgroup {med(t[:, 7] - t[:, 0]):8.0f}  = {med(t[:, 7] - t[:, 0]) / (tiles / 256):8.0f} per tile")
