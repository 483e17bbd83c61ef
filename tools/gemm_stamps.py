"""Diagnostic: per-workgroup phase times of the 256x256 GEMM from in-kernel s_memtime stamps
(needs asr_amd/libasrk_stamps.so from tools/build_diag.sh).  Not part of the product or the tests."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "asr_amd", "libasrk_stamps.so"))
from ctypes import c_float, c_int, c_int64, c_void_p  # noqa: E402

lib.asrk_gemm_bf16.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64, c_int,
                               c_int, c_int, c_int, c_int, c_float, c_void_p]
lib.asrk_debug_read_stamps.argtypes = [c_void_p, c_int]

M = int(sys.argv[1]) if len(sys.argv) > 1 else 96000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
K = int(sys.argv[3]) if len(sys.argv) > 3 else 768
gelu = int(sys.argv[4]) if len(sys.argv) > 4 else 0
x = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16)
w = (torch.randn((N, K), device="cuda") * K ** -0.5).to(torch.bfloat16)
bias = torch.randn((N,), device="cuda")
f32res = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # 1: fp32 output with in-place fp32 residual (out-proj / fc2)
out = torch.randn((M, N), device="cuda") if f32res else torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
p = lambda t: c_void_p(t.data_ptr())  # noqa: E731
st = c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    assert lib.asrk_gemm_bf16(p(x), K, p(w), p(bias), p(out) if f32res else None, N, M, p(out), N, 0 if f32res else 1, M, N, K, gelu, 1.0, st) == 0
torch.cuda.synchronize()
nb = ((M + 255) // 256) * ((N + 255) // 256)
nb = min(nb, 16384)
buf = np.zeros((nb, 8), dtype=np.uint64)
assert lib.asrk_debug_read_stamps(buf.ctypes.data_as(c_void_p), nb * 8) == 0
t = buf[:, :6].astype(np.int64)
t0 = t[:, 0].min()
names = ["prologue(load wait)", "first K tile", "rest of main loop", "epilogue issue", "store drain"]
d = np.diff(t, axis=1)
print(f"M={M} N={N} K={K} blocks={nb}; s_memtime ticks (shader clocks)")
for i, nme in enumerate(names):
    print(f"  {nme:22s} median {np.median(d[:, i]):8.0f}  p10 {np.percentile(d[:, i], 10):8.0f}  p90 {np.percentile(d[:, i], 90):8.0f}")
ep = buf[:, [3, 6, 7, 4]].astype(np.int64)
if ep[:, 1].max() > 0:
    de = np.diff(ep, axis=1)
    lab = ["  epi: pass 0 staged", "  epi: pass 0 add+store", "  epi: pass 1"] if f32res else ["  epi: value+LDS write", "  epi: barrier", "  epi: LDS read+store"]
    for i, nme in enumerate(lab):
        print(f"  {nme:22s} median {np.median(de[:, i]):8.0f}")
wl = np.zeros((nb, 2), dtype=np.uint64)
lib.asrk_debug_read_wall.argtypes = [c_void_p, c_int]
assert lib.asrk_debug_read_wall(wl.ctypes.data_as(c_void_p), nb * 2) == 0
dw = (wl[:, 1].astype(np.int64) - wl[:, 0].astype(np.int64)) * 10.0  # ns
dc = (t[:, 4] - t[:, 0]).astype(np.float64)
print(f"  shader clock during the tile: median {np.median(dc / dw) * 1000:.0f} MHz (s_memtime / s_memrealtime)")
tot = t[:, 5] - t[:, 0]
print(f"  total per block        median {np.median(tot):8.0f}; kernel span {t[:, 5].max() - t0} ticks")

