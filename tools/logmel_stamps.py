"""Cycles per phase of logmel_tile_kernel from the stamped diagnostic build (asr_amd/libasrk_stamps.so, tools/build_diag.sh):
python tools/logmel_stamps.py [n_mels] [clips]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["ASRK_LIB_PATH"] = os.path.join(ROOT, "asr_amd", "libasrk_stamps.so")
sys.path.insert(0, ROOT)
from asr_amd import LogMelFeatureExtractor, _lib  # noqa: E402

n_mels = int(sys.argv[1]) if len(sys.argv) > 1 else 80
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
lib = _lib.load()
fe = LogMelFeatureExtractor(feature_size=n_mels)
w = (torch.randn((B, 480000), device="cuda") * 0.1).clamp_(-1, 1)
for _ in range(3):
    fe.extract_from_device(w)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (1024 * 16))()
lib.asrk_debug_read_lmstamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.asrk_debug_read_lmstamps(buf, 1024 * 16) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 16)[:, :10].astype(np.float64)
tiles = (B * ((3000 + 23) // 24)) / 1024.0
names = ["sample loads", "window+DFT1+twiddle", "exchange", "(DFT2: moves)", "DFT2+mirror+power", "mel gather", "barrier B", "store pass", "barrier A", "loop"]
tot = a.sum(axis=1).mean()
print(f"n_mels {n_mels}, {B} clips, {tiles:.1f} tiles per workgroup; cycles per tile (wave 0, mean over workgroups):")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {a[:, i].mean() / tiles:9.0f}  {100 * a[:, i].mean() / tot:5.1f} %")
print(f"  total                  {tot / tiles:9.0f}")
