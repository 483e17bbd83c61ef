"""Per-phase s_memtime sums of the attention key loop (diagnostic build with -DASRK_ATTN_STAMPS):
    tools/build_variant.sh attnst "-DASRK_ATTN_STAMPS" asrk_attn.hip && ASRK_LIB_PATH=asr_amd/libasrk_attnst.so python tools/attn_stamps.py"""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import _lib
lib = _lib.load()
B, T, H = int(os.environ.get("B", 64)), 1500, 12
d = 64 * H
qkv = (torch.randn((B * T, 3 * d), device="cuda") * 0.5).to(torch.bfloat16)
out = torch.empty((B * T, d), device="cuda", dtype=torch.bfloat16)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    assert lib.asrk_attention_log2_bf16(ctypes.c_void_p(qkv.data_ptr()), ctypes.c_void_p(out.data_ptr()), B, T, H, st) == 0
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ["ASRK_LIB_PATH"])
buf = (ctypes.c_ulonglong * (1024 * 8))()
assert raw.asrk_debug_read_attn_stamps(buf, 1024 * 8) == 0
a = np.array(buf, dtype=np.float64).reshape(1024, 8)
nkt = a[:, 6].max()
names = ["wait DMA + barrier", "DMA issue", "QK (reads + MFMA issue)", "softmax", "PV (reads + MFMA issue)", "between tiles"]
tot = 0
for i, n in enumerate(names):
    v = a[:, i].mean() / nkt
    tot += v
    print(f"{n:28s} {v:8.0f} cycles per tile (wave 0 of the first 1024 workgroups)")
print(f"{'sum':28s} {tot:8.0f}")
