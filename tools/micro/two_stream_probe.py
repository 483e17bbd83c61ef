"""Probe: encoder on one stream over B clips, against two half-batches on two streams (two encoder objects, same
weights), to see whether kernels of one half fill the tails / memory-bound phases of the other."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd.encoder import WhisperEncoder  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.manual_seed(0)
e1 = WhisperEncoder().to("cuda")
e2 = WhisperEncoder().to("cuda")
e2.load_state_dict(e1.state_dict())
x = torch.randn((B, 80, 3000), device="cuda") * 0.3
_ = torch.randn((4096, 4096), device="cuda")
t0 = time.time()
while time.time() - t0 < 0.5:
    _ @ _
torch.cuda.synchronize()


def one():
    return e1(x)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
xa, xb = x[: B // 2].contiguous(), x[B // 2:].contiguous()


def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a = e1(xa)
    with torch.cuda.stream(s2):
        b = e2(xb)
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    return a, b


for name, fn in (("one stream", one), ("two streams", two), ("one stream", one), ("two streams", two)):
    for _i in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _i in range(10):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{name}: {ms:.3f} ms per {B} clips = {B * 30 / ms * 1e3:.0f} audio-s/s", flush=True)
ya, yb = two()
y = one()
torch.cuda.synchronize()
print("max diff", (torch.cat([ya, yb]) - y).abs().max().item())
