"""Diagnostic only: what the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches at the encoder's shapes."""
import time
import torch
wa = torch.randn(8192, 8192, device='cuda').bfloat16()
t_end = time.time() + 0.5
while time.time() < t_end:
    (wa @ wa).sum().item()
shapes = {"qkv": (96000, 2304, 768), "out": (96000, 768, 768), "fc1": (96000, 3072, 768), "fc2": (96000, 768, 3072), "sq4k": (4096, 4096, 4096), "sq8k": (8192, 8192, 8192)}
for name, (M, N, K) in shapes.items():
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
    for _ in range(5):
        y = x @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = x @ w.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print(f"{name:6s} M={M} N={N} K={K}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s (torch.matmul, no bias/epilogue)")
