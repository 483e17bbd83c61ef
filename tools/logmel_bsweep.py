"""Batch sweep of the log-mel front-end (run under rocprofv3 --kernel-trace; tools/logmel_bsweep.sh reads the trace):
duration of the tile kernel as a function of the number of clips separates per-launch costs from per-tile costs."""
import sys

import torch

from asr_amd import LogMelFeatureExtractor

BATCHES = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 8, 32, 64, 128, 256, 512]
fe = LogMelFeatureExtractor(feature_size=80)
g = torch.Generator(device="cuda").manual_seed(0)
w = (torch.randn((max(BATCHES), 480000), device="cuda", generator=g) * 0.1).clamp_(-1, 1)
for b in BATCHES:
    for _ in range(4):
        fe.extract_from_device(w[:b])
    torch.cuda.synchronize()
print("BATCHES", ",".join(map(str, BATCHES)))
