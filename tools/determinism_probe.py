"""Diagnostic: is the step bit-reproducible run after run?  python tools/determinism_probe.py [workload] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from asr_amd import LogMelFeatureExtractor, _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "pipeline-large"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
w = dict(bench.WORKLOADS[wl])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
_lib.load()
fe = LogMelFeatureExtractor(feature_size=w.get("n_mels", 80))
wave = bench.synth_audio(w["batch"], 0, dev)
enc, proj = bench.seeded_models(w, dev)
with torch.no_grad():
    f0 = fe.extract_from_device(wave, None)
    h0 = enc(f0)
    y0 = bench.encode_and_project(enc, proj, f0)
    bad = {"feat": 0, "hid": 0, "out": 0}
    for r in range(reps):
        f = fe.extract_from_device(wave, None)
        h = enc(f0)
        y = bench.encode_and_project(enc, proj, f0)
        torch.cuda.synchronize()
        for k, a, b in (("feat", f, f0), ("hid", h, h0), ("out", y, y0)):
            if not torch.equal(a, b):
                bad[k] += 1
                d = (a.float() - b.float()).abs()
                idx = torch.nonzero(d.flatten() > 0).flatten()
                print(f"rep {r}: {k} differs in {idx.numel()} elements, max {float(d.max()):.3e}; first flat index {int(idx[0])}, last {int(idx[-1])}, shape {tuple(a.shape)}")
print(wl, "mismatching reps:", bad, "of", reps)
# the same under a captured graph, alternating with stream launches
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    bench.encode_and_project(enc, proj, fe.extract_from_device(wave, None))
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g), torch.no_grad():
    yg = bench.encode_and_project(enc, proj, fe.extract_from_device(wave, None))
nbad = 0
for r in range(reps):
    with torch.no_grad():
        ye = bench.encode_and_project(enc, proj, fe.extract_from_device(wave, None))
    g.replay()
    torch.cuda.synchronize()
    for k, a in (("eager", ye), ("graph", yg)):
        if not torch.equal(a, y0):
            nbad += 1
            d = (a.float() - y0.float()).abs()
            idx = torch.nonzero(d.flatten() > 0).flatten()
            rows = torch.unique(idx // a.shape[-1])
            print(f"rep {r}: {k} differs in {idx.numel()} elements, max {float(d.max()):.3e}; rows {rows[:8].tolist()} .. {int(rows[-1])} ({rows.numel()} rows of {a.shape[0] * a.shape[1]})")
print(wl, "eager/graph alternating: mismatches", nbad, "of", 2 * reps)
