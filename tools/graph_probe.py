"""Diagnostic: does a captured HIP graph of the pipeline step beat stream launches?  (inter-kernel gaps without a profiler)
    python tools/graph_probe.py [workload] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from asr_amd import LogMelFeatureExtractor, _lib  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "pipeline-small"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = dict(bench.WORKLOADS[wl])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
_lib.load()
fe = LogMelFeatureExtractor(feature_size=w.get("n_mels", 80))
wave = bench.synth_audio(w["batch"], 0, dev)
enc, proj = bench.seeded_models(w, dev)


def step():
    with torch.no_grad():
        return bench.encode_and_project(enc, proj, fe.extract_from_device(wave, None))


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


eager = timed(step)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
graph = timed(g.replay)
eager2 = timed(step)
ref = step()
g.replay()
torch.cuda.synchronize()
for r in range(8):
    ref_r = step()
    g.replay()
    torch.cuda.synchronize()
    if not torch.equal(ref_r, out) or not torch.equal(ref_r, ref):
        d = (ref_r.float() - out.float()).abs()
        d2 = (ref_r.float() - ref.float()).abs()
        idx = torch.nonzero(d.flatten() > 0).flatten()
        print(f"  rep {r}: eager vs graph differ in {idx.numel()} elements (max {float(d.max()):.3e}), eager vs first eager max {float(d2.max()):.3e}",
              f"first index {int(idx[0]) if idx.numel() else -1} of shape {tuple(out.shape)}")
print(f"{wl}: stream launches {eager:.3f} / {eager2:.3f} ms per step, graph replay {graph:.3f} ms; same result: {bool(torch.equal(ref, out))}")

# which part differs under capture?
def cap(fn):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        o = fn()
    return gg, o


with torch.no_grad():
    feats = fe.extract_from_device(wave, None)
    hid = enc(feats)
    for name, fn, eager_out in (("log-mel", lambda: fe.extract_from_device(wave, None), feats), ("encoder", lambda: enc(feats), hid),
                                ("projector", lambda: proj(hid), proj(hid)), ("fused", lambda: bench.encode_and_project(enc, proj, feats), bench.encode_and_project(enc, proj, feats))):
        gg, o = cap(fn)
        gg.replay()
        torch.cuda.synchronize()
        a = o.clone()
        gg.replay()
        torch.cuda.synchronize()
        print(f"  {name}: graph == eager {bool(torch.equal(a, eager_out))}, replay == replay {bool(torch.equal(a, o))}, max diff {float((a.float() - eager_out.float()).abs().max()):.3e}")
