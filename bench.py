#!/usr/bin/env python
"""Benchmark of the audio hot path on MI355X: log-mel -> encoder -> projector on 30 s clips.

    python bench.py [--gpus N --steps K --warmup W] [--workload pipeline-small|pipeline-large|train-large|logmel|splice]

One "step" = one pass of the hot path over one batch of synthetic 16 kHz audio that is already
resident in HBM.  With N > 1 the driver launches one rank per GPU through torch.distributed.run;
every rank processes its own batch (clips are independent: weak scaling, no data-path
collective), the ranks meet at a barrier on both sides of the timed region and the slowest rank's
time is used.  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  pipeline-small  (default) 64 x 30 s per GPU, whisper-small dims (d768 L12 H12 ffn3072, 80 mel),
                  projector 768 -> 960 -> 960, bf16 MFMA: configs[2] fed by the configs[1] front-end,
                  i.e. the metric's "log-mel+encoder+projector"
  pipeline-large  16 x 30 s per GPU, large-v3 dims (d1280 L32 H20 ffn5120, 128 mel)
  train-large     configs[3]: pipeline-large plus projector backward (synthetic upstream gradient) and, for N > 1,
                  one flat RCCL all-reduce of the projector gradients per step
  logmel          256 x 30 s per GPU, front-end only (configs[1]), HBM-bound
  splice          SURVEY 8(f2): the embed splice of ASRModel.forward alone, 64 samples (HBM-bound copy)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "audio-sec/sec (RTFx) for log-mel+encoder+projector, 30 s clips, 1/2/4/8 GPUs"
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0      # HBM3E peak, same table

WORKLOADS = {
    "pipeline-small": dict(batch=64, d_model=768, layers=12, heads=12, ffn=3072, n_mels=80, text_dim=960),
    "pipeline-large": dict(batch=16, d_model=1280, layers=32, heads=20, ffn=5120, n_mels=128, text_dim=960),
    # BASELINE.json configs[3]: large-v3-class encoder forward + projector forward/backward + gradient all-reduce
    "train-large": dict(batch=16, d_model=1280, layers=32, heads=20, ffn=5120, n_mels=128, text_dim=960, train=True),
    "logmel": dict(batch=256, n_mels=80),
    # the embed splice of ASRModel.forward, 64 samples x (64 text tokens + 1500 audio
    # positions) x 960 (SmolLM2-360M width), bf16 like the decoder; HBM-bound copy
    "splice": dict(batch=64, text_len=64, text_dim=960),
}


def flops_per_clip(w):
    """Forward GEMM + attention flops per 30 s clip (2*M*N*K counting)."""
    d, L, F, C, T = w["d_model"], w["layers"], w["ffn"], w["n_mels"], w["text_dim"]
    stem = 2 * 3000 * d * 3 * C + 2 * 1500 * d * 3 * d
    layer = 2 * 1500 * (4 * d * d + 2 * d * F) + 4 * 1500 * 1500 * d
    proj = 2 * 1500 * (d * T + T * T)
    if w.get("train"):  # projector backward: dW1, dW2 and dH through W2 (no dX into the frozen encoder)
        proj += 2 * 1500 * (d * T + 2 * T * T)
    return stem + L * layer + proj


def synth_audio(batch, rank, device):
    # np.random.default_rng(1234 + rank), 0.1 * N(0,1) clipped to [-1, 1], 480000 samples
    rng = np.random.default_rng(1234 + rank)
    x = np.clip(0.1 * rng.standard_normal((min(batch, 8), 480000), dtype=np.float32), -1.0, 1.0)
    t = torch.from_numpy(x).to(device)
    reps = (batch + t.shape[0] - 1) // t.shape[0]
    if reps > 1:  # distinct clips by rolling, cheap to generate
        t = torch.cat([torch.roll(t, 7919 * i, dims=1) for i in range(reps)], dim=0)[:batch]
    return t.contiguous()


def seeded_models(w, device):
    from asr_amd.encoder import WhisperEncoder
    from asr_amd.projector import AudioProjector

    torch.manual_seed(0)
    enc = WhisperEncoder(w["d_model"], w["layers"], w["heads"], w["ffn"], w["n_mels"])
    for name, p in enc.named_parameters():  # random-init weights of the named architecture
        if p.dim() >= 2 and "embed_positions" not in name:
            torch.nn.init.normal_(p, std=1.0 / np.sqrt(p.shape[1] * (p.shape[2] if p.dim() == 3 else 1)))
    proj = AudioProjector(w["d_model"], w["text_dim"], out_dtype=torch.bfloat16)
    if not w.get("train"):
        proj.requires_grad_(False)
    return enc.to(device), proj.to(device)


def cpu_baseline(wname, w):
    """The CPU oracle (a port of the reference path, see oracle/) timed on this host's cores on a
    bounded sample of the same workload."""
    from oracle import encoder as oe
    from oracle import logmel as ol
    from oracle import projector as op

    cores = torch.get_num_threads()
    rng = np.random.default_rng(1234)
    if wname == "splice":
        from oracle import splice as osp

        n, Lt, D = 16, w["text_len"], w["text_dim"]
        text = rng.standard_normal((n, Lt, D)).astype(np.float32)
        audio = rng.standard_normal((n, 1500, D)).astype(np.float32)
        ids = rng.integers(0, 49000, size=(n, Lt), dtype=np.int64)
        ids[:, 5] = 49153
        labels = rng.integers(0, 49000, size=(n, Lt), dtype=np.int64)
        osp.splice(text, audio, ids, labels, 49153)
        t0 = time.perf_counter()
        for _ in range(5):
            osp.splice(text, audio, ids, labels, 49153)
        dt = (time.perf_counter() - t0) / 5
        return dict(value=n * 30.0 / dt, unit="audio-sec/sec", cores=1, kind="port",
                    sample=f"{n} samples x 5 reps, NumPy oracle splice (fp32, single thread)")
    if wname == "logmel":
        n = 8
        wave = np.clip(0.1 * rng.standard_normal((n, 480000), dtype=np.float32), -1, 1)
        ol.log_mel(wave[:1], w["n_mels"])
        t0 = time.perf_counter()
        ol.log_mel(wave, w["n_mels"])
        dt = time.perf_counter() - t0
        return dict(value=n * 30.0 / dt, unit="audio-sec/sec", cores=1, kind="port",
                    sample=f"{n} x 30 s clips, numpy/scipy oracle log-mel (single thread)")
    n = 2 if w["d_model"] <= 768 else 1
    wave = np.clip(0.1 * rng.standard_normal((n, 480000), dtype=np.float32), -1, 1)
    tsd = {}
    g = torch.Generator().manual_seed(0)

    def rnd(*shape, std):
        return torch.randn(shape, generator=g) * std

    d, F, C, T = w["d_model"], w["ffn"], w["n_mels"], w["text_dim"]
    tsd["conv1.weight"], tsd["conv1.bias"] = rnd(d, C, 3, std=(3 * C) ** -0.5), rnd(d, std=0.02)
    tsd["conv2.weight"], tsd["conv2.bias"] = rnd(d, d, 3, std=(3 * d) ** -0.5), rnd(d, std=0.02)
    tsd["embed_positions.weight"] = rnd(1500, d, std=0.1)
    for i in range(w["layers"]):
        p = f"layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            tsd[p + f"self_attn.{nm}.weight"] = rnd(d, d, std=d ** -0.5)
            if nm != "k_proj":
                tsd[p + f"self_attn.{nm}.bias"] = rnd(d, std=0.02)
        for ln in ("self_attn_layer_norm", "final_layer_norm"):
            tsd[p + ln + ".weight"], tsd[p + ln + ".bias"] = torch.ones(d), torch.zeros(d)
        tsd[p + "fc1.weight"], tsd[p + "fc1.bias"] = rnd(F, d, std=d ** -0.5), rnd(F, std=0.02)
        tsd[p + "fc2.weight"], tsd[p + "fc2.bias"] = rnd(d, F, std=F ** -0.5), rnd(d, std=0.02)
    tsd["layer_norm.weight"], tsd["layer_norm.bias"] = torch.ones(d), torch.zeros(d)
    psd = {"norm.weight": torch.ones(d), "linear_1.weight": rnd(T, d, std=0.01), "linear_1.bias": torch.zeros(T),
           "linear_2.weight": rnd(T, T, std=1.0), "linear_2.bias": torch.zeros(T)}

    def run(x):
        f = ol.log_mel(x, C)
        h = oe.encoder_forward(tsd, f, w["heads"], n_layers=w["layers"])
        return op.projector_forward(psd, h)

    run(wave[:1])  # warm-up
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        run(wave)
    dt = (time.perf_counter() - t0) / reps
    return dict(value=n * 30.0 / dt, unit="audio-sec/sec", cores=cores, kind="port",
                sample=f"{n} x 30 s clips x {reps} reps, oracle log-mel + fp32 encoder + projector (torch CPU, {cores} threads)")


def pmc_traffic(workload, batch, kernel):
    """HBM bytes per launch of the dominant kernel, from the PMC passes committed under profiles/
    (tools/pmc_traffic.py; rocprofv3 cannot run inside this process).  None when no pass matches this
    workload and batch."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
    except (OSError, ValueError):
        return None
    if t.get("workload") != workload or t.get("clips_per_gpu") != batch:
        return None
    fam = {"gemm_bf16_kernel": "gemm", "attention_kernel": "attention", "layernorm_kernel": "layernorm",
           "logmel_tile_kernel+logmel_finish_kernel": "logmel"}.get(kernel)
    k = t.get("kernels", {}).get(fam)
    return k["bytes_per_launch"] if k else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="pipeline-small", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="clips per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-input", nargs="?", const="sync", default=None, choices=["sync", "prefetch", "prefetch16"],
                    help="diagnostic: each step starts from pinned host memory; the contract's `value` is measured WITHOUT "
                         "this flag (inputs resident in HBM).  sync: H2D copy of the padded fp32 waveforms on the compute "
                         "stream; prefetch / prefetch16: asr_amd.collate.FeaturePrefetcher (fp32 / int16 ragged buffer, copy "
                         "on a side stream, pad + mask on the GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    device = torch.device("cuda", torch.cuda.current_device())

    from asr_amd import LogMelFeatureExtractor, _lib

    lib = _lib.load()
    w = dict(WORKLOADS[args.workload])
    if args.batch:
        w["batch"] = args.batch
    B = w["batch"]
    fe = LogMelFeatureExtractor(feature_size=w.get("n_mels", 80))
    wave = synth_audio(B, rank, device)
    prefetch_iter = None
    if args.host_input in ("prefetch", "prefetch16"):
        from asr_amd.collate import FeaturePrefetcher, pack_ragged

        w_np = wave.cpu().numpy()
        if args.host_input == "prefetch16":
            packed = pack_ragged([np.round(c * 32767).astype(np.int16) for c in w_np], np.int16)
        else:
            packed = pack_ragged(list(w_np), np.float32)

        def _forever():
            while True:
                yield packed

        prefetch_iter = iter(FeaturePrefetcher(fe, _forever(), depth=2))

        def fetch():
            raise RuntimeError("prefetch mode feeds features, not waveforms")
    elif args.host_input:
        wave_host = wave.cpu().pin_memory()
        wave_dev = wave

        def fetch():
            wave_dev.copy_(wave_host, non_blocking=True)
            return wave_dev
    else:
        def fetch():
            return wave

    def features():
        if prefetch_iter is not None:
            return next(prefetch_iter).input_features
        return fe.extract_from_device(fetch())
    if args.workload == "splice":
        from asr_amd.splice import splice_audio_embeddings

        g = torch.Generator(device=device).manual_seed(rank)
        Ltxt, Dtxt, chunk_id = w["text_len"], w["text_dim"], 49153
        text = torch.randn((B, Ltxt, Dtxt), device=device, generator=g).to(torch.bfloat16)
        audio = torch.randn((B, 1500, Dtxt), device=device, generator=g).to(torch.bfloat16)
        ids = torch.randint(0, 49000, (B, Ltxt), device=device, generator=g)
        ids[:, 5] = chunk_id
        labels = torch.randint(0, 49000, (B, Ltxt), device=device, generator=g)

        def step():
            return splice_audio_embeddings(text, audio, ids, labels, chunk_id)[0]
    elif args.workload == "logmel":
        def step():
            return features()
    elif w.get("train"):
        from asr_amd.distributed import FlatGradAllReducer

        enc, proj = seeded_models(w, device)
        reducer = FlatGradAllReducer(proj.parameters()) if dist is not None else None
        # the decoder is outside this path: its gradient w.r.t. the audio embeddings is a fixed synthetic tensor
        g_up = (torch.randn((B, 1500, w["text_dim"]), device=device) * 1e-2).to(torch.bfloat16)

        def step():
            feats = features()
            emb = proj(enc(feats))  # the encoder wrapper runs under no_grad (src/modeling.py:33-40)
            if reducer is not None:
                reducer.zero_()  # .grad tensors are views of the reducer's flat buffer
            else:
                for p_ in proj.parameters():
                    p_.grad = None
            emb.backward(g_up)
            if reducer is not None:
                reducer.all_reduce()
            return emb.detach()
    else:
        enc, proj = seeded_models(w, device)

        def step():
            with torch.no_grad():
                feats = features()
                return proj(enc(feats))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out.float()).all()

    # ---- instrumented pass (rank 0): HIP events around every launch of the hot kernels
    roofline = None
    stages = {}
    if rank == 0:
        lib.asrk_profile_enable(1)
        for _ in range(min(args.steps, 3)):
            step()
        torch.cuda.synchronize()
        lib.asrk_profile_enable(0)
        names = {_lib.PROF_GEMM: "gemm_bf16_kernel", _lib.PROF_ATTENTION: "attention_kernel",
                 _lib.PROF_LOGMEL: "logmel_tile_kernel+logmel_finish_kernel", _lib.PROF_NORM: "layernorm_kernel",
                 _lib.PROF_SPLICE: "splice_fwd_kernel"}
        tot = {}
        for cat, nm in names.items():
            ms, work, n = _lib.profile_read(cat)
            if n:
                tot[cat] = (ms, work, n)
                is_flops = cat in (_lib.PROF_GEMM, _lib.PROF_ATTENTION)
                ach = work / (ms * 1e-3) / (1e12 if is_flops else 1e9)
                stages[nm] = dict(ms_per_step=ms / min(args.steps, 3), launches_per_step=n // min(args.steps, 3),
                                  achieved=ach, unit="TFLOP/s" if is_flops else "GB/s",
                                  frac=ach / (PEAK_BF16_TFLOPS if is_flops else PEAK_HBM_GBS))
        dom = max(tot, key=lambda c: tot[c][0])
        ms, work, n = tot[dom]
        if dom in (_lib.PROF_GEMM, _lib.PROF_ATTENTION):
            ach = work / (ms * 1e-3) / 1e12
            roofline = dict(bound="mfma", kernel=names[dom], achieved=ach, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                            frac=ach / PEAK_BF16_TFLOPS, traffic=None, avg_launch_us=ms * 1e3 / n, launches=n)
        else:
            ach = work / (ms * 1e-3) / 1e9
            roofline = dict(bound="hbm", kernel=names[dom], achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s",
                            frac=ach / PEAK_HBM_GBS, traffic=None, avg_launch_us=ms * 1e3 / n, launches=n)

    if rank == 0 and roofline is not None:
        roofline["traffic"] = pmc_traffic(args.workload, B, roofline["kernel"])
    if rank == 0:
        value = world * B * 30.0 * args.steps / dt
        res = {
            "metric": METRIC, "value": value, "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.workload == "logmel" else "bf16", "data": "synthetic",
            "config": {"workload": args.workload, "clips_per_gpu": B, "clip_seconds": 30, "global_batch": world * B,
                       **{k: v for k, v in w.items() if k != "batch"},
                       "parallelism": f"dp{world} (independent clips, no data-path collective)"},
            "roofline": roofline, "stages": stages,
        }
        if args.host_input:
            res["data"] = "synthetic, copied from pinned host memory every step (diagnostic, not the contract's value)"
        if args.workload not in ("logmel", "splice"):
            fl = flops_per_clip(w)
            res["pipeline_tflops"] = fl * world * B * args.steps / dt / 1e12
            res["flops_per_clip"] = fl
        if not args.no_cpu_baseline and world == 1:
            torch.cuda.synchronize()
            res["cpu_baseline"] = cpu_baseline(args.workload, w)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
