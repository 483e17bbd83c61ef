#!/usr/bin/env python
"""Benchmark of the audio hot path on MI355X: log-mel -> encoder -> projector on 30 s clips.

    python bench.py [--gpus N --steps K --warmup W] [--workload pipeline-small|pipeline-large|train-large|logmel|splice]

One "step" = one pass of the hot path over one batch of synthetic 16 kHz audio that is already
resident in HBM.  With N > 1 there is one rank per GPU: either the driver starts them through
torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), or -- when
`--gpus N` is given and WORLD_SIZE is not set -- this script starts N fresh child processes itself
BEFORE it touches the GPU (launch_ranks below) and relays rank 0's line.  Every rank processes its
own batch (clips are independent: weak scaling, no data-path collective), the ranks meet at a
barrier on both sides of the timed region and the slowest rank's time is used.  Rank 0 prints ONE
JSON line.

Workloads (BASELINE.json configs):
  pipeline-small  (default) 64 x 30 s per GPU, whisper-small dims (d768 L12 H12 ffn3072, 80 mel),
                  projector 768 -> 960 -> 960, bf16 MFMA: configs[2] fed by the configs[1] front-end,
                  i.e. the metric's "log-mel+encoder+projector"
  pipeline-large  16 x 30 s per GPU, large-v3 dims (d1280 L32 H20 ffn5120, 128 mel)
  train-large     configs[3]: pipeline-large plus projector backward (synthetic upstream gradient) and, for N > 1,
                  one flat RCCL all-reduce of the projector gradients per step
  logmel          256 x 30 s per GPU, front-end only (configs[1]), HBM-bound; logmel-128: the same with 128 mel bins
  splice          SURVEY 8(f2): the embed splice of ASRModel.forward alone, 64 samples (HBM-bound copy)
  stub            no GPU: a small CPU matmul per step over gloo; exists so that the launcher, the barriers and the
                  max-over-ranks timing can be tested at world size 2 without a GPU (tests/test_bench_launcher.py)
  stub-train      no GPU: train-large's control flow (FlatGradAllReducer, every pass that calls step()) on a tiny CPU model
"""
import argparse
import json
import os
import socket
import subprocess
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
    "logmel-128": dict(batch=256, n_mels=128),  # configs[1] at the large-v3 front-end width
    # configs[4]'s front-end: clips of 1-30 s zero-padded to the 30 s window, valid lengths handed to the kernel (which
    # skips the tiles that lie in the padding); `value` counts the VALID audio seconds
    "logmel-ragged": dict(batch=256, n_mels=80, ragged=True),
    # the embed splice of ASRModel.forward, 64 samples x (64 text tokens + 1500 audio
    # positions) x 960 (SmolLM2-360M width), bf16 like the decoder; HBM-bound copy
    "splice": dict(batch=64, text_len=64, text_dim=960),
    "stub": dict(batch=4),
    # the train step's sequence of collectives (timed steps, instrumented steps, the all-reduce timing steps) on a CPU
    # stand-in for the projector over gloo: tests/test_bench_launcher.py checks at world size 2 that no rank issues a
    # collective the others do not
    "stub-train": dict(batch=4, train=True),
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


def encode_and_project(enc, proj, feats):
    from asr_amd.projector import encode_and_project as fused

    return fused(enc, proj, feats)


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
    """The CPU oracle (a port of the reference path, see oracle/) timed on this host's cores on a bounded sample of
    the same workload, at SURVEY.md section 8(d)'s sizes: front-end 16 clips, encoder + projector 4 clips fp32
    (1 clip at the large-v3 dimensions, whose forward is 6.5x the work), one warm-up and three timed repetitions,
    median reported."""
    from oracle import encoder as oe
    from oracle import logmel as ol
    from oracle import projector as op

    cores = torch.get_num_threads()
    rng = np.random.default_rng(1234)

    def timed(fn, reps=3):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    if wname == "splice":
        from oracle import splice as osp

        n, Lt, D = 16, w["text_len"], w["text_dim"]
        text = rng.standard_normal((n, Lt, D)).astype(np.float32)
        audio = rng.standard_normal((n, 1500, D)).astype(np.float32)
        ids = rng.integers(0, 49000, size=(n, Lt), dtype=np.int64)
        ids[:, 5] = 49153
        labels = rng.integers(0, 49000, size=(n, Lt), dtype=np.int64)
        osp.splice(text, audio, ids, labels, 49153)
        dt = timed(lambda: osp.splice(text, audio, ids, labels, 49153), reps=5)
        return dict(value=n * 30.0 / dt, unit="audio-sec/sec", cores=1, kind="port",
                    sample=f"{n} samples, median of 5, NumPy oracle splice (fp32, single thread)")
    if wname.startswith("logmel"):
        n = 16
        wave = np.clip(0.1 * rng.standard_normal((n, 480000), dtype=np.float32), -1, 1)
        ol.log_mel(wave[:1], w["n_mels"])
        dt = timed(lambda: ol.log_mel(wave, w["n_mels"]))
        return dict(value=n * 30.0 / dt, unit="audio-sec/sec", cores=1, kind="port",
                    sample=f"{n} x 30 s clips, 1 warm-up + median of 3, numpy/scipy oracle log-mel (single thread)")
    n = 4 if w["d_model"] <= 768 else 1
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
    dt = timed(lambda: run(wave))
    return dict(value=n * 30.0 / dt, unit="audio-sec/sec", cores=cores, kind="port",
                sample=f"{n} x 30 s clips, 1 warm-up + median of 3, oracle log-mel + fp32 encoder + projector "
                       f"(torch CPU, {cores} threads)")


def pmc_traffic(workload, batch, kernel):
    """(HBM bytes per launch of the dominant kernel, provenance) from the PMC passes committed under profiles/
    (tools/pmc_traffic.py; rocprofv3 cannot run inside this process).  (None, None) when no pass matches this workload
    and batch.  The provenance names the file and its git blob hash, so a line can be matched to the committed pass."""
    import hashlib

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")
    try:
        with open(path, "rb") as fh:
            raw = fh.read()
        t = json.loads(raw)
    except (OSError, ValueError):
        return None, None
    if t.get("workload") != workload or t.get("clips_per_gpu") != batch:
        return None, None
    k = t.get("kernels", {}).get(kernel)
    if not k:
        return None, None
    blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()  # == `git hash-object profiles/hbm_traffic.json`
    return k["bytes_per_launch"], {"file": "profiles/hbm_traffic.json", "git_blob": blob,
                                   "note": "separate rocprofv3 --pmc passes of this command, not collected in this run"}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, deadline_s=600.0):
    """`python bench.py --gpus N` without a launcher around it: start N fresh processes of this script, one per GPU,
    with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, BEFORE this process has made any GPU call (it
    never makes one: a process that has initialised the GPU must not be replaced or forked).  Rank 0's stdout -- the one
    JSON line -- is drained by a thread while EVERY child is polled: as soon as one rank exits non-zero (or the deadline
    passes) the others, which would sit in a barrier for as long as the collective's own timeout, are killed and the
    launcher returns non-zero within a fraction of a second."""
    import threading

    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes needs dmabuf IPC on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)  # EOF when rank 0 exits
    reader.start()
    rcs = [None] * n
    deadline = time.time() + deadline_s
    timed_out = False
    while any(rc is None for rc in rcs):
        for i, p_ in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p_.poll()
        if any(rc not in (None, 0) for rc in rcs):
            break  # one rank failed: the others would wait at a barrier for ever
        if time.time() > deadline:
            timed_out = True
            break
        time.sleep(0.05)
    for i, p_ in enumerate(procs):  # stop exactly the children started above that are still running
        if rcs[i] is None:
            p_.kill()
            rcs[i] = p_.wait()
    reader.join(timeout=10)
    out0 = b"".join(chunks)
    for line in out0.decode(errors="replace").splitlines():
        # the one JSON line goes to stdout; whatever else a library printed there (gloo / libdrm notices) to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr)
    sys.stdout.flush()
    bad = [(i, rc) for i, rc in enumerate(rcs) if rc != 0]
    if bad or timed_out:
        print(f"bench.py: ranks failed (rank, exit code): {bad}" + (" [deadline passed]" if timed_out else ""), file=sys.stderr)
        return 1
    return 0


# rocprof symbol of each profiling category of the library (include/asrk.h, asrk_profile_read)
PROF_SYMBOLS = {0: "gemm256p_kernel", 5: "gemm256_kernel", 6: "gemm_bf16_kernel", 1: "attention_kernel",
                2: "logmel_tile_kernel+logmel_finish_kernel", 3: "layernorm_kernel", 4: "splice_fwd_kernel"}
PROF_FLOPS = (0, 5, 6, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default 10; 100 for the sub-millisecond workloads logmel / logmel-128 / splice, whose "
                         "timed region would otherwise be dominated by its two synchronisations)")
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
    if args.steps is None:
        args.steps = 100 if (args.workload.startswith("logmel") or args.workload == "splice") else 10

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = not args.workload.startswith("stub")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
        world = dist.get_world_size()  # what is reported is what actually runs
        if os.environ.get("ASRK_BENCH_TEST_FAIL_RANK") == str(rank):
            sys.exit(3)  # test hook (tests/test_bench_launcher.py): a rank that dies after the group exists
    elif on_gpu:
        torch.cuda.set_device(0)
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but {world} rank(s) are running", file=sys.stderr)
    device = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")

    w = dict(WORKLOADS[args.workload])
    if args.batch:
        w["batch"] = args.batch
    B = w["batch"]
    lib = _lib = None
    comm_ms = []  # train-large, N > 1: host-observed time of each gradient all-reduce (enqueue -> complete)
    if on_gpu:
        from asr_amd import LogMelFeatureExtractor, _lib

        lib = _lib.load()
        fe = LogMelFeatureExtractor(feature_size=w.get("n_mels", 80))
        wave = synth_audio(B, rank, device)
    valid_lens = None
    if w.get("ragged"):
        valid_lens = torch.from_numpy(np.random.default_rng(99 + rank).integers(16000, 480001, size=B).astype(np.int32)).to(device)
        wave = wave * (torch.arange(wave.shape[1], device=device)[None, :] < valid_lens[:, None])
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
        return fe.extract_from_device(fetch(), valid_lens)
    if args.workload == "stub":
        gcpu = torch.Generator().manual_seed(rank)
        a_ = torch.randn((B, 256, 256), generator=gcpu)

        def step():
            return torch.bmm(a_, a_)
    elif args.workload == "splice":
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
    elif args.workload.startswith("logmel"):
        def step():
            return features()
    elif w.get("train"):
        from asr_amd.distributed import FlatGradAllReducer

        if on_gpu:
            enc, proj = seeded_models(w, device)
            # the decoder is outside this path: its gradient w.r.t. the audio embeddings is a fixed synthetic tensor
            g_up = (torch.randn((B, 1500, w["text_dim"]), device=device) * 1e-2).to(torch.bfloat16)

            def fwd():
                return encode_and_project(enc, proj, features())  # the encoder runs under no_grad (src/modeling.py:33-40)
        else:  # stub-train: the same control flow on a CPU stand-in
            torch.manual_seed(0)
            proj = torch.nn.Linear(64, 64)
            x_cpu = torch.randn((B, 64), generator=torch.Generator().manual_seed(rank))
            g_up = torch.ones((B, 64))

            def fwd():
                return proj(x_cpu)
        reducer = FlatGradAllReducer(proj.parameters()) if dist is not None else None
        timing = {"on": False}

        def now():  # GPU: an event on the current stream; CPU: the host clock
            if on_gpu:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                return e
            return time.perf_counter()

        def step():
            emb = fwd()
            if reducer is not None:
                reducer.zero_()  # .grad tensors are views of the reducer's flat buffer
            else:
                for p_ in proj.parameters():
                    p_.grad = None
            emb.backward(g_up)
            if reducer is not None:
                t_a = now() if timing["on"] else None
                reducer.all_reduce()
                timing["calls"] = timing.get("calls", 0) + 1
                if timing["on"]:
                    t_b = now()
                    if on_gpu:
                        t_b.synchronize()
                        comm_ms.append(t_a.elapsed_time(t_b))
                    else:
                        comm_ms.append((t_b - t_a) * 1e3)
            return emb.detach()
    else:
        enc, proj = seeded_models(w, device)

        def step():
            with torch.no_grad():
                feats = features()
                return encode_and_project(enc, proj, feats)

    def barrier():
        if on_gpu:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt_rank = time.perf_counter() - t0
    dt, rank_ms = dt_rank, [dt_rank / args.steps * 1e3]
    if dist is not None:
        tt = torch.tensor([dt_rank], device=device, dtype=torch.float64)
        gathered = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(gathered, tt)
        rank_ms = [float(t_.item()) / args.steps * 1e3 for t_ in gathered]
        dt = max(rank_ms) * args.steps / 1e3  # the slowest rank's time
    assert torch.isfinite(out.float()).all()

    # ---- instrumented pass (rank 0): HIP events around every launch of the hot kernels, on their launch stream
    roofline = None
    stages = {}
    if on_gpu or w.get("train"):
        # EVERY rank runs these steps (train-large's step contains the gradient all-reduce: the ranks' sequences of
        # collectives must stay identical); only rank 0 switches the event hook on and reads it
        n_inst = min(args.steps, 3)
        if rank == 0 and on_gpu:
            lib.asrk_profile_enable(1)
        for _ in range(n_inst):
            step()
        if on_gpu:
            torch.cuda.synchronize()
        if rank == 0 and on_gpu:
            lib.asrk_profile_enable(0)
    if rank == 0 and on_gpu:
        tot = {}
        fam = {}
        for cat, nm in PROF_SYMBOLS.items():
            ms, work, n = _lib.profile_read(cat)
            if n:
                tot[cat] = (ms, work, n)
                is_flops = cat in PROF_FLOPS
                ach = work / (ms * 1e-3) / (1e12 if is_flops else 1e9)
                stages[nm] = dict(ms_per_step=ms / n_inst, launches_per_step=n / n_inst, avg_launch_us=ms * 1e3 / n,
                                  achieved=ach, unit="TFLOP/s" if is_flops else "GB/s",
                                  frac=ach / (PEAK_BF16_TFLOPS if is_flops else PEAK_HBM_GBS))
                if nm.startswith("gemm"):
                    f_ = fam.setdefault("gemm (all three kernels)", [0.0, 0.0, 0])
                    f_[0] += ms
                    f_[1] += work
                    f_[2] += n
        for nm, (ms, work, n) in fam.items():
            ach = work / (ms * 1e-3) / 1e12
            stages[nm] = dict(ms_per_step=ms / n_inst, launches_per_step=n / n_inst, avg_launch_us=ms * 1e3 / n,
                              achieved=ach, unit="TFLOP/s", frac=ach / PEAK_BF16_TFLOPS)
        dom = max(tot, key=lambda c: tot[c][0])  # the kernel (rocprof symbol) with the most GPU time
        ms, work, n = tot[dom]
        if dom in PROF_FLOPS:
            ach = work / (ms * 1e-3) / 1e12
            roofline = dict(bound="mfma", kernel=PROF_SYMBOLS[dom], achieved=ach, peak=PEAK_BF16_TFLOPS, unit="TFLOP/s",
                            frac=ach / PEAK_BF16_TFLOPS, traffic=None, avg_launch_us=ms * 1e3 / n, launches=n)
        else:
            ach = work / (ms * 1e-3) / 1e9
            roofline = dict(bound="hbm", kernel=PROF_SYMBOLS[dom], achieved=ach, peak=PEAK_HBM_GBS, unit="GB/s",
                            frac=ach / PEAK_HBM_GBS, traffic=None, avg_launch_us=ms * 1e3 / n, launches=n)
        roofline["traffic"], roofline["traffic_source"] = pmc_traffic(args.workload, B, roofline["kernel"])
    if comm_ms is not None and w.get("train") and dist is not None:
        # the collective's own time, outside the timed region: three more steps with events around the all-reduce
        timing["on"] = True
        for _ in range(3):
            step()
        timing["on"] = False
        barrier()
        # every rank must have issued the same number of gradient all-reduces (reported, and checked by the gloo test)
        cnt = torch.tensor([timing.get("calls", 0)], device=device, dtype=torch.int64)
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        allreduce_calls = [int(c.item()) for c in cnts]
    if rank == 0:
        clip_s = 30.0 if valid_lens is None else float(valid_lens.sum().item()) / 16000.0 / B  # mean valid seconds (rank 0's)
        value = world * B * clip_s * args.steps / dt
        res = {
            "metric": METRIC, "value": value, "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "ms_per_step_ranks": rank_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.workload.startswith("logmel") or not on_gpu else "bf16", "data": "synthetic",
            "config": {"workload": args.workload, "clips_per_gpu": B, "clip_seconds": round(clip_s, 2), "window_seconds": 30,
                       "global_batch": world * B,
                       **{k: v for k, v in w.items() if k != "batch"},
                       "parallelism": f"dp{world} (independent clips, no data-path collective)"},
            "roofline": roofline, "stages": stages,
        }
        if comm_ms:
            res["grad_allreduce_ms"] = float(np.median(comm_ms))
            res["grad_allreduce_calls_ranks"] = allreduce_calls
        if args.host_input:
            res["data"] = "synthetic, copied from pinned host memory every step (diagnostic, not the contract's value)"
        if "d_model" in w:
            fl = flops_per_clip(w)
            res["pipeline_tflops"] = fl * world * B * args.steps / dt / 1e12
            res["flops_per_clip"] = fl
        if not args.no_cpu_baseline and world == 1 and on_gpu:
            torch.cuda.synchronize()
            res["cpu_baseline"] = cpu_baseline(args.workload, w)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
