"""CPU, world size 2 over gloo: `python bench.py --gpus 2` starts its own ranks (fresh child processes, before any GPU
call), the ranks meet at the barriers, rank 0 prints ONE JSON line that reports the world size that actually ran.
The `stub` workload is a small CPU matmul so that this runs without a GPU; the GPU workloads go through the same
launcher, barriers and max-over-ranks timing."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks():
    r = _run(["--gpus", "2", "--workload", "stub", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2 * d["config"]["clips_per_gpu"]
    assert len(d["ms_per_step_ranks"]) == 2 and d["ms_per_step"] == pytest.approx(max(d["ms_per_step_ranks"]))
    # whole-job aggregate over both ranks
    assert d["value"] == pytest.approx(2 * d["config"]["clips_per_gpu"] * 30.0 / (d["ms_per_step"] * 1e-3))


@pytest.mark.timeout(120)
def test_bench_single_rank_stub_and_failed_rank_exit_code():
    r = _run(["--workload", "stub", "--steps", "2", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip())["n_gpus"] == 1
    # a rank that cannot start (unknown workload -> argparse exits 2 in every child) makes the launcher fail too
    r = _run(["--gpus", "2", "--workload", "no-such-workload"])
    assert r.returncode != 0
