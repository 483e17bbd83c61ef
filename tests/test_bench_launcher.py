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


@pytest.mark.timeout(300)
def test_train_step_collective_sequence_is_the_same_on_every_rank():
    """`train-large`'s control flow on a CPU stand-in (workload stub-train, gloo, world size 2): the timed steps, the
    instrumented steps and the all-reduce timing steps all contain the gradient all-reduce, so every rank has to run
    all of them -- a rank-0-only pass would pair its all-reduces with the others' barrier and hang."""
    r = _run(["--gpus", "2", "--workload", "stub-train", "--steps", "4", "--warmup", "2"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["grad_allreduce_ms"] >= 0.0
    calls = d["grad_allreduce_calls_ranks"]
    assert len(calls) == 2 and calls[0] == calls[1] == 2 + 4 + 3 + 3  # warm-up + timed + instrumented + timing steps


@pytest.mark.timeout(120)
def test_launcher_stops_the_others_when_one_rank_dies_after_init():
    """Rank 1 exits with code 3 right after init_process_group (ASRK_BENCH_TEST_FAIL_RANK); rank 0 is then alone in a
    gloo collective.  The launcher has to notice, kill rank 0 and return non-zero within seconds, not after the
    collective's own timeout, and leave no child behind."""
    import time

    t0 = time.time()
    r = _run(["--gpus", "2", "--workload", "stub", "--steps", "3", "--warmup", "1"],
             env_extra={"ASRK_BENCH_TEST_FAIL_RANK": "1"}, timeout=100)
    took = time.time() - t0
    assert r.returncode != 0 and took < 30, (r.returncode, took, r.stderr[-1000:])
    assert "(1, 3)" in r.stderr  # rank 1, exit code 3, named in the launcher's message
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]  # no result line from a failed job
    # no child left: every bench.py process that has this launcher's MASTER_PORT in its environment is gone
    # (the launcher kill()s and wait()s each child it started; a zombie or survivor would still hold the gloo port)
    time.sleep(0.2)
    out = subprocess.run(["ps", "-eo", "pid,args"], capture_output=True, text=True).stdout
    assert not [ln for ln in out.splitlines() if "bench.py" in ln and "--workload stub" in ln and "pytest" not in ln], out
